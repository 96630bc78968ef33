"""Minimal ``DataProto`` (reference: roll/distributed/scheduler/protocol.py:145-733): a dict of batch tensors, a
dict of per-sample object arrays and a meta dict, with the chunk / concat / pop / rename / union operations the infer
pipeline uses.  tensordict and Ray are not required."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch


@dataclass
class DataProto:
    batch: Optional[Dict[str, torch.Tensor]] = None
    non_tensor_batch: Dict[str, np.ndarray] = field(default_factory=dict)
    meta_info: Dict = field(default_factory=dict)

    def __len__(self):
        if self.batch:
            return next(iter(self.batch.values())).shape[0]
        if self.non_tensor_batch:
            return len(next(iter(self.non_tensor_batch.values())))
        return 0

    @classmethod
    def from_single_dict(cls, data: Dict, meta_info=None) -> "DataProto":
        tensors = {k: v for k, v in data.items() if isinstance(v, torch.Tensor)}
        others = {k: (v if isinstance(v, np.ndarray) else np.array(v, dtype=object)) for k, v in data.items()
                  if not isinstance(v, torch.Tensor)}
        return cls(batch=tensors, non_tensor_batch=others, meta_info=dict(meta_info or {}))

    def pop(self, batch_keys=None, non_tensor_batch_keys=None, meta_info_keys=None) -> "DataProto":
        b = {k: self.batch.pop(k) for k in (batch_keys or [])}
        n = {k: self.non_tensor_batch.pop(k) for k in (non_tensor_batch_keys or [])}
        m = {k: self.meta_info.pop(k) for k in (meta_info_keys or [])}
        return DataProto(batch=b, non_tensor_batch=n, meta_info=m)

    def rename(self, old_keys=None, new_keys=None) -> "DataProto":
        old = [old_keys] if isinstance(old_keys, str) else list(old_keys or [])
        new = [new_keys] if isinstance(new_keys, str) else list(new_keys or [])
        for o, n in zip(old, new):
            self.batch[n] = self.batch.pop(o)
        return self

    def union(self, other: "DataProto") -> "DataProto":
        self.batch = {**(self.batch or {}), **(other.batch or {})}
        self.non_tensor_batch = {**self.non_tensor_batch, **other.non_tensor_batch}
        self.meta_info = {**self.meta_info, **other.meta_info}
        return self

    def chunk(self, chunks: int) -> List["DataProto"]:
        """Contiguous np.array_split-sized pieces (reference protocol.py:550-617)."""
        n = len(self)
        idx = np.array_split(np.arange(n), chunks)
        out = []
        for ix in idx:
            sl = slice(int(ix[0]), int(ix[-1]) + 1) if len(ix) else slice(0, 0)
            out.append(DataProto(batch={k: v[sl] for k, v in (self.batch or {}).items()},
                                 non_tensor_batch={k: v[sl] for k, v in self.non_tensor_batch.items()},
                                 meta_info=dict(self.meta_info)))
        return out

    @staticmethod
    def concat(data: List["DataProto"]) -> "DataProto":
        keys = data[0].batch.keys() if data[0].batch else []
        batch = {k: torch.cat([d.batch[k] for d in data], dim=0) for k in keys}
        nt = {k: np.concatenate([d.non_tensor_batch[k] for d in data], axis=0) for k in data[0].non_tensor_batch}
        return DataProto(batch=batch, non_tensor_batch=nt, meta_info=dict(data[0].meta_info))

    def to(self, device) -> "DataProto":
        if self.batch:
            self.batch = {k: v.to(device) for k, v in self.batch.items()}
        return self
