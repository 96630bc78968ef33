"""Trainer -> engine weight synchronisation, receiver side (SURVEY.md section 8(F) "next" row N4).

The reference pushes updated weights from the trainer to the inference engine after every optimisation step
(roll/distributed/strategy/megatron_strategy.py:411-448): parameters are packed into fixed-size int8 buckets (256 MiB),
a tensor may be split across consecutive buckets, and every bucket travels with a ``meta_infos`` dict

    name -> {"bucket_start", "tensor_start", "save_bytes", "tensor_meta": shape / dtype of the WHOLE tensor}

(roll/utils/send_recv_utils.py:64-179).  The engine side reassembles the tensors and hands every completed one to its
weight loader (roll/third_party/vllm/worker_helper.py:64-115).  ``BucketReceiver`` is that reassembly; ``BucketSender``
is the matching packer, used by the tests and by anything that wants to drive the receiver without the reference's
trainer.  Completed tensors go to ``Engine.load_weight`` (HF names, re-laid-out on the device).
"""
from __future__ import annotations

from typing import Dict, Iterator, Tuple

import torch


def _nbytes(t: torch.Tensor) -> int:
    return t.numel() * t.element_size()


def _meta_shape_dtype(tm) -> Tuple[Tuple[int, ...], torch.dtype]:
    """tensor_meta is a meta-device tensor on the wire inside one process and a dict once it crossed an RPC boundary."""
    if isinstance(tm, dict):
        return tuple(int(x) for x in tm["shape"]), tm["dtype"]
    return tuple(tm.shape), tm.dtype


class BucketSender:
    def __init__(self, bucket_size: int, device="cpu"):
        self.bucket_size, self.device = int(bucket_size), device
        self.buffer = torch.empty(self.bucket_size, dtype=torch.int8, device=device)
        self.write, self.meta = 0, {}

    def push(self, name: str, tensor: torch.Tensor) -> Iterator[Tuple[Dict, torch.Tensor]]:
        """Yields (meta_infos, buffer) every time a bucket fills up; the buffer is reused after the caller is done with it."""
        raw = tensor.detach().contiguous().view(-1).view(torch.int8)
        total, start = raw.numel(), 0
        while start < total:
            n = min(total - start, self.bucket_size - self.write)
            self.buffer[self.write:self.write + n].copy_(raw[start:start + n])
            self.meta[name] = {"bucket_start": self.write, "tensor_start": start, "save_bytes": n,
                               "tensor_meta": {"shape": list(tensor.shape), "dtype": tensor.dtype}}
            self.write += n
            start += n
            if self.write == self.bucket_size:
                yield self.meta, self.buffer
                self.write, self.meta = 0, {}

    def flush(self):
        """The last, partly filled bucket (or (None, None))."""
        if self.write == 0:
            return None, None
        out = (self.meta, self.buffer)
        self.write, self.meta = 0, {}
        return out


class BucketReceiver:
    def __init__(self):
        self.waiting: Dict[str, torch.Tensor] = {}
        self.filled: Dict[str, int] = {}

    def process_bucket(self, meta_infos: Dict, buffer: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Copies the bucket's pieces into their tensors; returns the tensors this bucket completed."""
        done = {}
        for name, m in meta_infos.items():
            shape, dtype = _meta_shape_dtype(m["tensor_meta"])
            t = self.waiting.get(name)
            if t is None:
                t = self.waiting[name] = torch.empty(shape, dtype=dtype, device=buffer.device)
                self.filled[name] = 0
            b0, t0, n = int(m["bucket_start"]), int(m["tensor_start"]), int(m["save_bytes"])
            if t0 != self.filled[name]:
                raise ValueError(f"bucket piece of {name} starts at byte {t0}, expected {self.filled[name]} (pieces arrive in order)")
            t.view(-1).view(torch.int8)[t0:t0 + n].copy_(buffer[b0:b0 + n])
            self.filled[name] = t0 + n
            if t0 + n == _nbytes(t):
                done[name] = self.waiting.pop(name)
                self.filled.pop(name)
        return done

    def clear(self):
        if self.waiting:
            raise RuntimeError(f"{len(self.waiting)} tensors were only partly received: {sorted(self.waiting)[:3]}")
