"""Config tree of the infer path.  The reference composes its YAML with hydra and maps it onto dataclasses with a
non-strict dacite (examples/start_rlvr_socioseg_pipeline_infer.py:24-30; roll/configs/*.py); hydra, omegaconf and
dacite are not available here, so the same YAML files are read with PyYAML, `${...}` interpolations are resolved and
the tree is exposed with attribute access under the same field names."""
from __future__ import annotations

import os
import re

import yaml


class Cfg(dict):
    """dict with attribute access; missing fields read as None (the reference uses non-strict dacite + getattr)."""

    def __getattr__(self, k):
        return self.get(k)

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return Cfg({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


_REF = re.compile(r"\$\{([A-Za-z0-9_.]+)\}")


def _resolve(node, root):
    if isinstance(node, dict):
        return {k: _resolve(v, root) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root) for v in node]
    if isinstance(node, str):
        def look(path):
            cur = root
            for part in path.split("."):
                cur = cur[part]
            return cur
        m = _REF.fullmatch(node)
        if m:
            return _resolve(look(m.group(1)), root)
        return _REF.sub(lambda mm: str(look(mm.group(1))), node)
    return node


def parse_device_mapping(s):
    """`list(range(a,b))` / `[0,1,2]` without eval (the reference evals this string, roll/configs/worker_config.py:110)."""
    if s is None or isinstance(s, list):
        return s
    m = re.fullmatch(r"\s*list\(range\(\s*(\d+)\s*,\s*(\d+)\s*\)\)\s*", str(s))
    if m:
        return list(range(int(m.group(1)), int(m.group(2))))
    return [int(x) for x in re.findall(r"\d+", str(s))]


def load_yaml_config(config_path: str, config_name: str) -> Cfg:
    """`--config_path infer --config_name rlvr_megatron` -> examples/infer/rlvr_megatron.yaml (hydra-style lookup
    relative to the launcher's directory); `defaults:` includes that do not exist are skipped (training-only files)."""
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "examples")
    cand = [os.path.join(config_path, config_name + ".yaml"), os.path.join(here, config_path, config_name + ".yaml")]
    path = next((c for c in cand if os.path.exists(c)), None)
    if path is None:
        raise FileNotFoundError(f"config {config_name}.yaml not found under {config_path}")
    raw = yaml.safe_load(open(path)) or {}
    raw.pop("defaults", None)
    raw.pop("hydra", None)
    return _wrap(_resolve(raw, raw))
