"""Where a ``model_args.model_name_or_path`` of the reference's YAML points on THIS machine (one policy for the LM and for SAM2).

The reference hands the string to ``from_pretrained`` / ``download_model`` (roll/utils/checkpoint_manager.py:33-44: a hub snapshot, file-locked);
there is no hub here, so:

  synthetic:<name>        random weights of that geometry, said loudly (tests, benchmarks, offline demos)
  an existing directory   used as it is
  a hub id                looked up in the local HuggingFace cache (``HF_HOME`` / ``HF_HUB_CACHE``; ``snapshot_download(local_files_only=True)``:
                          what a machine that once ran the reference has on disk) -- never fetched
  anything else           FileNotFoundError, as ``from_pretrained`` would raise; ``SR_ALLOW_SYNTHETIC_WEIGHTS=1`` turns it into a loud fallback to
                          synthetic weights (results from random weights must never pass for results by accident)
"""
from __future__ import annotations

import os
import warnings


def resolve(path: str, role: str, synthetic_default: str):
    """-> (kind, value): ("synthetic", name) | ("dir", local directory) | ("synthetic-fallback", synthetic_default)."""
    path = str(path or "")
    if not path:
        return "synthetic", synthetic_default
    if path.startswith("synthetic"):
        return "synthetic", path
    if os.path.isdir(path):
        return "dir", path
    looks_like_hub_id = not os.path.isabs(path) and path.count("/") == 1 and not path.startswith(".")
    if looks_like_hub_id:
        try:
            from huggingface_hub import snapshot_download
            snap = snapshot_download(repo_id=path, local_files_only=True)
            if os.path.isdir(snap):
                return "dir", snap
        except Exception:  # noqa: BLE001  (not in the cache / hub library absent: fall through to the refusal below)
            pass
    if os.environ.get("SR_ALLOW_SYNTHETIC_WEIGHTS") == "1":
        warnings.warn(f"{role}: nothing on disk for {path!r} (not a directory, not in the local HuggingFace cache): running {synthetic_default} "
                      f"with SYNTHETIC weights because SR_ALLOW_SYNTHETIC_WEIGHTS=1")
        return "synthetic-fallback", synthetic_default
    raise FileNotFoundError(f"{role}: no checkpoint for {path!r} -- not a directory and not in the local HuggingFace cache "
                            f"(HF_HOME={os.environ.get('HF_HOME', '~/.cache/huggingface')!r}; the hub is never contacted).  Point model_name_or_path at a "
                            f"checkpoint directory, use '{synthetic_default}' for random weights, or set SR_ALLOW_SYNTHETIC_WEIGHTS=1 to fall back to them on purpose")
