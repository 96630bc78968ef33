"""Model providers of the infer path (reference: roll/models/model_providers.py:515-562 ``sam2_seg_model_provider``).

The reference builds ``SAM2ImagePredictor(build_sam2(sam2_hiera_l.yaml, sam2_hiera_large.pt))`` from ``model_args.model_name_or_path``.
Here the provider returns ``socioreasoner_amd.sam2.Sam2Predictor`` (same ``set_image`` / ``predict`` contract) over the MI355X kernels:

  <dir>/sam2_hiera_large.pt        the sam2 package's checkpoint (``{"model": state_dict}``), renamed to HF parameter names
  <dir>/*.safetensors              an HF ``Sam2Model`` checkpoint
  synthetic:sam2-hiera-large | synthetic:sam2-tiny     random weights of that geometry (no checkpoint offline)

Precision.  The reference's provider ignores ``model_args.dtype`` (the YAML's ``dtype: bf16``, examples/infer/rlvr_megatron.yaml:112):
``build_sam2`` returns float32 modules and ``segment`` calls them without autocast (roll/distributed/strategy/seg_strategy.py:47-60).
So does this one: float32 (socioreasoner_amd.sam2, csrc/sam_f32.hip) unless ``model_args.sam2_compute_dtype: bf16`` (or the environment
variable SR_SAM2_DTYPE=bf16) opts into the bf16 kernels -- ~3x the throughput, masks equal to float32's only outside the bf16 noise band.

A hub id (``facebook/sam2-hiera-large`` in the shipped YAML) is looked up in the local HuggingFace cache; a ``model_name_or_path`` that is neither
``synthetic:*``, nor an existing directory, nor in that cache (a typo) RAISES, as the reference's ``build_sam2`` would;
SR_ALLOW_SYNTHETIC_WEIGHTS=1 turns that into a loud fallback to random weights (socioreasoner_amd/checkpoints.py: one policy for both models).
"""
from __future__ import annotations

import glob
import os


def _get(obj, name, default=None):
    if obj is None:
        return default
    return obj.get(name, default) if hasattr(obj, "get") else getattr(obj, name, default)


def sam2_seg_model_provider(model_args=None, training_args=None, is_trainable: bool = False):
    import torch
    from socioreasoner_amd import sam2
    if is_trainable:
        raise NotImplementedError("the MI355X SAM2 path is inference only")
    path = str(_get(model_args, "model_name_or_path", "") or "synthetic:sam2-hiera-large")
    dev = f"cuda:{torch.cuda.current_device()}"
    want = str(os.environ.get("SR_SAM2_DTYPE") or _get(model_args, "sam2_compute_dtype", "") or "float32").lower()
    if want not in ("float32", "fp32", "f32", "bf16", "bfloat16"):
        raise ValueError(f"sam2_compute_dtype {want!r}: float32 (the reference's precision) or bf16")
    dtype = torch.bfloat16 if want in ("bf16", "bfloat16") else torch.float32
    from socioreasoner_amd import checkpoints
    kind, path = checkpoints.resolve(path, "seg_infer (SAM2)", "synthetic:sam2-hiera-large")     # hub ids -> the local HF cache; else raises
    if kind != "dir":
        tiny = path.endswith("tiny")
        g = sam2.Sam2Geometry(image_size=256, embed_dims=(32, 64, 128, 256), heads=(1, 2, 4, 8), blocks=(1, 2, 3, 2), windows=(8, 4, 8, 4),
                              global_blocks=(4,), dec_mlp=256) if tiny else sam2.Sam2Geometry()
        eng = sam2.Sam2Engine(g, dev, dtype=dtype)
        eng.load_state_dict(sam2.synthetic_state_dict(g, seed=0))
        pred = sam2.Sam2Predictor(eng)
        pred.synthetic_weights = kind == "synthetic-fallback"
        return pred
    eng = sam2.Sam2Engine(sam2.Sam2Geometry(), dev, dtype=dtype)
    pt = os.path.join(path, "sam2_hiera_large.pt")
    if os.path.exists(pt):
        ck = torch.load(pt, map_location="cpu", weights_only=True)
        eng.load_state_dict(sam2.rename_sam2_checkpoint(ck.get("model", ck)))
    else:
        from safetensors.torch import load_file
        files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
        if not files:
            raise FileNotFoundError(f"no sam2_hiera_large.pt or *.safetensors under {path}")
        sd = {}
        for f in files:
            sd.update(load_file(f))
        eng.load_state_dict(sd)
    return sam2.Sam2Predictor(eng)
