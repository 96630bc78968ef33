"""Synthetic-weight generator (oracle side, numpy).  TEST INFRASTRUCTURE ONLY.

Real SocioReasoner-3B / Qwen2.5-VL-3B weights are not available offline
(SURVEY.md section 8(C)), so parity and benchmarks run on weights produced by a
counter-based integer hash of (seed, HF parameter name, linear element index).
The product has an independent device implementation of the same definition
(``socioreasoner_amd/csrc/elementwise.hip: k_synth_fill``); because the recipe is
integer arithmetic followed by ONE float32 multiply(+add) and a round-to-nearest
-even to bf16, both sides are bit-identical by construction and a test checks it.

Definition (all uint32 arithmetic wraps):
    key  = mix32(fnv1a32(name) ^ mix32(seed + 0x9E3779B9))
    h    = mix32(idx * 0x9E3779B1 + key)
    s    = b0 + b1 + b2 + b3 - 510          (bytes of h; Irwin-Hall(4), var 21845)
    val  = bf16_rne(float32(base) + float32(s) * float32(SCALE))
with SCALE = 0.02 / sqrt(21845) stored as the float32 literal below.  Linear
weights / biases / embeddings use base 0; norm weights use base 1.

Parameter names follow the reference-era HF checkpoint layout
(/root/reference/mcore_adapter/src/mcore_adapter/models/converter/template.py:845-899).
"""
from __future__ import annotations

import numpy as np
import torch

SCALE_F32 = np.float32(1.3531647e-4)  # 0.02 / sqrt(21845)


def fnv1a32(name: str) -> int:
    h = 2166136261
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 16777619) & 0xFFFFFFFF
    return h


def mix32_int(x: int) -> int:
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def tensor_key(name: str, seed: int) -> int:
    return mix32_int(fnv1a32(name) ^ mix32_int((seed + 0x9E3779B9) & 0xFFFFFFFF))


def _mix32_np(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint32, copy=True)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def _mix32_t(x: torch.Tensor) -> torch.Tensor:
    """mix32 on int64 tensors holding uint32 values (multiplications wrap mod 2^64; the low 32 bits are what counts)."""
    M = 0xFFFFFFFF
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & M
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & M
    return x ^ (x >> 16)


def synth_f32_torch(name: str, shape, seed: int = 0, base: float = 0.0, start: int = 0) -> torch.Tensor:
    """Same definition as synth_f32, evaluated with multi-threaded torch integer ops (the numpy path is single-threaded and
    needs minutes for the 3.75 G parameters of the full model); tests/test_oracle_golden.py checks the two bit for bit."""
    n = int(np.prod(shape))
    key = tensor_key(name, seed)
    out = torch.empty(n, dtype=torch.float32)
    CH = 1 << 24
    for lo in range(0, n, CH):
        hi = min(n, lo + CH)
        idx = torch.arange(start + lo, start + hi, dtype=torch.int64) & 0xFFFFFFFF
        h = _mix32_t((idx * 0x9E3779B1 + key) & 0xFFFFFFFF)
        s_ = (h & 255) + ((h >> 8) & 255) + ((h >> 16) & 255) + (h >> 24) - 510
        out[lo:hi] = np.float32(base) + s_.to(torch.float32) * float(SCALE_F32)
    return out.to(torch.bfloat16).to(torch.float32).reshape(tuple(shape))


def synth_f32(name: str, shape, seed: int = 0, base: float = 0.0, start: int = 0) -> np.ndarray:
    """float32 array holding bf16-representable values."""
    n = int(np.prod(shape))
    key = np.uint32(tensor_key(name, seed))
    out = np.empty(n, dtype=np.float32)
    CH = 1 << 24
    with np.errstate(over="ignore"):
        for lo in range(0, n, CH):
            hi = min(n, lo + CH)
            idx = np.arange(start + lo, start + hi, dtype=np.uint64).astype(np.uint32)
            h = _mix32_np(idx * np.uint32(0x9E3779B1) + key)
            s = ((h & np.uint32(255)) + ((h >> np.uint32(8)) & np.uint32(255))
                 + ((h >> np.uint32(16)) & np.uint32(255)) + (h >> np.uint32(24))).astype(np.int32) - 510
            v = np.float32(base) + s.astype(np.float32) * SCALE_F32
            out[lo:hi] = v
    t = torch.from_numpy(out).to(torch.bfloat16).to(torch.float32)  # RNE
    return t.numpy().reshape(shape)


def param_specs(cfg) -> list[tuple[str, tuple, float]]:
    """(hf_name, shape, base) for every parameter of a RefConfig, reference-era names."""
    v, t = cfg.vision, cfg.text
    P: list[tuple[str, tuple, float]] = []
    pd = v.in_channels * v.temporal_patch_size * v.patch_size * v.patch_size
    P.append(("visual.patch_embed.proj.weight", (v.hidden_size, pd), 0.0))
    for i in range(v.depth):
        p = f"visual.blocks.{i}."
        P += [
            (p + "norm1.weight", (v.hidden_size,), 1.0),
            (p + "norm2.weight", (v.hidden_size,), 1.0),
            (p + "attn.qkv.weight", (3 * v.hidden_size, v.hidden_size), 0.0),
            (p + "attn.qkv.bias", (3 * v.hidden_size,), 0.0),
            (p + "attn.proj.weight", (v.hidden_size, v.hidden_size), 0.0),
            (p + "attn.proj.bias", (v.hidden_size,), 0.0),
            (p + "mlp.gate_proj.weight", (v.intermediate_size, v.hidden_size), 0.0),
            (p + "mlp.gate_proj.bias", (v.intermediate_size,), 0.0),
            (p + "mlp.up_proj.weight", (v.intermediate_size, v.hidden_size), 0.0),
            (p + "mlp.up_proj.bias", (v.intermediate_size,), 0.0),
            (p + "mlp.down_proj.weight", (v.hidden_size, v.intermediate_size), 0.0),
            (p + "mlp.down_proj.bias", (v.hidden_size,), 0.0),
        ]
    mh = v.hidden_size * v.spatial_merge_size ** 2
    P += [
        ("visual.merger.ln_q.weight", (v.hidden_size,), 1.0),
        ("visual.merger.mlp.0.weight", (mh, mh), 0.0),
        ("visual.merger.mlp.0.bias", (mh,), 0.0),
        ("visual.merger.mlp.2.weight", (v.out_hidden_size, mh), 0.0),
        ("visual.merger.mlp.2.bias", (v.out_hidden_size,), 0.0),
    ]
    P.append(("model.embed_tokens.weight", (t.vocab_size, t.hidden_size), 0.0))
    kvd = t.num_key_value_heads * t.head_dim
    qd = t.num_attention_heads * t.head_dim
    for i in range(t.num_hidden_layers):
        p = f"model.layers.{i}."
        P += [
            (p + "input_layernorm.weight", (t.hidden_size,), 1.0),
            (p + "self_attn.q_proj.weight", (qd, t.hidden_size), 0.0),
            (p + "self_attn.q_proj.bias", (qd,), 0.0),
            (p + "self_attn.k_proj.weight", (kvd, t.hidden_size), 0.0),
            (p + "self_attn.k_proj.bias", (kvd,), 0.0),
            (p + "self_attn.v_proj.weight", (kvd, t.hidden_size), 0.0),
            (p + "self_attn.v_proj.bias", (kvd,), 0.0),
            (p + "self_attn.o_proj.weight", (t.hidden_size, qd), 0.0),
            (p + "post_attention_layernorm.weight", (t.hidden_size,), 1.0),
            (p + "mlp.gate_proj.weight", (t.intermediate_size, t.hidden_size), 0.0),
            (p + "mlp.up_proj.weight", (t.intermediate_size, t.hidden_size), 0.0),
            (p + "mlp.down_proj.weight", (t.hidden_size, t.intermediate_size), 0.0),
        ]
    P.append(("model.norm.weight", (t.hidden_size,), 1.0))
    return P


class LazyWeights(dict):
    """name -> float32 torch tensor (bf16-representable), generated on first use."""

    def __init__(self, cfg, seed: int = 0, cache: bool = True, fast: bool = False, source=None):
        """source: optional callable (name, shape, base) -> float32 tensor that materialises a parameter some faster way
        (bench.py hands in the device generator, which tests/test_gpu_parity.py::test_synth_fill_bit_exact pins to synth_f32)."""
        super().__init__()
        self.cfg, self.seed, self.cache, self.fast, self.source = cfg, seed, cache, fast, source
        self.specs = {n: (s, b) for n, s, b in param_specs(cfg)}

    def __missing__(self, name):
        if name == "lm_head.weight":  # tied (tie_word_embeddings for the 3B geometry)
            return self["model.embed_tokens.weight"]
        shape, base = self.specs[name]
        if self.source is not None:
            t = self.source(name, shape, base)
        else:
            t = synth_f32_torch(name, shape, self.seed, base) if self.fast else torch.from_numpy(synth_f32(name, shape, self.seed, base))
        if self.cache:
            self[name] = t
        return t

    def __contains__(self, name):
        return name in self.specs or name == "lm_head.weight"
