"""CPU restatement of the Qwen2.5-VL forward pass the reference executes.  TEST INFRASTRUCTURE ONLY.

The reference's ``actor_infer`` role runs this graph through third-party engines
(/root/reference/roll/distributed/strategy/vllm_strategy.py:79,127 and the HF-eager shape at
/root/reference/roll/distributed/strategy/hf_strategy.py:49-94).  The arithmetic itself is in the
un-vendored dependency ``transformers`` (``hf:`` = transformers/models/qwen2_5_vl/modeling_qwen2_5_vl.py,
version 5.15.0 in the build container), whose *bf16 eager* semantics are restated here:

  * all tensors are float32 arrays that hold bf16-representable values; ``r()`` marks every point
    where HF's bf16 eager path rounds (each op output), everything between two ``r()`` is float32;
  * deviation (documented in DESIGN.md): the LM-head returns un-rounded float32 logits
    (HF returns bf16 logits, hf:1386-1387), so that the 1e-3 logit tolerance is meaningful.

Pinned by tests/test_oracle_golden.py against fixtures generated from the real HF modules
(tools/make_golden.py; full 3B depth: tools/make_golden_full.py -> tests/golden/hf_full3b.npz).  Sequences are processed un-padded (the reference left-pads to 4096 and masks,
/root/reference/roll/datasets/collator.py:444-564; padding never changes un-masked results).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- config
@dataclass
class VisionCfg:
    depth: int = 32
    hidden_size: int = 1280
    num_heads: int = 16
    intermediate_size: int = 3420
    patch_size: int = 14
    temporal_patch_size: int = 2
    spatial_merge_size: int = 2
    window_size: int = 112
    fullatt_block_indexes: tuple = (7, 15, 23, 31)
    out_hidden_size: int = 2048
    in_channels: int = 3

    @property
    def head_dim(self):
        return self.hidden_size // self.num_heads


@dataclass
class TextCfg:
    num_hidden_layers: int = 36
    hidden_size: int = 2048
    num_attention_heads: int = 16
    num_key_value_heads: int = 2
    head_dim: int = 128
    intermediate_size: int = 11008
    vocab_size: int = 151936
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1000000.0
    mrope_section: tuple = (16, 24, 24)


@dataclass
class RefConfig:
    vision: VisionCfg = field(default_factory=VisionCfg)
    text: TextCfg = field(default_factory=TextCfg)
    image_token_id: int = 151655
    video_token_id: int = 151656
    vision_start_token_id: int = 151652
    vision_end_token_id: int = 151653


def config_3b() -> RefConfig:
    """SocioReasoner-3B = Qwen2.5-VL-3B geometry."""
    return RefConfig()


def config_tiny() -> RefConfig:
    """Small geometry with the true head dims (ViT 80, LM 128) for fast parity tests."""
    return RefConfig(
        vision=VisionCfg(depth=4, hidden_size=320, num_heads=4, intermediate_size=220,
                         fullatt_block_indexes=(1, 3), out_hidden_size=512),
        text=TextCfg(num_hidden_layers=3, hidden_size=512, num_attention_heads=4, num_key_value_heads=1,
                     intermediate_size=1000, vocab_size=2048),
        image_token_id=2040, video_token_id=2041, vision_start_token_id=2042, vision_end_token_id=2043,
    )


# ----------------------------------------------------------------------------- rounding helpers
def r(x: torch.Tensor) -> torch.Tensor:
    """Round to bf16 (nearest even) and return as float32 -- one HF bf16 op boundary."""
    return x.to(torch.bfloat16).to(torch.float32)


class QuantW:
    """fp8 (OCP e4m3fn) per-output-channel quantised Linear weight -- BASELINE.json configs[4], the repo's definition
    (include/socior.h sr_config.lm_weight_dtype): scale[n] = amax_n / 448 (1 for an all-zero row), q = fp8(W / scale),
    y = bf16((x . q^T) * scale + bias).  No reference implementation exists for this mode (the reference ships bf16)."""

    def __init__(self, w: torch.Tensor):
        amax = w.abs().amax(dim=1, keepdim=True)
        scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
        self.q8 = (w / scale).to(torch.float8_e4m3fn)
        self.q = self.q8.float()
        self.scale = scale[:, 0].contiguous()


def mx_quantize(x: torch.Tensor):
    """OCP Microscaling (MX) v1.0 quantisation of activations with e4m3 elements, the repo's definition of the fp8 PREFILL path's
    activation side (BASELINE.json configs[4] "CDNA4 fp8 MFMA"; the block-scaled K = 128 MFMA is the fp8 form that runs at twice
    the bf16 rate): along the last axis in blocks of 32, shared scale X = 2^(floor(log2(max|v|)) - 8) stored as e8m0 (exponent
    clamped to -127..127, an all-zero block takes -127), elements = e4m3(v / X), round to nearest even, saturating at +-448.
    -> (dequantised values x_q = element * X as float32 (exact), exponents int32 [..., K/32])."""
    K0 = x.shape[-1]
    if K0 % 32:             # the engine pads every K to a multiple of 64 with zeros; zeros change neither a block's maximum nor its sum
        x = F.pad(x, (0, 32 - K0 % 32))
    K = x.shape[-1]
    xb = x.reshape(*x.shape[:-1], K // 32, 32).float()
    amax = xb.abs().amax(-1)
    _, ex = torch.frexp(amax)                                   # amax = m * 2^ex, m in [0.5, 1): floor(log2) = ex - 1
    e = torch.where(amax > 0, ex - 1 - 8, torch.full_like(ex, -127)).clamp(-127, 127)
    scale = torch.ldexp(torch.ones_like(amax), e)
    el = (xb / scale[..., None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()
    return (el * scale[..., None]).reshape(x.shape)[..., :K0], e.to(torch.int32)


def linear(x, w, b=None):
    """bf16 nn.Linear: float32 accumulate, bias added in float32, one rounding (hf:88-96, 218, 626-629)."""
    if isinstance(w, QuantW):
        if getattr(w, "mx_act", False):                         # fp8 x fp8: MX-quantised activations against the fp8 weights
            x = mx_quantize(x)[0]
        y = (x @ w.q.t()) * w.scale
    else:
        y = x @ w.t()
    if b is not None:
        y = y + b
    return r(y)


class Fp8LmWeights:
    """View of a weight dict in which the LM decoder linears (q/k/v/o, gate/up/down) are fp8-quantised."""
    import re as _re
    _LM = _re.compile(r"model\.layers\.\d+\.(self_attn\.[qkvo]_proj|mlp\.(gate|up|down)_proj)\.weight$")

    def __init__(self, base, mx_act: bool = False):
        """mx_act: the linears quantise their INPUT too (mx_quantize) -- the engine's lm_weight_dtype = 2 does that in the prefill
        GEMMs (fp8 x fp8 block-scaled MFMA) and not in decode; toggle with set_mx_act around the calls."""
        self.base, self._cache, self.mx_act = base, {}, mx_act

    def set_mx_act(self, on: bool):
        self.mx_act = on
        for q in self._cache.values():
            q.mx_act = on

    def __getitem__(self, name):
        if self._LM.match(name):
            if name not in self._cache:
                self._cache[name] = QuantW(self.base[name])
                self._cache[name].mx_act = self.mx_act
            return self._cache[name]
        return self.base[name]


def rmsnorm(x, w, eps):
    """hf:65-79: float32 variance; ``.to(input_dtype)`` rounds before the (bf16) weight multiply."""
    var = x.pow(2).mean(-1, keepdim=True)
    xh = r(x * torch.rsqrt(var + eps))
    return r(w * xh)


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def silu_bf16(x):
    return r(x * torch.sigmoid(x))


def gelu_bf16(x):
    return r(F.gelu(x))  # exact erf form, nn.GELU() default (hf:142-147)


# ----------------------------------------------------------------------------- ViT host-side index math
def vision_position_ids(grid_thw, merge):
    """(h, w) position of every patch in merge-block order (hf: vision_utils.get_vision_position_ids)."""
    out = []
    for t, h, w in grid_thw:
        hp = torch.arange(h).unsqueeze(1).expand(-1, w)
        hp = hp.reshape(h // merge, merge, w // merge, merge).permute(0, 2, 1, 3).flatten()
        wp = torch.arange(w).unsqueeze(0).expand(h, -1)
        wp = wp.reshape(h // merge, merge, w // merge, merge).permute(0, 2, 1, 3).flatten()
        out.append(torch.stack([hp, wp], dim=-1).repeat(t, 1))
    return torch.cat(out, dim=0)


def vision_window_index(grid_thw, merge, window_size, patch_size):
    """hf: vision_utils.py:130-188.  Returns (window_index over merged units, cu_window_seqlens over patches)."""
    window_index, cu = [], [0]
    base = 0
    ws = window_size // merge // patch_size
    unit = merge * merge
    for t, h, w in grid_thw:
        lh, lw = h // merge, w // merge
        index = torch.arange(t * lh * lw).reshape(t, lh, lw)
        pad_h = ws - lh % ws
        pad_w = ws - lw % ws
        nh, nw = (lh + pad_h) // ws, (lw + pad_w) // ws
        ip = F.pad(index, (0, pad_w, 0, pad_h), "constant", -100)
        ip = ip.reshape(t, nh, ws, nw, ws).permute(0, 1, 3, 2, 4).reshape(t, nh * nw, ws, ws)
        seqlens = (ip != -100).sum([2, 3]).reshape(-1)
        ip = ip.reshape(-1)
        window_index.append(ip[ip != -100] + base)
        cu.extend((seqlens.cumsum(0) * unit + cu[-1]).tolist())
        base += t * lh * lw
    window_index = torch.cat(window_index)
    cu = torch.unique_consecutive(torch.tensor(cu, dtype=torch.int64))
    return window_index, cu


def vision_full_seqlens(grid_thw):
    """hf: one attention sequence per temporal frame (get_vision_attention_seqlens)."""
    cu = [0]
    for t, h, w in grid_thw:
        for _ in range(t):
            cu.append(cu[-1] + h * w)
    return torch.tensor(cu, dtype=torch.int64)


# ----------------------------------------------------------------------------- ViT
def vit_rotary_tables(cfg: VisionCfg, grid_thw, window_index):
    """hf:125-134 + 443-449: float32 cos/sin [N, head_dim] in window order."""
    merge = cfg.spatial_merge_size
    unit = merge * merge
    dim = cfg.head_dim // 2
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float) / dim))
    pos = vision_position_ids(grid_thw, merge)
    rp = (pos.unsqueeze(-1) * inv_freq).flatten(1)  # [N, dim]
    n = rp.shape[0]
    rp = rp.reshape(n // unit, unit, -1)[window_index].reshape(n, -1)
    emb = torch.cat((rp, rp), dim=-1)
    return emb.cos(), emb.sin()


def vit_attention(W, p, cfg: VisionCfg, x, cu, cos, sin):
    """hf:211-291 eager branch (per-chunk attention, softmax in float32, bf16 matmul outputs)."""
    n = x.shape[0]
    H, D = cfg.num_heads, cfg.head_dim
    qkv = linear(x, W[p + "attn.qkv.weight"], W[p + "attn.qkv.bias"]).reshape(n, 3, H, D)
    q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]
    c, s = cos.unsqueeze(-2), sin.unsqueeze(-2)
    q = r(q * c + rotate_half(q) * s)  # hf:160-171 float32 math, one rounding
    k = r(k * c + rotate_half(k) * s)
    scaling = torch.tensor(D ** -0.5, dtype=torch.float32)
    out = torch.empty(n, H, D)
    cu = [int(c_) for c_ in cu]
    for a, b in zip(cu[:-1], cu[1:]):
        qh = q[a:b].transpose(0, 1)  # [H, L, D]
        kh = k[a:b].transpose(0, 1)
        vh = v[a:b].transpose(0, 1)
        sc = r(qh @ kh.transpose(1, 2))
        sc = r(sc * scaling)
        pr = r(torch.softmax(sc, dim=-1))
        out[a:b] = r(pr @ vh).transpose(0, 1)
    return linear(out.reshape(n, H * D), W[p + "attn.proj.weight"], W[p + "attn.proj.bias"])


def vit_mlp(W, p, x):
    g = linear(x, W[p + "mlp.gate_proj.weight"], W[p + "mlp.gate_proj.bias"])
    u = linear(x, W[p + "mlp.up_proj.weight"], W[p + "mlp.up_proj.bias"])
    return linear(r(silu_bf16(g) * u), W[p + "mlp.down_proj.weight"], W[p + "mlp.down_proj.bias"])


def vit_block(W, i, cfg: VisionCfg, x, cu, cos, sin):
    """hf:294-322."""
    p = f"visual.blocks.{i}."
    x = r(x + vit_attention(W, p, cfg, rmsnorm(x, W[p + "norm1.weight"], 1e-6), cu, cos, sin))
    x = r(x + vit_mlp(W, p, rmsnorm(x, W[p + "norm2.weight"], 1e-6)))
    return x


def vit_merger(W, cfg: VisionCfg, x):
    """hf:137-150."""
    mh = cfg.hidden_size * cfg.spatial_merge_size ** 2
    y = rmsnorm(x, W["visual.merger.ln_q.weight"], 1e-6).reshape(-1, mh)
    y = gelu_bf16(linear(y, W["visual.merger.mlp.0.weight"], W["visual.merger.mlp.0.bias"]))
    return linear(y, W["visual.merger.mlp.2.weight"], W["visual.merger.mlp.2.bias"])


def vit_forward(W, cfg: RefConfig, pixel_values: torch.Tensor, grid_thw, return_hidden=False):
    """hf:408-474.  pixel_values float32 [N, C*T*p*p]; returns [N/merge^2, out_hidden] (bf16 values)."""
    vc = cfg.vision
    grid = [tuple(int(v) for v in g) for g in grid_thw]
    unit = vc.spatial_merge_size ** 2
    widx, cu_win = vision_window_index(grid, vc.spatial_merge_size, vc.window_size, vc.patch_size)
    cu_full = vision_full_seqlens(grid)
    x = linear(r(pixel_values.float()), W["visual.patch_embed.proj.weight"])  # hf:99-122 (Conv3d k=s, no bias)
    n = x.shape[0]
    x = x.reshape(n // unit, unit, -1)[widx].reshape(n, -1)
    cos, sin = vit_rotary_tables(vc, grid, widx)
    for i in range(vc.depth):
        cu = cu_full if i in vc.fullatt_block_indexes else cu_win
        x = vit_block(W, i, vc, x, cu, cos, sin)
    y = vit_merger(W, vc, x)
    y = y[torch.argsort(widx)]
    return (y, x) if return_hidden else y


# ----------------------------------------------------------------------------- LM
def mrope_tables(tc: TextCfg, pos3: torch.Tensor):
    """hf:486-539 + 557-599.  pos3 int64 [3, S] -> bf16-rounded cos/sin [S, head_dim] with the
    mrope_section interleave applied (channel chunk i takes axis i % 3)."""
    D = tc.head_dim
    inv_freq = 1.0 / (tc.rope_theta ** (torch.arange(0, D, 2, dtype=torch.float) / D))
    freqs = pos3.float().unsqueeze(-1) * inv_freq  # [3, S, D/2]  (== inv_freq @ pos, exact products)
    emb = torch.cat((freqs, freqs), dim=-1)
    cos, sin = r(emb.cos()), r(emb.sin())
    sec = list(tc.mrope_section) * 2
    cos = torch.cat([m[i % 3] for i, m in enumerate(cos.split(sec, dim=-1))], dim=-1)
    sin = torch.cat([m[i % 3] for i, m in enumerate(sin.split(sec, dim=-1))], dim=-1)
    return cos, sin


def lm_attention(W, p, tc: TextCfg, x, cos, sin, cache):
    """hf:602-689 eager, causal, GQA via repeat_kv.  x [S_new, hidden]; cache = dict(k,v) [S_past, KVH, D]."""
    s_new = x.shape[0]
    H, KVH, D = tc.num_attention_heads, tc.num_key_value_heads, tc.head_dim
    q = linear(x, W[p + "self_attn.q_proj.weight"], W[p + "self_attn.q_proj.bias"]).reshape(s_new, H, D)
    k = linear(x, W[p + "self_attn.k_proj.weight"], W[p + "self_attn.k_proj.bias"]).reshape(s_new, KVH, D)
    v = linear(x, W[p + "self_attn.v_proj.weight"], W[p + "self_attn.v_proj.bias"]).reshape(s_new, KVH, D)
    c, s = cos.unsqueeze(-2), sin.unsqueeze(-2)
    q = r(r(q * c) + r(rotate_half(q) * s))  # bf16 elementwise ops round individually
    k = r(r(k * c) + r(rotate_half(k) * s))
    if cache.get("k") is not None:
        k = torch.cat([cache["k"], k], dim=0)
        v = torch.cat([cache["v"], v], dim=0)
    cache["k"], cache["v"] = k, v
    s_tot = k.shape[0]
    g = H // KVH
    kh = k.transpose(0, 1).repeat_interleave(g, dim=0)  # [H, S_tot, D]
    vh = v.transpose(0, 1).repeat_interleave(g, dim=0)
    qh = q.transpose(0, 1)
    scaling = torch.tensor(D ** -0.5, dtype=torch.float32)
    sc = r(qh @ kh.transpose(1, 2))
    sc = r(sc * scaling)
    past = s_tot - s_new
    iq = torch.arange(s_new).unsqueeze(1) + past
    ik = torch.arange(s_tot).unsqueeze(0)
    sc = sc.masked_fill(ik > iq, float("-inf"))
    pr = r(torch.softmax(sc, dim=-1))
    o = r(pr @ vh).transpose(0, 1).reshape(s_new, H * D)
    return linear(o, W[p + "self_attn.o_proj.weight"])


def lm_mlp(W, p, x):
    g = linear(x, W[p + "mlp.gate_proj.weight"])
    u = linear(x, W[p + "mlp.up_proj.weight"])
    return linear(r(silu_bf16(g) * u), W[p + "mlp.down_proj.weight"])


def lm_layer(W, i, tc: TextCfg, x, cos, sin, cache):
    """hf:692-758."""
    p = f"model.layers.{i}."
    x = r(x + lm_attention(W, p, tc, rmsnorm(x, W[p + "input_layernorm.weight"], tc.rms_norm_eps), cos, sin, cache))
    x = r(x + lm_mlp(W, p, rmsnorm(x, W[p + "post_attention_layernorm.weight"], tc.rms_norm_eps)))
    return x


def embed_with_images(W, cfg: RefConfig, input_ids: torch.Tensor, image_embeds):
    """hf:1210-1216 masked_scatter; in-repo restatement mcore_adapter/.../modeling_qwen2_5_vl.py:294-315."""
    x = W["model.embed_tokens.weight"][input_ids].clone()
    if image_embeds is not None:
        mask = input_ids == cfg.image_token_id
        assert int(mask.sum()) == image_embeds.shape[0], "image tokens and image features do not match"
        x[mask] = image_embeds
    return x


def lm_forward(W, cfg: RefConfig, x, pos3, caches, all_logits=False):
    """Runs all layers on x [S_new, hidden]; returns float32 logits of the last (or every) position."""
    tc = cfg.text
    cos, sin = mrope_tables(tc, pos3)
    for i in range(tc.num_hidden_layers):
        x = lm_layer(W, i, tc, x, cos, sin, caches[i])
    h = rmsnorm(x if all_logits else x[-1:], W["model.norm.weight"], tc.rms_norm_eps)
    return h @ W["lm_head.weight"].t()  # float32, NOT rounded (documented deviation)


def new_caches(cfg: RefConfig):
    return [dict() for _ in range(cfg.text.num_hidden_layers)]


def greedy_argmax(logits: torch.Tensor) -> int:
    """Lowest index among maxima (torch.argmax semantics on CPU)."""
    m = logits.max()
    return int((logits == m).nonzero()[0, 0])


def generate_greedy(W, cfg: RefConfig, input_ids, pos3, image_embeds, max_new_tokens, eos_ids=(), force_tokens=None):
    """Prefill + greedy decode of one un-padded sequence.

    Decode positions follow the reference rule /root/reference/roll/utils/functionals.py:816-818
    (all three axes = last prompt position + 1, +2, ...).  ``force_tokens`` teacher-forces the fed-back
    token (used for per-step logit parity).  Returns (tokens, per-step float32 logits list).
    """
    caches = new_caches(cfg)
    x = embed_with_images(W, cfg, input_ids, image_embeds)
    logits = lm_forward(W, cfg, x, pos3, caches)[0]
    toks, all_logits = [], []
    nxt_pos = int(pos3.max()) + 1
    for step in range(max_new_tokens):
        all_logits.append(logits)
        tok = greedy_argmax(logits)
        toks.append(tok)
        if tok in eos_ids:
            break
        if step == max_new_tokens - 1:
            break
        feed = tok if force_tokens is None else int(force_tokens[step])
        x = W["model.embed_tokens.weight"][torch.tensor([feed])]
        p3 = torch.full((3, 1), nxt_pos, dtype=torch.int64)
        nxt_pos += 1
        logits = lm_forward(W, cfg, x, p3, caches)[0]
    return toks, all_logits
