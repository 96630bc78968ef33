"""Token choice for sampling callers of ``sr_decode_step`` (host-side policy on device-resident float32 logits).

The reference hands ``temperature / top_p / top_k / repetition_penalty`` to vLLM's sampler
(roll/distributed/strategy/vllm_strategy.py:289-309).  BASELINE.json's configurations are greedy, so this is not on the
measured hot path; it exists so that a YAML with the reference's sampling defaults produces samples instead of silently
decoding greedily.  Order of operations follows the vLLM sampler the reference configures: repetition penalty over
prompt + generated tokens, temperature, top-k, top-p, softmax, one categorical draw.  vLLM's random stream cannot be
reproduced, so parity for this path is distributional (see tests): only temperature == 0 / top_k == 1 is bit-pinned
(it must equal the greedy path).
"""
from __future__ import annotations

import torch


def is_greedy(gc: dict) -> bool:
    t = float(gc.get("temperature", 0) or 0)
    k = gc.get("top_k", -1)
    return t <= 1e-5 or (k is not None and int(k) == 1)


def sample(logits: torch.Tensor, temperature: float = 1.0, top_k: int = -1, top_p: float = 1.0,
           repetition_penalty: float = 1.0, seen: torch.Tensor | None = None,
           generator: torch.Generator | None = None) -> torch.Tensor:
    """logits float32 [B, V] (device) -> int64 [B].  ``seen`` bool [B, V]: tokens already in prompt or output."""
    x = logits.float().clone()
    if repetition_penalty and repetition_penalty != 1.0 and seen is not None:
        pen = torch.where(x > 0, x / repetition_penalty, x * repetition_penalty)
        x = torch.where(seen, pen, x)
    if temperature <= 1e-5:
        return x.argmax(dim=-1)
    x = x / temperature
    V = x.shape[-1]
    if top_k is not None and 0 < int(top_k) < V:
        kth = torch.topk(x, int(top_k), dim=-1).values[:, -1:]
        x = x.masked_fill(x < kth, float("-inf"))
    if top_p is not None and float(top_p) < 1.0:
        sx, si = torch.sort(x, dim=-1, descending=False)
        cp = torch.softmax(sx, dim=-1).cumsum(dim=-1)
        drop = cp <= (1.0 - float(top_p))
        drop[:, -1] = False                       # always keep the most likely token
        sx = sx.masked_fill(drop, float("-inf"))
        x = torch.empty_like(x).scatter_(-1, si, sx)
    p = torch.softmax(x, dim=-1)
    return torch.multinomial(p, 1, generator=generator)[:, 0]
