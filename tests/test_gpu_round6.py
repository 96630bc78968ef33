"""Round 6 GPU tests (through the C ABI): the counted ring loops of EVERY decode GEMV launch (LDS-staged <= 4-row launches and row-major x included)
give the bits of the conditional-refill loops of rounds 1-4."""
from dataclasses import replace

import numpy as np
import pytest
import torch

from tests.util import switch

pytestmark = pytest.mark.gpu


def _prompts(rng, n, lo=5, hi=40, vocab=2000):
    ids = [rng.integers(0, vocab, int(rng.integers(lo, hi))).astype(np.int64) for _ in range(n)]
    pos = [np.tile(np.arange(len(x)), (3, 1)).astype(np.int64) for x in ids]
    return ids, pos


@pytest.mark.parametrize("size,fp8", [("tiny", False), ("tiny", True), ("3b", False), ("3b", True)])
def test_counted_ring_loops_equal_the_conditional_refill_loops(monkeypatch, size, fp8):
    """SR_GEMV_COUNTED=1 (default): every fill of the GEMVs' weight / x ring is unconditional (chunk index clamped to the wave's last chunk, the last
    round peeled), so hipcc emits counted vmcnt waits instead of vmcnt(0) at the top of every round.  Round 5 did this for 5..32 rows with
    fragment-ordered x; round 6 for the LDS-staged launches of <= 4 rows (RMSNorm prologue, x from LDS) and for row-major x (the down-projection at
    <= 4 rows).  Same chunks, same MFMA order: every logit of every decode step equals SR_GEMV_COUNTED=0 at 1 / 2 / 3 / 4 / 5 / 16 / 17 / 32 rows, eager and
    graph-replayed -- tiny geometry (K = 512: one ring, no steady round) and the 3B geometry 4 layers deep (K = 2048 / 11008: 2 - 4 rounds per wave)."""
    from socioreasoner_amd.config import geometry_3b, geometry_tiny
    from socioreasoner_amd.engine import Engine
    if size == "tiny":
        geom, vocab, rows = geometry_tiny(), 2000, (1, 2, 3, 4, 5, 16, 17, 32)
    else:
        geom = geometry_3b()
        geom = replace(geom, text=replace(geom.text, num_hidden_layers=4), vision=replace(geom.vision, depth=2, fullatt_block_indexes=(1,)))
        vocab, rows = 150000, (1, 3, 4, 17, 32)
    rng = np.random.default_rng(66)
    ids, pos = _prompts(rng, 32, vocab=vocab)
    out = {}
    for flag in ("0", "1"):
        switch(monkeypatch, "SR_GEMV_COUNTED", flag)
        e = Engine(geom, max_patches=64, max_prefill_tokens=64 * 32, max_batch=32, max_ctx=128, max_new_tokens=16, lm_fp8=fp8)
        e.load_synthetic_weights(seed=0)
        res = []
        for B in rows:
            e.prefill(ids[:B], pos[:B])
            toks, tr = e.decode(10, trace=True, use_graph=False)
            e.prefill(ids[:B], pos[:B])
            toks_g, tr_g = e.decode(10, trace=True, use_graph=True)
            assert torch.equal(toks, toks_g) and torch.equal(tr, tr_g), (flag, B)
            assert bool(torch.isfinite(tr).all())
            res.append((toks.clone(), tr.clone()))
        out[flag] = res
        e.close()
    for B, (t0, r0), (t1, r1) in zip(rows, out["0"], out["1"]):
        assert torch.equal(t0, t1), B
        assert torch.equal(r0, r1), (B, float((r0 - r1).abs().max()))
