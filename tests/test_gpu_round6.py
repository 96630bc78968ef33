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


# ------------------------------------------------------------------------------------------------ configs[3] AS NAMED in the development layout
def test_configs3_eight_ranks_of_32_rows_equal_the_single_process_run():
    """BASELINE.json configs[3]: 8 x MI355X data-parallel, batch = 256 tiles -- 8 ranks x 32 rows.  No 8-GPU node is available to the builder, so the named
    workload runs in the development layout: `python bench.py --gpus 8 --batch 32 --waves 1` with SR_DIST_BACKEND=gloo (8 ranks = 8 engines of 10.7 GB, 8
    schedulers and 8 poll loops sharing this box's one device; host-staged exchange; on an 8-GPU node the SAME command runs one rank per GPU over RCCL and
    must print exchange.verified: true).  The line says n_gpus 8 / nranks 8 / dp8 / 256 result rows, and every tile's result row (128 greedy tokens + 2 IoU
    counts) equals, tile by tile, the row of ONE process that serves the same 256 tiles through 32 rows (8 waves, overlapped admission): sharding is
    np.array_split order, and a tile's tokens do not depend on which rank, wave or admission group served it.
    DP contract: /root/reference/roll/distributed/scheduler/decorator.py:106-181 (dispatch_dp_mp_compute), protocol.py:550-617 (chunk / concat)."""
    import json
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--batch", "32", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-latency", "--no-sam", "--no-pmc"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SR_DIST_BACKEND")}
    r8 = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--waves", "1"] + common, cwd=ROOT, env=dict(env, SR_DIST_BACKEND="gloo"), capture_output=True,
                        text=True, timeout=2400)
    assert r8.returncode == 0, r8.stdout[-2000:] + r8.stderr[-3000:]
    l8 = json.loads([ln for ln in r8.stdout.splitlines() if ln.startswith("{")][-1])
    assert l8["n_gpus"] == 8 and l8["config"]["exchange"]["nranks"] == 8 and l8["config"]["parallelism"] == "dp8" and l8["config"]["tiles_per_gpu_per_step"] == 32
    r1 = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--waves", "8"] + common, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-2000:]
    l1 = json.loads([ln for ln in r1.stdout.splitlines() if ln.startswith("{")][-1])
    assert len(l8["result_row_checksums"]) == 256 and l8["result_row_checksums"] == l1["result_row_checksums"]
    sc = l8["phase_ms_per_step"].get("scheduler") or {}
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump({"workload": "BASELINE.json configs[3] in the development layout: 8 gloo ranks x 32 rows on ONE MI355X (8 engines), 256 tiles, 1 step",
               "tiles_per_s_all_ranks_one_device": l8["value"], "ms_per_step": l8["ms_per_step"], "host_threads_per_rank": l8["host_threads_per_rank"],
               "host_ms_per_round_rank0": sc.get("host_ms_per_round"), "poll_wait_ms_per_round_rank0": sc.get("poll_wait_ms_per_round"),
               "rows_equal_single_process": True, "single_process_tiles_per_s": l1["value"]}, open(os.path.join(out, "r06_configs3_gloo.json"), "w"), indent=1)
