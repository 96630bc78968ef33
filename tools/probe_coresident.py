"""Can decode kernels CO-RESIDE with the admission's GEMM blocks on the same CUs (no CU masks)?  The 256-tile GEMM takes a whole CU
(128 KB LDS, 444 of 512 registers per SIMD lane), so decode blocks queue behind it; the 128-tile kernel (64 KB LDS, 72 registers, 2 blocks
per CU) leaves room for GEMV waves.  Run once per SR_GEMM256 setting (the switch is read once per process):
  SR_GEMM256=0 python tools/probe_coresident.py      (128-tile GEMMs)        python tools/probe_coresident.py     (default: 256-tile)
Decode on a HIGH-priority stream, admission on a low-priority one."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socioreasoner_amd import hostops, synthetic
from socioreasoner_amd.config import geometry_3b
from socioreasoner_amd.engine import Engine


def main():
    geom = geometry_3b()
    B, G = 32, 64
    grid = (1, 32, 32)
    dec = Engine(geom, max_patches=1024 * B, max_prefill_tokens=448 * B, max_batch=B, max_ctx=640, max_new_tokens=G)
    dec.load_synthetic_weights(seed=0)
    adm = Engine(geom, max_patches=1024 * B, max_prefill_tokens=448 * B, max_batch=B, max_ctx=640, max_new_tokens=G)
    adm.load_synthetic_weights(seed=0)
    imgs = [torch.from_numpy(synthetic.tile_pixels(i)).cuda() for i in range(B)]
    ids = [synthetic.tile_prompt(geom, i, grid) for i in range(B)]
    pos = [hostops.get_rope_index(torch.from_numpy(x)[None], [grid], None)[0][:, 0].numpy() for x in ids]

    def forward(e):
        pix = torch.cat([e.patchify(im) for im in imgs], dim=0)
        emb = e.vit_forward(pix, [grid] * B)
        e.prefill(ids, pos, emb)
    forward(dec); forward(adm)
    torch.cuda.synchronize()
    ev = lambda: torch.cuda.Event(enable_timing=True)
    hi, lo = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)

    def run(tag, do_dec, do_adm, n_adm=1):
        a0, a1, d0, d1 = ev(), ev(), ev(), ev()
        with torch.cuda.stream(hi):
            forward(dec)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if do_adm:
            with torch.cuda.stream(lo):
                a0.record()
                for _ in range(n_adm):
                    forward(adm)
                a1.record()
        if do_dec:
            with torch.cuda.stream(hi):
                d0.record()
                dec.decode(G)
                d1.record()
        torch.cuda.synchronize()
        out = {"gemm256": os.environ.get("SR_GEMM256", "default"), "config": tag, "wall_ms": round((time.perf_counter() - t0) * 1e3, 1)}
        if do_dec:
            out["decode_ms_per_step"] = round(d0.elapsed_time(d1) / (G - 1), 3)
        if do_adm:
            out["admit_ms_per_32_tiles"] = round(a0.elapsed_time(a1) / n_adm, 1)
        print(json.dumps(out), flush=True)
    for _ in range(2):
        run("decode alone", True, False)
        run("admission alone", False, True)
        run("both, no masks, decode high priority", True, True, 2)
    dec.close(); adm.close()


if __name__ == "__main__":
    main()
