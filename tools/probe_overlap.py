"""Can the admission of the next requests (ViT + prefill: MFMA-bound) run UNDER the decode steps of the running rows (HBM- / latency-
bound) on a CU-masked side stream?  Two engines (decode: 32 prefilled rows; admit: ViT + prefill of 32 tiles), measured alone and
together for several CU splits.  Prints one JSON line per configuration."""
import ctypes as C, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socioreasoner_amd import hostops, synthetic
from socioreasoner_amd.config import geometry_3b
from socioreasoner_amd.engine import Engine

hip = C.CDLL("libamdhip64.so")


def masked_stream(cu_lo, cu_hi):
    """stream whose kernels may only use CUs cu_lo .. cu_hi-1 of EVERY shader engine.  The runtime's mask bit i is CU (i / 32) of
    shader engine (i % 32) -- 32 SEs x 8 CUs on MI355X (first probe: setting the low k bits of every word confined the stream to k
    shader engines, i.e. k / 4 XCDs) -- so a full 32-bit word = one CU index across the whole chip."""
    words = (C.c_uint32 * 8)()
    for g in range(8):
        words[g] = 0xFFFFFFFF if cu_lo <= g < cu_hi else 0
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


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

    forward(dec)
    forward(adm)
    torch.cuda.synchronize()
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def run(tag, side, do_dec, do_adm, n_adm=2, main_s=None):
        main_s = main_s or dec_stream
        a0, a1, d0, d1 = ev(), ev(), ev(), ev()
        with torch.cuda.stream(main_s):
            forward(dec)                       # fresh rows for the decode
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if do_adm:
            with torch.cuda.stream(side):
                a0.record()
                for _ in range(n_adm):
                    forward(adm)
                a1.record()
        if do_dec:
            with torch.cuda.stream(main_s):    # NOT the null stream: masked streams are blocking streams and would serialise with it
                d0.record()
                dec.decode(G)
                d1.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        out = {"config": tag, "wall_ms": round(wall, 1)}
        if do_dec:
            out["decode_ms_per_step"] = round(d0.elapsed_time(d1) / (G - 1), 3)
        if do_adm:
            out["admit_ms_per_32_tiles"] = round(a0.elapsed_time(a1) / n_adm, 1)
        print(json.dumps(out), flush=True)

    full = torch.cuda.Stream()
    dec_stream = torch.cuda.Stream()
    run("decode alone", full, True, False)
    run("admission alone, unmasked side stream", full, False, True)
    run("both, unmasked side stream", full, True, True)
    for j in (2, 3, 4):
        side = masked_stream(0, j)
        rest = masked_stream(j, 8)
        run(f"admission alone on {32 * j} CUs ({j} of 8 per SE)", side, False, True, 1)
        run(f"decode alone on {32 * (8 - j)} CUs", side, True, False, 1, main_s=rest)
        run(f"both: admission on {32 * j} CUs, decode unmasked", side, True, True, 1)
        run(f"both: admission on {32 * j} CUs, decode on the other {32 * (8 - j)}", side, True, True, 1, main_s=rest)
    dec.close(); adm.close()


if __name__ == "__main__":
    main()
