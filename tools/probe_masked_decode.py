"""Why is a decode step 25 % slower while an admission is staged next to it?  Fewer CUs, or the company?  The same 63 decode steps of a
32-row engine (a) on an ordinary stream with the chip to itself, (b) ALONE on the CU-masked decode stream the scheduler uses (5 of the 8 CUs of
every shader engine), (c) on that stream next to an admission on the other 3 CUs, (d) alone on a 4-of-8 and a 6-of-8 stream, and (e) on the
ordinary (unmasked) stream next to a masked admission."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socioreasoner_amd import hostops, synthetic, streams
from socioreasoner_amd.config import geometry_3b
from socioreasoner_amd.engine import Engine


def main():
    geom = geometry_3b()
    B, G = 32, 64
    grid = (1, 32, 32)
    mk = lambda: Engine(geom, max_patches=1024 * B, max_prefill_tokens=448 * B, max_batch=B, max_ctx=640, max_new_tokens=G)
    dec, adm = mk(), mk()
    dec.load_synthetic_weights(seed=0); adm.load_synthetic_weights(seed=0)
    imgs = [torch.from_numpy(synthetic.tile_pixels(i)).cuda() for i in range(B)]
    ids = [synthetic.tile_prompt(geom, i, grid) for i in range(B)]
    pos = [hostops.get_rope_index(torch.from_numpy(x)[None], [grid], None)[0][:, 0].numpy() for x in ids]

    def forward(e):
        pix = torch.cat([e.patchify(im) for im in imgs], dim=0)
        emb = e.vit_forward(pix, [grid] * B)
        e.prefill(ids, pos, emb)
    plain = torch.cuda.Stream()
    with torch.cuda.stream(plain):
        forward(dec); forward(adm); dec.decode(G)
    torch.cuda.synchronize()
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def run(tag, dstream, astream=None, n_adm=2):
        a0, a1, d0, d1 = ev(), ev(), ev(), ev()
        with torch.cuda.stream(plain):
            forward(dec)
        torch.cuda.synchronize()
        if astream is not None:
            with torch.cuda.stream(astream):
                a0.record()
                for _ in range(n_adm):
                    forward(adm)
                a1.record()
        with torch.cuda.stream(dstream):
            d0.record()
            dec.decode(G)
            d1.record()
        torch.cuda.synchronize()
        out = {"config": tag, "decode_ms_per_step": round(d0.elapsed_time(d1) / (G - 1), 3)}
        if astream is not None:
            out["admit_ms_per_32_tiles"] = round(a0.elapsed_time(a1) / n_adm, 1)
        print(json.dumps(out), flush=True)
    s3 = streams.overlap_streams("cuda:0", 3)
    s4 = streams.overlap_streams("cuda:0", 4)
    s2 = streams.overlap_streams("cuda:0", 2)
    full_mask = streams.masked_stream("cuda:0", 0, 8)          # round 6: a stream created WITH a CU mask that enables every CU -- is there a cost of the masked queue itself?
    m7 = streams.masked_stream("cuda:0", 0, 7)
    for _ in range(2):
        run("(a) ordinary stream, alone", plain)
        run("(a') masked stream with ALL 8/8 CUs enabled, alone", full_mask)
        run("(a'') masked stream 7/8 CUs, alone", m7)
        run("(b) masked decode stream 5/8 CUs, alone", s3.decode)
        run("(d) masked decode stream 4/8 CUs, alone", s4.decode)
        run("(d) masked decode stream 6/8 CUs, alone", s2.decode)
        run("(b') masked stream with 3/8 CUs, alone", s3.admit)
        run("(c) masked decode 5/8 next to admission 3/8", s3.decode, s3.admit)
        run("(e) ordinary decode stream next to admission 3/8", plain, s3.admit)
    dec.close(); adm.close()


if __name__ == "__main__":
    main()
