"""GPU probe: SAM2 device path against the oracle, stage by stage (debugging aid for tests/test_gpu_sam2.py) + timings."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sam2_ref as S
from socioreasoner_amd import sam2, synthetic
tag = sys.argv[1] if len(sys.argv) > 1 else "tiny"
og = S.geometry_tiny() if tag == "tiny" else S.geometry_large()
g = sam2.Sam2Geometry(**{k: getattr(og, k) for k in sam2.Sam2Geometry.__dataclass_fields__})
W = S.synthetic_weights(og)
e = sam2.Sam2Engine(g, dtype=__import__("torch").bfloat16)
e.load_state_dict(W)
hw = 189 if tag == "tiny" else 756
img = synthetic.tile_pixels(7, hw, hw)
o = S.Sam2Oracle(W, og)
t0 = time.time(); o.set_image(img); print("oracle set_image %.1fs" % (time.time() - t0))
e.set_image(torch.from_numpy(img).cuda()); torch.cuda.synchronize()
def rep(name, mine, ref):
    d = (mine.float().cpu() - ref.float()).flatten()
    print(f"{name:12s} ref absmax {float(ref.abs().max()):8.3f}  err max {float(d.abs().max()):8.4f} rms {float(d.pow(2).mean().sqrt()):8.5f}")
for i, (x, ws) in enumerate(e.stage_out):
    G, C = e.grid[i], og.embed_dims[i]
    perm = torch.from_numpy(sam2.window_order(G, ws).astype(np.int64)).cuda()
    rep(f"stage{i}", x[perm][:, :C].reshape(G, G, C), o.stages[i])
for n, x, r in (("f0", e.f0, o.feats[0]), ("f1", e.f1, o.feats[1]), ("emb", e.emb, o.feats[2])):
    rep(n, x[:, : r.shape[-1]].reshape(r.shape), r)
masks, iou, low = o.predict([[60, 70]] if tag == "tiny" else [[250, 300]], [1], [30, 40, 120, 150] if tag == "tiny" else [100, 120, 500, 600])
lg, sc, lw = e.predict([[60, 70]] if tag == "tiny" else [[250, 300]], [1], [30, 40, 120, 150] if tag == "tiny" else [100, 120, 500, 600], return_logits=True)
rep("low", torch.from_numpy(lw), torch.from_numpy(low)); print("iou", sc, iou)
print("mask pixels differing", int(((lg > 0) != masks).sum()), "of", masks.size)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.time(); e.set_image(torch.from_numpy(img).cuda()); torch.cuda.synchronize(); t1 = time.time()
    acc = torch.zeros(hw, hw, dtype=torch.uint8, device="cuda")
    for k in range(4): e.predict_or(acc, None, None, [100 + 10 * k, 120, 500, 600] if tag != "tiny" else [30 + k, 40, 120, 150])
    torch.cuda.synchronize(); t2 = time.time()
    print("set_image %.2f ms, 4 objects %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
