"""The synthetic workload of the metric: inputs resident in HBM and ONE step of the hot path over one batch of tiles per GPU.

  uint8 tile -> patchify -> ViT (1024 patches) -> merger (256 tokens) -> LM prefill (448-token prompt) -> greedy decode of exactly 128 tokens
  (EOS ignored) -> raster tail (union of 4 756^2 masks -> nearest 768^2 -> IoU counts).

Nothing is skipped inside a step; the per-tile result row (128 tokens + 2 IoU counts) is what the ranks exchange.
"""
from __future__ import annotations

import time

import numpy as np
import torch

from .common import N_NEW, RAGGED_HI


class Workload:
    def __init__(self, args, rank, world, local):
        from socioreasoner_amd import hostops, synthetic
        from socioreasoner_amd.config import geometry_3b
        from socioreasoner_amd.engine import Engine
        self.args, self.rank, self.world = args, rank, world
        self.dev = torch.device(f"cuda:{local}")
        torch.cuda.set_device(self.dev)
        B = self.B = args.batch
        self.continuous = ((B > 1 and not args.static and not args.gather_logits) or args.continuous) and not (args.gather_logits or args.no_graph)
        self.overlap = self.continuous and not args.no_overlap
        self.geom = geom = geometry_3b()
        self.grid = (1, args.tile // 14, args.tile // 14)
        self.n_patch = self.grid[1] * self.grid[2]
        self.n_img = 2 if args.pair else 1
        self.s_prompt = 96 + 94 + self.n_img * (2 + self.n_patch // 4)
        self.quotes_mfma = args.tile == 448 and not args.pair           # (the algorithmic constants are the 448-tile counts)
        self.eng = Engine(geom, max_patches=self.n_patch * self.n_img * B, max_prefill_tokens=self.s_prompt * B, max_batch=B,
                          max_ctx=max(640, (self.s_prompt + RAGGED_HI + 63) // 64 * 64), max_new_tokens=RAGGED_HI, device=str(self.dev),
                          lm_fp8="mx" if args.fp8_mx else args.fp8,
                          kv_slots=2 * B if self.overlap else 0)      # spare KV slots: the next requests are prefilled while the current rows decode
        t0 = time.time()
        self.eng.load_synthetic_weights(seed=0)
        self.load_s = time.time() - t0
        # ---- synthetic inputs, resident in HBM.  Request k of a step is tile (rank * n_req + k)
        self.n_req = n_req = args.waves * B if self.continuous else B
        tiles = [rank * n_req + i for i in range(n_req)]
        dev = self.dev
        self.imgs = [[torch.from_numpy(synthetic.tile_pixels(self.n_img * i + j, args.tile, args.tile)).to(dev) for j in range(self.n_img)] for i in tiles]
        self.ids = [synthetic.tile_prompt(geom, i, self.grid, n_images=self.n_img) for i in tiles]
        self.pos3 = []
        for x in self.ids:
            p, _ = hostops.get_rope_index(torch.from_numpy(x)[None], [self.grid] * self.n_img, None, image_token_id=geom.image_token_id,
                                          vision_start_token_id=geom.vision_start_token_id)
            self.pos3.append(p[:, 0].numpy())
        mk = [synthetic.tile_masks(i) for i in tiles]
        self.masks = torch.from_numpy(np.stack([m for m, _ in mk], axis=1)).to(dev).contiguous()      # [4, n_req, 756, 756]
        self.gts = torch.from_numpy(np.stack([g for _, g in mk], axis=0)).to(dev).contiguous()         # [n_req, 768, 768]
        self.phase_ms = {"vit": 0.0, "prefill": 0.0, "decode": 0.0, "raster": 0.0}
        self.shares = []         # CU share (of 8 per shader engine) of every overlapped admission
        self.sched = {}          # continuous mode: requests admitted / staged under decode, decode steps alone / sharing the chip

    @staticmethod
    def ev():
        return torch.cuda.Event(enable_timing=True)

    def admit_share(self):
        return self.args.admit_cus if self.args.admit_cus == "auto" else float(self.args.admit_cus)

    def raster_tail(self, lo, n):
        """union of the 4 object masks of tiles lo..lo+n-1 -> nearest 756 -> 768 -> IoU counts vs the ground truth.  The tiles are stacked along
        the row axis: 756 / 768 = 63 / 64 is exact in binary, so the nearest-row rule of the stack equals the per-tile rule
        (floor(b * 756 + y * 63 / 64) = b * 756 + floor(y * 63 / 64)) and one launch serves all tiles."""
        from socioreasoner_amd import raster
        acc = torch.zeros(n * 756, 756, dtype=torch.uint8, device=self.dev)
        for j in range(4):
            raster.mask_union_(acc, self.masks[j, lo:lo + n].reshape(n * 756, 756))
        up = raster.resize_nearest(acc, n * 768, 768).reshape(n, 768, 768)
        return raster.iou_counts_batched(up, self.gts[lo:lo + n])          # one launch for the n tiles

    def steps_continuous(self, k_steps, phase_ms=None):
        """k_steps steps of waves x B requests each through B rows, served as ONE request stream: the later requests are admitted as rows free up
        (EOS is ignored by the metric, so all rows of a wave finish together; the point is the measured cost of the request-level path), step
        k + 1's first group is staged under step k's last rows like any other group; a step's raster tail and result exchange run when its last
        request completes.  --drain: one scheduler per step on an idle engine (rounds 1-3)."""
        from socioreasoner_amd import dp
        from socioreasoner_amd.serving import ContinuousBatcher, Request
        a, n_req, dev = self.args, self.n_req, self.dev
        res = None
        groups = [[s_] for s_ in range(k_steps)] if a.drain else [list(range(k_steps))]
        for grp in groups:
            cb = ContinuousBatcher(self.eng, eos=[], pad_id=0, steps_per_poll=a.poll, time_phases=phase_ms is not None, overlap=self.overlap,
                                   admit_cus_per_se=self.admit_share())
            reqs = [Request(ids=self.ids[k], pos3=self.pos3[k], max_new=N_NEW, images=self.imgs[k], grids=[self.grid] * self.n_img, tag=(s_, k))
                    for s_ in grp for k in range(n_req)]
            toks = {s_: [None] * n_req for s_ in grp}
            left = {s_: n_req for s_ in grp}
            spans, out = [], {}

            def finished(req, t):
                s_, k = req.tag
                toks[s_][k] = t
                left[s_] -= 1
                if left[s_] == 0:       # the step's last request: its raster tail + the one exchange of the step (every rank gets here in step order)
                    e0, e1 = self.ev(), self.ev()
                    e0.record()
                    counts = self.raster_tail(0, n_req)
                    e1.record()
                    spans.append((e0, e1))
                    r_ = torch.cat([torch.tensor(toks[s_], dtype=torch.int64, device=dev), counts], dim=1)
                    out[s_] = dp.all_gather_rows(r_, n_req * self.world) if self.world > 1 else r_
            cb.run_stream(reqs, finished)
            res = out[grp[-1]]
            if phase_ms is not None:
                for k, v in cb.phase_ms().items():
                    phase_ms[k] = phase_ms.get(k, 0.0) + v
                for k in ("admitted", "staged_shared", "steps", "steps_shared", "rounds", "host_ms", "poll_wait_ms"):
                    self.sched[k] = self.sched.get(k, 0) + cb.stats[k]
                self.shares.extend(cb.stats["shares"])
                self.sched["share_model"] = cb.stats.get("share_model")
                phase_ms["raster"] += sum(a_.elapsed_time(b_) for a_, b_ in spans)
        return res

    def step_static(self, nb, phase_ms=None, first=0, gather=True):
        """One static batch of nb tiles: patchify + ViT | prefill | 127 decode steps | raster tail, each bracketed by events."""
        from socioreasoner_amd import dp
        a, eng = self.args, self.eng
        e0, e1, e2, e3, e4 = (self.ev() for _ in range(5))
        e0.record()
        pix = torch.cat([eng.patchify(im) for grp in self.imgs[first:first + nb] for im in grp], dim=0)
        emb = eng.vit_forward(pix, [self.grid] * (nb * self.n_img))
        e1.record()
        first_logits = eng.prefill(self.ids[first:first + nb], self.pos3[first:first + nb], emb, return_logits=a.gather_logits)
        e2.record()
        if a.gather_logits:
            alltoks, bad = dp.decode_with_logits_gather(lambda t: eng.decode_step(t), first_logits, N_NEW, nb * self.world)
            assert bad == 0, f"{bad} on-device argmax results differ from the argmax of the gathered logits"
            toks = alltoks[self.rank * nb:(self.rank + 1) * nb]
        else:
            toks = eng.decode(N_NEW, use_graph=not a.no_graph)
        e3.record()
        counts = self.raster_tail(first, nb)
        e4.record()
        res = torch.cat([toks.to(torch.int64), counts], dim=1)             # [nb, 130] per-tile result row
        if self.world > 1 and nb == self.B and gather:
            res = dp.all_gather_rows(res, nb * self.world)                 # the one RCCL exchange of the step
        if phase_ms is not None:
            torch.cuda.synchronize(self.dev)
            for k, x, y in (("vit", e0, e1), ("prefill", e1, e2), ("decode", e2, e3), ("raster", e3, e4)):
                phase_ms[k] += x.elapsed_time(y)
        return res

    def run_steps(self, k_steps, rec=False):
        if self.continuous:
            return self.steps_continuous(k_steps, self.phase_ms if rec else None)
        r_ = None
        for _ in range(k_steps):
            r_ = self.step_static(self.B, self.phase_ms if rec else None)
        return r_
