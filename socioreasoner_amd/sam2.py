"""SAM2 (Hiera-L image encoder + prompt encoder + mask decoder) behind the reference's ``seg_infer`` role, on the MI355X kernels of
``libsocior.so`` (SURVEY.md "next" row N1).

Reference contract (/root/reference/roll/distributed/strategy/seg_strategy.py:26-72, roll/models/model_providers.py:515-562): the
provider returns a ``SAM2ImagePredictor`` over ``facebook/sam2-hiera-large``; ``segment`` resizes the sample to 756 x 756, calls
``set_image`` once and ``predict(point_coords, point_labels, box)`` per object (three masks + scores), ORs ``masks[argmax(scores)]``.
The network is the one of ``transformers.models.sam2.modeling_sam2`` ("hf:" below; same parameters as the sam2 package's checkpoint,
HF names) -- the oracle restates it (oracle/sam2_ref.py) and tests/test_gpu_sam2.py compares this module with HF's own outputs.

How it runs here.  Host code is Python (like the reference's); every FLOP is in ``libsocior.so``: all Linear / 1 x 1 / patch /
transposed convolutions are bf16 MFMA GEMMs (``sr_op_gemm``: bias, residual, GELU and destination row maps in the epilogue), all
attention goes through ``sr_op_attention`` (windowed, pooled-query and global blocks of Hiera; token <-> image attention of the
decoder), the passes in between are the coalesced kernels of csrc/sam.hip.  Layout decisions:

* tokens are channel-last bf16 matrices [rows][ld], ld = channels rounded up to whole 64-wide k-tiles, pad columns kept zero;
* head_dim 72 runs as 80: the qkv weight rows / proj weight columns of a head are padded with zeros once at load time;
* a stage's tokens live in WINDOW ORDER (window, y, x) from the patch embedding on (the GEMM's row map writes them there), so every
  window is a contiguous run of rows for the attention kernel and no partition / un-partition pass exists; a stage-entry block pools
  its queries 2 x 2 inside the windows, which leaves the next stage in window order of half the size -- for Hiera-L that IS the next
  stage's window except between stage 2 and 3 (one row gather); global-attention blocks do not care; the FPN's lateral GEMMs write
  image order through their row map;
* the prompt encoder (a dozen 256-vectors) runs on the host in float32.

Two storage modes.  ``dtype=torch.float32`` (the DEFAULT behind seg_infer) is the reference's own precision -- it builds the predictor
in float32 and calls it without autocast (model_providers.py:540-548; the YAML's ``dtype: bf16`` is never applied there): float32 token
matrices, the f32-input MFMA GEMM and the float32 attention of csrc/sam_f32.hip, sam.hip's passes instantiated for float; tests hold its
756 x 756 masks to HF-float32's EXACTLY (outside |logit| < 1e-3).  ``dtype=torch.bfloat16`` is the fast mode (bf16 storage / float32
accumulation like the LM path, ~3x the throughput): held to the distance HF-bf16 itself has from HF-float32 and to exact masks wherever
|logit| clears that band."""
from __future__ import annotations

import ctypes as C
import math
import gc
import os
from dataclasses import dataclass
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import lib as L

EPI_STORE, EPI_RESID, EPI_GELU, EPI_F32 = 0, 1, 3, 4
GELU_FAST = 0x800          # sr_op_gemm: the GELU epilogue through gelu_fast_f (csrc/common.h)
RELU = 0x1000              # sr_op_gemm with EPI_GELU: ReLU as the activation of the epilogue
_FEWQ = os.environ.get("SR_SAM_FEWQ", "1") != "0"
_WORK = np.dtype([("q_row0", "<i4"), ("seq_len", "<i4"), ("q_off", "<i4"), ("k_row0", "<i4"), ("vt_off", "<i8"), ("q_len", "<i4"), ("pad", "<i4")])
_HD_OK = (16, 32, 80, 128)


@dataclass
class Sam2Geometry:
    """sam2_hiera_l.yaml of the sam2 package (the checkpoint the reference loads)."""
    image_size: int = 1024
    embed_dims: Tuple[int, ...] = (144, 288, 576, 1152)
    heads: Tuple[int, ...] = (2, 4, 8, 16)
    blocks: Tuple[int, ...] = (2, 6, 36, 4)
    windows: Tuple[int, ...] = (8, 4, 16, 8)
    global_blocks: Tuple[int, ...] = (23, 33, 43)
    bkg_size: int = 7
    fpn_dim: int = 256
    top_down_levels: Tuple[int, ...] = (2, 3)
    dec_heads: int = 8
    dec_mlp: int = 2048
    dec_layers: int = 2
    n_mask_tokens: int = 4
    ln_eps: float = 1e-6

    def block_table(self):
        """(stage, dim_in, dim_out, heads, window (0 = global), pooled) per block (hf:457-501)."""
        out, k = [], 0
        for s, n in enumerate(self.blocks):
            for b in range(n):
                first = s > 0 and b == 0
                win = 0 if k in self.global_blocks else (self.windows[s - 1] if first else self.windows[s])
                out.append((s, self.embed_dims[s - 1] if first else self.embed_dims[s], self.embed_dims[s], self.heads[s], win, first))
                k += 1
        return out


def rup(x: int, m: int = 64) -> int:
    return (x + m - 1) // m * m


def _hd_pad(hd: int) -> int:
    for v in _HD_OK:
        if v >= hd:
            return v
    raise ValueError(f"head dim {hd} not supported")


def window_order(G: int, ws: int) -> np.ndarray:
    """perm[image index y * G + x] = row of that token when tokens are stored window by window (hf:397-425)."""
    y, x = np.divmod(np.arange(G * G), G)
    return (((y // ws) * (G // ws) + x // ws) * ws * ws + (y % ws) * ws + x % ws).astype(np.int32)


class Sam2Engine:
    def __init__(self, geometry: Optional[Sam2Geometry] = None, device="cuda:0", dtype=torch.float32):
        self.lib = L.load()
        if not torch.cuda.is_available():
            raise L.SocioRError("no GPU visible: the product path has no CPU fallback")
        self.g = geometry or Sam2Geometry()
        self.device = torch.device(device)
        if dtype not in (torch.bfloat16, torch.float32):
            raise ValueError("Sam2Engine dtype must be torch.float32 (the reference's precision) or torch.bfloat16")
        self.dt, self.f32 = dtype, dtype == torch.float32
        self._fn = lambda name: getattr(self.lib, name + ("_f32" if self.f32 else ""))
        g = self.g
        if g.image_size % 64:
            raise ValueError("image_size must be a multiple of 64")
        self.grid = [g.image_size // 4 // (1 << s) for s in range(4)]          # tokens per side of every stage
        for s in range(4):
            if self.grid[s] % g.windows[s] or (s and self.grid[s - 1] % g.windows[s - 1]) or g.windows[s] % 2:
                raise ValueError("window sizes must divide the stage grids (true for Hiera-L at 1024)")
        self.W: Dict[str, torch.Tensor] = {}
        self._bufs: Dict[str, torch.Tensor] = {}
        self._work: Dict[tuple, tuple] = {}
        self._idx: Dict[str, torch.Tensor] = {}
        self.image_set = False
        self._cur = None
        self._graphs: Dict[int, tuple] = {}
        self.graph_decode = os.environ.get("SR_SAM_GRAPH", "1") != "0"

    # ------------------------------------------------------------------ plumbing
    def _s(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _p(t, off_elems: int = 0):
        return C.c_void_p(t.data_ptr() + off_elems * t.element_size()) if t is not None else None

    def _ck(self, rc, what):
        if rc != 0:
            raise L.SocioRError(f"{what} failed with {rc}")

    def buf(self, name, rows, cols, dtype=None):
        """zero-initialised once; pad columns are never written afterwards"""
        dtype = dtype or self.dt
        t = self._bufs.get(name)
        if t is None or t.shape != (rows, cols) or t.dtype != dtype:
            t = self._bufs[name] = torch.zeros(rows, cols, dtype=dtype, device=self.device)
        return t

    def gemm(self, A, lda, Wn, M, out, ldo, epi=EPI_STORE, resid=None, rowmap=None, bias=True, a_off=0, out_off=0):
        w = self.W[Wn + ".weight"]
        N, K = w.shape
        b = self.W.get(Wn + ".bias") if bias else None
        self._ck(self._fn("sr_op_gemm")(self._p(A, a_off), lda, self._p(w), M, N, K, self._p(out, out_off), ldo, self._p(b), self._p(resid, out_off if resid is out else 0),
                                        self._p(rowmap), epi, self._s()), f"gemm {Wn}")

    def layernorm(self, x, ldx, name, out, ldo, rows, Cc, eps):
        self._ck(self._fn("sr_op_layernorm")(self._p(x), ldx, self._p(self.W[name + ".weight"]), self._p(self.W[name + ".bias"]), self._p(out), ldo, rows, Cc,
                                          C.c_float(eps), self._s()), "layernorm")

    def ew(self, a, lda, b, ldb, out, ldo, rows, Cc, mode):
        self._ck(self._fn("sr_op_ew")(self._p(a), lda, self._p(b), ldb, self._p(out), ldo, rows, Cc, mode, self._s()), "ew")

    def work(self, key, items) -> tuple:
        w = self._work.get(key)
        if w is None:
            arr = np.zeros(len(items), dtype=_WORK)
            for i, it in enumerate(items):
                arr[i] = it + (0,) * (7 - len(it))
            t = torch.from_numpy(arr.view(np.uint8).copy()).to(self.device)
            w = self._work[key] = (t, len(items))
        return w

    def attention(self, q, q_stride, k, k_off, k_stride, hdp, vt, vt_stride, out, out_stride, work, heads, scale, q_tile=64, v2_ok=0, v=None, v_off=0, v_stride=0):
        """bf16: V arrives transposed (``vt``); float32: V is read row-major where the projection left it (``v`` + ``v_off``, stride ``v_stride``)"""
        wt, n = work
        if self.f32:
            assert q_tile <= 64 and v is not None
            self._ck(self.lib.sr_op_attention_f32(self._p(q), q_stride, self._p(k, k_off), k_stride, self._p(v, v_off), v_stride, self._p(out), out_stride,
                                                  self._p(wt), n, heads, C.c_float(scale), hdp, self._s()), "attention_f32")
            return
        self._ck(self.lib.sr_op_attention(self._p(q), q_stride, self._p(k, k_off), k_stride, hdp, self._p(vt), vt_stride, hdp * vt_stride, self._p(out), out_stride,
                                          self._p(wt), n, heads, 1, C.c_float(scale), 0, hdp, q_tile, v2_ok, self._s()), "attention")

    # ------------------------------------------------------------------ weights (HF names -> device layouts)
    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        """HF ``Sam2Model`` state dict (or the sam2 checkpoint renamed to it).  Re-laid-out once: K padded to 64-multiples with zeros,
        head_dim padded to a kernel-supported width, 1 x 1 / patch / transposed convolutions flattened to GEMM operands."""
        g, dev = self.g, self.device
        self._graphs.clear()            # (captured launches hold the old weights' addresses)
        f = lambda name: sd[name].detach().float().cpu()

        def put(name, w2d, bias=None, n_pad=None):
            N, K = w2d.shape
            Np = n_pad or N
            out = torch.zeros(Np, rup(K), dtype=torch.float32)
            out[:N, :K] = w2d
            self.W[name + ".weight"] = out.to(self.dt).to(dev).contiguous()
            if bias is not None:
                bb = torch.zeros(Np, dtype=torch.float32)
                bb[:N] = bias
                self.W[name + ".bias"] = bb.to(self.dt).to(dev)

        def put_ln(name):
            self.W[name + ".weight"] = f(name + ".weight").to(self.dt).to(dev)
            self.W[name + ".bias"] = f(name + ".bias").to(self.dt).to(dev)

        pe = "vision_encoder.backbone."
        put(pe + "patch_embed.projection", f(pe + "patch_embed.projection.weight").reshape(g.embed_dims[0], -1), f(pe + "patch_embed.projection.bias"))
        # position embedding: bicubic-resized background grid + tiled window embedding (hf:645-651), stored in stage-0 window order.
        # Computed in float32 on the host and rounded once (HF's bf16 mode rounds the interpolation too: a parameter-only constant)
        G0 = self.grid[0]
        pos = torch.nn.functional.interpolate(f(pe + "pos_embed"), size=(G0, G0), mode="bicubic")
        win = f(pe + "pos_embed_window")
        pos = (pos.to(self.dt) + win.to(self.dt).tile(1, 1, G0 // win.shape[2], G0 // win.shape[3]))[0].permute(1, 2, 0).reshape(G0 * G0, -1)
        tab = torch.zeros(G0 * G0, rup(g.embed_dims[0]), dtype=self.dt)
        order0 = torch.from_numpy(window_order(G0, g.windows[0]).astype(np.int64))
        tab[order0, : g.embed_dims[0]] = pos
        self.W["pos_table"] = tab.to(dev)
        self.blocks = []
        for i, (s, din, dout, heads, win_, pooled) in enumerate(g.block_table()):
            p = f"{pe}blocks.{i}"
            hd = dout // heads
            hdp = _hd_pad(hd)
            HP = heads * hdp
            put_ln(p + ".layer_norm1")
            put_ln(p + ".layer_norm2")
            wq = f(p + ".attn.qkv.weight").reshape(3, heads, hd, din)
            bq = f(p + ".attn.qkv.bias").reshape(3, heads, hd)
            wq_p = torch.zeros(3, heads, hdp, din)
            wq_p[:, :, :hd] = wq
            bq_p = torch.zeros(3, heads, hdp)
            bq_p[:, :, :hd] = bq
            put(p + ".attn.qkv", wq_p.reshape(3 * HP, din), bq_p.reshape(-1))
            wp = f(p + ".attn.proj.weight").reshape(dout, heads, hd)
            wp_p = torch.zeros(dout, heads, hdp)
            wp_p[:, :, :hd] = wp
            put(p + ".attn.proj", wp_p.reshape(dout, HP), f(p + ".attn.proj.bias"))
            put(p + ".mlp.proj_in", f(p + ".mlp.proj_in.weight"), f(p + ".mlp.proj_in.bias"))
            put(p + ".mlp.proj_out", f(p + ".mlp.proj_out.weight"), f(p + ".mlp.proj_out.bias"))
            if din != dout:
                put(p + ".proj", f(p + ".proj.weight"), f(p + ".proj.bias"))
            self.blocks.append(dict(name=p, stage=s, din=din, dout=dout, heads=heads, hd=hd, hdp=hdp, HP=HP, win=win_, pooled=pooled))
        for j in range(4):
            n = f"vision_encoder.neck.convs.{j}"
            put(n, f(n + ".weight").reshape(g.fpn_dim, -1), f(n + ".bias"))
        Cd = g.fpn_dim
        self.W["no_mem"] = f("no_memory_embedding").reshape(-1).to(self.dt).to(dev)
        self.W["no_mask"] = f("prompt_encoder.no_mask_embed.weight").reshape(-1).to(self.dt).to(dev)
        put("mask_decoder.conv_s0", f("mask_decoder.conv_s0.weight").reshape(Cd // 8, Cd), f("mask_decoder.conv_s0.bias"))
        put("mask_decoder.conv_s1", f("mask_decoder.conv_s1.weight").reshape(Cd // 4, Cd), f("mask_decoder.conv_s1.bias"))
        md = "mask_decoder.transformer."
        attn_names = [f"{md}layers.{l}.{a}" for l in range(g.dec_layers) for a in ("self_attn", "cross_attn_token_to_image", "cross_attn_image_to_token")]
        for a in attn_names + [md + "final_attn_token_to_image"]:
            for q in ("q_proj", "k_proj", "v_proj", "o_proj"):
                put(f"{a}.{q}", f(f"{a}.{q}.weight"), f(f"{a}.{q}.bias"))
        for l in range(g.dec_layers):
            for k in (1, 2, 3, 4):
                put_ln(f"{md}layers.{l}.layer_norm{k}")
            put(f"{md}layers.{l}.mlp.proj_in", f(f"{md}layers.{l}.mlp.proj_in.weight"), f(f"{md}layers.{l}.mlp.proj_in.bias"))
            put(f"{md}layers.{l}.mlp.proj_out", f(f"{md}layers.{l}.mlp.proj_out.weight"), f(f"{md}layers.{l}.mlp.proj_out.bias"))
        put_ln(md + "layer_norm_final_attn")
        put_ln("mask_decoder.upscale_layer_norm")
        # transposed 2 x 2 / stride 2 convolutions as GEMMs: W[co*4 + dy*2 + dx][ci] (hf:1215-1221)
        for n in ("mask_decoder.upscale_conv1", "mask_decoder.upscale_conv2"):
            w = f(n + ".weight")                                        # [ci, co, 2, 2]
            put(n, w.permute(1, 2, 3, 0).reshape(-1, w.shape[0]), f(n + ".bias").repeat_interleave(4))
        for i in range(g.n_mask_tokens):
            n = f"mask_decoder.output_hypernetworks_mlps.{i}"
            for q in ("proj_in", "layers.0", "proj_out"):
                put(f"{n}.{q}", f(f"{n}.{q}.weight"), f(f"{n}.{q}.bias"))
        n = "mask_decoder.iou_prediction_head"
        put(n + ".proj_in", f(n + ".proj_in.weight"), f(n + ".proj_in.bias"))
        put(n + ".layers.0", f(n + ".layers.0.weight"), f(n + ".layers.0.bias"))
        put(n + ".proj_out", f(n + ".proj_out.weight"), f(n + ".proj_out.bias"), n_pad=16)
        # host-side constants of the prompt encoder and the decoder tokens (float32)
        self.h = {k: f(k) for k in ("prompt_encoder.shared_embedding.positional_embedding", "prompt_encoder.point_embed.weight",
                                    "prompt_encoder.not_a_point_embed.weight", "mask_decoder.obj_score_token.weight", "mask_decoder.iou_token.weight",
                                    "mask_decoder.mask_tokens.weight")}
        self.h_np = {"pe": self.h["prompt_encoder.shared_embedding.positional_embedding"].numpy().astype(np.float32),
                     "point": self.h["prompt_encoder.point_embed.weight"].numpy(), "nap": self.h["prompt_encoder.not_a_point_embed.weight"].numpy(),
                     "out_tokens": torch.cat([self.h["mask_decoder.obj_score_token.weight"], self.h["mask_decoder.iou_token.weight"],
                                              self.h["mask_decoder.mask_tokens.weight"]], dim=0).numpy()}
        m = g.image_size // 16
        ax = (torch.arange(m, dtype=torch.float32) + 0.5) / m
        yy, xx = torch.meshgrid(ax, ax, indexing="ij")
        self.W["image_pe"] = self._fourier(torch.stack([xx, yy], dim=-1).reshape(-1, 2)).to(self.dt).to(dev).contiguous()
        torch.cuda.synchronize(self.device)

    def _fourier(self, coords01) -> torch.Tensor:
        """random-Fourier position encoding (hf:727-749), float32 on the host.  numpy on purpose: a dozen 2-vectors through torch's
        CPU ops cost 18 ms per prompt on a 128-thread host (thread-pool wake-ups), 30 us here."""
        G = self.h_np["pe"]
        c = (2.0 * np.asarray(coords01, dtype=np.float32) - 1.0) @ G
        c = np.float32(2.0 * math.pi) * c
        return torch.from_numpy(np.concatenate([np.sin(c), np.cos(c)], axis=-1).astype(np.float32))

    # ------------------------------------------------------------------ image encoder
    def _index(self, name, arr) -> torch.Tensor:
        t = self._idx.get(name)
        if t is None:
            t = self._idx[name] = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int32)).to(self.device)
        return t

    def _window_work(self, n_win, n_keys, n_q, tile=64):
        items = []
        for w in range(n_win):
            for q0 in range(0, n_q, tile):
                items.append((w * n_q + q0, n_keys, q0, w * n_keys, w * n_keys, n_q if n_q != n_keys else 0))
        return items

    def set_image(self, img_u8: torch.Tensor):
        """uint8 HWC device tensor (756 x 756 in the reference's flow) -> image embedding + high-resolution features kept on the device."""
        self.set_images([img_u8])

    def set_images(self, imgs: Sequence[torch.Tensor]):
        """The image encoder over B images AT ONCE: their tokens are stacked along the row axis (image b = rows [b N, (b + 1) N) of every
        stage, each in window order), so every GEMM / LayerNorm / pooling launch covers all of them -- at one image the GEMMs of stages 3 / 4
        have 4096 / 1024 rows, too few for 256 CUs.  ``select(b)`` then points the decoder at image b's features."""
        g, lib, s = self.g, self.lib, self._s
        B = len(imgs)
        assert B >= 1
        S = g.image_size
        G0, C0 = self.grid[0], g.embed_dims[0]
        N0 = G0 * G0
        tag = f"@{B}"
        chw = self.buf("chw" + tag, B * 3, S * S)
        order0 = self._index("order0", window_order(G0, g.windows[0]))
        kp = rup(3 * 49)
        col = self.buf("im2col" + tag, B * N0, kp)
        self.orig_hws = []
        for b, im in enumerate(imgs):
            assert im.dtype == torch.uint8 and im.is_cuda and im.dim() == 3 and im.shape[2] == 3
            im = im.contiguous()
            self.orig_hws.append((int(im.shape[0]), int(im.shape[1])))
            self._ck(self._fn("sr_op_sam_preprocess")(self._p(im), int(im.shape[0]), int(im.shape[1]), self._p(chw, b * 3 * S * S), S, s()), "preprocess")
            self._ck(self._fn("sr_op_im2col")(self._p(chw, b * 3 * S * S), S, 7, 4, 3, self._p(col, b * N0 * kp), kp, self._p(order0), s()), "im2col")
        pos = self.W.get("pos_table" + tag)
        if pos is None:
            pos = self.W["pos_table" + tag] = self.W["pos_table"].repeat(B, 1).contiguous()
        x = self.buf("x0" + tag, B * N0, rup(C0))
        # x = bf16(pos + bf16(conv + bias)), rows already in window order (hf:664-665)
        self.gemm(col, kp, "vision_encoder.backbone.patch_embed.projection", B * N0, x, rup(C0), EPI_RESID, resid=pos)
        ends = set(np.cumsum(g.blocks) - 1)
        cur_ws = g.windows[0]                 # the window size the current token order is organised by
        stage_out = []
        for i, blk in enumerate(self.blocks):
            s_, din, dout, heads, hdp, HP, win, pooled = blk["stage"], blk["din"], blk["dout"], blk["heads"], blk["hdp"], blk["HP"], blk["win"], blk["pooled"]
            Gin = self.grid[s_ - 1] if pooled else self.grid[s_]
            Ni = Gin * Gin                     # tokens per image
            N = B * Ni
            name = blk["name"]
            scale = blk["hd"] ** -0.5
            sfx = ("p" if pooled else "") + tag
            xn = self.buf(f"xn{s_}{sfx}", N, rup(din))
            self.layernorm(x, rup(din), name + ".layer_norm1", xn, rup(din), N, din, g.ln_eps)
            qkv = self.buf(f"qkv{s_}{sfx}", N, 3 * HP)
            self.gemm(xn, rup(din), name + ".attn.qkv", N, qkv, 3 * HP)
            vts = rup(N) + 64
            vt = None
            vkw = dict(v=qkv, v_off=2 * HP, v_stride=3 * HP)          # float32: V is read where the qkv GEMM left it
            if not self.f32:
                vt = self.buf(f"vt{s_}{sfx}", HP, vts)
                self._ck(lib.sr_op_transpose(self._p(qkv, 2 * HP), 3 * HP, N, HP, self._p(vt), vts, s()), "transpose")
            if pooled:
                assert win > 0 and win == cur_ws, "stage-entry blocks pool inside the previous stage's windows"
                Nq, nw = N // 4, N // (win * win)
                xnew = self.buf(f"x{s_}{tag}", Nq, rup(dout))
                rfull = self.buf(f"rfull{s_}{tag}", N, rup(dout))
                self.gemm(xn, rup(din), name + ".proj", N, rfull, rup(dout))
                self._ck(self._fn("sr_op_maxpool_win")(self._p(rfull), rup(dout), dout, nw, win, self._p(xnew), rup(dout), s()), "maxpool")
                qp = self.buf(f"qp{s_}{tag}", Nq, HP)
                self._ck(self._fn("sr_op_maxpool_win")(self._p(qkv), 3 * HP, HP, nw, win, self._p(qp), HP, s()), "maxpool q")
                att = self.buf(f"att{s_}{tag}", Nq, rup(HP))
                wk = self.work(("pool", s_, B), self._window_work(nw, win * win, win * win // 4))
                self.attention(qp, HP, qkv, HP, 3 * HP, hdp, vt, vts, att, rup(HP), wk, heads, scale, **vkw)
                x, N, Ni, cur_ws = xnew, Nq, Ni // 4, win // 2
                self.gemm(att, rup(HP), name + ".attn.proj", N, x, rup(dout), EPI_RESID, resid=x)
            else:
                att = self.buf(f"att{s_}{tag}", N, rup(HP))
                if win > 0:
                    assert win == cur_ws
                    # windows of >= 128 tokens (stage 3: 16 x 16) go to the 8-wave kernel: 128 queries share each LDS-DMA-staged K / V^T tile
                    v2 = hdp == 80 and (win * win) % 128 == 0 and os.environ.get("SR_SAM_WIN_V2", "1") != "0" and not self.f32
                    tile = 128 if v2 else 64
                    wk = self.work(("win", s_, B, tile), self._window_work(N // (win * win), win * win, win * win, tile=tile))
                    self.attention(qkv, 3 * HP, qkv, HP, 3 * HP, hdp, vt, vts, att, rup(HP), wk, heads, scale, q_tile=tile, v2_ok=1 if v2 else 0, **vkw)
                else:           # global attention inside every image: token order does not matter
                    v2 = hdp == 80 and Ni % 8 == 0 and not self.f32
                    tile = 128 if v2 else 64
                    wk = self.work(("glob", s_, tile, B), [(b * Ni + q0, Ni, q0, b * Ni, b * Ni, 0) for b in range(B) for q0 in range(0, Ni, tile)])
                    self.attention(qkv, 3 * HP, qkv, HP, 3 * HP, hdp, vt, vts, att, rup(HP), wk, heads, scale, q_tile=tile, v2_ok=1 if v2 else 0, **vkw)
                self.gemm(att, rup(HP), name + ".attn.proj", N, x, rup(dout), EPI_RESID, resid=x)
            xn2 = self.buf(f"xn{s_}{tag}", N, rup(dout))
            self.layernorm(x, rup(dout), name + ".layer_norm2", xn2, rup(dout), N, dout, g.ln_eps)
            hid = self.buf(f"hid{s_}{tag}", N, 4 * dout)
            self.gemm(xn2, rup(dout), name + ".mlp.proj_in", N, hid, 4 * dout, EPI_GELU | (0 if self.f32 else GELU_FAST))
            self.gemm(hid, 4 * dout, name + ".mlp.proj_out", N, x, rup(dout), EPI_RESID, resid=x)
            if pooled and cur_ws != g.windows[s_]:
                # the pooled tokens sit in window order of size cur_ws; the stage's own windows are g.windows[s_]: one row gather
                Gs = self.grid[s_]
                a_, b_ = window_order(Gs, cur_ws), window_order(Gs, g.windows[s_])
                src = np.empty(Gs * Gs, dtype=np.int32)
                src[b_] = a_                               # row b_[i] of the new order takes row a_[i] of the old one
                src = np.concatenate([src + k * Gs * Gs for k in range(B)])
                x2 = self.buf(f"x{s_}g{tag}", N, rup(dout))
                # (a row copy: the float32 mode passes its rows as twice as many 2-byte elements)
                self._ck(lib.sr_op_gather_rows(self._p(x), self._p(self._index(f"regather{s_}{tag}", src)), self._p(x2), N, rup(dout) * (2 if self.f32 else 1), s()), "gather")
                x, cur_ws = x2, g.windows[s_]
            if i in ends:
                stage_out.append((x, cur_ws))
        # ---- FPN neck (hf:216-265): lateral 1 x 1 convolutions written in IMAGE order through the row map, one top-down step
        Cd = g.fpn_dim
        lat = []
        for lvl in range(4):
            xs, ws = stage_out[lvl]
            Gs = self.grid[lvl]
            inv1 = np.argsort(window_order(Gs, ws)).astype(np.int32)                                  # row r (window order) -> image index
            inv = self._index(f"to_image{lvl}_{ws}{tag}", np.concatenate([inv1 + k * Gs * Gs for k in range(B)]))
            o = self.buf(f"lat{lvl}{tag}", B * Gs * Gs, Cd)
            self.gemm(xs, rup(g.embed_dims[lvl]), f"vision_encoder.neck.convs.{3 - lvl}", B * Gs * Gs, o, Cd, EPI_STORE, rowmap=inv)
            lat.append(o)
        m2, n3 = self.grid[2] ** 2, self.grid[3] ** 2
        fpn2 = lat[2]
        if 2 in g.top_down_levels:
            fpn2 = self.buf("fpn2" + tag, B * m2, Cd)
            for b in range(B):
                self._ck(self._fn("sr_op_upsample2x_add")(self._p(lat[2], b * m2 * Cd), self._p(lat[3], b * n3 * Cd), self._p(fpn2, b * m2 * Cd), self.grid[2], Cd, Cd, s()), "upsample")
        emb = self.buf("emb" + tag, B * m2, Cd)
        self.ew(fpn2, Cd, self.W["no_mem"], 0, emb, Cd, B * m2, Cd, 1)                      # + no-memory embedding (hf:1499-1500)
        keys0 = self.buf("keys0" + tag, B * m2, Cd)
        self.ew(emb, Cd, self.W["no_mask"], 0, keys0, Cd, B * m2, Cd, 1)                    # + dense "no mask" embedding (hf:1193)
        f0 = self.buf("f0" + tag, B * self.grid[0] ** 2, Cd // 8)
        self.gemm(lat[0], Cd, "mask_decoder.conv_s0", B * self.grid[0] ** 2, f0, Cd // 8)
        f1 = self.buf("f1" + tag, B * self.grid[1] ** 2, Cd // 4)
        self.gemm(lat[1], Cd, "mask_decoder.conv_s1", B * self.grid[1] ** 2, f1, Cd // 4)
        self.stage_out, self.n_images = stage_out, B
        self._feat = (emb, keys0, f0, f1)
        self.select(0)

    def _set_current(self, emb, keys0, f0, f1, hw):
        """the decoder reads the image through FIXED buffers (its launches are replayed from a captured graph): 10 MB of copies per image"""
        cur = self._cur
        if cur is None:
            cur = self._cur = tuple(torch.empty_like(t) for t in (emb, keys0, f0, f1))
        for d, t in zip(cur, (emb, keys0, f0, f1)):
            d.copy_(t)
        self.emb, self.keys0, self.f0, self.f1 = cur
        self.orig_hw = hw
        self.image_set = True

    def select(self, b: int):
        """point the decoder at image b of the last set_images call"""
        emb, keys0, f0, f1 = self._feat
        m2, n0, n1 = self.grid[2] ** 2, self.grid[0] ** 2, self.grid[1] ** 2
        self._set_current(emb[b * m2:(b + 1) * m2], keys0[b * m2:(b + 1) * m2], f0[b * n0:(b + 1) * n0], f1[b * n1:(b + 1) * n1], self.orig_hws[b])

    def features(self) -> dict:
        """the selected image's decoder inputs as an object that outlives the next set_images call (an embedding cache: the reference's
        two stages segment the same satellite image)"""
        return {"emb": self.emb.clone(), "keys0": self.keys0.clone(), "f0": self.f0.clone(), "f1": self.f1.clone(), "orig_hw": self.orig_hw}

    def use_features(self, ft: dict):
        self._set_current(ft["emb"], ft["keys0"], ft["f0"], ft["f1"], ft["orig_hw"])

    # ------------------------------------------------------------------ prompt encoder (host, float32) + mask decoder
    def _tokens(self, coords: np.ndarray, labels: np.ndarray) -> torch.Tensor:
        """output tokens + sparse prompt embeddings (hf:791-813, 1175-1190): coords in the model's input frame, labels 1 / 0 clicks, 2 / 3 box
        corners; the encoder's padding point is appended."""
        g = self.g
        pts = np.concatenate([np.asarray(coords, dtype=np.float32) + 0.5, np.zeros((1, 2), np.float32)], axis=0)
        lab = np.concatenate([np.asarray(labels, dtype=np.int64), np.array([-1])])
        e = self._fourier(pts / np.float32(g.image_size)).numpy()
        e = np.where(lab[:, None] == -1, self.h_np["nap"], e)
        e = e + self.h_np["point"][np.clip(lab, 0, None)] * (lab >= 0)[:, None].astype(np.float32)
        return torch.from_numpy(np.concatenate([self.h_np["out_tokens"], e.astype(np.float32)], axis=0))

    TOK = 16        # row stride of one object's tokens in the decoder's token matrices (prompts of up to 16 tokens share one captured launch sequence)

    def _mha(self, name, q, k, v, q_side, k_side, Ts, internal, out, resid, tag):
        """Sam2Attention (hf:874-942) for every object of the batch: out = [resid +] o_proj(attention(q_proj(q), k_proj(k), v_proj(v))).
        ``q_side`` / ``k_side``: "tok" (object o = rows [o TOK, o TOK + Ts[o])) or "img" (rows [o m2, (o + 1) m2)); the objects are stacked
        along the rows of every operand, the attention work list keeps them apart."""
        g, Cd, NB, m2, TOK = self.g, self.g.fpn_dim, len(Ts), self.grid[2] ** 2, self.TOK
        hd = internal // g.dec_heads
        sq, sk = (TOK if q_side == "tok" else m2), (TOK if k_side == "tok" else m2)
        nq, nk = NB * sq, NB * sk
        tag = f"{tag}{NB}" if TOK == 16 else f"{tag}{NB}w{TOK}"          # (captured graphs hold addresses: a buffer never changes shape)
        qp = self.buf(f"d_qp_{tag}", max(nq, 16), internal)
        kp = self.buf(f"d_kp_{tag}", max(nk, 16), internal)
        vp = self.buf(f"d_vp_{tag}", max(nk, 16), internal)
        self.gemm(q, Cd, name + ".q_proj", nq, qp, internal)
        self.gemm(k, Cd, name + ".k_proj", nk, kp, internal)
        self.gemm(v, Cd, name + ".v_proj", nk, vp, internal)
        vts = rup(nk) + 64
        vt = None
        if not self.f32:
            vt = self.buf(f"d_vt_{tag}", internal, vts)
            self._ck(self.lib.sr_op_transpose(self._p(vp), internal, nk, internal, self._p(vt), vts, self._s()), "transpose")
        o = self.buf(f"d_o_{tag}", max(nq, 16), internal)
        items = []
        for ob, T in enumerate(Ts):
            lq, lk = (T if q_side == "tok" else m2), (T if k_side == "tok" else m2)
            items += [(ob * sq + q0, lk, q0, ob * sk, ob * sk, lq if lq != lk else 0) for q0 in range(0, lq, 64)]
        wk = self.work(("dec", q_side, k_side, Ts, TOK), items)
        # token -> image: <= 16 queries against m2 keys per object and head -- the kernel whose waves split the keys (q_tile = 16 says so)
        few = q_side == "tok" and k_side == "img" and hd == 16 and max(Ts) <= 16 and _FEWQ and not self.f32
        self.attention(qp, internal, kp, 0, internal, hd, vt, vts, o, internal, wk, g.dec_heads, hd ** -0.5, q_tile=16 if few else 64, v=vp, v_off=0, v_stride=internal)
        self.gemm(o, internal, name + ".o_proj", nq, out, Cd, EPI_RESID if resid is not None else EPI_STORE, resid=resid)

    def decode(self, coords: np.ndarray, labels: np.ndarray):
        """one object -> (low-resolution mask logits float32 [m4 * m4][16] (column i = mask token i), IoU-head logits float32 [1][16])"""
        low, iou = self.decode_many([(coords, labels)])
        return low, iou

    def decode_many(self, prompts: Sequence[Tuple[np.ndarray, np.ndarray]]):
        """The mask decoder for several objects of the CURRENT image at once (objects stacked along the rows of every launch: the decoder
        is ~110 dependent launches of a few microseconds, whatever the row count) -> (low [NB * m4 * m4][16] float32, iou [NB][16] float32)."""
        assert self.image_set, "set_image first"
        Cd, TOK = self.g.fpn_dim, self.TOK
        toks = [self._tokens(c, l) for c, l in prompts]
        Ts = tuple(int(t.shape[0]) for t in toks)
        NB = len(Ts)
        stride = max(TOK, max(Ts))
        if stride > TOK:               # a very long prompt: one object at a time, rows as wide as it needs, no captured graph
            assert NB == 1, "prompts of more than 10 points are decoded one object at a time"
            self.TOK, keep = (stride + 15) // 16 * 16, TOK
            try:
                tok = self.buf("d_tok_long", self.TOK, Cd)
                tok.zero_()
                tok[:Ts[0]].copy_(toks[0].to(self.dt))
                return self._decode_launches(Ts, tok)
            finally:
                self.TOK = keep
        tok = self.buf(f"d_tok{NB}", NB * TOK, Cd)
        host = torch.zeros(NB * TOK, Cd, dtype=self.dt)
        for ob, t in enumerate(toks):
            host[ob * TOK:ob * TOK + Ts[ob]] = t.to(self.dt)
        # pageable -> device in pieces of 16 KB (two objects): larger pageable copies wait for the stream to drain (the host then sits behind
        # the previous pass, 0.8 ms per call from three objects on); pinned staging buffers were measured too and cost MORE host time here
        # (0.5 ms per call for one object, 2 ms for four)
        step = (1 if self.f32 else 2) * TOK
        for r0 in range(0, NB * TOK, step):
            tok[r0:r0 + step].copy_(host[r0:r0 + step], non_blocking=True)
        # ~110 launches of a few microseconds each: issued one by one the host is the bottleneck (0.8 ms per call).  The launch sequence
        # is captured once per tuple of token counts (every buffer keeps its shape and address) and replayed.
        if not self.graph_decode:
            return self._decode_launches(Ts, tok)
        gr = self._graphs.get(Ts)
        if gr is None:
            out = self._decode_launches(Ts, tok)     # eager once: allocates every buffer / work list, raises kernel attributes
            torch.cuda.synchronize(self.device)
            cg = torch.cuda.CUDAGraph()
            # No Python garbage collection while the stream is capturing: a finaliser that touches the device (another engine's close(), a
            # tensor freed on a different stream) is an illegal call inside a capture and takes the process down (seen once in the GPU suite:
            # "Fatal Python error: Aborted ... Garbage-collecting" under this call).  torch.cuda.graph collects BEFORE it begins the capture.
            gc_was_on = gc.isenabled()
            gc.disable()
            try:
                with torch.cuda.graph(cg, capture_error_mode="thread_local"):
                    out = self._decode_launches(Ts, tok)
            finally:
                if gc_was_on:
                    gc.enable()
            gr = self._graphs[Ts] = (cg, out)
        gr[0].replay()
        return gr[1]

    def _decode_launches(self, Ts: Tuple[int, ...], tok: torch.Tensor):
        g, Cd, lib, TOK = self.g, self.g.fpn_dim, self.lib, self.TOK
        NB = len(Ts)
        m2 = self.grid[2] ** 2
        NT, NI = NB * TOK, NB * m2                                              # token rows, image rows
        sfx = f"{NB}" if TOK == 16 else f"{NB}w{TOK}"
        q = self.buf(f"d_q{sfx}", max(NT, 16), Cd)
        qx = self.buf(f"d_qx{sfx}", max(NT, 16), Cd)
        keys = self.buf(f"d_keys{sfx}", NI, Cd)
        kx = self.buf(f"d_kx{sfx}", NI, Cd)
        keys.view(NB, m2, Cd).copy_(self.keys0)
        pe = self.W.get(f"image_pe@{NB}")
        if pe is None:
            pe = self.W[f"image_pe@{NB}"] = self.W["image_pe"].repeat(NB, 1).contiguous()
        md = "mask_decoder.transformer."
        eps = 1e-5
        for l in range(g.dec_layers):
            p = f"{md}layers.{l}"
            if l == 0:
                self._mha(p + ".self_attn", tok, tok, tok, "tok", "tok", Ts, Cd, q, None, "self")
            else:
                self.ew(q, Cd, tok, Cd, qx, Cd, NT, Cd, 0)
                self._mha(p + ".self_attn", qx, qx, q, "tok", "tok", Ts, Cd, q, q, "self")
            self.layernorm(q, Cd, p + ".layer_norm1", q, Cd, NT, Cd, eps)
            self.ew(q, Cd, tok, Cd, qx, Cd, NT, Cd, 0)
            self.ew(keys, Cd, pe, Cd, kx, Cd, NI, Cd, 0)
            self._mha(p + ".cross_attn_token_to_image", qx, kx, keys, "tok", "img", Ts, Cd // 2, q, q, "t2i")
            self.layernorm(q, Cd, p + ".layer_norm2", q, Cd, NT, Cd, eps)
            hid = self.buf(f"d_hid{sfx}", max(NT, 16), g.dec_mlp)
            self.gemm(q, Cd, p + ".mlp.proj_in", NT, hid, g.dec_mlp, EPI_GELU | RELU)
            self.gemm(hid, g.dec_mlp, p + ".mlp.proj_out", NT, q, Cd, EPI_RESID, resid=q)
            self.layernorm(q, Cd, p + ".layer_norm3", q, Cd, NT, Cd, eps)
            self.ew(q, Cd, tok, Cd, qx, Cd, NT, Cd, 0)
            self._mha(p + ".cross_attn_image_to_token", kx, qx, q, "img", "tok", Ts, Cd // 2, keys, keys, "i2t")
            self.layernorm(keys, Cd, p + ".layer_norm4", keys, Cd, NI, Cd, eps)
        self.ew(q, Cd, tok, Cd, qx, Cd, NT, Cd, 0)
        self.ew(keys, Cd, pe, Cd, kx, Cd, NI, Cd, 0)
        self._mha(md + "final_attn_token_to_image", qx, kx, keys, "tok", "img", Ts, Cd // 2, q, q, "t2i")
        self.layernorm(q, Cd, md + "layer_norm_final_attn", q, Cd, NT, Cd, eps)
        # ---- upscaling (hf:1215-1221)
        G2, G1, G0 = self.grid[2], self.grid[1], self.grid[0]
        n1, n0 = G1 * G1, G0 * G0
        g1 = self.buf(f"d_g1{sfx}", NI, Cd)
        self.gemm(keys, Cd, "mask_decoder.upscale_conv1", NI, g1, Cd)
        u1 = self.buf(f"d_u1{sfx}", NB * n1, Cd // 4)
        for ob in range(NB):
            self._ck(self._fn("sr_op_pixel_shuffle_add")(self._p(g1, ob * m2 * Cd), Cd, self._p(self.f1), Cd // 4, self._p(u1, ob * n1 * (Cd // 4)), Cd // 4, G2, Cd // 4, self._s()), "shuffle1")
        self.layernorm(u1, Cd // 4, "mask_decoder.upscale_layer_norm", u1, Cd // 4, NB * n1, Cd // 4, 1e-6)
        self.ew(u1, Cd // 4, None, 0, u1, Cd // 4, NB * n1, Cd // 4, 3)
        g2 = self.buf(f"d_g2{sfx}", NB * n1, Cd // 2)
        self.gemm(u1, Cd // 4, "mask_decoder.upscale_conv2", NB * n1, g2, Cd // 2)
        u2 = self.buf(f"d_u2{sfx}", NB * n0, 64)                                 # Cd / 8 live channels, padded to one k-tile
        for ob in range(NB):
            self._ck(self._fn("sr_op_pixel_shuffle_add")(self._p(g2, ob * n1 * (Cd // 2)), Cd // 2, self._p(self.f0), Cd // 8, self._p(u2, ob * n0 * 64), 64, G1, Cd // 8, self._s()), "shuffle2")
        self.ew(u2, 64, None, 0, u2, 64, NB * n0, Cd // 8, 3)
        # ---- hypernetwork MLPs of the mask tokens, IoU head (hf:1223-1236): token t of every object = rows t, t + TOK, ... (lda = TOK * Cd)
        hyp = self.buf(f"d_hyp{sfx}", NB * 16, 64)                               # object o's 4 filters = rows 16 o .. 16 o + 3
        h1, h2 = self.buf(f"d_h1{sfx}", max(NB, 16), Cd), self.buf(f"d_h2{sfx}", max(NB, 16), Cd)
        for i in range(g.n_mask_tokens):
            n = f"mask_decoder.output_hypernetworks_mlps.{i}"
            self.gemm(q, TOK * Cd, n + ".proj_in", NB, h1, Cd, EPI_GELU | RELU, a_off=(2 + i) * Cd)
            self.gemm(h1, Cd, n + ".layers.0", NB, h2, Cd, EPI_GELU | RELU)
            self.gemm(h2, Cd, n + ".proj_out", NB, hyp, 16 * 64, out_off=i * 64)
        low = self.buf(f"d_low{sfx}", NB * n0, 16, torch.float32)
        for ob in range(NB):
            self._ck(self._fn("sr_op_gemm")(self._p(u2, ob * n0 * 64), 64, self._p(hyp, ob * 16 * 64), n0, 16, 64, self._p(low, ob * n0 * 16), 16, None, None, None, EPI_F32, self._s()), "mask gemm")
        n = "mask_decoder.iou_prediction_head"
        self.gemm(q, TOK * Cd, n + ".proj_in", NB, h1, Cd, EPI_GELU | RELU, a_off=1 * Cd)
        self.gemm(h1, Cd, n + ".layers.0", NB, h2, Cd, EPI_GELU | RELU)
        iou = self.buf(f"d_iou{sfx}", NB, 16, torch.float32)
        self.gemm(h2, Cd, n + ".proj_out", NB, iou, 16, EPI_F32)
        return low, iou

    # ------------------------------------------------------------------ the predictor's contract
    def prompt(self, point_coords=None, point_labels=None, box=None):
        """SAM2ImagePredictor's prompt preparation: coordinates scaled to the model frame, a box = two corner points labelled 2 / 3 in front
        of the clicks."""
        h, w = self.orig_hw
        cs, ls = [], []
        if box is not None:
            cs.append(np.asarray(box, dtype=np.float32).reshape(2, 2))
            ls.append(np.array([2, 3], dtype=np.int64))
        if point_coords is not None and len(point_coords):
            cs.append(np.asarray(point_coords, dtype=np.float32).reshape(-1, 2))
            ls.append(np.asarray(point_labels, dtype=np.int64).reshape(-1))
        if not cs:
            raise ValueError("a prompt needs a box or points")
        c = np.concatenate(cs, axis=0) * np.array([self.g.image_size / w, self.g.image_size / h], dtype=np.float32)
        return c.astype(np.float32), np.concatenate(ls)

    def predict_or(self, acc_u8: torch.Tensor, point_coords=None, point_labels=None, box=None, logits_out: Optional[torch.Tensor] = None):
        """One object of ``segment``'s loop, entirely on the device: decode, pick the mask with the highest predicted IoU, resize its logits
        to the image, threshold at 0, OR into ``acc_u8`` [h, w]."""
        return self.predict_or_many(acc_u8, [dict(point_coords=point_coords, point_labels=point_labels, box=box)], logits_out=logits_out)

    MAX_OBJECTS = 8     # objects per decoder pass

    def predict_or_many(self, acc_u8: torch.Tensor, prompts: Sequence[dict], logits_out: Optional[torch.Tensor] = None):
        """``segment``'s object loop for several objects per decoder pass (seg_strategy.py:47-60; the union does not depend on the order:
        the objects are sorted by prompt length so that equal multisets of lengths share a captured launch sequence)."""
        h, w = self.orig_hw
        assert acc_u8.shape == (h, w) and acc_u8.dtype == torch.uint8 and acc_u8.is_cuda
        assert logits_out is None or len(prompts) == 1
        return self.or_objects(acc_u8, [self.prompt(p.get("point_coords"), p.get("point_labels"), p.get("box")) for p in prompts], logits_out)

    def or_objects(self, acc_u8: torch.Tensor, prepared: Sequence[Tuple[np.ndarray, np.ndarray]], logits_out: Optional[torch.Tensor] = None):
        """``prepared``: (coords in the model frame, labels) per object, as ``prompt`` returns them"""
        h, w = self.orig_hw
        ps = sorted(prepared, key=lambda cl: len(cl[1]))
        long_ = [cl for cl in ps if len(cl[1]) + self.g.n_mask_tokens + 3 > self.TOK]
        ps = [cl for cl in ps if len(cl[1]) + self.g.n_mask_tokens + 3 <= self.TOK]
        n0, out = self.grid[0] ** 2, None
        for chunk in [ps[i:i + self.MAX_OBJECTS] for i in range(0, len(ps), self.MAX_OBJECTS)] + [[cl] for cl in long_]:
            low, iou = out = self.decode_many(chunk)
            for ob in range(len(chunk)):
                self._ck(self.lib.sr_op_mask_resize_or(self._p(low, ob * n0 * 16), 16, 1, self.g.n_mask_tokens - 1, self.grid[0], self._p(iou, ob * 16), self._p(acc_u8),
                                                       self._p(logits_out), h, w, self._s()), "mask resize")
        return out

    def predict(self, point_coords=None, point_labels=None, box=None, multimask_output: bool = True, return_logits: bool = False):
        """(masks [3, h, w] bool (or logits), scores [3], low-resolution logits [3, m, m]) as numpy, like SAM2ImagePredictor.predict."""
        if not multimask_output:
            raise NotImplementedError("the reference calls predict with the default multimask_output=True")
        h, w = self.orig_hw
        acc = torch.zeros(h, w, dtype=torch.uint8, device=self.device)
        lg = torch.empty(self.g.n_mask_tokens - 1, h, w, dtype=torch.float32, device=self.device)
        low, iou = self.predict_or(acc, point_coords, point_labels, box, logits_out=lg)
        m = self.grid[0]
        scores = torch.sigmoid(iou[0, 1:self.g.n_mask_tokens]).cpu().numpy()
        lows = low[:, 1:self.g.n_mask_tokens].t().reshape(-1, m, m).cpu().numpy()
        out = lg.cpu().numpy()
        return (out if return_logits else out > 0.0), scores, lows


def param_shapes(g: Sam2Geometry) -> List[Tuple[str, tuple, float, float]]:
    """(HF ``Sam2Model`` parameter name, shape, base, std) of everything ``load_state_dict`` reads.  std is what the synthetic mode draws
    (no checkpoint offline): N(0, 1 / fan_in) for weights, so that activations stay of order one through 48 blocks."""
    sp: List[Tuple[str, tuple, float, float]] = []

    def lin(name, out_f, in_f, k=0):
        sp.append((name + ".weight", (out_f, in_f) if k == 0 else (out_f, in_f, k, k), 0.0, 1.0 / math.sqrt(in_f * max(k, 1) ** 2)))
        sp.append((name + ".bias", (out_f,), 0.0, 0.02))

    def ln(name, c):
        sp.append((name + ".weight", (c,), 1.0, 0.02))
        sp.append((name + ".bias", (c,), 0.0, 0.02))

    C, d0 = g.fpn_dim, g.embed_dims[0]
    sp.append(("no_memory_embedding", (1, 1, C), 0.0, 0.02))
    sp.append(("vision_encoder.backbone.pos_embed", (1, d0, g.bkg_size, g.bkg_size), 0.0, 0.02))
    sp.append(("vision_encoder.backbone.pos_embed_window", (1, d0, g.windows[0], g.windows[0]), 0.0, 0.02))
    lin("vision_encoder.backbone.patch_embed.projection", d0, 3, 7)
    for i, (s, din, dout, heads, win, pooled) in enumerate(g.block_table()):
        p = f"vision_encoder.backbone.blocks.{i}"
        ln(p + ".layer_norm1", din)
        lin(p + ".attn.qkv", 3 * dout, din)
        lin(p + ".attn.proj", dout, dout)
        ln(p + ".layer_norm2", dout)
        lin(p + ".mlp.proj_in", 4 * dout, dout)
        lin(p + ".mlp.proj_out", dout, 4 * dout)
        if din != dout:
            lin(p + ".proj", dout, din)
    for j, c in enumerate(reversed(g.embed_dims)):
        lin(f"vision_encoder.neck.convs.{j}", C, c, 1)
    sp.append(("prompt_encoder.shared_embedding.positional_embedding", (2, C // 2), 0.0, 1.0))
    for n, r in (("prompt_encoder.no_mask_embed", 1), ("prompt_encoder.point_embed", 4), ("prompt_encoder.not_a_point_embed", 1), ("mask_decoder.iou_token", 1),
                 ("mask_decoder.mask_tokens", g.n_mask_tokens), ("mask_decoder.obj_score_token", 1)):
        sp.append((n + ".weight", (r, C), 0.0, 0.5))
    md = "mask_decoder.transformer."
    for a, internal in [(f"{md}layers.{l}.{a}", C if a == "self_attn" else C // 2) for l in range(g.dec_layers)
                        for a in ("self_attn", "cross_attn_token_to_image", "cross_attn_image_to_token")] + [(md + "final_attn_token_to_image", C // 2)]:
        for q in ("q_proj", "k_proj", "v_proj"):
            lin(f"{a}.{q}", internal, C)
        lin(f"{a}.o_proj", C, internal)
    for l in range(g.dec_layers):
        for k in (1, 2, 3, 4):
            ln(f"{md}layers.{l}.layer_norm{k}", C)
        lin(f"{md}layers.{l}.mlp.proj_in", g.dec_mlp, C)
        lin(f"{md}layers.{l}.mlp.proj_out", C, g.dec_mlp)
    ln(md + "layer_norm_final_attn", C)
    sp.append(("mask_decoder.upscale_conv1.weight", (C, C // 4, 2, 2), 0.0, 1.0 / math.sqrt(C)))
    sp.append(("mask_decoder.upscale_conv1.bias", (C // 4,), 0.0, 0.02))
    sp.append(("mask_decoder.upscale_conv2.weight", (C // 4, C // 8, 2, 2), 0.0, 1.0 / math.sqrt(C // 4)))
    sp.append(("mask_decoder.upscale_conv2.bias", (C // 8,), 0.0, 0.02))
    ln("mask_decoder.upscale_layer_norm", C // 4)
    for n, out_f in [(f"mask_decoder.output_hypernetworks_mlps.{i}", C // 8) for i in range(g.n_mask_tokens)] + [("mask_decoder.iou_prediction_head", g.n_mask_tokens)]:
        lin(n + ".proj_in", C, C)
        lin(n + ".layers.0", C, C)
        lin(n + ".proj_out", out_f, C)
    lin("mask_decoder.conv_s0", C // 8, C, 1)
    lin("mask_decoder.conv_s1", C // 4, C, 1)
    return sp


def synthetic_state_dict(g: Sam2Geometry, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Random-init weights of the right shapes (synthetic mode: no checkpoint offline)."""
    gen = torch.Generator().manual_seed(int(seed))
    return {n: (base + torch.randn(shape, generator=gen) * std).to(torch.bfloat16) for n, shape, base, std in param_shapes(g)}


# the sam2 package's checkpoint (sam2_hiera_large.pt, what the reference's provider loads: model_providers.py:540-541) -> HF names.
# UNPINNED: neither the package nor a checkpoint is available offline; the table follows the two public module trees.
_SAM2_RENAMES = [
    ("image_encoder.trunk.patch_embed.proj.", "vision_encoder.backbone.patch_embed.projection."),
    ("image_encoder.trunk.pos_embed_window", "vision_encoder.backbone.pos_embed_window"),
    ("image_encoder.trunk.pos_embed", "vision_encoder.backbone.pos_embed"),
    ("image_encoder.trunk.blocks.", "vision_encoder.backbone.blocks."),
    ("image_encoder.neck.convs.", "vision_encoder.neck.convs."),
    ("sam_prompt_encoder.pe_layer.positional_encoding_gaussian_matrix", "prompt_encoder.shared_embedding.positional_embedding"),
    ("sam_prompt_encoder.not_a_point_embed.", "prompt_encoder.not_a_point_embed."),
    ("sam_prompt_encoder.no_mask_embed.", "prompt_encoder.no_mask_embed."),
    ("sam_mask_decoder.", "mask_decoder."),
    ("no_mem_embed", "no_memory_embedding"),
]
_SAM2_INNER = [(".norm1.", ".layer_norm1."), (".norm2.", ".layer_norm2."), (".norm3.", ".layer_norm3."), (".norm4.", ".layer_norm4."),
               (".norm_final_attn.", ".layer_norm_final_attn."), (".out_proj.", ".o_proj."), (".conv.weight", ".weight"), (".conv.bias", ".bias"),
               ("output_upscaling.0.", "upscale_conv1."), ("output_upscaling.1.", "upscale_layer_norm."), ("output_upscaling.3.", "upscale_conv2.")]


def rename_sam2_checkpoint(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """sam2-package parameter names -> HF ``Sam2Model`` names (memory / video modules are dropped)."""
    out: Dict[str, torch.Tensor] = {}
    pts = {}
    for k, v in sd.items():
        if k.startswith(("memory_", "maskmem", "obj_ptr", "mask_downsample", "no_mem_pos_enc", "no_obj_", "sam_prompt_encoder.mask_downscaling")):
            continue
        if k.startswith("sam_prompt_encoder.point_embeddings."):
            pts[int(k.split(".")[2])] = v
            continue
        n = k
        for a, b in _SAM2_RENAMES:
            if n.startswith(a):
                n = b + n[len(a):]
                break
        for a, b in _SAM2_INNER:
            n = n.replace(a, b)
        if ".mlp.layers." in n and "hypernetworks" not in n and "iou_prediction" not in n and "obj_score" not in n:
            n = n.replace(".mlp.layers.0.", ".mlp.proj_in.").replace(".mlp.layers.1.", ".mlp.proj_out.")
        elif any(t in n for t in ("output_hypernetworks_mlps", "iou_prediction_head", "pred_obj_score_head")):
            n = n.replace(".layers.0.", ".proj_in.").replace(".layers.2.", ".proj_out.").replace(".layers.1.", ".layers.0.")
        out[n] = v
    if pts:
        out["prompt_encoder.point_embed.weight"] = torch.cat([pts[i].reshape(1, -1) for i in sorted(pts)], dim=0)
    return out


class Sam2Predictor:
    """What ``SegInferStrategy`` expects from its model provider: ``set_image(PIL / ndarray)`` and ``predict(**prompt)``; also
    ``segment_objects`` = the whole per-sample loop of seg_strategy.py:47-60 on the device."""

    def __init__(self, engine: Sam2Engine, batch: int = 16, cache_images: int = 256):
        self.model = self.engine = engine
        self.batch = max(1, int(batch))               # images per encoder pass (measured per tile: 5.41 / 5.19 / 5.10 ms at 8 / 12 / 16)
        self.cache_images = int(cache_images)         # embeddings kept (10 MB each at Hiera-L): stage 2 segments stage 1's image again
        self._cache: "OrderedDict[bytes, dict]" = OrderedDict()
        self.stats = {"images": 0, "encoded": 0, "cache_hits": 0, "encoder_passes": 0}
        self._upload = None                           # side stream of the image uploads (embed)
        # round 6: `prefetch` encodes images on a thread of its own (side stream) while the caller generates; the engine has ONE set of buffers,
        # so every user of it -- a prefetch chunk, embed, segment_batch -- holds this lock and leaves its stream drained when it lets go
        import threading
        self._lock = threading.RLock()
        self._pf_thread = None
        self._pf_stream = None
        self._pf_error = None
        self._want = 0                                # foreground callers waiting for the lock (the prefetch loop yields to them)

    @staticmethod
    def _host_u8(image) -> np.ndarray:
        arr = np.asarray(image.convert("RGB")) if hasattr(image, "convert") else np.asarray(image)
        return np.array(arr, dtype=np.uint8, order="C")          # (a copy: PIL hands out read-only buffers)

    def set_image(self, image):
        t = image if isinstance(image, torch.Tensor) else torch.from_numpy(self._host_u8(image))
        self.engine.set_image(t.to(self.engine.device))

    def predict(self, point_coords=None, point_labels=None, box=None, **kw):
        return self.engine.predict(point_coords, point_labels, box, **kw)

    def embed(self, images: Sequence) -> List[dict]:
        """image embeddings of a list of host images: the ones seen before come from the cache (keyed by a hash of the pixels), the rest go
        through the encoder ``batch`` at a time.  Same numbers either way: the encoder is deterministic and images do not interact."""
        import xxhash
        with self._lock:
            return self._embed_locked(images, xxhash)

    def _embed_locked(self, images, xxhash):
        arrs = [self._host_u8(im) for im in images]
        keys = [xxhash.xxh3_128_digest(a.data) + bytes(str(a.shape), "ascii") for a in arrs]
        feats: Dict[bytes, dict] = {}
        todo = []
        for k, a in zip(keys, arrs):
            if k in feats or k in todo:
                continue
            if k in self._cache:
                self._cache.move_to_end(k)
                feats[k] = self._cache[k]
                self.stats["cache_hits"] += 1
            else:
                todo.append(k)
        by_key = dict(zip(keys, arrs))
        main = torch.cuda.current_stream(self.engine.device)
        if self._upload is None:
            self._upload = torch.cuda.Stream(self.engine.device)
        for i in range(0, len(todo), self.batch):
            chunk = todo[i:i + self.batch]
            # pageable host -> device copies are stream-ordered AND host-synchronous: issued on the compute stream the host would sit behind
            # the previous chunk's encoder pass and the GPU would idle during the next uploads; on their own stream they overlap it
            with torch.cuda.stream(self._upload):
                dev_imgs = [torch.from_numpy(by_key[k]).to(self.engine.device, non_blocking=True) for k in chunk]
                up = torch.cuda.Event()
                up.record(self._upload)
            main.wait_event(up)
            for t in dev_imgs:
                t.record_stream(main)
            self.engine.set_images(dev_imgs)
            self.stats["encoder_passes"] += 1
            self.stats["encoded"] += len(chunk)
            for b, k in enumerate(chunk):
                self.engine.select(b)
                feats[k] = self.engine.features()
                if self.cache_images > 0:
                    self._cache[k] = feats[k]
                    while len(self._cache) > self.cache_images:
                        self._cache.popitem(last=False)
        self.stats["images"] += len(arrs)
        return [feats[k] for k in keys]

    def prefetch(self, images: Sequence, prepare=None, chunk: Optional[int] = None) -> None:
        """Start encoding `images` (host images, as `segment_batch` will receive them) on a background thread and a stream of its own; returns at once.
        `set_images` needs nothing but the pixels (seg_strategy.py:47-58), so the pipeline calls this before stage-1 generation: the encoder runs under
        the LM's generate call and `segment_batch` finds the embeddings in the cache (its own `embed` encodes whatever is still missing).  One chunk of
        `batch` images per lock hold, the side stream drained before the lock is released.  `prepare(image)` -- the caller's host-side preparation of one
        image, e.g. seg_infer's 756 x 756 resize -- runs on the thread too, chunk by chunk."""
        import threading
        self.wait_prefetch()
        images = list(images)
        dev = self.engine.device

        def work():
            from .serving import foreign_gpu_load
            foreign_gpu_load(True)           # the LM scheduler of this process takes no cost measurements next to this (serving.py)
            try:
                torch.cuda.set_device(dev)
                if self._pf_stream is None:
                    self._pf_stream = torch.cuda.Stream(dev)
                import time
                step = max(1, min(self.batch, int(chunk or self.batch)))      # images per lock hold (a foreground caller waits for at most one)
                for i in range(0, len(images), step):
                    while self._want > 0:            # a caller is waiting for the engine (segment_batch): a released threading lock goes to whoever asks
                        time.sleep(0.0005)           # first, which would be this loop again -- the foreground goes first
                    part = images[i:i + step] if prepare is None else [prepare(im) for im in images[i:i + step]]      # (host preparation outside the lock)
                    with self._lock, torch.cuda.stream(self._pf_stream):
                        self.embed(part)
                        self._pf_stream.synchronize()
            except Exception as e:  # noqa: BLE001  (reported by wait_prefetch / the next segment_batch; the embeddings it missed are encoded there)
                self._pf_error = e
            finally:
                foreign_gpu_load(False)
        self._pf_thread = threading.Thread(target=work, name="sam2-prefetch", daemon=True)
        self._pf_thread.start()

    def wait_prefetch(self) -> None:
        t, self._pf_thread = self._pf_thread, None
        if t is not None:
            t.join()
        if self._pf_error is not None:
            e, self._pf_error = self._pf_error, None
            raise RuntimeError("SAM2 prefetch failed") from e

    def segment_batch(self, images: Sequence, prompts: Sequence[Sequence[dict]]) -> List[torch.Tensor]:
        """seg_strategy.py:40-66 over a whole batch: embeddings (batched / cached), then the object loop of every sample on the device.
        A prefetch that is still running is not waited for (the streamed two-stage pipeline segments its first samples while later ones are still being
        encoded): this call goes ahead of the prefetch loop's next chunk, encodes what it misses itself, and the loop finds those in the cache."""
        if self._pf_error is not None:
            self.wait_prefetch()
        out = []
        self._want += 1
        try:
            with self._lock:
                for ft, vps in zip(self.embed(images), prompts):
                    self.engine.use_features(ft)
                    out.append(self.segment_objects(vps))
                if self._pf_thread is not None:      # the prefetch loop may take the engine next, on another stream: leave this one drained (as it does)
                    torch.cuda.current_stream(self.engine.device).synchronize()
        finally:
            self._want -= 1
        return out

    def segment_objects(self, prompts: Sequence[dict]) -> torch.Tensor:
        """OR of the best mask of every object prompt -> uint8 [h, w] on the device (objects whose prompt is malformed are skipped, as the
        reference's bare ``except: continue`` does)."""
        h, w = self.engine.orig_hw
        acc = torch.zeros(h, w, dtype=torch.uint8, device=self.engine.device)
        good = []
        for vp in prompts:
            try:
                kw = {}
                if "point_coords" in vp and "point_labels" in vp:
                    kw["point_coords"], kw["point_labels"] = vp["point_coords"], vp["point_labels"]
                if "box" in vp:
                    kw["box"] = vp["box"]
                good.append(self.engine.prompt(**kw))
            except (ValueError, KeyError, TypeError):
                continue
        if good:
            self.engine.or_objects(acc, good)
        return acc
