"""Python handle on one ``sr_engine`` (one per process / GPU).  PyTorch provides device memory and streams only."""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Sequence

import numpy as np
import torch

from . import lib as L
from .config import ModelGeometry


def _sr_config(g: ModelGeometry, max_patches, max_prefill_tokens, max_batch, max_ctx, max_new_tokens, lm_fp8=False, kv_slots=0) -> L.SrConfig:
    v, t = g.vision, g.text
    c = L.SrConfig()
    c.v_depth, c.v_hidden, c.v_heads, c.v_inter = v.depth, v.hidden_size, v.num_heads, v.intermediate_size
    c.v_patch, c.v_temporal, c.v_merge, c.v_window = v.patch_size, v.temporal_patch_size, v.spatial_merge_size, v.window_size
    c.v_out_hidden, c.v_in_ch = v.out_hidden_size, v.in_channels
    c.v_n_fullatt = len(v.fullatt_block_indexes)
    for i, b in enumerate(v.fullatt_block_indexes):
        c.v_fullatt[i] = b
    c.t_layers, c.t_hidden, c.t_heads, c.t_kv_heads = t.num_hidden_layers, t.hidden_size, t.num_attention_heads, t.num_key_value_heads
    c.t_head_dim, c.t_inter, c.t_vocab = t.head_dim, t.intermediate_size, t.vocab_size
    c.t_rms_eps, c.t_rope_theta = t.rms_norm_eps, t.rope_theta
    for i in range(3):
        c.mrope_section[i] = t.mrope_section[i]
    c.image_token_id = g.image_token_id
    c.max_patches, c.max_prefill_tokens, c.max_batch = max_patches, max_prefill_tokens, max_batch
    c.max_ctx, c.max_new_tokens = max_ctx, max_new_tokens
    c.kv_slots = int(kv_slots)
    c.lm_weight_dtype = 2 if lm_fp8 == "mx" else 1 if lm_fp8 else 0      # "mx": fp8 weights + MX fp8 activations in prefill (fp8 x fp8 MFMA)
    return c


class Engine:
    def __init__(self, geometry: ModelGeometry, *, max_patches=1024, max_prefill_tokens=512, max_batch=1, max_ctx=640,
                 max_new_tokens=128, device="cuda:0", lm_fp8=False, kv_slots: int = 0):
        """kv_slots: KV-cache slots (0 = max_batch); spare slots let the scheduler prefill the next requests while all rows decode"""
        self.lib = L.load()
        self.lib_held = L.load_held()          # the scheduler's queue-only calls, made without releasing the interpreter lock (lib.load_held)
        if not torch.cuda.is_available():
            raise L.SocioRError("no GPU visible: the product path has no CPU fallback")
        self.geom = geometry
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.cfg = _sr_config(geometry, max_patches, max_prefill_tokens, max_batch, max_ctx, max_new_tokens, lm_fp8, kv_slots)
        self.kv_slots = int(kv_slots) or int(max_batch)
        nbytes = self.lib.sr_workspace_bytes(C.byref(self.cfg))
        if nbytes == 0:
            raise L.SocioRError("invalid engine configuration: " + self.lib.sr_last_error(None).decode())
        self.workspace = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        base = (self.workspace.data_ptr() + 255) & ~255
        self._h = C.c_void_p()
        L.check(self.lib.sr_engine_create(C.byref(self.cfg), C.c_void_p(base), C.c_size_t(nbytes), C.byref(self._h)),
                None, "sr_engine_create")
        self.workspace_bytes = nbytes
        self.pixel_ld = self.lib.sr_pixel_ld(self._h)

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            torch.cuda.synchronize(self.device)
            self.lib.sr_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _s(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------ weights
    def load_weight(self, name: str, tensor: torch.Tensor):
        t = tensor.detach()
        if t.dtype not in (torch.bfloat16, torch.float32):
            t = t.float()
        t = t.to(self.device).contiguous()
        shape = (C.c_int64 * t.dim())(*t.shape)
        L.check(self.lib.sr_load_weight(self._h, name.encode(), C.c_void_p(t.data_ptr()), 0 if t.dtype == torch.bfloat16 else 1,
                                        shape, t.dim(), self._s()), self._h, f"sr_load_weight({name})")

    def load_state_dict(self, sd):
        for k, v in sd.items():
            self.load_weight(k, v)
        self.assert_ready()

    def load_safetensors_dir(self, path: str):
        """HF checkpoint directory (*.safetensors shards with HF names) -- real weights are not available offline."""
        import glob
        import os
        from safetensors import safe_open
        files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
        if not files:
            raise L.SocioRError(f"no *.safetensors under {path}")
        head = emb = None
        for f in files:
            with safe_open(f, framework="pt") as sf:
                for k in sf.keys():
                    t = sf.get_tensor(k)
                    if k.endswith("lm_head.weight"):
                        head = t                        # (the engine's LM head IS the embedding matrix: checked below, never loaded)
                        continue
                    if k.endswith("embed_tokens.weight"):
                        emb = t
                    self.load_weight(k, t)
        # decided from the checkpoint itself, whatever config.json says: an lm_head.weight that differs from the embedding is an UNTIED head
        if head is not None and emb is not None and (head.shape != emb.shape or not torch.equal(head, emb)):
            raise L.SocioRError("this checkpoint has an untied LM head (lm_head.weight differs from model.embed_tokens.weight): the engine uses the "
                                "embedding matrix as the head and would compute different logits")
        self.assert_ready()

    def load_synthetic_weights(self, seed: int = 0):
        """Device-side counter-based generator (definition: oracle/weights.py; implementation: k_synth_fill)."""
        specs = self.geom.param_specs()
        biggest = max(int(np.prod(s)) for _, s, _ in specs)
        tmp = torch.empty(biggest, dtype=torch.bfloat16, device=self.device)
        for name, shape, base in specs:
            n = int(np.prod(shape))
            L.check(self.lib.sr_synth_fill(C.c_void_p(tmp.data_ptr()), n, name.encode(), seed, C.c_float(base), self._s()),
                    None, "sr_synth_fill")
            sh = (C.c_int64 * len(shape))(*shape)
            L.check(self.lib.sr_load_weight(self._h, name.encode(), C.c_void_p(tmp.data_ptr()), 0, sh, len(shape), self._s()),
                    self._h, f"sr_load_weight({name})")
        torch.cuda.synchronize(self.device)
        del tmp
        self.assert_ready()

    def assert_ready(self):
        buf = C.create_string_buffer(200)
        n = self.lib.sr_weights_missing(self._h, buf, 200)
        if n:
            raise L.SocioRError(f"{n} parameters missing, e.g. {buf.value.decode()}")
        L.check(self.lib.sr_finalize_weights(self._h, self._s()), self._h, "sr_finalize_weights")   # fp8 mode: quantise

    # ------------------------------------------------------------------ ViT
    def patchify(self, img_u8: torch.Tensor) -> torch.Tensor:
        """uint8 HWC cuda image -> bf16 [N, pixel_ld] (K1)."""
        assert img_u8.dtype == torch.uint8 and img_u8.is_cuda and img_u8.dim() == 3 and img_u8.shape[2] == 3
        img_u8 = img_u8.contiguous()
        h, w = int(img_u8.shape[0]), int(img_u8.shape[1])
        p = self.geom.vision.patch_size
        out = torch.empty((h // p) * (w // p), self.pixel_ld, dtype=torch.bfloat16, device=self.device)
        L.check(self.lib.sr_patchify_u8(self._h, C.c_void_p(img_u8.data_ptr()), h, w, C.c_void_p(out.data_ptr()), self._s()),
                self._h, "sr_patchify_u8")
        return out

    def vit_forward(self, pixels: torch.Tensor, grid_thw: Sequence[Sequence[int]]) -> torch.Tensor:
        """pixels: bf16 [N, pixel_ld] (from patchify) or float32 [N, C*T*p*p] (HF processor layout) -> bf16 [N/4, H]."""
        grid = np.ascontiguousarray(np.asarray(grid_thw, dtype=np.int64).reshape(-1, 3))
        n = int((grid[:, 0] * grid[:, 1] * grid[:, 2]).sum())
        pixels = pixels.contiguous()
        if pixels.dtype == torch.bfloat16:
            assert pixels.shape == (n, self.pixel_ld), (pixels.shape, n, self.pixel_ld)
            dt = 0
        else:
            assert pixels.dtype == torch.float32 and pixels.shape[0] == n
            dt = 1
        m2 = self.geom.vision.spatial_merge_size ** 2
        out = torch.empty(n // m2, self.geom.vision.out_hidden_size, dtype=torch.bfloat16, device=self.device)
        L.check(self.lib.sr_vit_forward(self._h, C.c_void_p(pixels.data_ptr()), dt, grid.ctypes.data_as(L._i64p), len(grid),
                                        C.c_void_p(out.data_ptr()), self._s()), self._h, "sr_vit_forward")
        return out

    def vit_plan(self) -> dict:
        """Which attention kernels the last vit_forward's grids take (sr_vit_plan)."""
        out = np.zeros(4, dtype=np.int32)
        L.check(self.lib.sr_vit_plan(self._h, out.ctypes.data_as(L._i32p)), self._h, "sr_vit_plan")
        return {"window_items": int(out[0]), "full_items": int(out[1]), "windows_all_64": bool(out[2]), "full_aligned": bool(out[3])}

    # ------------------------------------------------------------------ LM
    def prefill(self, ids: Sequence[np.ndarray], pos3: Sequence[np.ndarray], image_embeds: torch.Tensor | None = None,
                slots: Iterable[int] | None = None, return_logits: bool = False):
        """ids[b]: int64 [S_b]; pos3[b]: int64 [3, S_b] (un-padded).  Fills KV slots, returns float32 logits [B, V] if asked."""
        B = len(ids)
        lens = np.array([len(x) for x in ids], dtype=np.int32)
        flat = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.int64) for x in ids]))
        p3 = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.int64).reshape(3, -1) for p in pos3], axis=1))
        sl = np.arange(B, dtype=np.int32) if slots is None else np.asarray(list(slots), dtype=np.int32)
        logits = torch.empty(B, self.geom.text.vocab_size, dtype=torch.float32, device=self.device) if return_logits else None
        n_img = 0 if image_embeds is None else int(image_embeds.shape[0])
        if image_embeds is not None:
            image_embeds = image_embeds.contiguous()
            assert image_embeds.dtype == torch.bfloat16
        L.check(self.lib.sr_prefill(self._h, flat.ctypes.data_as(L._i64p), p3.ctypes.data_as(L._i64p),
                                    lens.ctypes.data_as(L._i32p), sl.ctypes.data_as(L._i32p), B,
                                    C.c_void_p(image_embeds.data_ptr()) if image_embeds is not None else None, n_img,
                                    C.c_void_p(logits.data_ptr()) if logits is not None else None, self._s()),
                self._h, "sr_prefill")
        self._last_B = B
        return logits

    def decode_sample(self, max_new: int, temperature: float, top_k: int, top_p: float = 1.0, repetition_penalty: float = 1.0,
                      seed: int = 0, eos: Sequence[int] = (), pad_id: int = 0, use_graph: bool = True) -> torch.Tensor:
        """Sampled decode on the device for the sequences of the last prefill -> int32 tokens [B, max_new]."""
        B = self._last_B
        toks = torch.empty(B, max_new, dtype=torch.int32, device=self.device)
        eos_a = np.asarray(list(eos), dtype=np.int32)
        done = C.c_int(0)
        L.check(self.lib.sr_decode_sample(self._h, B, max_new, eos_a.ctypes.data_as(L._i32p) if len(eos_a) else None, len(eos_a), pad_id,
                                          C.c_float(temperature), int(top_k), C.c_float(top_p), C.c_float(repetition_penalty), int(seed) & 0xffffffff,
                                          C.c_void_p(toks.data_ptr()), 1 if use_graph else 0, self._s(), C.byref(done)), self._h, "sr_decode_sample")
        self.steps_done = done.value
        return toks

    def forward_logits(self, ids: Sequence[np.ndarray], pos3: Sequence[np.ndarray], image_embeds: torch.Tensor | None = None):
        """Teacher-forced forward: float32 logits [n_tok, V] for every position of the packed sequences."""
        B = len(ids)
        lens = np.array([len(x) for x in ids], dtype=np.int32)
        flat = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.int64) for x in ids]))
        p3 = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.int64).reshape(3, -1) for p in pos3], axis=1))
        out = torch.empty(len(flat), self.geom.text.vocab_size, dtype=torch.float32, device=self.device)
        n_img = 0 if image_embeds is None else int(image_embeds.shape[0])
        if image_embeds is not None:
            image_embeds = image_embeds.contiguous()
        L.check(self.lib.sr_forward_logits(self._h, flat.ctypes.data_as(L._i64p), p3.ctypes.data_as(L._i64p), lens.ctypes.data_as(L._i32p), B,
                                           C.c_void_p(image_embeds.data_ptr()) if image_embeds is not None else None, n_img,
                                           C.c_void_p(out.data_ptr()), self._s()), self._h, "sr_forward_logits")
        self._last_B = B
        return out

    def decode(self, max_new: int, eos: Sequence[int] = (), pad_id: int = 0, trace: bool = False,
               forced: torch.Tensor | None = None, use_graph: bool = True):
        """Greedy decode for the sequences of the last prefill.  Returns int32 tokens [B, max_new] (and, with
        trace=True, the float32 logits [max_new, B, V] that produced them)."""
        B = self._last_B
        toks = torch.empty(B, max_new, dtype=torch.int32, device=self.device)
        tr = torch.empty(max_new, B, self.geom.text.vocab_size, dtype=torch.float32, device=self.device) if trace else None
        eos_a = np.asarray(list(eos), dtype=np.int32)
        if forced is not None:
            forced = forced.to(device=self.device, dtype=torch.int32).contiguous()
            assert forced.shape == (B, max_new)
        done = C.c_int(0)
        L.check(self.lib.sr_decode(self._h, None, B, max_new, eos_a.ctypes.data_as(L._i32p) if len(eos_a) else None, len(eos_a),
                                   pad_id, C.c_void_p(toks.data_ptr()), C.c_void_p(tr.data_ptr()) if trace else None,
                                   C.c_void_p(forced.data_ptr()) if forced is not None else None, 1 if use_graph else 0,
                                   self._s(), C.byref(done)), self._h, "sr_decode")
        self.steps_done = done.value
        return (toks, tr) if trace else toks

    # ------------------------------------------------------------------ continuous batching (rows with independent lifecycles)
    def rows_begin(self):
        L.check(self.lib.sr_rows_begin(self._h, self._s()), self._h, "sr_rows_begin")
        self._last_B = self.cfg.max_batch

    def rows_sampling(self, temperature: float, top_k: int, top_p: float = 1.0, seed: int = 0):
        """After rows_begin: all rows sample (k_sample) instead of taking the arg-max; temperature 0 = greedy again."""
        L.check(self.lib.sr_rows_sampling(self._h, C.c_float(temperature), int(top_k), C.c_float(top_p), int(seed) & 0xffffffff), self._h, "sr_rows_sampling")

    def admit(self, rows: Sequence[int], ids: Sequence[np.ndarray], pos3: Sequence[np.ndarray], max_new: Sequence[int],
              image_embeds: torch.Tensor | None = None, return_logits: bool = False):
        """Prefill sequences into free batch rows without disturbing running ones; row i stops after max_new[i] tokens."""
        n = len(ids)
        lens = np.array([len(x) for x in ids], dtype=np.int32)
        flat = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.int64) for x in ids]))
        p3 = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.int64).reshape(3, -1) for p in pos3], axis=1))
        rw = np.asarray(list(rows), dtype=np.int32)
        mn = np.asarray(list(max_new), dtype=np.int32)
        logits = torch.empty(n, self.geom.text.vocab_size, dtype=torch.float32, device=self.device) if return_logits else None
        n_img = 0 if image_embeds is None else int(image_embeds.shape[0])
        if image_embeds is not None:
            image_embeds = image_embeds.contiguous()
            assert image_embeds.dtype == torch.bfloat16
        L.check(self.lib.sr_admit(self._h, flat.ctypes.data_as(L._i64p), p3.ctypes.data_as(L._i64p), lens.ctypes.data_as(L._i32p),
                                  rw.ctypes.data_as(L._i32p), mn.ctypes.data_as(L._i32p), n,
                                  C.c_void_p(image_embeds.data_ptr()) if image_embeds is not None else None, n_img,
                                  C.c_void_p(logits.data_ptr()) if logits is not None else None, self._s()), self._h, "sr_admit")
        return logits

    def admit_stage(self, kv_slots: Sequence[int], ids: Sequence[np.ndarray], pos3: Sequence[np.ndarray], max_new: Sequence[int],
                    image_embeds: torch.Tensor | None = None):
        """First half of an admission, on the CURRENT torch stream (normally a CU-masked side stream): prefill into spare KV slots, LM
        head, first token.  No row state is touched; admit_commit installs the sequences into rows later."""
        n = len(ids)
        lens = np.array([len(x) for x in ids], dtype=np.int32)
        flat = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.int64) for x in ids]))
        p3 = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.int64).reshape(3, -1) for p in pos3], axis=1))
        sl = np.asarray(list(kv_slots), dtype=np.int32)
        mn = np.asarray(list(max_new), dtype=np.int32)
        n_img = 0 if image_embeds is None else int(image_embeds.shape[0])
        if image_embeds is not None:
            image_embeds = image_embeds.contiguous()
            assert image_embeds.dtype == torch.bfloat16
        L.check(self.lib.sr_admit_stage(self._h, flat.ctypes.data_as(L._i64p), p3.ctypes.data_as(L._i64p), lens.ctypes.data_as(L._i32p),
                                        sl.ctypes.data_as(L._i32p), mn.ctypes.data_as(L._i32p), n,
                                        C.c_void_p(image_embeds.data_ptr()) if image_embeds is not None else None, n_img, None, self._s()),
                self._h, "sr_admit_stage")

    def admit_commit(self, rows: Sequence[int]):
        """Second half, on the decode stream between two steps: batch row rows[i] takes over the i-th staged sequence (and its KV slot)."""
        rw = np.asarray(list(rows), dtype=np.int32)
        L.check(self.lib_held.sr_admit_commit(self._h, rw.ctypes.data_as(L._i32p), len(rw), self._s()), self._h, "sr_admit_commit")

    def rows_step(self, n_steps: int, eos: Sequence[int] = (), pad_id: int = 0):
        eos_a = np.asarray(list(eos), dtype=np.int32)
        L.check(self.lib_held.sr_rows_step(self._h, n_steps, eos_a.ctypes.data_as(L._i32p) if len(eos_a) else None, len(eos_a), pad_id, self._s()),
                self._h, "sr_rows_step")

    def rows_set_cus(self, n_cus: int):
        """hint: the decode steps queued after this on the current stream run on n_cus compute units (0 = the whole chip); results do not depend on it"""
        L.check(self.lib_held.sr_rows_set_cus(self._h, int(n_cus), self._s()), self._h, "sr_rows_set_cus")

    def rows_poll(self):
        """-> (finished flags, generated-token counts), numpy int32 [max_batch]; synchronises."""
        fin = np.zeros(self.cfg.max_batch, dtype=np.int32)
        cnt = np.zeros(self.cfg.max_batch, dtype=np.int32)
        L.check(self.lib.sr_rows_poll(self._h, fin.ctypes.data_as(L._i32p), cnt.ctypes.data_as(L._i32p), self._s()), self._h, "sr_rows_poll")
        return fin, cnt

    def rows_abort(self, rows: Sequence[int]):
        """The named rows stop now: the next rows_poll reports them finished, row and KV slot can be re-used at once."""
        rw = np.asarray(list(rows), dtype=np.int32)
        L.check(self.lib.sr_rows_abort(self._h, rw.ctypes.data_as(L._i32p), len(rw), self._s()), self._h, "sr_rows_abort")

    def row_tokens(self, row: int, n: int) -> torch.Tensor:
        out = torch.empty(n, dtype=torch.int32, device=self.device)
        L.check(self.lib_held.sr_rows_read(self._h, row, C.c_void_p(out.data_ptr()), n, self._s()), self._h, "sr_rows_read")
        return out

    def rows_tokens(self, rows: Sequence[int], counts: Sequence[int]) -> torch.Tensor:
        """the generated tokens of several rows as ONE device tensor int32 [len(rows), max(counts)] (row i valid up to counts[i]): stream-ordered
        device-to-device copies, so a row may be re-used by a later call on the same stream before the host has looked at the result"""
        out = torch.empty(len(rows), max(max(counts, default=0), 1), dtype=torch.int32, device=self.device)
        for i, (row, n) in enumerate(zip(rows, counts)):
            L.check(self.lib_held.sr_rows_read(self._h, int(row), C.c_void_p(out[i].data_ptr()), int(n), self._s()), self._h, "sr_rows_read")
        return out

    def decode_step(self, last_ids: torch.Tensor | None = None, return_logits: bool = True):
        """One forward pass with the token choice left to the caller (sampling, logits verification): feeds
        last_ids [B] int64 (None = the engine's greedy token) and returns (float32 logits [B, V] or None, greedy ids [B] int64)."""
        B = self._last_B
        if last_ids is not None:
            last_ids = last_ids.to(device=self.device, dtype=torch.int64).contiguous()
            assert last_ids.shape == (B,)
        logits = torch.empty(B, self.geom.text.vocab_size, dtype=torch.float32, device=self.device) if return_logits else None
        nxt = torch.empty(B, dtype=torch.int64, device=self.device)
        L.check(self.lib.sr_decode_step(self._h, C.c_void_p(last_ids.data_ptr()) if last_ids is not None else None, B,
                                        C.c_void_p(logits.data_ptr()) if logits is not None else None, C.c_void_p(nxt.data_ptr()),
                                        self._s()), self._h, "sr_decode_step")
        return logits, nxt
