"""ctypes binding of ``libsocior.so`` (C ABI: include/socior.h).  No fallback: without the HIP library the
product path raises -- a CPU path silently taking over would void every parity and performance claim."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsocior.so")


class SrConfig(C.Structure):
    _fields_ = [
        ("v_depth", C.c_int32), ("v_hidden", C.c_int32), ("v_heads", C.c_int32), ("v_inter", C.c_int32),
        ("v_patch", C.c_int32), ("v_temporal", C.c_int32), ("v_merge", C.c_int32), ("v_window", C.c_int32),
        ("v_out_hidden", C.c_int32), ("v_in_ch", C.c_int32), ("v_n_fullatt", C.c_int32), ("v_fullatt", C.c_int32 * 16),
        ("t_layers", C.c_int32), ("t_hidden", C.c_int32), ("t_heads", C.c_int32), ("t_kv_heads", C.c_int32),
        ("t_head_dim", C.c_int32), ("t_inter", C.c_int32), ("t_vocab", C.c_int32),
        ("t_rms_eps", C.c_float), ("t_rope_theta", C.c_float), ("mrope_section", C.c_int32 * 3),
        ("image_token_id", C.c_int32),
        ("max_patches", C.c_int32), ("max_prefill_tokens", C.c_int32), ("max_batch", C.c_int32),
        ("max_ctx", C.c_int32), ("max_new_tokens", C.c_int32), ("lm_weight_dtype", C.c_int32), ("kv_slots", C.c_int32),
    ]


class SocioRError(RuntimeError):
    pass


_lib = None
_vp, _i, _i64p, _i32p = C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int32)

# name -> (restype, argtypes); every symbol include/socior.h declares
SIGNATURES = {
    "sr_version": (C.c_int, []),
    "sr_workspace_bytes": (C.c_size_t, [C.POINTER(SrConfig)]),
    "sr_engine_create": (C.c_int, [C.POINTER(SrConfig), _vp, C.c_size_t, C.POINTER(_vp)]),
    "sr_engine_destroy": (C.c_int, [_vp]),
    "sr_last_error": (C.c_char_p, [_vp]),
    "sr_load_weight": (C.c_int, [_vp, C.c_char_p, _vp, _i, _i64p, _i, _vp]),
    "sr_weights_missing": (C.c_int, [_vp, C.c_char_p, C.c_size_t]),
    "sr_synth_fill": (C.c_int, [_vp, C.c_int64, C.c_char_p, C.c_uint32, C.c_float, _vp]),
    "sr_pixel_ld": (C.c_int, [_vp]),
    "sr_patchify_u8": (C.c_int, [_vp, _vp, _i, _i, _vp, _vp]),
    "sr_vit_forward": (C.c_int, [_vp, _vp, _i, _i64p, _i, _vp, _vp]),
    "sr_prefill": (C.c_int, [_vp, _i64p, _i64p, _i32p, _i32p, _i, _vp, _i, _vp, _vp]),
    "sr_decode": (C.c_int, [_vp, _i32p, _i, _i, _i32p, _i, C.c_int32, _vp, _vp, _vp, _i, _vp, C.POINTER(C.c_int)]),
    "sr_decode_step": (C.c_int, [_vp, _vp, _i, _vp, _vp, _vp]),
    "sr_finalize_weights": (C.c_int, [_vp, _vp]),
    "sr_decode_sample": (C.c_int, [_vp, _i, _i, _i32p, _i, C.c_int32, C.c_float, _i, C.c_float, C.c_float, C.c_uint32, _vp, _i, _vp, C.POINTER(C.c_int)]),
    "sr_op_sample": (C.c_int, [_vp, _i, _i, C.c_float, _i, C.c_float, C.c_float, _vp, C.c_uint32, _vp, _vp, _vp, _i, _i, _vp]),
    "sr_forward_logits": (C.c_int, [_vp, _i64p, _i64p, _i32p, _i, _vp, _i, _vp, _vp]),
    "sr_op_quant_f8": (C.c_int, [_vp, _i, _i, _vp, _vp, _vp]),
    "sr_op_gemv_f8": (C.c_int, [_vp, _i, _vp, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, C.c_float, _i, _vp]),
    "sr_rows_begin": (C.c_int, [_vp, _vp]),
    "sr_rows_sampling": (C.c_int, [_vp, C.c_float, _i, C.c_float, C.c_uint32]),
    "sr_admit": (C.c_int, [_vp, _i64p, _i64p, _i32p, _i32p, _i32p, _i, _vp, _i, _vp, _vp]),
    "sr_admit_stage": (C.c_int, [_vp, _i64p, _i64p, _i32p, _i32p, _i32p, _i, _vp, _i, _vp, _vp]),
    "sr_admit_commit": (C.c_int, [_vp, _i32p, _i, _vp]),
    "sr_rows_step": (C.c_int, [_vp, _i, _i32p, _i, C.c_int32, _vp]),
    "sr_rows_set_cus": (C.c_int, [_vp, _i, _vp]),
    "sr_op_gemv_set_cus": (C.c_int, [_i, _vp]),
    "sr_rows_poll": (C.c_int, [_vp, _i32p, _i32p, _vp]),
    "sr_rows_read": (C.c_int, [_vp, _i, _vp, _i, _vp]),
    "sr_rows_abort": (C.c_int, [_vp, _i32p, _i, _vp]),
    "sr_vit_plan": (C.c_int, [_vp, _i32p]),
    "sr_mask_union": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "sr_resize_nearest_u8": (C.c_int, [_vp, _i, _i, _vp, _i, _i, _vp]),
    "sr_iou_counts": (C.c_int, [_vp, _vp, C.c_size_t, _vp, _vp]),
    "sr_iou_counts_batched": (C.c_int, [_vp, _vp, C.c_size_t, _i, _vp, _vp]),
    "sr_render_overlay": (C.c_int, [_vp, _i, _i, _vp, _i, _i, _vp, _i, _vp]),
    "sr_op_gemm": (C.c_int, [_vp, _i, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _vp]),
    "sr_op_quant_mx": (C.c_int, [_vp, _i, _i, _i, _vp, _vp, _i, _vp]),
    "sr_op_gemm_mx": (C.c_int, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp]),
    "sr_op_gemv": (C.c_int, [_vp, _i, _vp, _i, _i, _i, _vp, _i, _i, _vp]),
    "sr_op_gemv_fused": (C.c_int, [_vp, _i, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, C.c_float, _vp, _i, _vp, _vp, _vp, _vp]),
    "sr_op_gemv_f32_blocks": (C.c_int, [_i, _i, _i, _i]),
    "sr_op_attn_decode": (C.c_int, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, C.c_float, _vp, _vp]),
    "sr_op_attention": (C.c_int, [_vp, _i, _vp, _i, C.c_longlong, _vp, _i, C.c_longlong, _vp, _i, _vp, _i, _i, _i, C.c_float, _i, _i, _i, _i, _vp]),
    "sr_op_sam_preprocess": (C.c_int, [_vp, _i, _i, _vp, _i, _vp]),
    "sr_op_im2col": (C.c_int, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "sr_op_layernorm": (C.c_int, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, C.c_float, _vp]),
    "sr_op_maxpool_win": (C.c_int, [_vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "sr_op_ew": (C.c_int, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "sr_op_transpose": (C.c_int, [_vp, _i, _i, _i, _vp, _i, _vp]),
    "sr_op_upsample2x_add": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "sr_op_pixel_shuffle_add": (C.c_int, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp]),
    "sr_op_mask_resize_or": (C.c_int, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp]),
    "sr_op_gather_rows": (C.c_int, [_vp, _vp, _vp, _i, _i, _vp]),
    "sr_op_gemm_f32": (C.c_int, [_vp, _i, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _vp]),
    "sr_op_attention_f32": (C.c_int, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, C.c_float, _i, _vp]),
    "sr_op_attention_f32_causal": (C.c_int, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, C.c_float, _i, _vp]),
    "sr_op_rmsnorm_f32": (C.c_int, [_vp, _i, _vp, _vp, _i, _i, _i, C.c_float, _vp]),
    "sr_op_rope_f32": (C.c_int, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "sr_op_rope_table_f32": (C.c_int, [_vp, _i, _vp, _i, _vp, _vp, _vp]),
    "sr_op_sam_preprocess_f32": (C.c_int, [_vp, _i, _i, _vp, _i, _vp]),
    "sr_op_im2col_f32": (C.c_int, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "sr_op_layernorm_f32": (C.c_int, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, C.c_float, _vp]),
    "sr_op_maxpool_win_f32": (C.c_int, [_vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "sr_op_ew_f32": (C.c_int, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "sr_op_upsample2x_add_f32": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "sr_op_pixel_shuffle_add_f32": (C.c_int, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp]),
    "sr_op_rmsnorm": (C.c_int, [_vp, _vp, _vp, _i, _i, C.c_float, _vp]),
    "sr_op_resid_rmsnorm": (C.c_int, [_vp, _vp, _i, _vp, _vp, _i, _i, C.c_float, _vp]),
    "sr_op_argmax": (C.c_int, [_vp, _i, _i, _vp, _vp]),
    "sr_switches_reload": (C.c_int, []),
}


def load():
    """Loads libsocior.so (built by ``__graft_entry__.build()`` / ``make -C socioreasoner_amd/csrc``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SocioRError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)   # AttributeError here = ABI drift between header and library
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


_lib_held = None
# the scheduler's per-round calls that only QUEUE work (no wait inside): see load_held
HELD = ("sr_rows_step", "sr_rows_set_cus", "sr_rows_read", "sr_admit_commit")


def load_held():
    """The same library through ctypes.PyDLL, for the few calls of a scheduling round that only queue work: a CDLL call releases the interpreter
    lock and has to win it back afterwards -- with other Python threads busy (the streamed two-stage pipeline: host flow, SAM2 prefetch) that is up to
    one switch interval per call, five calls per round, between two chunks of decode steps.  Never used for a call that waits for the device."""
    global _lib_held
    if _lib_held is None:
        load()
        lib = C.PyDLL(LIB_PATH)
        for name in HELD:
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = SIGNATURES[name]
        _lib_held = lib
    return _lib_held


def check(rc: int, engine=None, what: str = ""):
    if rc != 0:
        msg = load().sr_last_error(engine).decode("utf-8", "replace")
        raise SocioRError(f"{what or 'libsocior'} failed ({rc}): {msg}")


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def reload_switches():
    """The library reads its SR_* switches once (and at every engine creation); call this after changing one in ``os.environ``."""
    return load().sr_switches_reload()
