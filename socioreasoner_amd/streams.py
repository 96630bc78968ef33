"""HIP streams restricted to a subset of the compute units (hipExtStreamCreateWithCUMask), used to run the admission of the next
requests (ViT + prefill: MFMA-bound) UNDER the decode steps of the running rows (HBM- / latency-bound) instead of in front of them.

Unmasked streams do not give that overlap on MI355X: the prefill GEMM grids own every CU for hundreds of microseconds at a time and the
decode step's short kernels queue behind them (measured: both simply run back to back).  With disjoint CU sets the two really run side
by side (tools/probe_overlap.py: admission of 32 tiles on 96 CUs 288 ms instead of 128 ms, decode step on the other 160 CUs 3.16 ms
instead of 2.49 ms -- together 15 % less time per 32-tile wave than one after the other).

Mask layout (established by the same probe): bit i of the mask is CU (i // 32) of shader engine (i % 32) -- MI355X has 32 shader engines
of 8 CUs -- so one full 32-bit word selects one CU index across the whole chip and every XCD / HBM channel keeps being used evenly.
Masked streams are ordinary (blocking) streams: they synchronise with the NULL stream, so their partner must not be the null stream.
"""
from __future__ import annotations

import ctypes as C

import torch

_hip = None


def _rt():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
    return _hip


def _stream_with_mask(dev, mask_words) -> "torch.cuda.ExternalStream":
    mask = (C.c_uint32 * len(mask_words))(*mask_words)
    s = C.c_void_p()
    with torch.cuda.device(dev):
        rc = _rt().hipExtStreamCreateWithCUMask(C.byref(s), len(mask_words), mask)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed with {rc}")
    return torch.cuda.ExternalStream(s.value, device=dev)


def _cu_words(device):
    return max(1, torch.cuda.get_device_properties(torch.device(device)).multi_processor_count // 32)


def masked_stream(device, cu_lo: int, cu_hi: int) -> "torch.cuda.ExternalStream":
    """A stream whose kernels may only run on CUs cu_lo .. cu_hi - 1 of every shader engine (0 <= cu_lo < cu_hi <= 8 on MI355X)."""
    dev = torch.device(device)
    words = _cu_words(dev)
    if not (0 <= cu_lo < cu_hi <= words):
        raise ValueError(f"CU range {cu_lo}..{cu_hi} outside 0..{words}")
    return _stream_with_mask(dev, [0xFFFFFFFF if cu_lo <= g < cu_hi else 0 for g in range(words)])


def split_masks(words: int, admit_cus_per_se: float):
    """(admission mask, decode mask): the admission gets CU indices 0 .. floor(x) - 1 of every shader engine and, for a fractional x,
    CU index floor(x) of every other shader engine (x = 2.5 -> 80 of 256 CUs, 2 of the 4 shader engines of every XCD)."""
    full = int(admit_cus_per_se)
    frac = admit_cus_per_se - full
    adm = [0xFFFFFFFF if g < full else 0 for g in range(words)]
    if frac > 0 and full < words:
        adm[full] = 0x55555555
    if not any(adm) or all(a == 0xFFFFFFFF for a in adm):
        raise ValueError(f"admission share {admit_cus_per_se} of {words} CUs per shader engine leaves one side empty")
    return adm, [a ^ 0xFFFFFFFF for a in adm]


_sets = {}


def overlap_streams(device, admit_cus_per_se: float = 3) -> "OverlapStreams":
    """One stream set per (device, split) for the life of the process: schedulers come and go (one per generate call), HIP streams
    created with a CU mask are never handed back by torch."""
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), float(admit_cus_per_se))
    if key not in _sets:
        _sets[key] = OverlapStreams(dev, admit_cus_per_se)
    return _sets[key]


class OverlapStreams:
    """decode_full: unmasked, used while no admission is in flight; decode / admit: the two halves of the chip."""

    def __init__(self, device, admit_cus_per_se: float = 3):
        dev = torch.device(device)
        adm, dec = split_masks(_cu_words(dev), admit_cus_per_se)
        self.decode_full = torch.cuda.Stream(dev)
        self.admit = _stream_with_mask(dev, adm)
        self.decode = _stream_with_mask(dev, dec)
        self.decode_cus = sum(bin(w).count("1") for w in dec)          # CUs of the decode half (the engine's sr_rows_set_cus hint)
