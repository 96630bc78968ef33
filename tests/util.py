"""Shared helpers for parity tests."""
import numpy as np
import torch


def bits_to_f32(bits: np.ndarray) -> torch.Tensor:
    """uint16 bf16 bit patterns -> float32 tensor."""
    return torch.from_numpy(bits.astype(np.uint16).view(np.int16).copy()).view(torch.bfloat16).float()


def bf16_ulp(x: torch.Tensor) -> torch.Tensor:
    """Spacing of bf16 numbers at |x| (>= the smallest normal spacing)."""
    e = torch.floor(torch.log2(x.abs().clamp_min(1e-30)))
    return torch.pow(2.0, e - 7)


def bf16_compare(got: torch.Tensor, want: torch.Tensor):
    """Returns (max error in bf16 ulps of ``want``, fraction of elements that differ, max abs error)."""
    got, want = got.float().flatten(), want.float().flatten()
    d = (got - want).abs()
    # near-zero outputs of a cancelling sum carry the absolute error of the typical output: floor the
    # magnitude at the tensor rms so that "1 ulp" means one bf16 step of a typically sized element
    floor = want.pow(2).mean().sqrt()
    ulps = d / bf16_ulp(torch.maximum(torch.maximum(got.abs(), want.abs()), floor))
    return float(ulps.max()), float((d > 0).float().mean()), float(d.max())


def assert_bf16_close(got, want, max_ulp=1.0, max_frac=0.01, what=""):
    mu, frac, mad = bf16_compare(got, want)
    assert mu <= max_ulp + 1e-6 and frac <= max_frac, f"{what}: max {mu:.2f} ulp, {frac:.4%} differ, max abs {mad:.3e}"
    return mu, frac, mad


def tile16x64(w: torch.Tensor) -> torch.Tensor:
    """Row-major [N, K] -> the engine's fragment-ordered layout (socioreasoner_amd/csrc/common.h tiled_offset):
    [n/16][k/64][kstep=(k%16)/8][lane=((k%64)/16)*16 + n%16][k%8], returned flat with the same number of elements."""
    N, K = w.shape
    assert N % 16 == 0 and K % 64 == 0
    return w.reshape(N // 16, 16, K // 64, 4, 2, 8).permute(0, 2, 4, 3, 1, 5).contiguous().reshape(N, K)


def tile8(q: torch.Tensor) -> torch.Tensor:
    """Row-major [N, K] bytes -> the fp8 fragment order (socioreasoner_amd/csrc/common.h tiled8_offset):
    [n/16][k/64][lane=((k%64)/16)*16 + n%16][k%16]."""
    N, K = q.shape
    assert N % 16 == 0 and K % 64 == 0
    return q.reshape(N // 16, 16, K // 64, 4, 16).permute(0, 2, 3, 1, 4).contiguous().reshape(N, K)


def switch(monkeypatch, name, value):
    """Set (value=None: delete) an SR_* switch of the library for the rest of the test."""
    from socioreasoner_amd import lib
    if value is None:
        monkeypatch.delenv(name, raising=False)
    else:
        monkeypatch.setenv(name, str(value))
    lib.reload_switches()
