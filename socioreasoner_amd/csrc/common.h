// Shared device helpers for the SocioReasoner MI355X (gfx950) hot path.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // storage type for bf16 everywhere (bit pattern)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ---- bf16 <-> f32.  f2bf lowers to v_cvt_pk_bf16_f32 on gfx950 (round-to-nearest-even), the same
// rounding torch applies at every bf16 op boundary of the HF eager path this library reproduces.
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }
__device__ __forceinline__ uint32_t pack2(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }
__device__ __forceinline__ float lo16(uint32_t u);
__device__ __forceinline__ float hi16(uint32_t u);
// a = rbf(a), b = rbf(b) with ONE v_cvt_pk_bf16_f32 (rbf converts its value alone and wastes the instruction's second lane)
__device__ __forceinline__ void rbf2(float& a, float& b) {
    typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const uint32_t u = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2_t{a, b}, bf2_t));
    a = __uint_as_float(u << 16);
    b = __uint_as_float(u & 0xffff0000u);
}
__device__ __forceinline__ float lo16(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi16(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// HF: F.silu on a bf16 tensor = float32 x * sigmoid(x), one rounding (hf:85-96 act_fn)
__device__ __forceinline__ float silu_f(float x) { return x * (1.0f / (1.0f + __expf(-x))); }
// nn.GELU() (erf form), float32 internally
__device__ __forceinline__ float gelu_f(float x) { return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f)); }
// The same function for callers that opt in (GemmArgs.gelu_fast; the SAM2 image encoder): x * Phi(x) with Phi from erfc(z) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-z^2), t = 1 / (1 + p z), z = |x| / sqrt(2) >= 0
// (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 absolute on erfc: the same size as the float32 erff's own error after the 1 + erf
// cancellation at negative x, and three orders below the bf16 rounding every caller applies next).  ~14 instructions against the device
// library erff's ~40: the GELU epilogue of the short-K GEMMs of SAM2's
// first stages is mostly this function (encoder over 8 tiles: 41.9 -> 38.8 ms).  The LM's merger keeps erff: its float32-truth band
// (tests/test_gpu_round3.py) was measured with it.
__device__ __forceinline__ float gelu_fast_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float half_erfc = 0.5f * poly * __expf(-z * z);
    return x * (x >= 0.f ? 1.0f - half_erfc : half_erfc);
}

// Fragment-ordered ("tiled16x64") weight layout of a [N, K] matrix (N % 16 == 0, K % 64 == 0): every 16-row x 64-k
// block is 2 KB contiguous, stored in the order the decode GEMV's MFMA A-operand wants it --
//   [n / 16][k / 64][kstep = (k % 16) / 8][lane = ((k % 64) / 16) * 16 + n % 16][k % 8]
// so one wave instruction of the weight stream reads 1 KB fully contiguous (measured +22..28 % HBM throughput over
// 16 rows x 64 B fragments of a row-major matrix).  Any 8 consecutive k of one row stay contiguous (16 B), which is all
// the GEMM's LDS-DMA staging and the embedding gather need.
__host__ __device__ __forceinline__ size_t tiled_offset(size_t n, size_t k, size_t K) {
    return (((n >> 4) * (K >> 6) + (k >> 6)) * 2 + ((k >> 3) & 1)) * 512 + ((((k >> 4) & 3) << 4) + (n & 15)) * 8 + (k & 7);
}

// fp8 (OCP e4m3fn) weights of the decode stream (BASELINE.json configs[4]): W[n, k] = q[n, k] * scale[n], q in fp8.
// Fragment-ordered layout of q: every 16-row x 64-k block is 1 KB contiguous,
//   [n / 16][k / 64][lane = ((k % 64) / 16) * 16 + n % 16][k % 16]
// i.e. a lane's 16 bytes are the two 8-k MFMA steps (k % 16 < 8, >= 8) of its (row, k-group) -- the same k permutation
// the bf16 stream uses, so the x operand addressing is unchanged.
__host__ __device__ __forceinline__ size_t tiled8_offset(size_t n, size_t k, size_t K) {
    return (((n >> 4) * (K >> 6) + (k >> 6)) * 64 + (((k >> 4) & 3) << 4) + (n & 15)) * 16 + (k & 15);
}
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// 4 fp8 in one dword -> 4 bf16 (exact: every e4m3 value is a bf16 value); v_cvt_scalef32_pk_bf16_fp8 with scale 1
__device__ __forceinline__ void f8x4_to_bf16(uint32_t v, uint32_t& lo, uint32_t& hi) {
    lo = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v, 1.0f, false));
    hi = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v, 1.0f, true));
}

#define SR_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
