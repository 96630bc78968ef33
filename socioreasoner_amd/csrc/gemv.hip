// Weight-streaming skinny GEMM for autoregressive decode (SURVEY.md 2.3 K13/K16/K17 at M = batch <= 32):
//   out[M,N] = x[M,K] . W[N,K]^T.   HBM-bound: every weight byte is read exactly once per step, straight from
//   HBM into VGPRs (no LDS round trip: the operand is not shared between waves), 16 B per lane, 128 B per row
//   per k-chunk, non-temporal (each CU reads its slice once), deep unroll so that >= 8 KB per wave is in flight.
// Each wave owns one 16-row tile of W (two tiles -- gate and up -- for the SwiGLU epilogue) over its K slice and
// feeds it to the 16x16x32 MFMA as the A operand; x (tiny, L2 resident) is the B operand, rows >= M read as zero.
// The k-slot -> k mapping is permuted (lane group g covers k = g*16 .. g*16+15 of each 64-chunk as two MFMA steps)
// so that a lane's two 16-byte loads are contiguous; both operands use the same permutation, so the sum is unchanged.
#include "kernels.h"
#include <type_traits>

namespace {

constexpr int WAVES = 4, UNROLL = 4;

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__device__ __forceinline__ u32x4 ldg_nt(const bf16_t* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
}
__device__ __forceinline__ bf16x8 as_frag(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

template <int MODE, int MT>
__global__ __launch_bounds__(WAVES * 64) void k_gemv(GemvArgs p, int ntiles) {
    constexpr int T = (MODE == GV_SWIGLU) ? 2 : 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int tile = blockIdx.x * WAVES + wave;   // in units of T 16-row tiles
    if (tile >= ntiles) return;
    const int nchunks = p.K / 64;
    const int per = nchunks / (MODE == GV_PARTIAL ? p.ksplit : 1);
    const int c0 = (MODE == GV_PARTIAL ? blockIdx.y : 0) * per;

    const bf16_t* wrow[T];
#pragma unroll
    for (int t = 0; t < T; ++t) wrow[t] = p.W + (size_t)(tile * T * 16 + t * 16 + fr) * p.K + fg * 16;
    const bf16_t* xrow[MT];
    bool xok[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int m = mt * 16 + fr;
        xok[mt] = m < p.M;
        xrow[mt] = p.x + (size_t)(xok[mt] ? m : 0) * p.ldx + fg * 16;
    }

    f32x4 acc[T][MT];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto body = [&](int c, auto UN) {
        constexpr int U = decltype(UN)::value;
        u32x4 w[U][T][2], xv[U][MT][2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                w[u][t][0] = ldg_nt(wrow[t] + (size_t)(c + u) * 64);
                w[u][t][1] = ldg_nt(wrow[t] + (size_t)(c + u) * 64 + 8);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (xok[mt]) {
                    xv[u][mt][0] = *reinterpret_cast<const u32x4*>(xrow[mt] + (size_t)(c + u) * 64);
                    xv[u][mt][1] = *reinterpret_cast<const u32x4*>(xrow[mt] + (size_t)(c + u) * 64 + 8);
                } else {
                    xv[u][mt][0] = u32x4{0, 0, 0, 0};
                    xv[u][mt][1] = u32x4{0, 0, 0, 0};
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(w[u][t][0]), as_frag(xv[u][mt][0]),
                                                                          acc[t][mt], 0, 0, 0);
                    acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(w[u][t][1]), as_frag(xv[u][mt][1]),
                                                                          acc[t][mt], 0, 0, 0);
                }
    };
    int c = c0;
    const int cend = c0 + per;
    for (; c + UNROLL <= cend; c += UNROLL) body(c, std::integral_constant<int, UNROLL>{});
    for (; c < cend; ++c) body(c, std::integral_constant<int, 1>{});

    // lane owns batch row m = mt*16 + fr and output columns tile*16 + fg*4 + {0..3}
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + fr;
        if (m >= p.M) continue;
        if constexpr (MODE == GV_SWIGLU) {
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float g = rbf(acc[0][mt][r]), u = rbf(acc[1][mt][r]);
                o[r] = rbf(silu_f(g)) * u;
            }
            uint2 v = {pack2(o[0], o[1]), pack2(o[2], o[3])};
            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * (p.N / 2) + tile * 16 + fg * 4) = v;
        } else {
            const int n = tile * 16 + fg * 4;
            float* o = reinterpret_cast<float*>(p.out) +
                       ((size_t)(MODE == GV_PARTIAL ? blockIdx.y : 0) * p.M + m) * p.N + n;
            *reinterpret_cast<float4*>(o) = float4{acc[0][mt][0], acc[0][mt][1], acc[0][mt][2], acc[0][mt][3]};
        }
    }
}

template <int MODE>
int launch_m(hipStream_t s, const GemvArgs& a) {
    const int T = (MODE == GV_SWIGLU) ? 2 : 1;
    const int ntiles = a.N / (16 * T);
    dim3 grid(cdiv(ntiles, WAVES), MODE == GV_PARTIAL ? a.ksplit : 1);
    if (a.M <= 16) hipLaunchKernelGGL((k_gemv<MODE, 1>), grid, dim3(WAVES * 64), 0, s, a, ntiles);
    else hipLaunchKernelGGL((k_gemv<MODE, 2>), grid, dim3(WAVES * 64), 0, s, a, ntiles);
    SR_CHECK_LAUNCH();
    return 0;
}

}  // namespace

int launch_gemv(hipStream_t s, const GemvArgs& a, int mode) {
    if (a.M <= 0) return 0;
    if (a.M > 32 || a.K % 64 != 0 || a.N % 16 != 0) return -22;
    if (mode == GV_SWIGLU && a.N % 32 != 0) return -22;
    if (mode == GV_PARTIAL && (a.ksplit < 1 || (a.K / 64) % a.ksplit != 0)) return -22;
    switch (mode) {
        case GV_PARTIAL: return launch_m<GV_PARTIAL>(s, a);
        case GV_SWIGLU: return launch_m<GV_SWIGLU>(s, a);
        case GV_F32: return launch_m<GV_F32>(s, a);
    }
    return -22;
}
