// Weight-streaming skinny GEMM for autoregressive decode (SURVEY.md 2.3 K12/K13/K16/K17 at M = batch <= 32):
//   out[M,N] = f(x)[M,K] . W[N,K]^T.   HBM-bound: every weight byte is read exactly once per step, straight from
//   HBM into VGPRs (no LDS round trip: the operand is not shared between waves), 16 B per lane, 128 B per row
//   per k-chunk, non-temporal (each CU reads its slice once), deep unroll so that >= 8 KB per wave is in flight.
// Weights are normally stored fragment-ordered (tiled16x64, common.h): one wave instruction then reads 1 KB contiguous
// (+22..28 % throughput over row-major 16 x 64 B fragments); row-major is kept for the op-level tests.
// A wave owns one 16-row tile of W (two tiles -- gate and up -- for the SwiGLU epilogue) over a K slice and feeds it
// to the 16x16x32 MFMA as the A operand; x (tiny) is the B operand, rows >= M read as zero.  The k-slot -> k mapping
// is permuted (lane group g covers k = g*16 .. g*16+15 of each 64-chunk as two MFMA steps) so that a lane's two
// 16-byte loads are contiguous; both operands use the same permutation, so the sum is unchanged.
// K is split over the KP waves of a block (deterministic LDS reduction, fixed order) so that even N = 2048 outputs
// give every CU work, and optionally over gridDim.y blocks (PARTIAL mode: float32 slabs summed by the consumer).
// Fusions (each removes a launch from the 36-layer decode chain):
//   NORM prologue : h = bf16(x + bf16(sum slabs)) (pending residual of the previous down-projection), RMSNorm
//                   (hf:65-79) of h into LDS as the B operand; block 0 writes h back for the later residual add;
//   BIAS epilogue : bf16(acc + bias)            (q/k/v Linear, hf:626-629)
//   RESID epilogue: x = bf16(x + bf16(acc))      (o_proj + residual, hf:744-748) in place
//   SWIGLU        : bf16(bf16(silu(bf16 g)) * bf16 u)   (hf:541-554)
//   F32 + argmax  : float32 logits and a per-block (max, lowest index) pair for the greedy token (hf:1386-1387).
#include "kernels.h"
#include "rownorm.h"
#include <math.h>
#include <stdlib.h>
#include <type_traits>

// (Measured dead ends of round 3 -- RMSNorm folded into / deferred behind the consuming GEMV, 8- and 16-wave blocks, a 3-deep gate/up ring --
// are no longer compiled into the library: tools/experiments/README.md.)

int gemv_prepare_px();

namespace {

// -DSR_GEMV_TIMING (tools/probe_gemv_timeline.py builds its own library with it; never the product build): wave 0 of every block records the
// 100 MHz clock at entry / when its ring is filled and the k loop starts / when the k loop is over / at the end
#ifdef SR_GEMV_TIMING
__device__ long long g_tgv[16384 * 4];
#define TGV(slot) do { if (threadIdx.x == 0) { const int b_ = blockIdx.y * gridDim.x + blockIdx.x; if (b_ < 16384) g_tgv[b_ * 4 + (slot)] = wall_clock64(); } } while (0)
#else
#define TGV(slot)
#endif

// -DSR_EXP_NOX (tools/probe_gemv_masked.py builds its own library with it; never the product build): every x fragment load of the counted loops reads
// chunk 0 again -- L1 hits, WRONG results, the same instruction stream: what would a launch cost if the activations cost no L2 traffic?
#ifdef SR_EXP_NOX
#define SR_XC(c) 0
#else
#define SR_XC(c) (c)
#endif
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__device__ __forceinline__ u32x4 ldg_nt(const bf16_t* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
}
__device__ __forceinline__ bf16x8 as_frag(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// WAVES = 4 : the dispatched configuration -- many small blocks, K split over the 4 waves.
// STAGE     : x (optionally + pending residual slabs, optionally RMS-normalised) is prepared in LDS by the prologue
//             (used for batches <= 4, where the prologue is a few KB per block).
template <int MODE, int MT, int KP, bool STAGE, int WAVES, bool F8 = false>
__global__ __launch_bounds__(WAVES * 64) void k_gemv(GemvArgs p, int ntiles) {
    constexpr int T = (MODE == GV_SWIGLU) ? 2 : 1;     // 16-row W tiles per wave
    constexpr int TPB = WAVES / KP;                    // wave-tiles per block
    constexpr int NT = WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TGV(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (wave-uniform for the compiler too: scalar branches)
    const int fr = lane & 15, fg = lane >> 4;
    const int tp = wave / KP, kp = wave % KP;
    const int tile = (int)blockIdx.x * TPB + tp;            // in units of T tiles
    const bool active = tile < ntiles;
    const int nchunks = p.K / 64;
    const int ks = (MODE == GV_PARTIAL) ? p.ksplit : 1;
    const int per = (nchunks + ks * KP - 1) / (ks * KP);          // uneven split allowed (K/64 = 172 for the 3B MLP)
    const int c0 = min(((MODE == GV_PARTIAL ? blockIdx.y : 0) * KP + kp) * per, nchunks);

    // ---------------------------------------------------------------- weight ring: U chunk slots (16 loads = 16 KB per
    // wave) stay in flight; a slot is refilled right after its MFMAs issue, so the compiler's counted vmcnt only ever
    // waits for the oldest slot.  The first fills do not depend on x: they go out before the prologue's second pass.
    // F8: the stream is the fp8 image (1 KB per 16 x 64 block): ONE 16-byte load per lane and chunk, widened to the two
    // bf16 MFMA operands at consume time (exact), the per-channel scale multiplies the float32 accumulator.
    constexpr int U = ((F8 && STAGE) ? 16 : 8) / T;          // (gate / up at 17..32 rows: 4 -> 184 registers, 2 blocks per CU; x rides the ring in registers when it is not staged)
    constexpr int WL = F8 ? 1 : 2;
    u32x4 w[U][T][WL];
    const int cend = min(c0 + per, nchunks);
    const bf16_t* wrow[T];
    const bool tiled = p.w_tiled != 0;    // fragment-ordered weights [n_tile][chunk][kstep][lane][8] (common.h)
#pragma unroll
    for (int t = 0; t < T; ++t)
        wrow[t] = tiled ? p.W + (size_t)((active ? tile : 0) * T + t) * 16 * p.K + lane * 8
                        : p.W + (size_t)((active ? tile : 0) * T * 16 + t * 16 + fr) * p.K + fg * 16;
    const unsigned char* w8row[T];
#pragma unroll
    for (int t = 0; t < T; ++t) w8row[t] = F8 ? p.W8 + (size_t)((active ? tile : 0) * T + t) * 16 * p.K + lane * 16 : nullptr;
    auto fill_w = [&](int u, int c) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if constexpr (F8) {
                w[u][t][0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w8row[t] + (size_t)c * 1024));
            } else if (tiled) {
                w[u][t][0] = ldg_nt(wrow[t] + (size_t)c * 1024);
                w[u][t][1] = ldg_nt(wrow[t] + (size_t)c * 1024 + 512);
            } else {
                w[u][t][0] = ldg_nt(wrow[t] + (size_t)c * 64);
                w[u][t][1] = ldg_nt(wrow[t] + (size_t)c * 64 + 8);
            }
        }
    };
    // counted form (round 5: fragment-ordered x at 5..32 rows; round 6: every launch of this kernel, the LDS-staged ones of <= 4 rows included): every fill of the ring is
    // UNCONDITIONAL -- the chunk index is clamped to the wave's last chunk (at most U - 1 redundant 1-KB loads per wave, L2 hits), the steady
    // rounds refill every slot and only the peeled last round asks whether a chunk exists.  With a load behind `if (chunk exists)` hipcc's vmcnt
    // bookkeeping collapses to `vmcnt(0)` at the top of every round (see k_gemv32): the whole ring landed before a round's first MFMA.
    const bool counted = p.counted && active && cend > c0;
    auto first_fills = [&]() {
        if (counted) {
#pragma unroll
            for (int u = 0; u < U; ++u) fill_w(u, min(c0 + u, cend - 1));
        } else if (active) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (c0 + u < cend) fill_w(u, c0 + u);
        }
    };
    // epilogue operands (bias / residual) are fetched first: they are the oldest loads in flight, so the epilogue never
    // waits a memory round trip for them
    uint2 ep_bias[MT], ep_res[MT];
    if constexpr (MODE == GV_BIAS || MODE == GV_RESID) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            ep_bias[mt] = uint2{0, 0};
            ep_res[mt] = uint2{0, 0};
            const int m = mt * 16 + fr, n = tile * 16 + fg * 4;
            if (active && kp == 0 && m < p.M) {
                if (MODE == GV_BIAS && p.bias) ep_bias[mt] = *reinterpret_cast<const uint2*>(p.bias + n);
                if (MODE == GV_RESID) ep_res[mt] = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(p.out) + (size_t)m * p.ldo + n);
            }
        }
    }

    // ---------------------------------------------------------------- STAGE prologue: x -> LDS
    // layout: xn[m][K + 8] bf16 (row pad 16 B: conflict-free ds_read_b128 across rows); red[(K/512)][32] float after it.
    // The in-block K reduction later reuses the xn region.
    const int xs = p.K + 8;
    bf16_t* xn = reinterpret_cast<bf16_t*>(smem);
    const size_t xn_bytes = STAGE ? ((size_t)p.M * xs * 2 + 15) / 16 * 16 : 0;
    float* red = reinterpret_cast<float*>(smem + xn_bytes);
    if constexpr (STAGE) {
        const int nch = p.K / 8;                      // 16-byte chunks per row; nch % 64 == 0 (K % 512 == 0)
        const int nseg = nch / 64;                    // 64-chunk segments per row: one wave-iteration each
        const bool norm = p.norm_w != nullptr;
        // pass 1: h = x (+ pending residual) -> LDS (bf16); per-(row, segment) sum of squares
        auto finish = [&](int m, int c, uint4 u, const float (&a)[8]) {
            float v[8] = {lo16(u.x), hi16(u.x), lo16(u.y), hi16(u.y), lo16(u.z), hi16(u.z), lo16(u.w), hi16(u.w)};
            if (p.n_slabs > 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = rbf(v[e] + rbf(a[e]));
                u = uint4{pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
                if (blockIdx.x == 0 && blockIdx.y == 0) *reinterpret_cast<uint4*>(p.x_out + (size_t)m * p.ldx + c * 8) = u;
            }
            *reinterpret_cast<uint4*>(xn + (size_t)m * xs + c * 8) = u;
            if (norm) {
                float ss = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) ss += v[e] * v[e];
                ss = wave_sum(ss);                    // the 64 tasks of a wave-iteration share (m, segment)
                if (lane == 0) red[(c >> 6) * 32 + m] = ss;
            }
        };
        const int total = p.M * nch;
        const bool pre = total <= NT && p.n_slabs <= 2;
        uint4 wu_pre = uint4{0, 0, 0, 0};
        if (pre) {
            // one task per thread (batch 1 at K = 2048): its loads go out BEFORE the first weight fills -- memory
            // returns in order, so x issued behind 16 KB of weights would wait for them, and weights issued behind the
            // consumption of x would start a DRAM round trip late
            const int m = tid / nch, c = tid % nch;
            const bool on = tid < total;
            uint4 u = uint4{0, 0, 0, 0};
            float4 s0[2] = {float4{0.f, 0.f, 0.f, 0.f}, float4{0.f, 0.f, 0.f, 0.f}}, s1[2] = {s0[0], s0[0]};
            if (on) {
                u = *reinterpret_cast<const uint4*>(p.x + (size_t)m * p.ldx + c * 8);
                if (norm) wu_pre = *reinterpret_cast<const uint4*>(p.norm_w + c * 8);
#pragma unroll
                for (int sl = 0; sl < 2; ++sl)
                    if (sl < p.n_slabs) {
                        const float4* pp = reinterpret_cast<const float4*>(p.slabs + ((size_t)sl * p.M + m) * p.K + c * 8);
                        s0[sl] = pp[0];
                        s1[sl] = pp[1];
                    }
            }
            first_fills();
            if (on) {
                float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sl = 0; sl < 2; ++sl)      // slab order 0, 1: same sums as the general path
                    if (sl < p.n_slabs) {
                        a[0] += s0[sl].x; a[1] += s0[sl].y; a[2] += s0[sl].z; a[3] += s0[sl].w;
                        a[4] += s1[sl].x; a[5] += s1[sl].y; a[6] += s1[sl].z; a[7] += s1[sl].w;
                    }
                finish(m, c, u, a);
            }
        } else {
            for (int task = tid; task < total; task += NT) {
                const int m = task / nch, c = task % nch;
                const uint4 u = *reinterpret_cast<const uint4*>(p.x + (size_t)m * p.ldx + c * 8);
                float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                for (int sl = 0; sl < p.n_slabs; ++sl) {
                    const float4* pp = reinterpret_cast<const float4*>(p.slabs + ((size_t)sl * p.M + m) * p.K + c * 8);
                    float4 p0 = pp[0], p1 = pp[1];
                    a[0] += p0.x; a[1] += p0.y; a[2] += p0.z; a[3] += p0.w;
                    a[4] += p1.x; a[5] += p1.y; a[6] += p1.z; a[7] += p1.w;
                }
                finish(m, c, u, a);
            }
            first_fills();
        }
        __syncthreads();
        if (norm) {
            // pass 2: xn = bf16(w * bf16(h * rs)) in place  (hf:65-79)
            for (int task = tid; task < p.M * nch; task += NT) {
                const int m = task / nch, c = task % nch;
                float tot = 0.f;
                for (int sg = 0; sg < nseg; ++sg) tot += red[sg * 32 + m];
                const float rs = 1.0f / sqrtf(tot / (float)p.K + p.eps);
                uint4 u = *reinterpret_cast<const uint4*>(xn + (size_t)m * xs + c * 8);
                const uint4 wu = pre ? wu_pre : *reinterpret_cast<const uint4*>(p.norm_w + c * 8);
                float v[8] = {lo16(u.x), hi16(u.x), lo16(u.y), hi16(u.y), lo16(u.z), hi16(u.z), lo16(u.w), hi16(u.w)};
                float wv[8] = {lo16(wu.x), hi16(wu.x), lo16(wu.y), hi16(wu.y), lo16(wu.z), hi16(wu.z), lo16(wu.w), hi16(wu.w)};
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = wv[e] * rbf(v[e] * rs);
                *reinterpret_cast<uint4*>(xn + (size_t)m * xs + c * 8) =
                    uint4{pack2(o[0], o[1]), pack2(o[2], o[3]), pack2(o[4], o[5]), pack2(o[6], o[7])};
            }
            __syncthreads();
        }
    }

    f32x4 acc[T][MT];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (active) {
        const bf16_t* xrow[MT];
        bool xok[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = mt * 16 + fr;
            xok[mt] = m < p.M;
            const int mm = xok[mt] ? m : 0;
            xrow[mt] = (STAGE ? xn + (size_t)mm * xs : p.x + (size_t)mm * p.ldx) + fg * 16;
        }
        // x fragments: from LDS (STAGE) they are read at consume time; from global they ride the ring with the weights
        constexpr int XU = STAGE ? 1 : U;
        u32x4 xv[XU][MT][2];
        auto fill_x = [&](int u, int c) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (!STAGE && p.x_tiled) {      // fragment-ordered x: 1 KB contiguous per wave instruction; rows >= M of the 16-row group only
                                                // feed output columns that are never stored
                    const bf16_t* xt = p.x + ((size_t)(mt * nchunks + c) * 2) * 512 + lane * 8;
                    xv[u][mt][0] = *reinterpret_cast<const u32x4*>(xt);
                    xv[u][mt][1] = *reinterpret_cast<const u32x4*>(xt + 512);
                } else if (xok[mt]) {
                    xv[u][mt][0] = *reinterpret_cast<const u32x4*>(xrow[mt] + (size_t)c * 64);
                    xv[u][mt][1] = *reinterpret_cast<const u32x4*>(xrow[mt] + (size_t)c * 64 + 8);
                } else {
                    xv[u][mt][0] = u32x4{0, 0, 0, 0};
                    xv[u][mt][1] = u32x4{0, 0, 0, 0};
                }
            }
        };
        auto mfmas = [&](int u, int xs_) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                u32x4 w0, w1;
                if constexpr (F8) {
                    const u32x4 q = w[u][t][0];
                    uint32_t d[8];
                    f8x4_to_bf16(q[0], d[0], d[1]); f8x4_to_bf16(q[1], d[2], d[3]);
                    f8x4_to_bf16(q[2], d[4], d[5]); f8x4_to_bf16(q[3], d[6], d[7]);
                    w0 = u32x4{d[0], d[1], d[2], d[3]};
                    w1 = u32x4{d[4], d[5], d[6], d[7]};
                } else {
                    w0 = w[u][t][0];
                    w1 = w[u][t][WL - 1];
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(w0), as_frag(xv[xs_][mt][0]), acc[t][mt], 0, 0, 0);
                    acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(w1), as_frag(xv[xs_][mt][1]), acc[t][mt], 0, 0, 0);
                }
            }
        };
        if (counted) {
            const int rounds = (cend - c0 + U - 1) / U;
            // x without a condition around the loads: both layouts through one base pointer + two strides (fragment-ordered: 1 KB per 16-row group and
            // chunk, halves 512 elements apart; row-major: rows >= M read row 0 -- valid memory, their output columns are never stored)
            const bf16_t* xb[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xb[mt] = p.x_tiled ? p.x + (size_t)mt * nchunks * 1024 + lane * 8 : xrow[mt];
            const size_t x_cs = p.x_tiled ? 1024 : 64, x_h = p.x_tiled ? 512 : 8;
            auto fill_xt = [&](int u, int c) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    xv[STAGE ? 0 : u][mt][0] = *reinterpret_cast<const u32x4*>(xb[mt] + (size_t)SR_XC(c) * x_cs);
                    xv[STAGE ? 0 : u][mt][1] = *reinterpret_cast<const u32x4*>(xb[mt] + (size_t)SR_XC(c) * x_cs + x_h);
                }
            };
            if constexpr (!STAGE) {
                // x in front of the weights in the first ring (in-order return: x comes from L2, W from HBM -- see below)
#pragma unroll
                for (int u = 0; u < U; ++u) fill_xt(u, min(c0 + u, cend - 1));
                first_fills();
            }
            TGV(1);
            for (int r = 0; r + 1 < rounds; ++r) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int cn = min(c0 + (r + 1) * U + u, cend - 1);
                    if constexpr (STAGE) fill_x(0, c0 + r * U + u);      // (x fragments from LDS at consume time)
                    mfmas(u, STAGE ? 0 : u);
                    fill_w(u, cn);
                    if constexpr (!STAGE) fill_xt(u, cn);
                }
            }
            const int last = c0 + (rounds - 1) * U;
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (last + u < cend) {
                    if constexpr (STAGE) fill_x(0, last + u);
                    mfmas(u, STAGE ? 0 : u);
                }
        }
        if (!counted) {
        if constexpr (!STAGE) {
            // x BEFORE the weights: a wave's loads return in issue order, x comes from L2 (the previous launch wrote it) and W from HBM.
            // Behind the weights the x loads of the first ring only started to arrive when ALL of it had landed (launches whose whole K
            // fits the first ring -- qkv, o_proj: 3.2 us for W, then 1.3 us of x through the CU's load path, tools/probe_gemv_timeline.py);
            // in front of them they travel during the HBM latency.
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (c0 + u < cend) fill_x(u, c0 + u);
        }
        if constexpr (!STAGE) first_fills();
        TGV(1);
        for (int c = c0; c < cend; c += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (c + u < cend) {
                    constexpr int xu_static = 0;
                    const int xu = STAGE ? xu_static : u;
                    if constexpr (STAGE) fill_x(0, c + u);
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        u32x4 w0, w1;
                        if constexpr (F8) {
                            const u32x4 q = w[u][t][0];
                            uint32_t d[8];
                            f8x4_to_bf16(q[0], d[0], d[1]); f8x4_to_bf16(q[1], d[2], d[3]);
                            f8x4_to_bf16(q[2], d[4], d[5]); f8x4_to_bf16(q[3], d[6], d[7]);
                            w0 = u32x4{d[0], d[1], d[2], d[3]};
                            w1 = u32x4{d[4], d[5], d[6], d[7]};
                        } else {
                            w0 = w[u][t][0];
                            w1 = w[u][t][WL - 1];
                        }
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(w0), as_frag(xv[STAGE ? 0 : u][mt][0]), acc[t][mt], 0, 0, 0);
                            acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(w1), as_frag(xv[STAGE ? 0 : u][mt][1]), acc[t][mt], 0, 0, 0);
                        }
                    }
                    (void)xu;
                    if (c + U + u < cend) {
                        fill_w(u, c + U + u);
                        if constexpr (!STAGE) fill_x(u, c + U + u);
                    }
                }
            }
        }
        }      // !counted
    }

    TGV(2);
    // ---------------------------------------------------------------- in-block K reduction (fixed order kp = 1, 2, 3)
    // reuses the staged-x region once every wave is done reading it
    f32x4* rbuf = reinterpret_cast<f32x4*>(STAGE ? smem : reinterpret_cast<unsigned char*>(red + 128));   // [TPB][KP-1][T*MT][64]
    if constexpr (KP > 1) {
        if constexpr (STAGE) __syncthreads();
        if (kp > 0) {
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    rbuf[((tp * (KP - 1) + (kp - 1)) * (T * MT) + t * MT + mt) * 64 + lane] = acc[t][mt];
        }
        __syncthreads();
        if (kp == 0) {
#pragma unroll
            for (int k = 1; k < KP; ++k)
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        f32x4 o = rbuf[((tp * (KP - 1) + (k - 1)) * (T * MT) + t * MT + mt) * 64 + lane];
                        acc[t][mt][0] += o[0]; acc[t][mt][1] += o[1]; acc[t][mt][2] += o[2]; acc[t][mt][3] += o[3];
                    }
        }
    }

    // ---------------------------------------------------------------- epilogue (wave kp == 0 of each tile)
    // lane owns batch row m = mt*16 + fr and output columns tile*16 + fg*4 + {0..3}
    if constexpr (F8) {
        if (active && kp == 0) {        // per-output-channel scale of the fp8 weights (engine row order, gate / up interleaved)
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const float4 sc = *reinterpret_cast<const float4*>(p.w_scale + (size_t)(tile * T + t) * 16 + fg * 4);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    acc[t][mt][0] *= sc.x; acc[t][mt][1] *= sc.y; acc[t][mt][2] *= sc.z; acc[t][mt][3] *= sc.w;
                }
            }
        }
    }
    float bestv[MT];
    int besti[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { bestv[mt] = -INFINITY; besti[mt] = 0x7fffffff; }
    if (active && kp == 0) {
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
                bf16_t* ob = reinterpret_cast<bf16_t*>(p.out);
                *reinterpret_cast<uint2*>(ob + (p.out_tiled ? tiled_offset((size_t)m, (size_t)(tile * 16 + fg * 4), (size_t)(p.N / 2))
                                                            : (size_t)m * (p.N / 2) + tile * 16 + fg * 4)) = v;
            } else if constexpr (MODE == GV_BIAS || MODE == GV_RESID) {
                const int n = tile * 16 + fg * 4;
                bf16_t* optr = reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.ldo + n;
                float o[4] = {acc[0][mt][0], acc[0][mt][1], acc[0][mt][2], acc[0][mt][3]};
                if (MODE == GV_BIAS && p.bias) {
                    const uint2 b = ep_bias[mt];
                    o[0] += lo16(b.x); o[1] += hi16(b.x); o[2] += lo16(b.y); o[3] += hi16(b.y);
                }
                if constexpr (MODE == GV_RESID) {
                    const uint2 rv = ep_res[mt];
                    o[0] = lo16(rv.x) + rbf(o[0]); o[1] = hi16(rv.x) + rbf(o[1]);
                    o[2] = lo16(rv.y) + rbf(o[2]); o[3] = hi16(rv.y) + rbf(o[3]);
                }
                const uint2 ov = uint2{pack2(o[0], o[1]), pack2(o[2], o[3])};
                *reinterpret_cast<uint2*>(optr) = ov;
            } else {
                const int n = tile * 16 + fg * 4;
                const size_t oi = ((size_t)(MODE == GV_PARTIAL ? blockIdx.y : 0) * p.M + m) * p.N + n;
                const float4 ov = float4{acc[0][mt][0], acc[0][mt][1], acc[0][mt][2], acc[0][mt][3]};
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + oi) = ov;
                if constexpr (MODE == GV_F32) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (acc[0][mt][r] > bestv[mt]) { bestv[mt] = acc[0][mt][r]; besti[mt] = n + r; }
                }
            }
        }
    }
    if constexpr (MODE == GV_F32) {
        // fused greedy argmax: (max, lowest index) of this block's logits per batch row -> amax[m][blockIdx.x]
        if (p.amax_val) {
            if constexpr (STAGE) __syncthreads();      // KP == 1 here; the staged x is dead, reuse its region
            float* av = reinterpret_cast<float*>(rbuf);                 // [WAVES][32]
            int* ai = reinterpret_cast<int*>(av + WAVES * 32);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float bv = bestv[mt];
                int bi = besti[mt];
#pragma unroll
                for (int o = 16; o <= 32; o <<= 1) {     // combine the 4 lane groups that hold the same batch row
                    const float ov = __shfl_xor(bv, o, 64);
                    const int oi = __shfl_xor(bi, o, 64);
                    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
                }
                if (fg == 0) { av[wave * 32 + mt * 16 + fr] = bv; ai[wave * 32 + mt * 16 + fr] = bi; }
            }
            __syncthreads();
            if (tid < p.M) {
                float bv = av[tid];
                int bi = ai[tid];
                for (int w = 1; w < WAVES; ++w) {
                    const float ov = av[w * 32 + tid];
                    const int oi = ai[w * 32 + tid];
                    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
                }
                p.amax_val[(size_t)tid * gridDim.x + blockIdx.x] = bv;
                p.amax_idx[(size_t)tid * gridDim.x + blockIdx.x] = bi;
            }
        }
    }
    TGV(3);
}

// ---------------------------------------------------------------------------------------------- 17..32 rows, x STATIONARY in registers (round 6)
// Why: at 32 rows every wave of k_gemv re-reads as many bytes of x from L2 as it streams weights from HBM, and a CU's load path carries both.  On the whole chip
// that costs the gate/up launch ~2 us of 19; on the scheduler's 160-CU decode stream (5 of 8 CUs per shader engine while an admission is staged: two thirds of the
// headline's decode steps) it is THE limit -- tools/probe_gemv_masked.py, profiles/r06_probe_gemv_masked.jsonl: gate/up 24.9 us with the x loads, 18.0 with them
// pinned to L1.  A wave of this launch only ever needs ITS K quarter of x (32 rows x K / 4 bf16 <= 32 KB = 128 registers per lane): here it loads that quarter ONCE,
// keeps it in registers, and walks weight tiles it pulls from a ticket counter (persistent: one block per CU -- ~300 registers per wave --, blocks that find the
// counter exhausted leave: the launch balances itself on any CU count).  Per tile the arithmetic is k_gemv<SWIGLU, 2, 4, false, 4>'s -- the same 4 waves = 4 K
// quarters, the same chunk order and MFMA order per wave, the same in-block reduction order, the same epilogue -- so the result is bit-identical
// (tests/test_gpu_round6.py).  The ring holds a wave's WHOLE K slice (U = PER chunks): a slot is refilled with the NEXT tile's chunk right after it is consumed,
// so the weight stream never ramps between tiles.  (A first version kept x in LDS, 128 KB per block by LDS-DMA: its up-front copy made the first tile cost twice
// a tile's bytes per CU -- 7.6 us of prologue -- and lost to the streaming kernel on the whole chip.)  On the whole chip even this form is 1 % slower per step
// than the streaming kernel (688 tiles over 256 persistent blocks are 2.7 rounds that quantise to 3): the engine takes it on a CU-limited stream only
// (sr_rows_set_cus; bit 2 of SR_GEMV_XLDS forces it everywhere).
// Tickets: ONE counter word saturates at ~88 dequeues per microsecond on this chip (MI355X guide, "dequeue") -- 944 tickets would cost the launch 10 us.  The
// counter is sharded 8 ways: block b draws from shard b % 8 (its XCD under round-robin dispatch; nothing depends on that being true), whose k-th ticket is tile
// 8 k + shard.  A shard is served by every 8th block, on the whole chip and on a CU-masked stream alike (the first blocks dispatched are 0 .. n - 1), so the
// shards stay balanced without stealing; a block leaves when its shard is exhausted.
template <bool F8, int PER>
__global__ __launch_bounds__(256) void k_gemv_px(GemvArgs p, int ntiles) {
    constexpr int T = 2, MT = 2, KP = 4, U = PER;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TGV(0);
    const int tid = threadIdx.x, lane = tid & 63, kp = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int nchunks = p.K / 64;                          // = 4 PER (K = 256 PER <= 2048, checked by the launcher): every wave owns PER chunks = its whole ring
    constexpr int per = PER;
    const int c0 = kp * per;
    unsigned* shard = p.px_counter + (blockIdx.x & 7) * 16;                                  // (64 bytes apart)
    const int sh = blockIdx.x & 7;
    f32x4* rbuf = reinterpret_cast<f32x4*>(smem);                                            // [2 parities][KP - 1][T * MT][64]
    int* s_tile = reinterpret_cast<int*>(smem + 2 * (KP - 1) * T * MT * 64 * 16);            // [4]: tile of iteration i at s_tile[i & 3]
    // this wave's K quarter of x: chunk c0 + u, 16-row group mt, k-step half h -- loaded once, kept for every tile (rows >= M of the second group only feed output
    // columns that are never stored)
    u32x4 xr[U][MT][2];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const bf16_t* xt = p.x + ((size_t)(mt * nchunks + c0 + u) * 2) * 512 + lane * 8;
            xr[u][mt][0] = *reinterpret_cast<const u32x4*>(xt);
            xr[u][mt][1] = *reinterpret_cast<const u32x4*>(xt + 512);
        }
    // tickets of iterations 0 and 1 (wave 0), published through LDS
    if (tid == 0) {
        const unsigned k0 = atomicAdd(shard, 2u);
        s_tile[0] = (int)(k0 * 8u) + sh;
        s_tile[1] = (int)((k0 + 1u) * 8u) + sh;
    }
    __syncthreads();
    int tile = s_tile[0];
    u32x4 w[U][T][F8 ? 1 : 2];
    auto fill_w = [&](int u, int tile, int c) {           // weight chunk c of `tile` into ring slot u (fragment-ordered weights only)
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if constexpr (F8) {
                w[u][t][0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p.W8 + (size_t)(tile * T + t) * 16 * p.K + lane * 16 + (size_t)c * 1024));
            } else {
                const bf16_t* wr = p.W + (size_t)(tile * T + t) * 16 * p.K + lane * 8 + (size_t)c * 1024;
                w[u][t][0] = ldg_nt(wr);
                w[u][t][1] = ldg_nt(wr + 512);
            }
        }
    };
    if (tile < ntiles) {
#pragma unroll
        for (int u = 0; u < U; ++u) fill_w(u, tile, c0 + u);
    }
    TGV(1);
    // one tile: consume the ring slot by slot (x from this wave's registers), refilling every slot with the SAME chunk of tile `nxt` when REFILL.  Two straight-line
    // instantiations (steady state / last tile of the block) instead of a refill behind a condition: hipcc then counts its vmcnt waits (see k_gemv32)
    auto run_tile = [&](auto refill_, int nxt, f32x4 (&acc)[T][MT]) {
        constexpr bool REFILL = decltype(refill_)::value;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (u < per) {
                const int c = c0 + u;
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    u32x4 w0, w1;
                    if constexpr (F8) {
                        const u32x4 q = w[u][t][0];
                        uint32_t d[8];
                        f8x4_to_bf16(q[0], d[0], d[1]); f8x4_to_bf16(q[1], d[2], d[3]);
                        f8x4_to_bf16(q[2], d[4], d[5]); f8x4_to_bf16(q[3], d[6], d[7]);
                        w0 = u32x4{d[0], d[1], d[2], d[3]};
                        w1 = u32x4{d[4], d[5], d[6], d[7]};
                    } else {
                        w0 = w[u][t][0];
                        w1 = w[u][t][F8 ? 0 : 1];
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(w0), as_frag(xr[u][mt][0]), acc[t][mt], 0, 0, 0);
                        acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(w1), as_frag(xr[u][mt][1]), acc[t][mt], 0, 0, 0);
                    }
                }
                if constexpr (REFILL) fill_w(u, nxt, c);
            }
        }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    for (int it = 0;; ++it) {
        if (tile >= ntiles) break;                        // (block-uniform: every wave read the same s_tile word)
        unsigned ticket = 0;
        if (tid == 0) ticket = atomicAdd(shard, 1u) * 8u + (unsigned)sh;      // the tile of iteration it + 2: requested now, stored behind the k loop, published by the barrier
        const int next = s_tile[(it + 1) & 3];
        f32x4 acc[T][MT];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (next < ntiles) run_tile(T_{}, next, acc);
        else run_tile(F_{}, 0, acc);
        if (tid == 0) s_tile[(it + 2) & 3] = (int)ticket;
        // in-block K reduction, fixed order kp = 1, 2, 3 (k_gemv's); the buffer alternates so that a fast wave's partials of the next tile never meet a slow reader
        f32x4* rb = rbuf + (size_t)(it & 1) * (KP - 1) * T * MT * 64;
        if (kp > 0) {
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) rb[((kp - 1) * (T * MT) + t * MT + mt) * 64 + lane] = acc[t][mt];
        }
        __syncthreads();
        if (kp == 0) {
#pragma unroll
            for (int k = 1; k < KP; ++k)
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const f32x4 o = rb[((k - 1) * (T * MT) + t * MT + mt) * 64 + lane];
                        acc[t][mt][0] += o[0]; acc[t][mt][1] += o[1]; acc[t][mt][2] += o[2]; acc[t][mt][3] += o[3];
                    }
            if constexpr (F8) {
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const float4 sc = *reinterpret_cast<const float4*>(p.w_scale + (size_t)(tile * T + t) * 16 + fg * 4);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        acc[t][mt][0] *= sc.x; acc[t][mt][1] *= sc.y; acc[t][mt][2] *= sc.z; acc[t][mt][3] *= sc.w;
                    }
                }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = mt * 16 + fr;
                if (m >= p.M) continue;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float g = rbf(acc[0][mt][r]), u = rbf(acc[1][mt][r]);
                    o[r] = rbf(silu_f(g)) * u;
                }
                bf16_t* ob = reinterpret_cast<bf16_t*>(p.out);
                *reinterpret_cast<uint2*>(ob + (p.out_tiled ? tiled_offset((size_t)m, (size_t)(tile * 16 + fg * 4), (size_t)(p.N / 2))
                                                            : (size_t)m * (p.N / 2) + tile * 16 + fg * 4)) = uint2{pack2(o[0], o[1]), pack2(o[2], o[3])};
            }
        }
        tile = next;
    }
    TGV(2);
    // the last block out re-arms the counters for the next launch (stream order makes the zeros visible to it)
    __syncthreads();
    if (tid == 0) {
        if (atomicAdd(p.px_counter + 8 * 16, 1u) == gridDim.x - 1) {
#pragma unroll
            for (int i = 0; i <= 8; ++i) p.px_counter[i * 16] = 0u;
        }
    }
    TGV(3);
}

// ---------------------------------------------------------------------------------------------- batches 17..32
// 32x32x16 MFMA variant: a wave owns a 32-row weight tile (two 16-row tiles of the fragment-ordered layout) and all
// 32 batch rows are ONE B operand, so x is read once per 32 weight rows instead of once per 16 (at M = 32 the x
// fragments otherwise cost twice the weight bytes on the CU's load path).  D[i = weight row][j = batch row]:
// lane holds batch row l & 31 and weight rows (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5) -> 4 consecutive outputs per
// register quad, and with the gate/up interleave rows r and r + 16 (gate / up of one column) sit in the same lane.
typedef __attribute__((ext_vector_type(16))) float f32x16;

// cross-wave reduction buffer of the 32-row-tile kernels: a lane's 16 accumulators as four 16-byte quads, quad-major ([slot][quad][lane]) -- with
// the lane-major f32x16 layout of rounds 2-4 (64 B per lane) every fourth lane of a ds_write_b128 / ds_read_b128 group shared its banks: PMC round 5,
// SQ_LDS_BANK_CONFLICT = 0.75 of the LDS-array cycles of the 32-row down-projection (profiles/r05_pmc_lds_all.txt).  Same values, same sum order.
__device__ __forceinline__ void rb_store(unsigned char* smem, int slot, int lane, const f32x16& a) {
    float4* b = reinterpret_cast<float4*>(smem) + (size_t)slot * 256 + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) b[q * 64] = float4{a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
}
__device__ __forceinline__ void rb_add(const unsigned char* smem, int slot, int lane, f32x16& a) {
    const float4* b = reinterpret_cast<const float4*>(smem) + (size_t)slot * 256 + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 o = b[q * 64];
        a[4 * q] += o.x; a[4 * q + 1] += o.y; a[4 * q + 2] += o.z; a[4 * q + 3] += o.w;
    }
}

// AW (round 5, "all refills unconditional"): hipcc's vmcnt bookkeeping gives up on the ring loop as written below -- every refill sits behind `if (chunk
// exists)`, and at the loop header the counts of the entry edge and the back edge merge to "unknown": the generated code waits for vmcnt(0) at the top of
// every round (disassembly, round 5), i.e. the whole ring lands before the first MFMA of a round and the refills of a round only start to arrive when the
// round is over -- a ring in name only.  Here the steady rounds refill EVERY slot (chunk index clamped to the wave's last chunk: at most U - 1 redundant
// 1-KB loads per wave, L2 hits) and only the peeled last round asks whether a chunk exists -- no load sits behind a condition, the waits are counted.
template <int MODE, int KP, bool AW>
__global__ __launch_bounds__(256) void k_gemv32(GemvArgs p, int ntiles32) {
    constexpr int WAVES = 4, TPB = WAVES / KP, U = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TGV(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = AW ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6;      // (AW: the chunk range must be wave-uniform for the compiler too)
    const int fr = lane & 15, half = (lane >> 4) & 1, kg = lane >> 5, m = lane & 31;
    const int tp = wave / KP, kp = wave % KP;
    const int tile = blockIdx.x * TPB + tp;
    const bool active = tile < ntiles32;
    const int nchunks = p.K / 64;
    const int ks = (MODE == GV_PARTIAL) ? p.ksplit : 1;
    const int per = (nchunks + ks * KP - 1) / (ks * KP);
    const int c0 = min(((MODE == GV_PARTIAL ? blockIdx.y : 0) * KP + kp) * per, nchunks);
    const int cend = min(c0 + per, nchunks);
    const int t16 = (active ? tile : 0) * 2 + half;             // 16-row tile index of this lane's weight row
    const bf16_t* wbase = p.w_tiled ? p.W + (size_t)t16 * nchunks * 1024 + kg * 512 + fr * 8
                                    : p.W + (size_t)(t16 * 16 + fr) * p.K + kg * 8;
    const size_t w_c = p.w_tiled ? 1024 : 64, w_s = p.w_tiled ? 128 : 16;   // element strides per chunk / per k16 step
    const bool xok = m < p.M;
    // fragment-ordered x (x_tiled): element (row m, k = c*64 + st*16 + kg*8 + e) sits at tile (m / 16, c), k-step half kg, lane st*16 + m % 16
    const bf16_t* xbase = p.x_tiled ? p.x + ((size_t)(m >> 4) * nchunks * 2 + kg) * 512 + (m & 15) * 8
                                    : p.x + (size_t)(xok ? m : 0) * p.ldx + kg * 8;
    const size_t x_c = p.x_tiled ? 1024 : 64, x_s = p.x_tiled ? 128 : 16;

    u32x4 w[U][4], xv[U][4];
    auto fill_w = [&](int u, int c) {
#pragma unroll
        for (int st = 0; st < 4; ++st) w[u][st] = ldg_nt(wbase + (size_t)c * w_c + st * w_s);
    };
    auto fill_x = [&](int u, int c) {
#pragma unroll
        for (int st = 0; st < 4; ++st)
            xv[u][st] = (xok || p.x_tiled) ? *reinterpret_cast<const u32x4*>(xbase + (size_t)c * x_c + st * x_s) : u32x4{0, 0, 0, 0};
    };
    auto fill = [&](int u, int c) { fill_w(u, c); fill_x(u, c); };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if constexpr (AW) {
        const int n = cend - c0;
        if (active && n > 0) {
            // (rows >= M of an un-tiled x read row 0 -- valid memory; their output columns are never stored)
            auto fill_a = [&](int u, int c) {
                c = min(c, cend - 1);
#pragma unroll
                for (int st = 0; st < 4; ++st) xv[u][st] = *reinterpret_cast<const u32x4*>(xbase + (size_t)SR_XC(c) * x_c + st * x_s);
#pragma unroll
                for (int st = 0; st < 4; ++st) w[u][st] = ldg_nt(wbase + (size_t)c * w_c + st * w_s);
            };
            const int rounds = (n + U - 1) / U;
#pragma unroll
            for (int u = 0; u < U; ++u) fill_a(u, c0 + u);
            TGV(1);
            for (int r = 0; r + 1 < rounds; ++r) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
#pragma unroll
                    for (int st = 0; st < 4; ++st)
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(w[u][st]), as_frag(xv[u][st]), acc, 0, 0, 0);
                    fill_a(u, c0 + (r + 1) * U + u);
                }
            }
            const int last = c0 + (rounds - 1) * U;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (last + u < cend) {
#pragma unroll
                    for (int st = 0; st < 4; ++st)
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(w[u][st]), as_frag(xv[u][st]), acc, 0, 0, 0);
                }
            }
        }
    } else
    if (active) {
        // first ring: x (L2) in front of the weights (HBM) -- loads return in issue order, see k_gemv
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (c0 + u < cend) fill_x(u, c0 + u);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (c0 + u < cend) fill_w(u, c0 + u);
        TGV(1);
        for (int c = c0; c < cend; c += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (c + u < cend) {
#pragma unroll
                    for (int st = 0; st < 4; ++st)
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(w[u][st]), as_frag(xv[u][st]), acc, 0, 0, 0);
                    if (c + U + u < cend) fill(u, c + U + u);
                }
            }
        }
    }
    TGV(2);
    if constexpr (KP > 1) {
        if (kp > 0) rb_store(smem, tp * (KP - 1) + (kp - 1), lane, acc);          // [TPB][KP-1] slots
        __syncthreads();
        if (kp == 0) {
#pragma unroll
            for (int k = 1; k < KP; ++k) rb_add(smem, tp * (KP - 1) + (k - 1), lane, acc);
        }
    }
    float bestv = -INFINITY;
    int besti = 0x7fffffff;
    if (active && kp == 0 && xok) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int nl = 8 * g4 + 4 * kg;                      // first of this quad's 4 consecutive weight rows
            if constexpr (MODE == GV_SWIGLU) {
                if (g4 < 2) {
                    float o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float g = rbf(acc[g4 * 4 + r]), u = rbf(acc[(g4 + 2) * 4 + r]);
                        o[r] = rbf(silu_f(g)) * u;
                    }
                    bf16_t* ob = reinterpret_cast<bf16_t*>(p.out);
                    *reinterpret_cast<uint2*>(ob + (p.out_tiled ? tiled_offset((size_t)m, (size_t)(tile * 16 + nl), (size_t)(p.N / 2))
                                                                : (size_t)m * (p.N / 2) + tile * 16 + nl)) = uint2{pack2(o[0], o[1]), pack2(o[2], o[3])};
                }
            } else if constexpr (MODE == GV_BIAS || MODE == GV_RESID) {
                const int n = tile * 32 + nl;
                bf16_t* optr = reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.ldo + n;
                float o[4] = {acc[g4 * 4], acc[g4 * 4 + 1], acc[g4 * 4 + 2], acc[g4 * 4 + 3]};
                if (MODE == GV_BIAS && p.bias) {
                    const uint2 b = *reinterpret_cast<const uint2*>(p.bias + n);
                    o[0] += lo16(b.x); o[1] += hi16(b.x); o[2] += lo16(b.y); o[3] += hi16(b.y);
                }
                if constexpr (MODE == GV_RESID) {
                    const uint2 rv = *reinterpret_cast<const uint2*>(optr);
                    o[0] = lo16(rv.x) + rbf(o[0]); o[1] = hi16(rv.x) + rbf(o[1]);
                    o[2] = lo16(rv.y) + rbf(o[2]); o[3] = hi16(rv.y) + rbf(o[3]);
                }
                const uint2 ov = uint2{pack2(o[0], o[1]), pack2(o[2], o[3])};
                *reinterpret_cast<uint2*>(optr) = ov;
            } else {
                const int n = tile * 32 + nl;
                const size_t oi = ((size_t)(MODE == GV_PARTIAL ? blockIdx.y : 0) * p.M + m) * p.N + n;
                const float4 ov = float4{acc[g4 * 4], acc[g4 * 4 + 1], acc[g4 * 4 + 2], acc[g4 * 4 + 3]};
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + oi) = ov;
                if constexpr (MODE == GV_F32) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (acc[g4 * 4 + r] > bestv || (acc[g4 * 4 + r] == bestv && n + r < besti)) { bestv = acc[g4 * 4 + r]; besti = n + r; }
                }
            }
        }
    }
    if constexpr (MODE == GV_F32) {
        if (p.amax_val) {                                          // KP == 1: smem is free
            float* av = reinterpret_cast<float*>(smem);            // [WAVES][32]
            int* ai = reinterpret_cast<int*>(av + WAVES * 32);
            const float ov = __shfl_xor(bestv, 32, 64);
            const int oi = __shfl_xor(besti, 32, 64);
            if (ov > bestv || (ov == bestv && oi < besti)) { bestv = ov; besti = oi; }
            if (kg == 0) { av[wave * 32 + m] = bestv; ai[wave * 32 + m] = besti; }
            __syncthreads();
            if (tid < p.M) {
                float bv = av[tid];
                int bi = ai[tid];
                for (int w2 = 1; w2 < WAVES; ++w2) {
                    const float v2 = av[w2 * 32 + tid];
                    const int i2 = ai[w2 * 32 + tid];
                    if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; }
                }
                p.amax_val[(size_t)tid * gridDim.x + blockIdx.x] = bv;
                p.amax_idx[(size_t)tid * gridDim.x + blockIdx.x] = bi;
            }
        }
    }
    TGV(3);
}

// ---------------------------------------------------------------------------------------------- 17..32 rows, split-K down-projection with x STATIONARY (round 6)
// The same idea for the 32-row-tile split-K launch (k_gemv32<GV_PARTIAL, 4>: grid = 64 tiles x 4 K slabs, every wave reads 44 KB of weights and 44 KB of x): a
// block is pinned to ONE K slab, its 4 waves keep their K sixteenth of x (<= PER chunks x 4 k-steps = 176 registers per lane at K = 11008) for every tile they
// walk, and the ring holds a wave's whole slice of a tile (PER slots), refilled with the next tile's chunks as they are consumed.  With one tile per block (the
// whole chip: 256 blocks, 64 tiles x 4 slabs) this is the streaming kernel with a deeper ring; with fewer CUs than units -- the scheduler's CU-masked decode
// stream, 160 CUs -- a block's second tile costs only its weights.  The tiles are dealt STATICALLY: blocks 0 .. L - 1 take part (L = the CU count the host wrote
// to px_counter[160] -- sr_rows_set_cus --, 0 = the grid), block b owns slab b % ksplit and tiles b / ksplit, + L / ksplit, ...; the other blocks leave at once.
// (Tickets, as in k_gemv_px, cannot serve a launch with about one unit per block: a block has to hold the NEXT tile while it works, and the early blocks would
// take every tile.)  Any L gives the same bits: per tile the chunk order, the MFMA order, the in-block reduction order and the slab stores are k_gemv32's.
template <int PER, bool F8 = false>
__global__ __launch_bounds__(256) void k_gemv32_px(GemvArgs p, int ntiles32) {
    constexpr int KP = 4, WL = F8 ? 2 : 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TGV(0);
    const int tid = threadIdx.x, lane = tid & 63, kp = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, half = (lane >> 4) & 1, kg = lane >> 5, m = lane & 31;
    const int nchunks = p.K / 64, ks = p.ksplit;
    const int per = (nchunks + ks * KP - 1) / (ks * KP);                 // <= PER, and every wave owns >= 1 chunk (launcher)
    const unsigned lim = p.px_counter[10 * 16];
    const int nb = (lim != 0u && lim < gridDim.x) ? max((int)lim, ks) : (int)gridDim.x;        // (the grid is a multiple of ksplit, >= ksplit)
    const int per_slab = nb / ks;
    const int slab = blockIdx.x % ks;
    int tile = blockIdx.x / ks;
    if ((int)blockIdx.x >= per_slab * ks || tile >= ntiles32) return;
    const int c0 = (slab * KP + kp) * per, cend = min(c0 + per, nchunks), n = cend - c0;
    // this wave's K sixteenth of x, fragment order (k_gemv32; F8: k_gemv32g's operand order, MFMA step st = 2 j + h reads lane group 2 j + kg of k-step half h):
    // loaded once.  Slots u >= n hold a copy of the last chunk and are never multiplied
    const bf16_t* xbase = F8 ? p.x + ((size_t)(m >> 4) * nchunks * 2) * 512 + (kg * 16 + (m & 15)) * 8
                             : p.x + ((size_t)(m >> 4) * nchunks * 2 + kg) * 512 + (m & 15) * 8;
    u32x4 xr[PER][4], w[PER][WL];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int c = min(c0 + u, cend - 1);
#pragma unroll
        for (int st = 0; st < 4; ++st)
            xr[u][st] = *reinterpret_cast<const u32x4*>(xbase + (size_t)c * 1024 + (F8 ? (st >> 1) * 256 + (st & 1) * 512 : st * 128));
    }
    auto fill_w = [&](int u, int tile, int c) {            // no load behind a condition: the chunk index is clamped instead (counted vmcnt waits, see k_gemv32)
        c = min(c, cend - 1);
        if constexpr (F8) {
            const unsigned char* wb = p.W8 + ((size_t)(tile * 2 + half) * nchunks + c) * 1024 + kg * 256 + fr * 16;
#pragma unroll
            for (int j = 0; j < 2; ++j) w[u][j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wb + j * 512));
        } else {
            const bf16_t* wb = p.W + ((size_t)(tile * 2 + half) * nchunks + c) * 1024 + kg * 512 + fr * 8;
#pragma unroll
            for (int st = 0; st < 4; ++st) w[u][st] = ldg_nt(wb + st * 128);
        }
    };
#pragma unroll
    for (int u = 0; u < PER; ++u) fill_w(u, tile, c0 + u);
    TGV(1);
    auto run_tile = [&](auto refill_, int nxt, f32x16& acc) {
        constexpr bool REFILL = decltype(refill_)::value;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            if (u < n) {
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    u32x4 wop;
                    if constexpr (F8) {
                        const u32x4 q = w[u][st >> 1];
                        uint32_t d[4];
                        f8x4_to_bf16(q[(st & 1) * 2], d[0], d[1]);
                        f8x4_to_bf16(q[(st & 1) * 2 + 1], d[2], d[3]);
                        wop = u32x4{d[0], d[1], d[2], d[3]};
                    } else wop = w[u][st];
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(wop), as_frag(xr[u][st]), acc, 0, 0, 0);
                }
            }
            if constexpr (REFILL) fill_w(u, nxt, c0 + u);
        }
    };
    const bool xok = m < p.M;
    for (int it = 0;; ++it) {
        const int next = tile + per_slab;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        if (next < ntiles32) run_tile(std::true_type{}, next, acc);
        else run_tile(std::false_type{}, 0, acc);
        // in-block K reduction, fixed order kp = 1, 2, 3; the buffer alternates (a fast wave's partials of the next tile never meet a slow reader)
        unsigned char* rb = smem + (size_t)(it & 1) * (KP - 1) * 64 * sizeof(f32x16);
        if (kp > 0) rb_store(rb, kp - 1, lane, acc);
        __syncthreads();
        if (kp == 0) {
#pragma unroll
            for (int k = 1; k < KP; ++k) rb_add(rb, k - 1, lane, acc);
            if (xok) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    float4 o = float4{acc[g4 * 4], acc[g4 * 4 + 1], acc[g4 * 4 + 2], acc[g4 * 4 + 3]};
                    if constexpr (F8) {        // per-output-channel scale of the fp8 weights
                        const float4 sc = *reinterpret_cast<const float4*>(p.w_scale + (size_t)tile * 32 + 8 * g4 + 4 * kg);
                        o.x *= sc.x; o.y *= sc.y; o.z *= sc.z; o.w *= sc.w;
                    }
                    const size_t oi = ((size_t)slab * p.M + m) * p.N + tile * 32 + 8 * g4 + 4 * kg;
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + oi) = o;
                }
            }
        }
        if (next >= ntiles32) break;
        tile = next;
    }
    TGV(3);
}

// ---------------------------------------------------------------------------------------------- 17..32 rows, LM head with x resident in LDS (round 6)
// The head keeps its summation order -- ONE wave per 32-row vocabulary tile, all 32 chunks of K = 2048 in sequence (k_gemv32<GV_F32, 1>) -- so a wave needs ALL of x
// (128 KB), which no register file holds: here the block copies x to LDS once and its 4 waves walk (groups of 4) vocabulary tiles, reading the activation fragments
// with ds_read_b128 (conflict-free: 16 lanes read 256 contiguous bytes) while the weights stream through an 8-slot ring that is refilled across tile boundaries.  On
// 160 CUs the streaming head is bound by the CUs' load path, half of which carries x re-reads from L2 (152 us against 113 on the whole chip); the whole chip is
// HBM-bound either way, so this form is used on a CU-limited stream only (sr_rows_set_cus).  Groups are dealt statically to the first L blocks (L = the hinted CU
// count).  Logits and the per-group arg-max partials are the streaming kernel's, bit for bit.
template <int U>
__global__ __launch_bounds__(256) void k_gemv32_hpx(GemvArgs p, int ntiles32, int ngroups) {
    constexpr int NCH = 32;                     // K = 2048 (launcher)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);                               // [2 row groups][32 chunks][2 halves][64 lanes][8]: x as it lies in memory
    float* av = reinterpret_cast<float*>(smem + (size_t)32 * 2048 * 2);        // [4 waves][32 rows]
    int* ai = reinterpret_cast<int*>(av + 4 * 32);
    TGV(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, half = (lane >> 4) & 1, kg = lane >> 5, m = lane & 31;
    const unsigned lim = p.px_counter[10 * 16];
    const int nb = (lim != 0u && lim < gridDim.x) ? (int)lim : (int)gridDim.x;
    int g = blockIdx.x;
    if (g >= nb || g >= ngroups) return;
    u32x4 w[U][4];
    auto wptr = [&](int tile, int c) { return p.W + ((size_t)(tile * 2 + half) * NCH + c) * 1024 + kg * 512 + fr * 8; };
    auto fill_w = [&](int u, int tile, int c) {
        const bf16_t* wb = wptr(tile, c);
#pragma unroll
        for (int st = 0; st < 4; ++st) w[u][st] = ldg_nt(wb + st * 128);
    };
    int tile = min(g * 4 + wave, ntiles32 - 1);           // (a wave past the last tile repeats it and stores nothing)
#pragma unroll
    for (int u = 0; u < U; ++u) fill_w(u, tile, u);
    // x -> LDS by LDS-DMA (one wave instruction = 1 KB, no register round trip), 32 KB per wave, behind the first ring of weights; the barrier drains it
    {
        typedef __attribute__((address_space(1))) const void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int kb = wave * 32 + i;
            __builtin_amdgcn_global_load_lds((gptr_t)(p.x + (size_t)kb * 512 + lane * 8), (lptr_t)(xs + kb * 512), 16, 0, 0);
        }
    }
    __syncthreads();
    TGV(1);
    const bf16_t* xl = xs + ((size_t)(m >> 4) * NCH * 2 + kg) * 512 + (m & 15) * 8;
    auto run_tile = [&](auto refill_, int cur, int nxt, f32x16& acc) {
        constexpr bool REFILL = decltype(refill_)::value;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int u = c % U;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const u32x4 xv = *reinterpret_cast<const u32x4*>(xl + (size_t)c * 1024 + st * 128);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(w[u][st]), as_frag(xv), acc, 0, 0, 0);
            }
            if (c + U < NCH) fill_w(u, cur, c + U);
            else if constexpr (REFILL) fill_w(u, nxt, c + U - NCH);
            // hipcc otherwise sinks every refill down to its use eight chunks later (one fully unrolled basic block, "fewer live registers"): the ring
            // would be one slot deep -- vmcnt(0) in front of every MFMA (disassembly)
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const bool xok = m < p.M;
    for (;;) {
        const int gnext = g + nb;
        const bool more = gnext < ngroups;
        const int tnext = min(gnext * 4 + wave, ntiles32 - 1);
        const bool active = g * 4 + wave < ntiles32;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        if (more) run_tile(std::true_type{}, tile, tnext, acc);
        else run_tile(std::false_type{}, tile, 0, acc);
        float bestv = -INFINITY;
        int besti = 0x7fffffff;
        if (active && xok) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int n = tile * 32 + 8 * g4 + 4 * kg;
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.N + n) = float4{acc[g4 * 4], acc[g4 * 4 + 1], acc[g4 * 4 + 2], acc[g4 * 4 + 3]};
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (acc[g4 * 4 + r] > bestv || (acc[g4 * 4 + r] == bestv && n + r < besti)) { bestv = acc[g4 * 4 + r]; besti = n + r; }
            }
        }
        if (p.amax_val) {        // the group's arg-max partial, as one block of the streaming launch computes it
            const float ov = __shfl_xor(bestv, 32, 64);
            const int oi = __shfl_xor(besti, 32, 64);
            if (ov > bestv || (ov == bestv && oi < besti)) { bestv = ov; besti = oi; }
            if (kg == 0) { av[wave * 32 + m] = bestv; ai[wave * 32 + m] = besti; }
            __syncthreads();
            if (tid < p.M) {
                float bv = av[tid];
                int bi = ai[tid];
                for (int w2 = 1; w2 < 4; ++w2) {
                    const float v2 = av[w2 * 32 + tid];
                    const int i2 = ai[w2 * 32 + tid];
                    if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; }
                }
                p.amax_val[(size_t)tid * ngroups + g] = bv;
                p.amax_idx[(size_t)tid * ngroups + g] = bi;
            }
            __syncthreads();
        }
        if (!more) break;
        g = gnext;
        tile = tnext;
    }
    TGV(3);
}

// ---------------------------------------------------------------------------------------------- batches 33..128
// The same 32-row-tile kernel for G groups of 32 batch rows (round 4: the reference's request-level mode keeps up to 128 requests in flight
// per worker, /root/reference/roll/distributed/scheduler/generate_scheduler.py:57): a wave streams its weight tile ONCE and multiplies it with
// the x fragments of every group (G accumulators), so the bytes per decode step stay those of batch 32 while the rows per step double or
// quadruple.  The x fragments of all groups ride the ring with the weights (ring depth 3 at G = 2, 2 at G = 4: 16 registers per group and
// slot); at G = 4 a wave reads 4 x as many x bytes (from L2) as weight bytes (from HBM).  Per row the arithmetic is that of k_gemv32 (same
// MFMA, same k order, same split over waves and the same reduction order), so a row's result does not depend on which group it sits in.
// F8 (round 5: configs[4]'s fp8 weights at 33..128 rows): the stream is the fp8 image (tiled8, common.h: 16 bytes = 16 consecutive k of one row).
// A lane loads TWO 16-byte units per 64-k chunk -- k-groups 2 j + kg, j = 0 / 1 -- and widens each (exactly) to the bf16 operands of two MFMA steps
// (k % 16 < 8, >= 8), so MFMA step s = 2 j + h of lane half kg covers k = (2 j + kg) * 16 + h * 8 .. + 7; x is addressed with the same permutation.
// The per-channel scale multiplies the float32 sums in the epilogue.  Half the weight bytes per step of the bf16 stream, the same x bytes.
template <int MODE, int KP, int G, bool F8 = false>
__global__ __launch_bounds__(256, G <= 2 ? 2 : 1) void k_gemv32g(GemvArgs p, int ntiles32) {
    constexpr int WAVES = 4, TPB = WAVES / KP, U = (G <= 2) ? 3 : 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // BIAS / RESID / SWIGLU launches may split the ROWS over gridDim.y (32 G rows per y): the narrow matrices (N = 2048 / 2560: 64 / 80 tiles)
    // otherwise leave three quarters of the CUs without a block, and their few MB of weights cost nothing to stream twice
    const int fr = lane & 15, half = (lane >> 4) & 1, kg = lane >> 5, m0 = (lane & 31) + (MODE != GV_PARTIAL ? (int)blockIdx.y * 32 * G : 0);
    const int tp = wave / KP, kp = wave % KP;
    const int tile = blockIdx.x * TPB + tp;
    const bool active = tile < ntiles32;
    const int nchunks = p.K / 64;
    const int ks = (MODE == GV_PARTIAL) ? p.ksplit : 1;
    const int per = (nchunks + ks * KP - 1) / (ks * KP);
    const int c0 = min(((MODE == GV_PARTIAL ? blockIdx.y : 0) * KP + kp) * per, nchunks);
    const int cend = min(c0 + per, nchunks);
    const int t16 = (active ? tile : 0) * 2 + half;
    const bf16_t* wbase = F8 ? nullptr : p.w_tiled ? p.W + (size_t)t16 * nchunks * 1024 + kg * 512 + fr * 8
                                                   : p.W + (size_t)(t16 * 16 + fr) * p.K + kg * 8;
    const unsigned char* w8base = F8 ? p.W8 + (size_t)t16 * nchunks * 1024 + kg * 256 + fr * 16 : nullptr;
    const size_t w_c = p.w_tiled ? 1024 : 64, w_s = p.w_tiled ? 128 : 16;
    // x of row group g = x of group 0 + g * x_g elements (fragment order: two 16-row tiles further; row-major: 32 rows further).  Fragment-
    // ordered x holds whole 16-row groups: rows beyond ceil16(M) do not exist -- never read them (m_rd); row-major x: rows beyond M
    // (F8: MFMA step s = 2 j + h reads k-group 2 j + kg, k half h -- fragment order: k-step half h, lane group 2 j + kg; see above)
    const bf16_t* xbase = F8 ? (p.x_tiled ? p.x + ((size_t)(m0 >> 4) * nchunks * 2) * 512 + (kg * 16 + (m0 & 15)) * 8 : p.x + (size_t)m0 * p.ldx + kg * 16)
                             : (p.x_tiled ? p.x + ((size_t)(m0 >> 4) * nchunks * 2 + kg) * 512 + (m0 & 15) * 8 : p.x + (size_t)m0 * p.ldx + kg * 8);
    const size_t x_c = p.x_tiled ? 1024 : 64, x_s = p.x_tiled ? 128 : 16, x_g = p.x_tiled ? (size_t)nchunks * 2048 : (size_t)32 * p.ldx;
    const size_t x8_j = p.x_tiled ? 256 : 32, x8_h = p.x_tiled ? 512 : 8;           // F8: element strides of j (k-group pair) and h (k half)
    const int m_rd = p.x_tiled ? (p.M + 15) / 16 * 16 : p.M;
    constexpr int WL = F8 ? 2 : 4;
    u32x4 w[U][WL], xv[U][G][4];
    auto fill_w = [&](int u, int c) {
        if constexpr (F8) {
#pragma unroll
            for (int j = 0; j < 2; ++j) w[u][j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w8base + (size_t)c * 1024 + j * 512));
        } else {
#pragma unroll
            for (int st = 0; st < 4; ++st) w[u][st] = ldg_nt(wbase + (size_t)c * w_c + st * w_s);
        }
    };
    auto fill_x = [&](int u, int c) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const bf16_t* xp = F8 ? xbase + g * x_g + (size_t)c * x_c + (st >> 1) * x8_j + (st & 1) * x8_h : xbase + g * x_g + (size_t)c * x_c + st * x_s;
                xv[u][g][st] = (m0 + 32 * g < m_rd) ? *reinterpret_cast<const u32x4*>(xp) : u32x4{0, 0, 0, 0};
            }
        }
    };
    auto fill = [&](int u, int c) { fill_w(u, c); fill_x(u, c); };
    f32x16 acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[g][i] = 0.f;
    auto consume = [&](int u) {
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            u32x4 wop;
            if constexpr (F8) {
                const u32x4 q = w[u][st >> 1];
                uint32_t d[4];
                f8x4_to_bf16(q[(st & 1) * 2], d[0], d[1]);
                f8x4_to_bf16(q[(st & 1) * 2 + 1], d[2], d[3]);
                wop = u32x4{d[0], d[1], d[2], d[3]};
            } else wop = w[u][st];
#pragma unroll
            for (int g = 0; g < G; ++g)
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(wop), as_frag(xv[u][g][st]), acc[g], 0, 0, 0);
        }
    };
    if (active) {
        // first ring: x (L2) in front of the weights (HBM), see k_gemv
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (c0 + u < cend) fill_x(u, c0 + u);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (c0 + u < cend) fill_w(u, c0 + u);
        for (int c = c0; c < cend; c += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (c + u < cend) {
                    consume(u);
                    if (c + U + u < cend) fill(u, c + U + u);
                }
            }
        }
    }
    if constexpr (KP > 1) {
        if (kp > 0) {                                             // [TPB][KP-1][G] slots, quad-major (rb_store)
#pragma unroll
            for (int g = 0; g < G; ++g) rb_store(smem, (tp * (KP - 1) + (kp - 1)) * G + g, lane, acc[g]);
        }
        __syncthreads();
        if (kp == 0) {
#pragma unroll
            for (int k = 1; k < KP; ++k)
#pragma unroll
                for (int g = 0; g < G; ++g) rb_add(smem, (tp * (KP - 1) + (k - 1)) * G + g, lane, acc[g]);
        }
    }
    if constexpr (F8) {
        if (active && kp == 0) {        // per-output-channel scale of the fp8 weights (engine row order): weight row of acc[i] = tile * 32 + 8 (i / 4) + 4 kg + i % 4
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 sc = *reinterpret_cast<const float4*>(p.w_scale + (size_t)tile * 32 + 8 * g4 + 4 * kg);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    acc[g][g4 * 4] *= sc.x; acc[g][g4 * 4 + 1] *= sc.y; acc[g][g4 * 4 + 2] *= sc.z; acc[g][g4 * 4 + 3] *= sc.w;
                }
            }
        }
    }
    float bestv[G];
    int besti[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { bestv[g] = -INFINITY; besti[g] = 0x7fffffff; }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int m = m0 + 32 * g;
        if (!(active && kp == 0 && m < p.M)) continue;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int nl = 8 * g4 + 4 * kg;
            if constexpr (MODE == GV_SWIGLU) {
                if (g4 < 2) {
                    float o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float gt = rbf(acc[g][g4 * 4 + r]), up = rbf(acc[g][(g4 + 2) * 4 + r]);
                        o[r] = rbf(silu_f(gt)) * up;
                    }
                    bf16_t* ob = reinterpret_cast<bf16_t*>(p.out);
                    *reinterpret_cast<uint2*>(ob + (p.out_tiled ? tiled_offset((size_t)m, (size_t)(tile * 16 + nl), (size_t)(p.N / 2))
                                                                : (size_t)m * (p.N / 2) + tile * 16 + nl)) = uint2{pack2(o[0], o[1]), pack2(o[2], o[3])};
                }
            } else if constexpr (MODE == GV_BIAS || MODE == GV_RESID) {
                const int n = tile * 32 + nl;
                bf16_t* optr = reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.ldo + n;
                float o[4] = {acc[g][g4 * 4], acc[g][g4 * 4 + 1], acc[g][g4 * 4 + 2], acc[g][g4 * 4 + 3]};
                if (MODE == GV_BIAS && p.bias) {
                    const uint2 b = *reinterpret_cast<const uint2*>(p.bias + n);
                    o[0] += lo16(b.x); o[1] += hi16(b.x); o[2] += lo16(b.y); o[3] += hi16(b.y);
                }
                if constexpr (MODE == GV_RESID) {
                    const uint2 rv = *reinterpret_cast<const uint2*>(optr);
                    o[0] = lo16(rv.x) + rbf(o[0]); o[1] = hi16(rv.x) + rbf(o[1]);
                    o[2] = lo16(rv.y) + rbf(o[2]); o[3] = hi16(rv.y) + rbf(o[3]);
                }
                *reinterpret_cast<uint2*>(optr) = uint2{pack2(o[0], o[1]), pack2(o[2], o[3])};
            } else {
                const int n = tile * 32 + nl;
                float* o = reinterpret_cast<float*>(p.out) + ((size_t)(MODE == GV_PARTIAL ? blockIdx.y : 0) * p.M + m) * p.N + n;
                *reinterpret_cast<float4*>(o) = float4{acc[g][g4 * 4], acc[g][g4 * 4 + 1], acc[g][g4 * 4 + 2], acc[g][g4 * 4 + 3]};
                if constexpr (MODE == GV_F32) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (acc[g][g4 * 4 + r] > bestv[g] || (acc[g][g4 * 4 + r] == bestv[g] && n + r < besti[g])) { bestv[g] = acc[g][g4 * 4 + r]; besti[g] = n + r; }
                }
            }
        }
    }
    if constexpr (MODE == GV_F32) {
        if (p.amax_val) {                                          // KP == 1: smem is free
            float* av = reinterpret_cast<float*>(smem);            // [WAVES][32 G]
            int* ai = reinterpret_cast<int*>(av + WAVES * 32 * G);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float bv = bestv[g];
                int bi = besti[g];
                const float ov = __shfl_xor(bv, 32, 64);
                const int oi = __shfl_xor(bi, 32, 64);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
                if (kg == 0) { av[wave * 32 * G + 32 * g + m0] = bv; ai[wave * 32 * G + 32 * g + m0] = bi; }
            }
            __syncthreads();
            if (tid < p.M) {
                float bv = av[tid];
                int bi = ai[tid];
                for (int w2 = 1; w2 < WAVES; ++w2) {
                    const float v2 = av[w2 * 32 * G + tid];
                    const int i2 = ai[w2 * 32 * G + tid];
                    if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; }
                }
                p.amax_val[(size_t)tid * gridDim.x + blockIdx.x] = bv;
                p.amax_idx[(size_t)tid * gridDim.x + blockIdx.x] = bi;
            }
        }
    }
}

template <int MODE, int KP, int G, bool F8 = false>
int launch_32g(hipStream_t s, const GemvArgs& a) {
    constexpr int TPB = 4 / KP;
    const int ntiles = a.N / 32;
    dim3 grid(cdiv(ntiles, TPB), MODE == GV_PARTIAL ? a.ksplit : cdiv(a.M, 32 * G));
    size_t smem = (size_t)TPB * (KP > 1 ? KP - 1 : 0) * G * 64 * sizeof(f32x16);
    if (smem < (size_t)4 * 32 * G * 8) smem = (size_t)4 * 32 * G * 8;
    hipLaunchKernelGGL((k_gemv32g<MODE, KP, G, F8>), grid, dim3(256), smem, s, a, ntiles);
    SR_CHECK_LAUNCH();
    return 0;
}
template <int MODE, int G, bool F8 = false>
int launch_32g_kp(hipStream_t s, const GemvArgs& a, int kp) {
    if constexpr (MODE == GV_F32) return launch_32g<MODE, 1, G, false>(s, a);
    else if constexpr (F8) return kp == 4 ? launch_32g<MODE, 4, G, true>(s, a) : -22;        // (the fp8 stream: in-block K split 4 only, as at <= 32 rows)
    else return kp == 4 ? launch_32g<MODE, 4, G>(s, a) : kp == 2 ? launch_32g<MODE, 2, G>(s, a) : launch_32g<MODE, 1, G>(s, a);
}
template <int G, bool F8 = false>
int launch_32g_mode(hipStream_t s, const GemvArgs& a, int mode, int kp) {
    switch (mode) {
        case GV_PARTIAL: return launch_32g_kp<GV_PARTIAL, G, F8>(s, a, kp);
        case GV_SWIGLU: return launch_32g_kp<GV_SWIGLU, G, F8>(s, a, kp);
        case GV_F32: return F8 ? -22 : launch_32g_kp<GV_F32, G, false>(s, a, kp);
        case GV_BIAS: return launch_32g_kp<GV_BIAS, G, F8>(s, a, kp);
        case GV_RESID: return launch_32g_kp<GV_RESID, G, F8>(s, a, kp);
    }
    return -22;
}

// Measured and dropped (round 2): sharing x through LDS at batches 5..32 -- the 4 waves of a block own 4 DIFFERENT weight tiles
// over the SAME k range, the block stages each 64-wide x chunk once (double-buffered groups of 4 chunks, one barrier per group), so
// the loads per weight chunk drop from 2 + 4 to 2 + 1/4.  Correct, but slower everywhere: gate/up 25.3 vs 23.1 us at M = 32 (22.0 vs
// 18.9 at M = 16) with 172 blocks instead of 688, the split-K down-projection 18.6 vs 13.7 us.  Giving every CU several independent
// waves matters more than the x re-reads; what does help the x path is the fragment-ordered x below (x_tiled).
template <int MODE, int KP>
int launch_32(hipStream_t s, const GemvArgs& a) {
    constexpr int TPB = 4 / KP;
    const int ntiles = a.N / 32;
    dim3 grid(cdiv(ntiles, TPB), MODE == GV_PARTIAL ? a.ksplit : 1);
    size_t smem = (size_t)TPB * (KP > 1 ? KP - 1 : 0) * 64 * sizeof(f32x16);
    if (smem < 4 * 32 * 8) smem = 4 * 32 * 8;
    if (sr_switches().gemv_counted) hipLaunchKernelGGL((k_gemv32<MODE, KP, true>), grid, dim3(256), smem, s, a, ntiles);
    else hipLaunchKernelGGL((k_gemv32<MODE, KP, false>), grid, dim3(256), smem, s, a, ntiles);
    SR_CHECK_LAUNCH();
    return 0;
}

int g_px_cus = 0;
// the persistent x-stationary launch (SWIGLU, 17..32 rows, fragment-ordered x and weights, K = 256 / 512 / 1024 / 2048)
bool px_ok(const GemvArgs& a, int mode) {
    return a.px_counter && (sr_switches().gemv_xlds & 1) && mode == GV_SWIGLU && a.M > 16 && a.M <= 32 && a.x_tiled && a.w_tiled && !a.norm_w && (a.K == 256 || a.K == 512 || a.K == 1024 || a.K == 2048)
           && a.N % 32 == 0 && (!a.W8 || a.w_scale);
}
template <bool F8, int PER>
int launch_px_per(hipStream_t s, const GemvArgs& a, int grid, size_t smem, int ntiles) {
    hipLaunchKernelGGL((k_gemv_px<F8, PER>), dim3(grid), dim3(256), smem, s, a, ntiles);
    SR_CHECK_LAUNCH();
    return 0;
}
constexpr size_t kPxSmem = (size_t)2 * 3 * 4 * 64 * 16 + 64;     // reduction buffers (2 parities) + tile slots
template <bool F8>
int launch_px(hipStream_t s, const GemvArgs& a) {
    const int ntiles = a.N / 32;
    if (!g_px_cus) { if (int rc = gemv_prepare_px()) return rc; }      // (engines call it at sr_engine_create: a decode step is a stream capture)
    const int n_cu = g_px_cus;
    int grid = ntiles < n_cu ? ntiles : n_cu;      // one block per CU (~300 registers per wave); on a CU-masked stream the surplus blocks start late, find no ticket and leave
    switch (a.K / 256) {
        case 8: return launch_px_per<F8, 8>(s, a, grid, kPxSmem, ntiles);
        case 4: return launch_px_per<F8, 4>(s, a, grid, kPxSmem, ntiles);
        case 2: return launch_px_per<F8, 2>(s, a, grid, kPxSmem, ntiles);
        case 1: return launch_px_per<F8, 1>(s, a, grid, kPxSmem, ntiles);
    }
    return -22;
}

// the x-stationary split-K launch (PARTIAL, 17..32 rows, bf16 fragment-ordered x and weights, 4 waves = 4 in-block K parts, <= 11 chunks per wave)
bool px32_ok(const GemvArgs& a, int mode) {
    if (!(a.px_counter && (sr_switches().gemv_xlds & 2) && mode == GV_PARTIAL && a.M > 16 && a.M <= 32 && a.x_tiled && a.w_tiled && !a.norm_w && (!a.W8 || a.w_scale)
          && a.N % 32 == 0 && a.ksplit >= 4 && a.ksplit <= 8)) return false;          // (ksplit >= 4: where the streaming path runs the 32-row-tile kernels, use_32 / the fp8 branch of launch_gemv)
    const int nch = a.K / 64, per = (nch + a.ksplit * 4 - 1) / (a.ksplit * 4);
    return per >= 3 && per <= 11 && (a.ksplit * 4 - 1) * per < nch;
}
template <bool F8>
int launch_px32(hipStream_t s, const GemvArgs& a) {
    const int ntiles = a.N / 32;
    const size_t smem = (size_t)2 * 3 * 64 * sizeof(f32x16);
    if (!g_px_cus) { if (int rc = gemv_prepare_px()) return rc; }
    int grid = g_px_cus / a.ksplit * a.ksplit;           // one block per CU (~400 registers per wave)
    if (grid > ntiles * a.ksplit) grid = ntiles * a.ksplit;
    const int nch = a.K / 64, per = (nch + a.ksplit * 4 - 1) / (a.ksplit * 4);
    if (per <= 4) hipLaunchKernelGGL((k_gemv32_px<4, F8>), dim3(grid), dim3(256), smem, s, a, ntiles);
    else if (per <= 8) hipLaunchKernelGGL((k_gemv32_px<8, F8>), dim3(grid), dim3(256), smem, s, a, ntiles);
    else hipLaunchKernelGGL((k_gemv32_px<11, F8>), dim3(grid), dim3(256), smem, s, a, ntiles);
    SR_CHECK_LAUNCH();
    return 0;
}

// the LM head with x resident in LDS (F32, 17..32 rows, bf16 fragment-ordered x and weights, K = 2048, whole groups of 4 tiles)
bool hpx_ok(const GemvArgs& a, int mode) {
    return a.px_counter && (sr_switches().gemv_xlds & 8) && mode == GV_F32 && a.M > 16 && a.M <= 32 && a.x_tiled && a.w_tiled && !a.norm_w && !a.W8 && a.K == 2048
           && a.N % 32 == 0 && (a.amax_val == nullptr) == (a.amax_idx == nullptr);
}
int launch_hpx(hipStream_t s, const GemvArgs& a) {
    const int ntiles = a.N / 32, ngroups = cdiv(ntiles, 4);
    const size_t smem = (size_t)32 * 2048 * 2 + 4 * 32 * 8;
    if (!g_px_cus) { if (int rc = gemv_prepare_px()) return rc; }
    const int grid = ngroups < g_px_cus ? ngroups : g_px_cus;
    hipLaunchKernelGGL((k_gemv32_hpx<8>), dim3(grid), dim3(256), smem, s, a, ntiles, ngroups);
    SR_CHECK_LAUNCH();
    return 0;
}

size_t stage_bytes(const GemvArgs& a) { return ((size_t)a.M * (a.K + 8) * 2 + 15) / 16 * 16; }

template <int MODE, int MT, int KP, bool STAGE, int WAVES, bool F8 = false>
int launch_k(hipStream_t s, const GemvArgs& a) {
    constexpr int T = (MODE == GV_SWIGLU) ? 2 : 1;
    constexpr int TPB = WAVES / KP;
    const int ntiles = a.N / (16 * T);
    dim3 grid(cdiv(ntiles, TPB), MODE == GV_PARTIAL ? a.ksplit : 1);
    size_t red = (size_t)TPB * (KP > 1 ? KP - 1 : 0) * T * MT * 64 * sizeof(f32x4);
    if (MODE == GV_F32) red = red > (size_t)WAVES * 32 * 8 ? red : (size_t)WAVES * 32 * 8;
    size_t smem;
    if (STAGE) {                                       // [xn | row sums]; the reduction reuses the xn region
        smem = stage_bytes(a);
        if (red > smem) smem = red;
        smem += 128 * sizeof(float);
    } else smem = 128 * sizeof(float) + red;
    if (smem > 160 * 1024) return -12;
    static size_t attr = 0;
    if (smem > 64 * 1024 && smem > attr) {
        hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemv<MODE, MT, KP, STAGE, WAVES, F8>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (r != hipSuccess) return (int)r;
        attr = smem;
    }
    hipLaunchKernelGGL((k_gemv<MODE, MT, KP, STAGE, WAVES, F8>), grid, dim3(WAVES * 64), smem, s, a, ntiles);
    SR_CHECK_LAUNCH();
    return 0;
}

// x can be staged in LDS when a whole padded copy fits and the prologue's row segmentation works
bool can_stage(const GemvArgs& a) { return a.K % 512 == 0 && stage_bytes(a) + 1024 <= 150 * 1024; }
// (16-wave blocks that stage x once for 4 weight tiles were measured SLOWER at M = 32 -- qkv 30 vs 11 us, gate/up 34 vs 29 us: a CU ingests
// only ~50-100 GB/s, so a serial 128-640 KB prologue per block costs more than the x re-reads it saves, and only 172 of 256 CUs get a block.)

template <int MODE, int KP, bool F8 = false>
int launch_small(hipStream_t s, const GemvArgs& a) {      // 4-wave blocks
    const bool stage = a.norm_w != nullptr;
    if constexpr (MODE == GV_BIAS || MODE == GV_SWIGLU || MODE == GV_F32) {
        if (stage) return a.M <= 16 ? launch_k<MODE, 1, KP, true, 4, F8>(s, a) : launch_k<MODE, 2, KP, true, 4, F8>(s, a);
    }
    return a.M <= 16 ? launch_k<MODE, 1, KP, false, 4, F8>(s, a) : launch_k<MODE, 2, KP, false, 4, F8>(s, a);
}
}  // namespace

// attribute + CU count of the persistent x-resident kernel, outside of any stream capture (sr_engine_create; the op-level entry points call it lazily)
int gemv_prepare_px() {
    for (const void* f : {reinterpret_cast<const void*>(k_gemv_px<false, 8>), reinterpret_cast<const void*>(k_gemv_px<true, 8>),
                          reinterpret_cast<const void*>(k_gemv_px<false, 4>), reinterpret_cast<const void*>(k_gemv_px<true, 4>),
                          reinterpret_cast<const void*>(k_gemv_px<false, 2>), reinterpret_cast<const void*>(k_gemv_px<true, 2>),
                          reinterpret_cast<const void*>(k_gemv_px<false, 1>), reinterpret_cast<const void*>(k_gemv_px<true, 1>)}) {
        hipError_t r = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
        if (r != hipSuccess) return (int)r;
    }
    {
        hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemv32_hpx<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
        if (r != hipSuccess) return (int)r;
    }
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return -5;
    g_px_cus = pr.multiProcessorCount;
    return 0;
}

#ifdef SR_GEMV_TIMING
extern "C" __attribute__((visibility("default"))) int sr_dbg_gemv_times(long long* host_out, int n_blocks) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_tgv), (size_t)n_blocks * 4 * sizeof(long long), 0, hipMemcpyDeviceToHost);
}
#endif

// Also measured at M = 32: doubling the weight rows a wave owns per x fragment (half the x load instructions) is slower -- gate/up 25.5 vs
// 22.7 us, qkv 11.1 vs 10.4 us: the halved wave count costs more than the saved L2 reads.
// measured at M = 32 (us, 16-row / 32-row variant): LM head 187 / 140, down-projection with 4 slabs 17.4 / 13.4;
// qkv 9.6 / 11.5, o_proj 9.2 / 9.9, gate/up 22.9 / 25.4 (the 32-row tiles halve the wave count of the small launches)
bool use_32(const GemvArgs& a, int mode) {
    if (a.force32 && a.N % 32 == 0 && !a.norm_w) return true;
    if (a.M <= 16 || a.N % 32 != 0 || a.norm_w) return false;
    return mode == GV_F32 || (mode == GV_PARTIAL && a.ksplit >= 4);
}

int gemv_f32_blocks(int N, int M, int K, int has_norm) {
    GemvArgs a{};
    a.M = M; a.K = K; a.N = N;
    a.norm_w = has_norm ? reinterpret_cast<const bf16_t*>(&a) : nullptr;     // only its null-ness matters here
    return use_32(a, GV_F32) ? cdiv(N / 32, 4) : cdiv(N / 16, 4);
}

// vocabulary rows covered by one block of the F32 (LM head) launch = rows per entry of the argmax partials
int gemv_f32_block_rows(int N, int M, int K, int has_norm) {
    GemvArgs a{};
    a.M = M; a.K = K; a.N = N;
    a.norm_w = has_norm ? reinterpret_cast<const bf16_t*>(&a) : nullptr;
    return use_32(a, GV_F32) ? 128 : 64;
}

// largest in-block K split that leaves >= 2 chunks per wave
int gemv_pick_kp(int K, int ksplit, int want) {
    const int ch = K / 64 / (ksplit > 0 ? ksplit : 1);
    for (int kp = want; kp > 1; kp >>= 1)
        if (ch / kp >= 2) return kp;
    return 1;
}

int launch_gemv(hipStream_t s, const GemvArgs& a_, int mode) {
    GemvArgs a = a_;
    a.counted = sr_switches().gemv_counted;      // (SR_GEMV_COUNTED=0: the conditional-refill loops of rounds 1-4, A/B and bit-identity hook)
    if (a.M <= 0) return 0;
    if (a.M > 128 || a.K % 64 != 0 || a.N % 16 != 0) return -22;
    if (a.M > 32) {      // 33..128 rows: the 32-row-tile kernel over 2 / 4 row groups per weight pass (bf16 stream, no fused norm)
        if (a.N % 32 != 0 || a.norm_w || a.n_slabs) return -22;
        if (a.W8 && (!a.w_scale || mode == GV_F32)) return -22;
        if (mode == GV_PARTIAL && (a.ksplit < 1 || a.ksplit > a.K / 64)) return -22;
        if (a.out_tiled && (mode != GV_SWIGLU || (a.N / 2) % 64 != 0)) return -22;
        const int kp32 = gemv_pick_kp(a.K, mode == GV_PARTIAL ? a.ksplit : 1, mode == GV_F32 ? 1 : 4);
        // NARROW matrices (q/k/v, o_proj: N = 2560 / 2048 = 80 / 64 tiles): one 32-row group per block and the groups over gridDim.y -- 2 - 4 x the
        // blocks, and their 8 - 10 MB of weights cost nothing to stream once per group (measured against two groups per block: decode step
        // 3.24 -> 3.09 ms at 64 rows, 4.20 -> 4.15 ms at 128; bit-identical)
        const bool narrow = (mode == GV_BIAS || mode == GV_RESID) && a.N <= 4096;
        if (a.W8) {     // fp8 weight stream (round 5): the same row-group kernel on the tiled8 image
            if (narrow) return launch_32g_mode<1, true>(s, a, mode, kp32);
            return a.M <= 64 ? launch_32g_mode<2, true>(s, a, mode, kp32) : launch_32g_mode<4, true>(s, a, mode, kp32);
        }
        if (narrow) return launch_32g_mode<1>(s, a, mode, kp32);
        return a.M <= 64 ? launch_32g_mode<2>(s, a, mode, kp32) : launch_32g_mode<4>(s, a, mode, kp32);
    }
    if (mode == GV_SWIGLU && a.N % 32 != 0) return -22;
    if (mode == GV_PARTIAL && (a.ksplit < 1 || a.ksplit > a.K / 64)) return -22;
    if (a.norm_w && !(mode == GV_BIAS || mode == GV_SWIGLU || mode == GV_F32)) return -22;
    if (a.norm_w && !can_stage(a)) return -22;
    if (a.n_slabs > 0 && (!a.norm_w || !a.slabs || !a.x_out)) return -22;
    if (a.x_tiled && a.norm_w) return -22;                   // fragment-ordered x: the un-staged paths only
    if (a.out_tiled && (mode != GV_SWIGLU || (a.N / 2) % 64 != 0)) return -22;
    int want = mode == GV_F32 ? 1 : 4;
    const int kp = gemv_pick_kp(a.K, mode == GV_PARTIAL ? a.ksplit : 1, want);
    if (kp == 4 && px_ok(a, mode)) return a.W8 ? launch_px<true>(s, a) : launch_px<false>(s, a);
    if (kp == 4 && px32_ok(a, mode)) return a.W8 ? launch_px32<true>(s, a) : launch_px32<false>(s, a);
    if (hpx_ok(a, mode)) return launch_hpx(s, a);
    if (a.W8) {          // fp8 weight stream (decode of the quantised LM linears); the in-block K split is always 4 there
        if (!a.w_scale || mode == GV_F32) return -22;
        if (kp != 4) return -22;
        // 17..32 rows, 4-slab down-projection: the 32-row-tile kernel (x read once per 32 weight rows), as for the bf16 stream -- measured on the fp8
        // stream at 32 rows: decode step 2.211 -> 2.180 ms; gate/up unchanged (2.216), q/k/v + o_proj slower (2.323) and therefore left on 16-row tiles
        if (mode == GV_PARTIAL && a.M > 16 && a.ksplit >= 4 && a.N % 32 == 0 && !a.norm_w) return launch_32g_mode<1, true>(s, a, mode, kp);
        switch (mode) {
            case GV_PARTIAL: return launch_small<GV_PARTIAL, 4, true>(s, a);
            case GV_SWIGLU: return launch_small<GV_SWIGLU, 4, true>(s, a);
            case GV_BIAS: return launch_small<GV_BIAS, 4, true>(s, a);
            case GV_RESID: return launch_small<GV_RESID, 4, true>(s, a);
        }
        return -22;
    }
    if (use_32(a, mode)) {
        switch (mode) {
            case GV_PARTIAL: return kp == 4 ? launch_32<GV_PARTIAL, 4>(s, a) : kp == 2 ? launch_32<GV_PARTIAL, 2>(s, a) : launch_32<GV_PARTIAL, 1>(s, a);
            case GV_SWIGLU: return kp == 4 ? launch_32<GV_SWIGLU, 4>(s, a) : kp == 2 ? launch_32<GV_SWIGLU, 2>(s, a) : launch_32<GV_SWIGLU, 1>(s, a);
            case GV_F32: return launch_32<GV_F32, 1>(s, a);
            case GV_BIAS: return kp == 4 ? launch_32<GV_BIAS, 4>(s, a) : kp == 2 ? launch_32<GV_BIAS, 2>(s, a) : launch_32<GV_BIAS, 1>(s, a);
            case GV_RESID: return kp == 4 ? launch_32<GV_RESID, 4>(s, a) : kp == 2 ? launch_32<GV_RESID, 2>(s, a) : launch_32<GV_RESID, 1>(s, a);
        }
    }
    switch (mode) {
        case GV_PARTIAL: return kp == 4 ? launch_small<GV_PARTIAL, 4>(s, a) : kp == 2 ? launch_small<GV_PARTIAL, 2>(s, a) : launch_small<GV_PARTIAL, 1>(s, a);
        case GV_SWIGLU: return kp == 4 ? launch_small<GV_SWIGLU, 4>(s, a) : kp == 2 ? launch_small<GV_SWIGLU, 2>(s, a) : launch_small<GV_SWIGLU, 1>(s, a);
        case GV_F32: return launch_small<GV_F32, 1>(s, a);
        case GV_BIAS: return kp == 4 ? launch_small<GV_BIAS, 4>(s, a) : kp == 2 ? launch_small<GV_BIAS, 2>(s, a) : launch_small<GV_BIAS, 1>(s, a);
        case GV_RESID: return kp == 4 ? launch_small<GV_RESID, 4>(s, a) : kp == 2 ? launch_small<GV_RESID, 2>(s, a) : launch_small<GV_RESID, 1>(s, a);
    }
    return -22;
}
