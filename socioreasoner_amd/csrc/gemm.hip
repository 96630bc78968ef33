// bf16 MFMA GEMM for the MFMA-bound rows of the hot path:
//   out[M,N] = A[M,K] . W[N,K]^T  with W in nn.Linear layout, float32 accumulation, fused epilogues that
//   reproduce HF's bf16 rounding points (hf: transformers/models/qwen2_5_vl/modeling_qwen2_5_vl.py:85-96, 137-150,
//   211-322, 541-554, 602-758).
// Tile: BM x 128 x 64 (BM = 128 or 64), 256 threads = 4 waves in 2x2, each wave (BM/2) x 64 as 16x16x32 MFMA tiles.
// LDS: double-buffered, filled by LDS-DMA (global_load_lds), rows of 64 bf16 (128 B) with the 16-byte chunk index
// XOR-ed with (row & 7) so that the ds_read_b128 fragment reads (lane = row, 4 lane-groups = 4 k-chunks) are
// bank-conflict free.
// The MFMA is issued as D = Wfrag x Afrag, i.e. D[i = n][j = m]: a lane then owns 4 CONSECUTIVE output columns of
// one output row, which makes the epilogue an 8-byte store per lane and keeps gate/up pairs in the same lane.
// Measured alternative (round 1): a 256 x 128 x 64, 8-wave, THREE-stage variant (prefetch distance 2, counted vmcnt waits + bare
// s_barrier instead of __syncthreads' vmcnt(0) drain, fragment double-buffering) was correct but 5-10 % SLOWER than this kernel on
// every hot-path shape (lm gate/up 911 vs 980 TF/s, ViT qkv 663 vs 723): with two 4-wave blocks per CU the memory round trip is
// already covered by the other block, and one 8-wave block per CU barriers twice as many waves per k-tile.  Under MFMA load the
// chip clocks ~2.0 GHz (guide: DVFS), i.e. ~2.1 PF effective peak; this kernel runs 0.9-1.0 PF on random data at large M.
// A two-stage 256 x 128 tile with 8 waves (one block per CU) lost 5-25 %, a 64 x 128 tile (three blocks per CU) 10-18 %: two
// 128 x 128 blocks per CU, each covering the other's LDS-DMA wait, is the sweet spot of this structure (PMC on lm gate/up: MFMA pipe
// 46 % busy at ~2.0 GHz; waves 34 % issuing, 37 % issue-stalled, 30 % parked; LDS instructions 3 % of wave time, no bank conflicts).
// Also measured: requesting both k-steps' fragments up front (sched_group_barrier: 16 ds_read, then 32 MFMA) lets the compiler hoist
// the tile barrier above the MFMAs; +4..6 % on two shapes, -3..20 % on the others (lm qkv 725 vs 910 TF/s).  Not kept.
#include "kernels.h"
#include <stdlib.h>

namespace {

constexpr int BN = 128, BK = 64;

template <int BM, int NS = 2>
struct Smem {
    bf16_t a[NS][BM * BK];
    bf16_t w[NS][BN * BK];
};

__device__ __forceinline__ int swz(int row, int chunk) { return row * (BK * 2) + ((chunk ^ (row & 7)) << 4); }

// NS = LDS stages.  2: the double-buffered loop described above (one __syncthreads per k-tile, the compiler drains the LDS-DMA in front
// of it).  NS > 2 (small M: a few hundred blocks, each walking 20 .. 172 k-tiles): a ring with NS - 1 k-tiles of LDS-DMA in flight, ONE
// counted s_waitcnt vmcnt + raw s_barrier per k-tile -- with one k-tile in flight the loop ran at the memory round trip (0.8 us per
// k-tile: the batch-1 LM down-projection, K = 11008, took 140 us on 112 CUs); same k order, bit-identical results.
template <int BM, int EPI, int NW = 4, int NS = 2>
__global__ __launch_bounds__(NW * 64) void k_gemm(GemmArgs p, int ntm, int ntn) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem<BM, NS>& sm = *reinterpret_cast<Smem<BM, NS>*>(smem_raw);
    constexpr int MI = BM / 32;       // 16-row m-tiles per wave (waves are arranged 2 x NW/2)
    constexpr int WNC = NW / 2;       // wave columns
    constexpr int NJ = BN / WNC / 16; // 16-col n-tiles per wave: 4 (NW = 4) or 2 (NW = 8)

    // ---- block -> tile: XCD-aware (block b runs on XCD b % 8: give each XCD a contiguous run of tiles), then
    // grouped ordering (8 m-tiles share their W panels while walking n)
    int bid = blockIdx.x, nblk = gridDim.x;
    {
        int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    constexpr int GROUP = 8;
    int per_group = GROUP * ntn;
    int gid = bid / per_group;
    int first_m = gid * GROUP;
    int gsz = min(ntm - first_m, GROUP);
    int tm = first_m + (bid % per_group) % gsz;
    int tn = (bid % per_group) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WNC, wn = wave % WNC;
    const int fr = lane & 15, fg = lane >> 4;

    // ---- global -> LDS staging by LDS-DMA (global_load_lds, 16 B per lane): no VGPR round trip and no ds_write
    // (a register-staged ds_write_b128 costs ~13 LDS cycles per wave-instruction on CDNA4 and made the loop LDS-bound).
    // One wave instruction fills 1 KB of LDS linearly = 8 tile rows x 128 B; the XOR swizzle is applied on the SOURCE
    // side: lane l fetches logical chunk (l & 7) ^ (row & 7) of row (l >> 3), so that LDS holds physical chunk l & 7.
    // Rows are clamped so edge tiles stay in bounds (their results are never stored).
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    constexpr int A_G = BM / 8 / NW, W_G = BN / 8 / NW;  // 8-row groups (1-KB LDS-DMA instructions) per wave
    const int lr = lane >> 3, lc = ((lane & 7) ^ lr) * 8;
    const bf16_t* asrc[A_G];
    const bf16_t* wsrc[W_G];
#pragma unroll
    for (int i = 0; i < A_G; ++i) asrc[i] = p.A + (size_t)min(m0 + (wave * A_G + i) * 8 + lr, p.M - 1) * p.lda + lc;
    // W source.  Row-major: like A (XOR swizzle on the source side).  Fragment-ordered (tiled16x64, common.h): every
    // (16-row tile, 64-k chunk) block is 2 KB contiguous, so one wave instruction copies 1 KB *contiguous* global memory
    // (one kstep half of a tile) linearly into LDS and the LDS image simply keeps the fragment order.
    const size_t w_kt_stride = p.w_tiled ? 1024 : BK;   // elements between consecutive 64-wide k tiles
#pragma unroll
    for (int i = 0; i < W_G; ++i) {
        if (p.w_tiled) {
            const int q = wave * W_G + i;                // 0..15: (n_tile_local = q >> 1, kstep = q & 1)
            const size_t nt = min(n0 / 16 + (q >> 1), p.N / 16 - 1);
            wsrc[i] = p.W + nt * (size_t)(p.K / 64) * 1024 + (q & 1) * 512 + lane * 8;
        } else {
            const size_t r = min(n0 + (wave * W_G + i) * 8 + lr, p.N - 1);
            wsrc[i] = p.W + r * p.K + lc;
        }
    }
    auto stage = [&](int buf, int kt) {
#pragma unroll
        for (int i = 0; i < A_G; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[i] + kt * BK), (lptr_t)(sm.a[buf] + (wave * A_G + i) * 512), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < W_G; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + (size_t)kt * w_kt_stride), (lptr_t)(sm.w[buf] + (wave * W_G + i) * 512), 16, 0, 0);
    };

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // k-tiles of this block: all of them, or slice blockIdx.y of a split-K launch (EPI_F32 partial products)
    const int nk_all = p.K / BK, per = (nk_all + (int)gridDim.y - 1) / (int)gridDim.y;
    const int kt0 = min((int)blockIdx.y * per, nk_all), nk = min(kt0 + per, nk_all);
    auto compute = [&](int cur) {
        const unsigned char* ab = reinterpret_cast<const unsigned char*>(sm.a[cur]);
        const unsigned char* wb = reinterpret_cast<const unsigned char*>(sm.w[cur]);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[MI], wf[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                int row = wm * (BM / 2) + i * 16 + fr;
                af[i] = *reinterpret_cast<const bf16x8*>(ab + swz(row, kk * 4 + fg));
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (p.w_tiled) {     // LDS keeps the fragment order: [tile][kstep = fg & 1][lane' = (kk*2 + fg/2)*16 + fr][16 B]
                    wf[j] = *reinterpret_cast<const bf16x8*>(wb + (wn * NJ + j) * 2048 + (fg & 1) * 1024 + (((kk * 2 + (fg >> 1)) * 16 + fr) << 4));
                } else {
                    int row = wn * (NJ * 16) + j * 16 + fr;
                    wf[j] = *reinterpret_cast<const bf16x8*>(wb + swz(row, kk * 4 + fg));
                }
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
        }
    };
    if constexpr (NS == 2) {
        if (kt0 < nk) stage(0, kt0);
        __syncthreads();               // the compiler drains the LDS-DMA (vmcnt(0)) in front of the barrier
        for (int kt = kt0; kt < nk; ++kt) {
            const int cur = (kt - kt0) & 1;
            if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
            compute(cur);
            __syncthreads();
        }
    } else {
        // ring: stages kt .. kt + NS - 2 are in flight or landed when k-tile kt is computed.  A stage is G = A_G + W_G LDS-DMA
        // instructions per wave; "vmcnt(G * j)" = everything but the newest j stages of THIS wave has landed, the barrier extends that
        // to every wave's share and also says that all waves are done reading the buffer the next issue overwrites (it was read in the
        // previous iteration).
        constexpr int G = A_G + W_G;
        static_assert(NS <= 6 && G * (NS - 2) <= 60, "one wait case per stage count; vmcnt is a 6-bit counter");
#pragma unroll
        for (int j = 0; j < NS - 1; ++j)
            if (kt0 + j < nk) stage(j, kt0 + j);
        int buf = 0;
        for (int kt = kt0; kt < nk; ++kt) {
            const int ahead = min(nk - 1 - kt, NS - 2);      // stages newer than kt that this wave has issued
            switch (ahead) {                                  // (immediate operand: one case per count)
                case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(G) : "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(G * 2) : "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(G * 3) : "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(G * 4) : "memory"); break;
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (kt + NS - 1 < nk) stage(buf == 0 ? NS - 1 : buf - 1, kt + NS - 1);      // = (buf + NS - 1) % NS: the buffer of k-tile kt - 1
            compute(buf);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's fragment reads are done before it can reach the next barrier
            __builtin_amdgcn_sched_barrier(0);
            buf = buf + 1 == NS ? 0 : buf + 1;
        }
    }

    // ---- epilogue.  lane owns row m = .. + fr and columns n = .. + fg*4 + {0,1,2,3} of every 16x16 tile.
    // 16-byte accesses (same scheme as gemm256.hip): v_permlane16_swap trades the packed halves of two neighbouring tiles between
    // the lane pairs (l, l + 16), after which even 16-lane rows hold 8 consecutive columns of the first tile and odd rows 8 of the
    // second -- one dwordx4 store (and residual load) per lane and tile pair instead of two dwordx2.
    auto widen = [&](uint2 ta, uint2 tb) -> uint4 {
        const auto sx = __builtin_amdgcn_permlane16_swap(ta.x, tb.x, false, false);
        const auto sy = __builtin_amdgcn_permlane16_swap(ta.y, tb.y, false, false);
        return uint4{sx[0], sy[0], sx[1], sy[1]};
    };
    const int odd = fg & 1, half8 = (fg >> 1) * 8;
    int orows[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * (BM / 2) + i * 16 + fr;
        orows[i] = (m < p.M && p.rowmap) ? p.rowmap[m] : m;
    }
    // EPI_RESID: every residual load of the lane before the first store (the residual may alias the output, so the compiler keeps a load
    // behind each earlier store: a load -> wait -> store round trip per 16 rows otherwise; see gemm256.hip)
    uint4 rl[MI][NJ / 2 > 0 ? NJ / 2 : 1];
    if constexpr (EPI == EPI_RESID) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int jp = 0; jp < NJ / 2; ++jp) {
                const int n = n0 + wn * (NJ * 16) + (2 * jp + odd) * 16 + half8;
                rl[i][jp] = (m0 + wm * (BM / 2) + i * 16 + fr < p.M && n < p.N) ? *reinterpret_cast<const uint4*>(p.resid + (size_t)orows[i] * p.ldo + n) : uint4{0, 0, 0, 0};
            }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * (BM / 2) + i * 16 + fr;
        const bool rok = m < p.M;                              // (masked rows still take part in the lane exchange)
        const int orow = orows[i];
        if constexpr (EPI == EPI_SWIGLU) {
            uint2 t2[NJ / 2];
#pragma unroll
            for (int jp = 0; jp < NJ / 2; ++jp) {
                const int ng = n0 + wn * (NJ * 16) + jp * 32 + fg * 4;       // gate columns (interleaved index)
                const int nu = ng + 16;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float g = acc[i][2 * jp][r], u = acc[i][2 * jp + 1][r];
                    if (ng < p.N) {
                        if (p.w_scale) { g *= p.w_scale[ng + r]; u *= p.w_scale[nu + r]; }
                        if (p.bias) { g += bf2f(p.bias[ng + r]); u += bf2f(p.bias[nu + r]); }
                    }
                    g = rbf(g); u = rbf(u);
                    o[r] = rbf(silu_f(g)) * u;
                }
                t2[jp] = uint2{pack2(o[0], o[1]), pack2(o[2], o[3])};
            }
            if constexpr (NJ == 4) {
                const uint4 v = widen(t2[0], t2[1]);
                const int ngs = n0 + wn * 64 + odd * 32;                     // gate-column base of the output tile this lane stores
                const int no = (n0 + wn * 64) / 2 + odd * 16 + half8;
                // out_tiled: the activation leaves fragment-ordered (tiled16x64 of [ceil16(M)][ldo]: 8 consecutive k of a row stay 16 contiguous
                // bytes) -- it is the x operand of the decode down-projection GEMV at more than 64 batch rows
                if (rok && ngs < p.N) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + (p.out_tiled ? tiled_offset((size_t)orow, (size_t)no, (size_t)p.ldo) : (size_t)orow * p.ldo + no)) = v;
            } else {
                const int ng = n0 + wn * (NJ * 16) + fg * 4;
                const int no = (n0 + wn * (NJ * 16)) / 2 + fg * 4;
                if (rok && ng < p.N) *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + (p.out_tiled ? tiled_offset((size_t)orow, (size_t)no, (size_t)p.ldo) : (size_t)orow * p.ldo + no)) = t2[0];
            }
        } else if constexpr (EPI == EPI_F32) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = n0 + wn * (NJ * 16) + j * 16 + fg * 4;
                if (!rok || n >= p.N) continue;
                float o[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if (p.w_scale) {
                    const float4 sc = *reinterpret_cast<const float4*>(p.w_scale + n);
                    o[0] *= sc.x; o[1] *= sc.y; o[2] *= sc.z; o[3] *= sc.w;
                }
                if (p.bias) { o[0] += bf2f(p.bias[n]); o[1] += bf2f(p.bias[n + 1]); o[2] += bf2f(p.bias[n + 2]); o[3] += bf2f(p.bias[n + 3]); }
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + ((size_t)blockIdx.y * p.M + orow) * p.ldo + n) = float4{o[0], o[1], o[2], o[3]};
            }
        } else {
            uint2 rv[NJ], t4[NJ];
            if constexpr (EPI == EPI_RESID) {
#pragma unroll
                for (int jp = 0; jp < NJ / 2; ++jp) {
                    const uint4 l4 = rl[i][jp];
                    const auto sx = __builtin_amdgcn_permlane16_swap(l4.x, l4.z, false, false);
                    const auto sy = __builtin_amdgcn_permlane16_swap(l4.y, l4.w, false, false);
                    rv[2 * jp] = uint2{sx[0], sy[0]};
                    rv[2 * jp + 1] = uint2{sx[1], sy[1]};
                }
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = n0 + wn * (NJ * 16) + j * 16 + fg * 4;
                float o[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if (n < p.N) {
                    if (p.w_scale) {
                        const float4 sc = *reinterpret_cast<const float4*>(p.w_scale + n);
                        o[0] *= sc.x; o[1] *= sc.y; o[2] *= sc.z; o[3] *= sc.w;
                    }
                    if (p.bias) {
                        const uint2 b = *reinterpret_cast<const uint2*>(p.bias + n);
                        o[0] += lo16(b.x); o[1] += hi16(b.x); o[2] += lo16(b.y); o[3] += hi16(b.y);
                    }
                }
                if constexpr (EPI == EPI_RESID) {
                    o[0] = lo16(rv[j].x) + rbf(o[0]); o[1] = hi16(rv[j].x) + rbf(o[1]);
                    o[2] = lo16(rv[j].y) + rbf(o[2]); o[3] = hi16(rv[j].y) + rbf(o[3]);
                } else if constexpr (EPI == EPI_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = p.gelu_fast == 2 ? fmaxf(rbf(o[r]), 0.f) : p.gelu_fast ? gelu_fast_f(rbf(o[r])) : gelu_f(rbf(o[r]));
                }
                t4[j] = uint2{pack2(o[0], o[1]), pack2(o[2], o[3])};
            }
#pragma unroll
            for (int jp = 0; jp < NJ / 2; ++jp) {
                const uint4 v = widen(t4[2 * jp], t4[2 * jp + 1]);
                const int n = n0 + wn * (NJ * 16) + (2 * jp + odd) * 16 + half8;
                if (rok && n < p.N) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)orow * p.ldo + n) = v;
            }
        }
    }
}

// raises the kernel's dynamic-LDS limit once per process (never inside a stream capture: gemm_prepare_decode below)
template <int BM, int EPI, int NW, int NS>
int ensure_attr() {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm<BM, EPI, NW, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Smem<BM, NS>));
        if (r != hipSuccess) return (int)r;
        attr_done = true;
    }
    return 0;
}

template <int BM, int EPI, int NW = 4, int NS = 2>
int launch_t(hipStream_t s, const GemmArgs& a) {
    int ntm = cdiv(a.M, BM), ntn = cdiv(a.N, BN);
    size_t smem = sizeof(Smem<BM, NS>);
    if (int rc = ensure_attr<BM, EPI, NW, NS>()) return rc;
    hipLaunchKernelGGL((k_gemm<BM, EPI, NW, NS>), dim3(ntm * ntn, EPI == EPI_F32 && a.ksplit > 1 ? a.ksplit : 1), dim3(NW * 64), smem, s, a, ntm, ntn);
    SR_CHECK_LAUNCH();
    return 0;
}

template <int EPI>
int launch_e(hipStream_t s, const GemmArgs& a) {
    // 128-row tiles only when they still give every CU work; otherwise 64-row tiles double the block count
    long blocks128 = (long)cdiv(a.M, 128) * cdiv(a.N, BN);
    const int bm = sr_switches().gemm_bm;                      // tuning hook for tools/bench_gemm.py: 64 forces the 64-row tile
    if (bm == 64) return launch_t<64, EPI>(s, a);
    if (bm == 1288) return launch_t<128, EPI, 8>(s, a);      // 128-row tile, 8 waves (2 x 4) of 64 x 32
    if (blocks128 >= 512) {
        // 8 waves (2 x 4, 64 x 32 each; four waves per SIMD with two blocks per CU) measured +2..4 % on the SwiGLU / residual shapes
        // (lm gate/up 984 vs 943 TF/s, ViT gate/up 821 vs 797, ViT down 874 vs 856) and -5 % on the ViT qkv store shape
        if constexpr (EPI == EPI_SWIGLU || EPI == EPI_RESID) return launch_t<128, EPI, 8>(s, a);
        return launch_t<128, EPI>(s, a);
    }
    // at most one block per CU, each walking the whole K: the 6-stage ring (5 k-tiles of LDS-DMA in flight; 144 KB of LDS).  With more
    // blocks than CUs the three co-resident blocks of the double-buffered loop cover each other better (ViT qkv at batch 1, 480 blocks:
    // 6.06 vs 6.47 ms per ViT pass)
    const int ring = sr_switches().gemm_ring;                 // tuning / test hook: 0 = never, 2 = whenever K allows
    const long blocks64 = (long)cdiv(a.M, 64) * cdiv(a.N, BN);
    if (a.K / BK >= 8 && (ring == 2 || (ring == 1 && blocks64 <= 256))) return launch_t<64, EPI, 4, 6>(s, a);
    return launch_t<64, EPI>(s, a);
}

}  // namespace

// The decode layer of an engine with more than 64 batch rows launches its gate/up as a tile GEMM INSIDE the captured decode step: the
// attribute call of that kernel must have happened before the capture starts (sr_engine_create calls this)
int gemm_prepare_decode() {
    if (int rc = ensure_attr<64, EPI_SWIGLU, 4, 2>()) return rc;
    return ensure_attr<64, EPI_SWIGLU, 4, 6>();
}

// does the dispatch below send `a` to the 256-tile kernel?  (SR_GEMM256: 0 = never, 2 = whenever the shape is supported -- tuning hook)
static bool picks_256(const GemmArgs& a) {
    const int g256 = sr_switches().gemm256;
    if (a.force_tile == 256) return true;
    return a.force_tile != 128 && g256 && gemm256_supports(a) && (g256 == 2 || (long)cdiv(a.M, 256) * (a.N / 256) >= 384);
}
bool lmqkv_ok(const GemmArgs& a);
bool gemm_fuses_lmqkv(const GemmArgs& a) {
    if (!sr_switches().fuse_qkv) return false;                 // tuning / test hook: 0 = separate rotary + cache-write launch as in round 1
    return a.K % BK == 0 && picks_256(a) && gemm256_supports(a) && lmqkv_ok(a);
}

bool vitqkv_ok(const GemmArgs& a);
bool gemm_fuses_vitqkv(const GemmArgs& a) {
    if (!sr_switches().fuse_qkv) return false;                 // the same hook as above
    return a.K % BK == 0 && picks_256(a) && gemm256_supports(a) && vitqkv_ok(a);
}

int launch_gemm(hipStream_t s, const GemmArgs& a, int epi) {
    if (a.M <= 0) return 0;
    if (a.K % BK != 0 || a.N % 16 != 0 || (epi == EPI_SWIGLU && a.N % 32 != 0)) return -22;
    if (a.out_tiled) {      // fragment-ordered SwiGLU output: this file's kernel only (the decode gate/up at more than 64 rows), whole 64-wide k chunks
        if (epi != EPI_SWIGLU || a.ldo % 64 != 0 || a.rowmap || a.ksplit > 1 || a.force_tile == 256) return -22;
        // (one 128-row tile per W panel -- 172 blocks, the weights read once -- was measured and is no faster: 4.14 vs 4.15 ms per step)
        return launch_e<EPI_SWIGLU>(s, a);
    }
    if (a.ksplit > 1) {      // split-K partial products (small M): this file's kernel, float32 slabs, no row map
        // (a bias would be added once per slab -- the float32 epilogue adds it in every block: refused, the consumer of the slabs adds it)
        if (epi != EPI_F32 || a.rowmap || a.bias || a.ksplit > a.K / BK) return -22;
        return launch_e<EPI_F32>(s, a);
    }
    // large M: the 256 x 256 8-phase kernel (gemm256.hip)
    if (picks_256(a) || epi == EPI_LMQKV || epi == EPI_VITQKV) return launch_gemm256(s, a, epi);
    switch (epi) {
        case EPI_STORE: return launch_e<EPI_STORE>(s, a);
        case EPI_RESID: return launch_e<EPI_RESID>(s, a);
        case EPI_SWIGLU: return launch_e<EPI_SWIGLU>(s, a);
        case EPI_GELU: return launch_e<EPI_GELU>(s, a);
        case EPI_F32: return launch_e<EPI_F32>(s, a);
    }
    return -22;
}
