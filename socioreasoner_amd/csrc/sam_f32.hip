// SAM2 at the REFERENCE's precision: float32 storage, float32 MFMA.
//
// The reference builds its segmenter as SAM2ImagePredictor(build_sam2(cfg, ckpt)) and calls set_image / predict without autocast
// (/root/reference/roll/models/model_providers.py:540-548, roll/distributed/strategy/seg_strategy.py:47-60; the YAML's `dtype: bf16`,
// examples/infer/rlvr_megatron.yaml:112, is never applied): every Linear, attention and normalisation of Hiera-L and of the mask decoder
// runs in float32 there, and north_star asks for EXACT decoded mask pixels.  The bf16 kernels (gemm*.hip, attention.hip) cannot give
// that (1 000-1 500 of 571 536 pixels differ: profiles/r03_sam2_parity.json); these two kernels are the float32 mode of
// socioreasoner_amd/sam2.py (Sam2Engine(dtype=torch.float32)), the passes in between are sam.hip's templates instantiated for float.
//
// gfx950 has no xf32 / tf32 path: f32-input MFMA (v_mfma_f32_32x32x2_f32, v_mfma_f32_16x16x4_f32) runs at the float32 VECTOR rate,
// 157 TF/s = 1/16 of the bf16 rate (MI355X_MICROARCH.md "Matrix cores"), each product and sum an exact float32 fmaf chain -- the
// arithmetic torch's CPU float32 path performs, up to the order of the sums.
#include "kernels.h"
#include "rownorm.h"     // (sr_rsrc: buffer descriptors)
#include <type_traits>
#include <math.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

// ------------------------------------------------------------------------------------------------ GEMM
// out[M][N] = act(A[M][K] . W[N][K]^T + bias) (+ resid), all float32, row-major, K % 16 == 0, N % 4 == 0.
// 128 x 128 x 16 tiles, 4 waves (2 x 2), each wave 64 x 64 as 2 x 2 v_mfma_f32_32x32x2_f32.  The MFMA is issued as D = W_frag x A_frag,
// so a lane owns 4 CONSECUTIVE output columns of one row (16-byte epilogue accesses).  LDS holds both operands k-major ([k][row], pitch
// 132: fragment reads are one conflict-free ds_read_b32 per operand and k-step); tiles are staged global -> registers -> LDS, the loads
// of k-tile i + 1 in flight under the MFMAs of k-tile i (32 MFMAs x 64 cycles per wave and k-tile: the matrix pipe is the bound, LDS
// at ~25 % of its cycles).
constexpr int GB = 128, GK = 16, GP = GB + 4;

__global__ __launch_bounds__(256, 2) void k_gemm_f32(GemmF32Args p, int n_tiles_n) {
    __shared__ float As[2][GK][GP];
    __shared__ float Ws[2][GK][GP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = blockIdx.x;
    const int m0 = (tile / n_tiles_n) * GB, n0 = (tile % n_tiles_n) * GB;      // neighbouring blocks share the A rows; W (small) stays in L2
    const int lr = tid >> 2, lc = (tid & 3) * 4;                                // staging: rows lr, lr + 64; k = lc .. lc + 3
    const float* arow[2];
    const float* wrow[2];
    bool aok[2], wok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int gm = m0 + lr + 64 * j, gn = n0 + lr + 64 * j;
        aok[j] = gm < p.M;
        wok[j] = gn < p.N;
        arow[j] = p.A + (size_t)(aok[j] ? gm : 0) * p.lda + lc;
        wrow[j] = p.W + (size_t)(wok[j] ? gn : 0) * p.K + lc;
    }
    float4 ra[2], rw[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            ra[j] = aok[j] ? *reinterpret_cast<const float4*>(arow[j] + k0) : float4{0.f, 0.f, 0.f, 0.f};
            rw[j] = wok[j] ? *reinterpret_cast<const float4*>(wrow[j] + k0) : float4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto lstore = [&](int b) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = lr + 64 * j;
            As[b][lc + 0][r] = ra[j].x; As[b][lc + 1][r] = ra[j].y; As[b][lc + 2][r] = ra[j].z; As[b][lc + 3][r] = ra[j].w;
            Ws[b][lc + 0][r] = rw[j].x; Ws[b][lc + 1][r] = rw[j].y; Ws[b][lc + 2][r] = rw[j].z; Ws[b][lc + 3][r] = rw[j].w;
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[j][i][v] = 0.f;
    const int nk = p.K / GK;
    gload(0);
    lstore(0);
    __syncthreads();
    const int kg = lane >> 5, col = lane & 31;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * GK);
#pragma unroll
        for (int s = 0; s < GK / 2; ++s) {
            const int kk = 2 * s + kg;
            float wf[2], af[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) wf[j] = Ws[cur][kk][wn * 64 + j * 32 + col];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = As[cur][kk][wm * 64 + i * 32 + col];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j], af[i], acc[j][i], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(cur ^ 1);
        __syncthreads();
    }
    // epilogue: D[n][m], lane (kg, col): m = col, n = 8 * (v / 4) + 4 * kg + v % 4.
    // Instantiated per (bias, residual, activation) case behind block-uniform branches.  The residual may alias the output (the blocks add
    // in place), so inside one store loop every residual load waited for the previous store: the four float4s of a (row group, column
    // half) are fetched together, ahead of its stores -- four round trips per tile instead of sixteen.  (All sixteen up front cost 86 more
    // registers and two of the four blocks a CU holds: batched encoder 156 -> 164 ms.  With four blocks per CU the others' k loops hide what
    // is left.)  Same arithmetic, same order.
    auto epilogue = [&](auto hb_, auto hr_, auto act_) {
        constexpr bool HB = decltype(hb_)::value, HR = decltype(hr_)::value;
        constexpr int ACT = decltype(act_)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + wm * 64 + i * 32 + col;
            if (m >= p.M) continue;
            const size_t orow = (size_t)(p.rowmap ? p.rowmap[m] : m) * p.ldo;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float4 rv[4];
                if constexpr (HR) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = n0 + wn * 64 + j * 32 + 8 * q + 4 * kg;
                        rv[q] = n < p.N ? *reinterpret_cast<const float4*>(p.resid + orow + n) : float4{0.f, 0.f, 0.f, 0.f};
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n0 + wn * 64 + j * 32 + 8 * q + 4 * kg;
                    if (n >= p.N) continue;
                    float o[4];
                    float4 bq = float4{0.f, 0.f, 0.f, 0.f};
                    if constexpr (HB) bq = *reinterpret_cast<const float4*>(p.bias + n);
                    const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = acc[j][i][4 * q + e];
                        if constexpr (HB) x += bb[e];
                        if constexpr (ACT == 1) x = gelu_f(x);
                        else if constexpr (ACT == 2) x = fmaxf(x, 0.f);
                        o[e] = x;
                    }
                    if constexpr (HR) {
                        const float4 r = rv[q];
                        o[0] = r.x + o[0]; o[1] = r.y + o[1]; o[2] = r.z + o[2]; o[3] = r.w + o[3];
                    }
                    *reinterpret_cast<float4*>(p.out + orow + n) = float4{o[0], o[1], o[2], o[3]};
                }
            }
        }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    auto by_act = [&](auto hb_, auto hr_) {
        if (p.act == 1) epilogue(hb_, hr_, std::integral_constant<int, 1>{});
        else if (p.act == 2) epilogue(hb_, hr_, std::integral_constant<int, 2>{});
        else epilogue(hb_, hr_, std::integral_constant<int, 0>{});
    };
    if (p.bias) { if (p.resid) by_act(T_{}, T_{}); else by_act(T_{}, F_{}); }
    else { if (p.resid) by_act(F_{}, T_{}); else by_act(F_{}, F_{}); }
}

// ------------------------------------------------------------------------------------------------ GEMM on the bf16 matrix pipe (round 5)
// The same contract -- float32 operands in, float32 out, float32-grade results -- at the bf16 MFMA rate instead of the float32 one (1/16 of
// it): every float32 operand is split EXACTLY into three bf16 terms, x = hi + mid + lo (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid);
// 3 x 8 significand bits cover float32's 24, the two subtractions are exact), and a . w is formed from the six partial products whose
// weight is >= 2^-16 of the leading one:
//     hi.hi                                  -> accumulator `main`
//     hi.mid + mid.hi + mid.mid + hi.lo + lo.hi   -> accumulator `corr`        (dropped: mid.lo, lo.mid, lo.lo, <= 3 x 2^-24 relative)
// Each bf16 x bf16 product is exact in float32 (16 significand bits) and the MFMA sums them in float32; `corr` (2^-8 of `main`) is kept apart so
// that `main` sees exactly as many float32 roundings as the f32-input MFMA chain does, and is added once at the end.  The dropped terms are of
// the size of ONE float32 rounding of the product -- below the round-off of the K-term sum that any float32 GEMM (torch's CPU path included,
// whose summation order differs from ours anyway) carries: tests/test_gpu_round4.py::test_gemm_f32_vs_float64 holds this kernel to the bound
// it held the f32-input kernel to, and SAM2's masks keep their pixel-exactness against HF float32 (tests/test_gpu_sam2.py).
// 6 x v_mfma_f32_32x32x16_bf16 (32 cycles each) replace 8 x v_mfma_f32_32x32x2_f32 (64 cycles each) per 16 k: 2.67 x fewer matrix-pipe cycles.
// Tiles as above (128 x 128 x 16, 4 waves 2 x 2, D = W_frag x A_frag -> the same accumulator layout and epilogue).  The split happens on the
// way from global memory to LDS: thread t owns 8 consecutive k of one row (row t / 2, k half t % 2) of each operand; LDS holds three bf16
// planes per operand as [k half][row][8 k] (16 bytes per entry: fragment reads are conflict-free ds_read_b128, stores conflict-free
// ds_write_b128 with the second k half displaced by 64 bytes).
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4s;
constexpr int FGROUP = 8;                         // m-tiles that walk the n-tiles together (tile order below)
constexpr int SP_HALF = 128 * 16 + 64;            // bytes between the two k halves of a plane
constexpr int SP_PLANE = 2 * SP_HALF;             // bytes per plane
constexpr int SP_OPER = 3 * SP_PLANE;             // bytes per operand (hi, mid, lo)

// x = hi + mid + lo, two values per dword of each plane: one v_cvt_pk_bf16_f32 (round to nearest even) per pair and level, the bf16 terms
// widened back by a shift / a mask, the remainders by one exact subtraction each -- 11 VALU instructions per pair of elements
__device__ __forceinline__ uint32_t cvt_pk(float a, float b) {
    typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
    typedef float f2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2_t{a, b}, bf2_t));
}
__device__ __forceinline__ void split3(const float (&x)[8], u32x4s& hi, u32x4s& mid, u32x4s& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float a = x[2 * e], b = x[2 * e + 1];
        const uint32_t h = cvt_pk(a, b);
        const float ra = a - lo16(h), rb = b - hi16(h);              // exact
        const uint32_t m = cvt_pk(ra, rb);
        const float sa = ra - lo16(m), sb = rb - hi16(m);            // exact
        hi[e] = h; mid[e] = m; lo[e] = cvt_pk(sa, sb);
    }
}

__global__ __launch_bounds__(256, 2) void k_gemm_f32s(GemmF32Args p, int n_tiles_n) {
    __shared__ __attribute__((aligned(16))) unsigned char sm[2][2][SP_OPER];     // [buffer][A | W]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // tile order (as gemm256.hip): block b runs on XCD b % 8 -> every XCD gets a contiguous run of tiles, and inside it FGROUP m-tiles walk the n-tiles
    // together, so that the blocks in flight on an XCD share their W panels (and their A panels) in ITS L2.  With the plain m-major order the
    // XCDs each streamed all of W once per m-tile from the memory side (1.4 GB for the 8192 x 4608 x 1152 launch against 59 MB of operands)
    int bid = blockIdx.x;
    {
        const int nblk = gridDim.x, q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int ntm = gridDim.x / n_tiles_n, per_group = FGROUP * n_tiles_n;
    const int gid = bid / per_group, first_m = gid * FGROUP, gsz = min(ntm - first_m, FGROUP);
    const int m0 = (first_m + (bid % per_group) % gsz) * GB, n0 = ((bid % per_group) / gsz) * GB;
    const int sr = tid >> 1, sk = tid & 1;                                       // staging: row sr, k half sk
    // rows beyond M / N are read from the last valid row instead: an output element depends on ITS row of A and ITS row of W only, and the
    // epilogue never stores rows >= M or columns >= N -- no zero fill, no conditional loads.  Buffer loads (descriptor + per-thread byte offset
    // fixed for the whole k loop + the k-tile's offset in an SGPR): no per-iteration address arithmetic in VGPRs.
    const __amdgpu_buffer_rsrc_t ra_ = sr_rsrc(p.A, (unsigned)(((size_t)(p.M - 1) * p.lda + p.K) * 4));
    const __amdgpu_buffer_rsrc_t rw_ = sr_rsrc(p.W, (unsigned)((size_t)p.N * p.K * 4));
    const int arow_i = min(m0 + sr, p.M - 1), wrow_i = min(n0 + sr, p.N - 1);
    const unsigned avo = ((unsigned)arow_i * p.lda + sk * 8) * 4;
    const unsigned wvo = ((unsigned)wrow_i * p.K + sk * 8) * 4;
    // TWO k-tiles of global loads in flight per thread (register sets 0 / 1 alternate): with one, a k-tile lasted one memory round trip
    // (~3 200 cycles against the 768 of its 24 MFMAs: 2 blocks per CU do not cover it the way the f32-input kernel's four do)
    u32x4s ra[2][2], rw[2][2];
    auto gload = [&](auto set_, int k0) {
        constexpr int S = decltype(set_)::value;
        ra[S][0] = __builtin_amdgcn_raw_buffer_load_b128(ra_, avo, k0 * 4, 0);
        ra[S][1] = __builtin_amdgcn_raw_buffer_load_b128(ra_, avo + 16, k0 * 4, 0);
        rw[S][0] = __builtin_amdgcn_raw_buffer_load_b128(rw_, wvo, k0 * 4, 0);
        rw[S][1] = __builtin_amdgcn_raw_buffer_load_b128(rw_, wvo + 16, k0 * 4, 0);
    };
    const int soff = sk * SP_HALF + sr * 16;
    auto lstore = [&](auto set_, int b) {
        constexpr int S = decltype(set_)::value;
        u32x4s h, m, l;
        float xa[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) xa[e] = __uint_as_float(ra[S][e >> 2][e & 3]);
        split3(xa, h, m, l);
        *reinterpret_cast<u32x4s*>(sm[b][0] + soff) = h;
        *reinterpret_cast<u32x4s*>(sm[b][0] + SP_PLANE + soff) = m;
        *reinterpret_cast<u32x4s*>(sm[b][0] + 2 * SP_PLANE + soff) = l;
        float xw[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) xw[e] = __uint_as_float(rw[S][e >> 2][e & 3]);
        split3(xw, h, m, l);
        *reinterpret_cast<u32x4s*>(sm[b][1] + soff) = h;
        *reinterpret_cast<u32x4s*>(sm[b][1] + SP_PLANE + soff) = m;
        *reinterpret_cast<u32x4s*>(sm[b][1] + 2 * SP_PLANE + soff) = l;
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    f32x16 acc[2][2], cor[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) { acc[j][i][v] = 0.f; cor[j][i][v] = 0.f; }
    const int nk = p.K / GK;
    gload(S0{}, 0);
    gload(S1{}, min(1, nk - 1) * GK);
    lstore(S0{}, 0);
    __syncthreads();
    const int kg = lane >> 5, col = lane & 31;
    const int roff = kg * SP_HALF + col * 16;                                     // fragment: row col of its 32-row tile, k half kg
    // k-tile kt: LDS buffer kt & 1 holds it, register set (kt + 1) & 1 holds k-tile kt + 1 (loaded one iteration ago), set kt & 1 is free for kt + 2
    auto ktile = [&](auto cur_, int kt) {
        constexpr int cur = decltype(cur_)::value;
        using NXT = std::integral_constant<int, cur ^ 1>;
        gload(cur_, min(kt + 2, nk - 1) * GK);      // UNCONDITIONAL (past the end: the last k-tile again, never stored): behind a branch the compiler cannot count
                                                    // these loads and makes the store of k-tile kt + 1 below wait for ALL of them (vmcnt(0) instead of vmcnt(5))
        bf16x8 af[2][3], wf[2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                af[t][pl] = *reinterpret_cast<const bf16x8*>(sm[cur][0] + pl * SP_PLANE + roff + (wm * 64 + t * 32) * 16);
                wf[t][pl] = *reinterpret_cast<const bf16x8*>(sm[cur][1] + pl * SP_PLANE + roff + (wn * 64 + t * 32) * 16);
            }
        // six partial products, the smallest first; each pass runs over the four (j, i) accumulators, so an MFMA never waits for the one before it
        // (back-to-back MFMAs on ONE accumulator issue at their 64-cycle latency instead of the pipe's 32-cycle rate)
#define SR_PASS(ACC, WP, AP)                                                                                               \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                      \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                  \
                ACC[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j][WP], af[i][AP], ACC[j][i], 0, 0, 0);
        SR_PASS(cor, 2, 0)
        SR_PASS(cor, 0, 2)
        SR_PASS(cor, 1, 1)
        SR_PASS(cor, 1, 0)
        SR_PASS(cor, 0, 1)
        SR_PASS(acc, 0, 0)
#undef SR_PASS
        if (kt + 1 < nk) lstore(NXT{}, cur ^ 1);        // (waits for the loads of k-tile kt + 1 only: those of kt + 2 stay in flight)
        __syncthreads();
    };
    for (int kt = 0; kt < nk; kt += 2) {
        ktile(S0{}, kt);
        if (kt + 1 < nk) ktile(S1{}, kt + 1);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[j][i][v] += cor[j][i][v];
    // epilogue: identical to k_gemm_f32's (same accumulator layout)
    auto epilogue = [&](auto hb_, auto hr_, auto act_) {
        constexpr bool HB = decltype(hb_)::value, HR = decltype(hr_)::value;
        constexpr int ACT = decltype(act_)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + wm * 64 + i * 32 + col;
            if (m >= p.M) continue;
            const size_t orow = (size_t)(p.rowmap ? p.rowmap[m] : m) * p.ldo;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float4 rv[4];
                if constexpr (HR) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = n0 + wn * 64 + j * 32 + 8 * q + 4 * kg;
                        rv[q] = n < p.N ? *reinterpret_cast<const float4*>(p.resid + orow + n) : float4{0.f, 0.f, 0.f, 0.f};
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n0 + wn * 64 + j * 32 + 8 * q + 4 * kg;
                    if (n >= p.N) continue;
                    float o[4];
                    float4 bq = float4{0.f, 0.f, 0.f, 0.f};
                    if constexpr (HB) bq = *reinterpret_cast<const float4*>(p.bias + n);
                    const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = acc[j][i][4 * q + e];
                        if constexpr (HB) x += bb[e];
                        if constexpr (ACT == 1) x = gelu_f(x);
                        else if constexpr (ACT == 2) x = fmaxf(x, 0.f);
                        o[e] = x;
                    }
                    if constexpr (HR) {
                        const float4 r = rv[q];
                        o[0] = r.x + o[0]; o[1] = r.y + o[1]; o[2] = r.z + o[2]; o[3] = r.w + o[3];
                    }
                    *reinterpret_cast<float4*>(p.out + orow + n) = float4{o[0], o[1], o[2], o[3]};
                }
            }
        }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    auto by_act = [&](auto hb_, auto hr_) {
        if (p.act == 1) epilogue(hb_, hr_, std::integral_constant<int, 1>{});
        else if (p.act == 2) epilogue(hb_, hr_, std::integral_constant<int, 2>{});
        else epilogue(hb_, hr_, std::integral_constant<int, 0>{});
    };
    if (p.bias) { if (p.resid) by_act(T_{}, T_{}); else by_act(T_{}, F_{}); }
    else { if (p.resid) by_act(F_{}, T_{}); else by_act(F_{}, F_{}); }
}

// ------------------------------------------------------------------------------------------------ attention
// softmax(q k^T * scale) v per work item (AttnWork, kernels.h: one tile of <= 64 queries of one sequence / window), float32 throughout,
// online softmax over 64-key tiles.  One block = 4 waves x 16 queries of ONE head.  Everything is computed transposed, as in
// attention.hip: S^T = K . Q^T (v_mfma_f32_16x16x4_f32: A = K rows from LDS, B = Q^T held in registers for the whole pass), whose
// accumulator layout -- lane (g, c) holds keys 4 g .. 4 g + 3 of query c -- IS the B operand of O^T += V^T . P^T for the key
// permutation the V^T fragment reads follow, so P never moves between lanes.  K rows have pitch HD + 2 and V rows HD + 4 floats: both
// fragment reads are conflict-free ds_read_b32.
template <int HD>
// Registers: 196 at head_dim 80 with the next tile's 40 in flight = 2 blocks per CU instead of the 3 the LDS would allow.  Measured (SAM2 encoder, 8 tiles, same box):
// 98.2 ms of attention without the prefetch, 98.1 with it at 3 blocks per CU (28 dwords spilled), 106.9 with the prefetch issued behind the S^T MFMAs, **91.5** like this.
__global__ __launch_bounds__(256, 2) void k_attn_f32(AttnF32Args p) {
    constexpr int PK = HD + 2, PV = HD + 4, NS = HD / 4, ND = HD / 16;
    __shared__ __attribute__((aligned(16))) float Ks[64 * PK];
    __shared__ __attribute__((aligned(16))) float Vs[64 * PV];
    const AttnWork w = p.work[blockIdx.x];
    const int h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int nq = min(64, (w.q_len ? w.q_len : w.seq_len) - w.q_off);
    const int qi = wave * 16 + c;                                   // this lane's query inside the tile
    // causal (the LM of the float32 verification path): the item's queries are the LAST (q_len or seq_len) positions of its keys; query q_off + qi
    // sees keys 0 .. its own position.  Key 0 is visible to every query, so the running maximum is finite from the first tile on.
    const int kmax = p.causal ? w.seq_len - (w.q_len ? w.q_len : w.seq_len) + w.q_off + qi : 0x7fffffff;
    const bool qok = qi < nq;
    const bool wave_live = wave * 16 < nq;
    float qreg[NS];
    {
        const float* qp = p.q + (size_t)(w.q_row0 + (qok ? qi : 0)) * p.q_stride + h * HD + g;
#pragma unroll
        for (int s = 0; s < NS; ++s) qreg[s] = qok ? qp[4 * s] : 0.f;
    }
    f32x4 o[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mrun = -INFINITY, lrun = 0.f;
    const float* kbase = p.k + (size_t)w.k_row0 * p.k_stride + h * HD;
    const float* vbase = p.v + (size_t)w.k_row0 * p.v_stride + h * HD;
    // the K / V rows of a tile travel through registers into LDS; the NEXT tile's loads are issued right behind the barrier that publishes the current
    // one (round 5: they used to be issued, waited for and stored between the two barriers -- a memory round trip per tile with nothing under it)
    constexpr int NL = HD / 16;                                     // float4 per thread, tile and operand (64 keys x HD / 4 chunks over 256 threads)
    float4 kreg[NL], vreg[NL];
    auto fetch = [&](int j0) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = tid + i * 256, r = e / (HD / 4), d4 = (e % (HD / 4)) * 4;
            kreg[i] = float4{0.f, 0.f, 0.f, 0.f};
            vreg[i] = kreg[i];
            if (j0 + r < w.seq_len) {
                kreg[i] = *reinterpret_cast<const float4*>(kbase + (size_t)(j0 + r) * p.k_stride + d4);
                vreg[i] = *reinterpret_cast<const float4*>(vbase + (size_t)(j0 + r) * p.v_stride + d4);
            }
        }
    };
    fetch(0);
    for (int j0 = 0; j0 < w.seq_len; j0 += 64) {
        __syncthreads();                                            // the previous tile's fragment reads are done
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = tid + i * 256, r = e / (HD / 4), d4 = (e % (HD / 4)) * 4;
            float2* kd = reinterpret_cast<float2*>(Ks + r * PK + d4);          // (pitch HD + 2: rows are 8-byte aligned)
            kd[0] = float2{kreg[i].x, kreg[i].y};
            kd[1] = float2{kreg[i].z, kreg[i].w};
            *reinterpret_cast<float4*>(Vs + r * PV + d4) = vreg[i];
        }
        __syncthreads();
        if (j0 + 64 < w.seq_len) fetch(j0 + 64);
        if (!wave_live) continue;               // (round 5) a wave without a query keeps only the barriers: Hiera's 196-token windows leave 3 of a window's 16 waves empty
        // live 16-key sub-tiles of this 64-key tile (round 5): a window of 196 keys ends in a tile of 4 -- its three dead sub-tiles used to cost as much
        // as live ones (S = -inf, P = 0, 0 x V added: skipping them changes no bit)
        const int nt = min(4, (w.seq_len - j0 + 15) >> 4);
        // S^T tile t: keys 16 t + 4 g + v (v = 0..3) of query c
        f32x4 sc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) sc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* kr = Ks + c * PK + g;
        if (nt == 4) {
#pragma unroll
            for (int s = 0; s < NS; ++s)        // four independent accumulator chains: the 40-cycle dependent latency of the 16x16x4 form stays hidden
#pragma unroll
                for (int t = 0; t < 4; ++t) sc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[16 * t * PK + 4 * s], qreg[s], sc[t], 0, 0, 0);
        } else {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (t < nt) sc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[16 * t * PK + 4 * s], qreg[s], sc[t], 0, 0, 0);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int key = j0 + 16 * t + 4 * g + v;
                const bool live = key < w.seq_len && key <= kmax;
                sc[t][v] = live ? sc[t][v] * p.scale : -INFINITY;
                mx = fmaxf(mx, sc[t][v]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(mrun, mx);                         // finite: key j0 of every tile is live
        const float alpha = expf(mrun - mnew);                      // (exp(-inf) = 0 on the first tile)
        float ps = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                sc[t][v] = expf(sc[t][v] - mnew);
                ps += sc[t][v];
            }
        lrun = lrun * alpha + ps;
        mrun = mnew;
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            o[d][0] *= alpha; o[d][1] *= alpha; o[d][2] *= alpha; o[d][3] *= alpha;
        }
        // O^T[d][q] += sum_key V[key][d] P[key][q]: A = V^T fragment (row d = 16 dt + c, k = g -> key 16 t + 4 g + v), B = P^T (this lane's sc[t][v])
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t >= nt) continue;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float* vr = Vs + (16 * t + 4 * g + v) * PV + c;
#pragma unroll
                for (int d = 0; d < ND; ++d) o[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[16 * d], sc[t][v], o[d], 0, 0, 0);
            }
        }
    }
    lrun += __shfl_xor(lrun, 16, 64);
    lrun += __shfl_xor(lrun, 32, 64);
    if (!qok) return;
    const float inv = 1.0f / lrun;
    float* op = p.out + (size_t)(w.q_row0 + qi) * p.out_stride + h * HD + 4 * g;       // O^T accumulator: d = 16 dt + 4 g + v, query c
#pragma unroll
    for (int d = 0; d < ND; ++d) *reinterpret_cast<float4*>(op + 16 * d) = float4{o[d][0] * inv, o[d][1] * inv, o[d][2] * inv, o[d][3] * inv};
}

}  // namespace

int launch_gemm_f32(hipStream_t s, const GemmF32Args& a) {
    if (a.M <= 0 || a.N <= 0) return 0;
    if (a.K % GK || a.N % 4 || a.ldo % 4 || a.lda % 4 || !a.W || (((uintptr_t)a.A | (uintptr_t)a.W | (uintptr_t)a.out | (uintptr_t)a.resid | (uintptr_t)a.bias) & 15)) return -22;      // (bias is read 16 bytes at a time)
    const int tn = cdiv(a.N, GB), tm = cdiv(a.M, GB);
    // default: the split-bf16 form on the bf16 matrix pipe (SR_SAM_F32_SPLIT=0: the f32-input MFMA of round 4); the row stride of A and the
    // rows of W must keep the staging's 16-byte loads of 8 consecutive k aligned (lda % 4, K % 16: checked above)
    const bool fits32 = ((size_t)a.M * a.lda + a.K) * 4 < (1ull << 32) && (size_t)3 * a.N * a.K * 4 < (1ull << 32);      // buffer descriptors address 4 GB
    if (sr_switches().sam_f32_split && fits32) hipLaunchKernelGGL(k_gemm_f32s, dim3((unsigned)tn * (unsigned)tm), dim3(256), 0, s, a, tn);
    else hipLaunchKernelGGL(k_gemm_f32, dim3((unsigned)tn * (unsigned)tm), dim3(256), 0, s, a, tn);
    SR_CHECK_LAUNCH();
    return 0;
}

int launch_attn_f32(hipStream_t s, const AttnF32Args& a, int head_dim) {
    if (a.n_work <= 0) return 0;
    if ((a.q_stride | a.k_stride | a.v_stride | a.out_stride) % 4 || (((uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v | (uintptr_t)a.out) & 15)) return -22;
    const dim3 grid(a.n_work, a.n_heads);
    switch (head_dim) {
        case 16: hipLaunchKernelGGL(k_attn_f32<16>, grid, dim3(256), 0, s, a); break;
        case 32: hipLaunchKernelGGL(k_attn_f32<32>, grid, dim3(256), 0, s, a); break;
        case 80: hipLaunchKernelGGL(k_attn_f32<80>, grid, dim3(256), 0, s, a); break;
        case 128: hipLaunchKernelGGL(k_attn_f32<128>, grid, dim3(256), 0, s, a); break;      // (the LM's heads: float32 verification path only)
        default: return -22;
    }
    SR_CHECK_LAUNCH();
    return 0;
}
