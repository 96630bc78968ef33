// 256 x 256 x 64 bf16 MFMA GEMM with an 8-phase, ping-pong schedule for the large-M rows of the hot path (ViT and LM prefill at
// batch >= 8: SURVEY.md 2.3 K5, K8, K9, K13, K16).  Same contract and epilogues as gemm.hip (out = A[M,K] . W[N,K]^T, float32
// accumulation, HF's bf16 rounding points); this file only changes HOW the tile is fed.
//
// Why: the 128 x 128 kernel of gemm.hip tops out at 0.9-1.0 PF (MFMA pipe 46 % busy): every k-tile ends in a barrier that first
// drains the LDS-DMA queue (vmcnt(0)), so loads never span a barrier.  Here (MI355X guide, "256^2 8-phase template"):
//   * 8 waves (2 x 4), each 128 x 64 of the tile = 8 x 4 MFMA tiles of 16 x 16 x 32 -> 64 MFMAs per wave and k-tile, issued
//     as four "quadrant" phases of 16 (4 m-tiles x 2 n-tiles x 2 k-steps); fragments are (re)loaded per quadrant, so only
//     32 + 32 VGPRs hold operands next to the 128 accumulator registers;
//   * LDS: 2 k-tile buffers x (A 32 KB + W 32 KB) = 128 KB, filled by LDS-DMA in four 16 KB UNITS per k-tile, one unit per
//     phase: a unit is the set of rows ALL waves read in one phase (UA0 = the first 64 rows of both A halves, UA1 = the
//     second 64, UB0 / UB1 = the first / second 32 columns of every wave's 64), so a unit's buffer is free again as soon as
//     its phase is over and the next-but-one k-tile's unit can be streamed into it two phases later;
//   * loads stay in flight across barriers: raw s_barrier (no vmcnt drain) and ONE counted s_waitcnt vmcnt(4) per k-tile
//     (two units = 4 LDS-DMA instructions per wave may still be flying); a unit is read one phase after the wait + barrier
//     that retires it, and re-staged >= 2 phases after its last read;
//   * the two wave rows run half a phase apart (wm = 1 passes one extra barrier up front): on every SIMD one wave is in its
//     MFMA segment while its partner issues ds_reads / LDS-DMA, s_setprio(1) around the MFMA cluster.
// Phase plan of k-tile t (reads come from buffer t & 1):
//     p0: read UB0(t) [4 x b128] + UA0(t) [8]   stage UA1(t+1)        p1: read UB1(t) [4]      stage UB1(t+1)
//     p2: read UA1(t) [8]                       stage UA0(t+2)        p3: (B0 kept in VGPRs)   stage UB0(t+2); vmcnt(4)
#include "kernels.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int BM2 = 256, BN2 = 256, BK2 = 64;
constexpr int A_BYTES = BM2 * BK2 * 2, W_BYTES = BN2 * BK2 * 2, BUF_BYTES = A_BYTES + W_BYTES;   // 32 KB + 32 KB

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ int swz2(int row, int chunk) { return row * (BK2 * 2) + ((chunk ^ (row & 7)) << 4); }

// MX = true: the fp8 path of BASELINE.json configs[4].  Operands are OCP e4m3 bytes, 128 of them (= the same 128 bytes) per tile row
// and k-tile, multiplied by v_mfma_scale_f32_16x16x128_f8f6f4 -- the block-scaled form, which is the one that runs at twice the bf16
// rate (plain fp8 MFMA does not).  Everything about the data movement stays: same units, same swizzle, same phase plan; the byte
// geometry of a k-tile is identical, it just spans 128 k instead of 64.  Operand layout (established on the hardware by
// tools/mx_probe): lane (g = l >> 4, r = l & 15) supplies row r's bytes k = 16g..16g+15 and 64+16g..64+16g+15 of the 128-chunk --
// exactly the two 16-byte pieces the bf16 kernel's two k-steps read -- and the e8m0 scale of the row's 32-wide k-block b comes from
// lane g = b.  Activations: fp8 bytes [M][K] row-major + scales [K/128][ceil256(M)][4] (k_quant_mx_act); the 1 KB of scales of a
// k-tile rides along with unit UA1 (one more LDS-DMA by wave 0).  Weights: the decode stream's fp8 image (tiled8, common.h), whose
// 2 KB per (16-row tile, 128 k) already IS two operand fragments; its per-output-channel float32 scale stays in the epilogue, the
// block scale of the weight operand is 1 (e8m0 127).
// -DSR_G256_TIMING (tools/probe_gemm256_timeline.py builds its own library with it; never the product build): wave 0 of every block
// records the 100 MHz clock at entry / after the prologue / after the k loop / after the epilogue's stores have drained, and the CU it ran on
#ifdef SR_G256_TIMING
__device__ long long g_t256[16384 * 6];
#define T256(slot, v) do { if (threadIdx.x == 0 && blockIdx.x < 16384) g_t256[blockIdx.x * 6 + (slot)] = (v); } while (0)
// the phases of ONE steady-state k-tile (t == 5) in shader clocks (s_memtime), for the first wave of each wave row (threads 0 and 256)
__device__ long long g_p256[1024 * 2 * 6 + 2048];      // (+ per block: s_memtime at the start and at the end of the k loop, next to T256's wall clock)
#define C256(slot) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_p256[1024 * 2 * 6 + blockIdx.x * 2 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define P256(slot) do { if (t == 5 && (threadIdx.x & 255) == 0 && blockIdx.x < 1024) g_p256[(blockIdx.x * 2 + (threadIdx.x >> 8)) * 6 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define T256(slot, v)
#define P256(slot)
#define C256(slot)
#endif
template <int EPI, bool MX>
__global__ __launch_bounds__(512) void k_gemm256(GemmArgs p, int ntm, int ntn, int GROUP) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // ONE array (a second __shared__ object de-pipelines)
    T256(0, wall_clock64());
    T256(4, ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4));
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    {   // XCD-aware: block b runs on XCD b % 8 -> give every XCD a contiguous run of tiles (bijective for any grid size)
        const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // GROUP m-tiles share their W panels while walking n
    const int per_group = GROUP * ntn;
    const int gid = bid / per_group, first_m = gid * GROUP, gsz = min(ntm - first_m, GROUP);
    const int tm = first_m + (bid % per_group) % gsz, tn = (bid % per_group) / gsz;
    const int m0 = tm * BM2, n0 = tn * BN2;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const bool tiled = p.w_tiled != 0;

    // ---- LDS-DMA sources.  Every unit is 16 pieces of 1 KB (8 tile rows x 128 B, or one k-step half of a fragment-ordered
    // W tile); wave w issues pieces 2w and 2w + 1.
    const int lr = lane >> 3, lc = ((lane & 7) ^ lr) * 16;                    // XOR swizzle on the SOURCE side (LDS-DMA writes linearly); bytes
    constexpr int ES = MX ? 1 : 2;                                            // bytes per element
    const unsigned char* Ab = reinterpret_cast<const unsigned char*>(p.A);
    const unsigned char* Wb = reinterpret_cast<const unsigned char*>(p.W);
    const int nk = p.K / (MX ? 128 : BK2);                                    // k-tiles of 128 bytes per row
    const unsigned char* asrc[2][2];                                          // [unit half][piece]
    int adst[2][2];
    const unsigned char* wsrc[2][2];
    int wdst[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int g = wave * 2 + i;                                       // 8-row group inside the unit: 0..15
            const int ru = g * 8;                                             // first unit row of the group
            const int trow = (ru < 64 ? ru : ru + 64) + h * 64;               // tile row: unit h = rows [h*64, +64) of both A halves
            asrc[h][i] = Ab + (size_t)min(m0 + trow + lr, p.M - 1) * p.lda * ES + lc;
            adst[h][i] = trow * 128;
            if (tiled) {      // piece = (n-tile of the unit, k-step): the unit's 8 n-tiles are tiles {wn*4 + h*2 + (0,1)}
                const int ntu = g >> 1, ks = g & 1;
                const int ntile = (ntu >> 1) * 4 + h * 2 + (ntu & 1);
                // bf16 tiled16x64: 2 KB per (tile, 64 k) = two k-step halves; fp8 tiled8: 1 KB per (tile, 64 k), two of them per k-tile
                wsrc[h][i] = Wb + (size_t)(n0 / 16 + ntile) * (size_t)nk * 2048 + ks * 1024 + lane * 16;
                wdst[h][i] = ntile * 2048 + ks * 1024;
            } else {          // row-major: unit h = n-rows [wn*64 + h*32, +32) of every wave column
                const int nrow = (ru >> 5) * 64 + h * 32 + (ru & 31);
                wsrc[h][i] = Wb + (size_t)min(n0 + nrow + lr, p.N - 1) * p.K * ES + lc;
                wdst[h][i] = nrow * 128;
            }
        }
    const size_t w_kt = tiled ? 2048 : 128;                                   // bytes per k-tile
    // MX: the k-tile's activation scales, 256 rows x 4 bytes = 1 KB, [k-tile][row][4]; staged by wave 0 together with unit UA1
    const unsigned char* ssrc = MX ? p.a_scale + (size_t)m0 * 4 + lane * 16 : nullptr;
    auto stage_a = [&](int h, int kt) {
        unsigned char* base = smem + (kt & 1) * BUF_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[h][i] + (size_t)kt * 128), (lptr_t)(base + adst[h][i]), 16, 0, 0);
        if constexpr (MX) {
            if (h == 1 && wave == 0)
                __builtin_amdgcn_global_load_lds((gptr_t)(ssrc + (size_t)kt * p.a_rows_pad * 4), (lptr_t)(smem + 2 * BUF_BYTES + (kt & 1) * 1024), 16, 0, 0);
        }
    };
    auto stage_w = [&](int h, int kt) {
        unsigned char* base = smem + (kt & 1) * BUF_BYTES + A_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[h][i] + (size_t)kt * w_kt), (lptr_t)(base + wdst[h][i]), 16, 0, 0);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // operand fragments of one quadrant.  bf16: [tile][k-step] 4 VGPRs each; MX: ONE 8-VGPR operand per tile (the two 16-byte pieces
    // are loaded straight into its halves, so that no register moves are needed to form the 8-register MFMA source)
    typedef __attribute__((ext_vector_type(8))) int i32x8;
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    struct Frag { bf16x8 h[2]; };
    typedef typename std::conditional<MX, i32x8, Frag>::type frag_t;
    frag_t af[4], wf0[2], wf1[2];

    // fragment addresses (bytes inside a buffer)
    int a_off[2][2];                                 // [m-half quadrant base: tile 0][k-step] -> + t * 16 rows * 128 B
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        a_off[0][kk] = swz2(wm * 128 + fr, kk * 4 + fg);          // rows (wm*128 + t*16 + fr): the swizzle only depends on fr & 7
        a_off[1][kk] = swz2(wm * 128 + 64 + fr, kk * 4 + fg);
    }
    int w_off[2][2];                                 // [n-half][k-step] of the half's tile 0 -> + t * (2048 or 16 * 128)
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
            w_off[nh][kk] = A_BYTES + (!tiled ? swz2(wn * 64 + nh * 32 + fr, kk * 4 + fg)
                                       : MX ? (wn * 4 + nh * 2) * 2048 + kk * 1024 + (lane << 4)       // tiled8: [64-k half][lane][16 B]
                                            : (wn * 4 + nh * 2) * 2048 + (fg & 1) * 1024 + (((kk * 2 + (fg >> 1)) * 16 + fr) << 4));
    const int w_tile = tiled ? 2048 : 16 * 128;

    auto load_frag = [&](const unsigned char* p0, const unsigned char* p1, frag_t& f) {
        if constexpr (MX) {
            const i32x4 lo = *reinterpret_cast<const i32x4*>(p0), hi = *reinterpret_cast<const i32x4*>(p1);
            f = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        } else {
            f.h[0] = *reinterpret_cast<const bf16x8*>(p0);
            f.h[1] = *reinterpret_cast<const bf16x8*>(p1);
        }
    };
    auto read_a = [&](const unsigned char* buf, int mh) {
#pragma unroll
        for (int t = 0; t < 4; ++t) load_frag(buf + a_off[mh][0] + t * 2048, buf + a_off[mh][1] + t * 2048, af[t]);
    };
    auto read_w = [&](const unsigned char* buf, int nh, frag_t (&wf)[2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) load_frag(buf + w_off[nh][0] + t * w_tile, buf + w_off[nh][1] + t * w_tile, wf[t]);
    };
    int asc[4];                                      // MX: e8m0 scale of (m-tile row, this lane's k-block fg) for the 4 m-tiles of the current half
    auto read_scales = [&](int t, int mh) {
        if constexpr (MX) {
            const unsigned char* sb = smem + 2 * BUF_BYTES + (t & 1) * 1024;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                asc[i] = *reinterpret_cast<const int*>(sb + (wm * 128 + (mh * 4 + i) * 16 + fr) * 4) >> (8 * fg);   // the MFMA reads byte 0
        }
    };
    auto quad = [&](int mh, int nh, frag_t (&wf)[2]) {
        if constexpr (MX) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[mh * 4 + i][nh * 2 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[j], af[i], acc[mh * 4 + i][nh * 2 + j], 0, 0, 0, 127, 0, asc[i]);
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[mh * 4 + i][nh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j].h[kk], af[i].h[kk], acc[mh * 4 + i][nh * 2 + j], 0, 0, 0);
        }
    };
    // MX only: hipcc (ROCm 7.2) does not keep the block-scaled MFMAs inside their phase -- sched_barrier notwithstanding it sinks
    // the quadrants of three phases into one 24-MFMA run at the end of the k-tile (ping-pong gone, every fragment alive at once,
    // spills).  An empty volatile asm that "modifies" the quadrant's accumulators before and after the MFMAs ties them to the
    // barriers (volatile asm keeps its order relative to the barrier builtins and the other asm statements).
    auto pin = [&](int mh, int nh) {
        if constexpr (MX) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(acc[mh * 4 + i][nh * 2 + j]));
        }
    };
#define PHASE_SYNC_COMPUTE(MH, NH, WF)                       \
    __builtin_amdgcn_sched_barrier(0);                       \
    __builtin_amdgcn_s_barrier();                            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
    __builtin_amdgcn_sched_barrier(0);                       \
    pin(MH, NH);                                             \
    __builtin_amdgcn_s_setprio(1);                           \
    quad(MH, NH, WF);                                        \
    __builtin_amdgcn_s_setprio(0);                           \
    pin(MH, NH);                                             \
    __builtin_amdgcn_sched_barrier(0);                       \
    __builtin_amdgcn_s_barrier();                            \
    __builtin_amdgcn_sched_barrier(0);

    // the lane's 16 bias values are fetched here, under the prologue's wait (8 registers through the k loop; the MX kernel has none to spare
    // and fetches them at the head of its epilogue)
    uint2 bia[4];
    if constexpr (!MX) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bia[j] = p.bias ? *reinterpret_cast<const uint2*>(p.bias + n0 + wn * 64 + j * 16 + fg * 4) : uint2{0, 0};
    }
    // ---- prologue: k-tile 0 completely, plus the two units of k-tile 1 the steady state has already issued by then
    stage_a(0, 0); stage_w(0, 0); stage_w(1, 0); stage_a(1, 0);
    if (nk > 1) { stage_a(0, 1); stage_w(MX ? 1 : 0, 1); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();       // the second wave row runs half a phase behind the first
    T256(1, wall_clock64());
    C256(0);

    for (int t = 0; t < nk; ++t) {
        const unsigned char* buf = smem + (t & 1) * BUF_BYTES;
        P256(0);
        // p0
        read_w(buf, 0, wf0);
        read_a(buf, 0);
        read_scales(t, 0);
        if (t + 1 < nk) stage_a(1, t + 1);
        PHASE_SYNC_COMPUTE(0, 0, wf0)
        P256(1);
        // p1
        read_w(buf, 1, wf1);
        if (t + 1 < nk) stage_w(MX ? 0 : 1, t + 1);       // MX re-reads UB0 in p3, so UB0 / UB1 trade places in the staging order
        PHASE_SYNC_COMPUTE(0, 1, wf1)
        P256(2);
        // p2
        read_a(buf, 1);
        read_scales(t, 1);
        if (t + 2 < nk) stage_a(0, t + 2);
        PHASE_SYNC_COMPUTE(1, 1, wf1)
        P256(3);
        // p3 (MX: the 8-register operands leave no room to keep B0 alive through p1 / p2 -- it is re-read here, like the guide's
        // template does.  UB0 is then busy until p3, so MX stages UA1(t+1) | UB0(t+1) | UA0(t+2) | UB1(t+2) in p0..p3: every unit is
        // still re-staged >= 2 phases after its last read, and vmcnt(4) at p3 still retires everything k-tile t + 1 needs)
        if constexpr (MX) read_w(buf, 0, wf0);
        if (t + 2 < nk) { stage_w(MX ? 1 : 0, t + 2); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PHASE_SYNC_COMPUTE(1, 0, wf0)
        P256(4);
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();       // balance the extra barrier of the other wave row
#undef PHASE_SYNC_COMPUTE
    T256(2, wall_clock64());
    C256(1);

    // ---- epilogue (same rounding points as gemm.hip).  lane owns row m = .. + fr and columns n = .. + fg*4 + {0..3}.
    // Column-only operands (bias, fp8 scale) and the row map are fetched once, up front: no memory wait inside the store loop.
    int orow[8];
    bool rok[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + wm * 128 + i * 16 + fr;
        rok[i] = m < p.M;
        orow[i] = (rok[i] && p.rowmap) ? p.rowmap[m] : m;
    }
    float4 scl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + fg * 4;
        if constexpr (MX) bia[j] = p.bias ? *reinterpret_cast<const uint2*>(p.bias + n) : uint2{0, 0};
        scl[j] = p.w_scale ? *reinterpret_cast<const float4*>(p.w_scale + n) : float4{1.f, 1.f, 1.f, 1.f};
    }
    // 16-byte stores: a lane owns 4 consecutive columns (8 B) of a row in every 16-column tile; the lane 16 further on owns the next
    // 4.  v_permlane16_swap trades the odd 16-lane rows of one register with the even rows of another, so after swapping the packed
    // halves of two neighbouring tiles (A, B) every even-row lane holds 8 consecutive columns of tile A and every odd-row lane 8
    // consecutive columns of tile B: one dwordx4 store per lane and tile pair instead of two dwordx2 (16 rows x 64 B per
    // instruction instead of 16 x 32 B).
    auto widen = [&](uint2 ta, uint2 tb) -> uint4 {
        const auto sx = __builtin_amdgcn_permlane16_swap(ta.x, tb.x, false, false);
        const auto sy = __builtin_amdgcn_permlane16_swap(ta.y, tb.y, false, false);
        return uint4{sx[0], sy[0], sx[1], sy[1]};
    };
    if constexpr (EPI == EPI_VITQKV) {
        // ViT qkv Linear with k_vit_rope + k_vit_vtranspose folded in.  The q / k channels of every head are in the paired order
        // (vit_qk_perm): inside each 16-column tile columns 0..7 are x1 of 8 rotary pairs and 8..15 their x2, i.e. the partner of a
        // lane's 4 columns sits in lane ^ 32.  Sections (q | k | v, C columns each) start on tile boundaries: the role is block-uniform.
        const VitRope& f = p.vrope;
        const int sec = n0 / f.C, half = f.hd / 2;
        if (sec < 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                uint2 t4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ct = n0 - sec * f.C + wn * 64 + j * 16;               // first column of the tile inside its section
                    const int jl = (ct % f.hd) / 16 * 8 + (fg & 1) * 4;             // rotary pair of this lane's first column
                    const float4 c4 = rok[i] ? *reinterpret_cast<const float4*>(f.cos_t + (size_t)orow[i] * half + jl) : float4{0.f, 0.f, 0.f, 0.f};
                    const float4 s4 = rok[i] ? *reinterpret_cast<const float4*>(f.sin_t + (size_t)orow[i] * half + jl) : float4{0.f, 0.f, 0.f, 0.f};
                    float o[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    if (p.bias) { o[0] += lo16(bia[j].x); o[1] += hi16(bia[j].x); o[2] += lo16(bia[j].y); o[3] += hi16(bia[j].y); }
                    const uint2 own = uint2{pack2(o[0], o[1]), pack2(o[2], o[3])};   // the Linear's bf16 output
                    const uint2 oth = uint2{(uint32_t)__shfl_xor((int)own.x, 32, 64), (uint32_t)__shfl_xor((int)own.y, 32, 64)};
                    const float me[4] = {lo16(own.x), hi16(own.x), lo16(own.y), hi16(own.y)};
                    const float ot[4] = {lo16(oth.x), hi16(oth.x), lo16(oth.y), hi16(oth.y)};
                    const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r)      // float32 mul, mul, add (hf:124-135); x1 holder (fg < 2): x1 c - x2 s, x2 holder: x2 c + x1 s
                        o[r] = __fadd_rn(__fmul_rn(me[r], cc[r]), __fmul_rn(fg < 2 ? -ot[r] : ot[r], ss[r]));
                    t4[j] = uint2{pack2(o[0], o[1]), pack2(o[2], o[3])};
                }
                const int odd_ = fg & 1, half8_ = (fg >> 1) * 8;
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    const uint4 v = widen(t4[2 * jp], t4[2 * jp + 1]);
                    const int n = n0 + wn * 64 + (2 * jp + odd_) * 16 + half8_;
                    if (rok[i]) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)orow[i] * p.ldo + n) = v;
                }
            }
        } else {
            // V^T[channel][row]: transposed through LDS (the staging buffers are dead), 4 row tiles per pass, so that every lane stores
            // 8 consecutive rows of one channel with one 16-byte store
            constexpr int EX_RS = 136;
            unsigned char* ex = smem + wave * 64 * EX_RS;
            __syncthreads();                                     // every wave is done with the staging buffers
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const int i = pass * 4 + ii;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float o[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                        if (p.bias) { o[0] += lo16(bia[j].x); o[1] += hi16(bia[j].x); o[2] += lo16(bia[j].y); o[3] += hi16(bia[j].y); }
                        *reinterpret_cast<uint2*>(ex + (ii * 16 + fr) * EX_RS + (j * 16 + fg * 4) * 2) = uint2{pack2(o[0], o[1]), pack2(o[2], o[3])};
                    }
                }
                __syncthreads();
                const int ch = n0 - 2 * f.C + wn * 64 + lane;                          // V channel (= head * hd + d) of this lane
                const int rbase = m0 + wm * 128 + pass * 64;
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    uint32_t w4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint32_t lo = *reinterpret_cast<const unsigned short*>(ex + (g * 8 + 2 * e) * EX_RS + lane * 2);
                        const uint32_t hi = *reinterpret_cast<const unsigned short*>(ex + (g * 8 + 2 * e + 1) * EX_RS + lane * 2);
                        w4[e] = lo | (hi << 16);
                    }
                    if (rbase + g * 8 < p.M) *reinterpret_cast<uint4*>(f.vt + (size_t)ch * f.vt_stride + rbase + g * 8) = uint4{w4[0], w4[1], w4[2], w4[3]};
                }
                __syncthreads();
            }
        }
        return;
    }
    if constexpr (EPI == EPI_LMQKV) {
        // q/k/v Linear of the LM prefill with k_lm_rope_prefill folded in (same bf16 rounding points, bit-identical results):
        //   q heads : bias, round, mRoPE (hf:557-599: three bf16 ops) -> out;   k heads: the same -> K cache row (slot, idx)
        //   v heads : bias, round -> V^T cache column (slot, idx)
        // The 256-column tile holds whole heads (128 columns = the two wave columns 2h, 2h + 1): the rotary partner of column d
        // (d +- 64) lives in the neighbouring wave at the same (row, column offset) and is traded through LDS (the staging buffers are
        // dead by now), 4 row tiles per pass.  q / k / v sections start on tile boundaries (launcher check), so the role is block-uniform.
        const QkvRope& f = p.rope;
        const int q_cols = f.n_q_heads * 128, qk_cols = q_cols + f.n_kv_heads * 128;
        const bool is_v = n0 >= qk_cols, is_k = !is_v && n0 >= q_cols;
        const int hi = wn & 1;                                   // 0: this wave holds x1 (d < 64), 1: x2 (d >= 64)
        const int kvh = ((is_v ? n0 - qk_cols : n0 - q_cols) >> 7) + (wn >> 1);
        constexpr int EX_RS = 136;                               // exchange row stride in bytes (64 columns + 8 B pad)
        unsigned char* ex = smem;
        __syncthreads();                                         // every wave is done with the staging buffers
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {      // (fully unrolled: a runtime `pass` would index the accumulators dynamically -> scratch)
            uint2 own[4][4];
            int pos3v[4][3], slot[4], cidx[4];
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int i = pass * 4 + ii;
                const int m = m0 + wm * 128 + i * 16 + fr;
                const int mc = rok[i] ? m : 0;
                if (!is_v) {
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) pos3v[ii][ax] = f.pos3[(size_t)ax * f.n_tok + mc];
                }
                if (is_k) { slot[ii] = f.tok_slot[mc]; cidx[ii] = f.tok_idx[mc]; }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float o[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    if (p.w_scale) { o[0] *= scl[j].x; o[1] *= scl[j].y; o[2] *= scl[j].z; o[3] *= scl[j].w; }
                    if (p.bias) { o[0] += lo16(bia[j].x); o[1] += hi16(bia[j].x); o[2] += lo16(bia[j].y); o[3] += hi16(bia[j].y); }
                    own[ii][j] = uint2{pack2(o[0], o[1]), pack2(o[2], o[3])};
                    *reinterpret_cast<uint2*>(ex + (wave * 64 + ii * 16 + fr) * EX_RS + (j * 16 + fg * 4) * 2) = own[ii][j];
                }
            }
            if (is_v) {
                // V^T[channel][token]: transposed through the exchange buffer so that a lane owns one channel and 8 consecutive rows.  When
                // those rows are 8 consecutive, 8-aligned cache positions of one sequence (always, for prompts that start on a multiple of 8
                // packed rows) they leave as one 16-byte store, otherwise token by token.
                __syncthreads();
                const unsigned char* exw = ex + wave * 64 * EX_RS + lane * 2;
                const int rbase = m0 + wm * 128 + pass * 64;
#pragma unroll 2
                for (int g = 0; g < 8; ++g) {
                    const int r0 = rbase + g * 8;
                    if (r0 >= p.M) break;
                    uint32_t w4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint32_t lo = *reinterpret_cast<const unsigned short*>(exw + (g * 8 + 2 * e) * EX_RS);
                        const uint32_t hi16v = *reinterpret_cast<const unsigned short*>(exw + (g * 8 + 2 * e + 1) * EX_RS);
                        w4[e] = lo | (hi16v << 16);
                    }
                    const int r7 = min(r0 + 7, p.M - 1);
                    const int s0 = f.tok_slot[r0], i0 = f.tok_idx[r0], s7 = f.tok_slot[r7], i7 = f.tok_idx[r7];
                    const size_t chan = (size_t)kvh * 128 + hi * 64 + lane;
                    if (r0 + 7 < p.M && s7 == s0 && i7 == i0 + 7 && (i0 & 7) == 0) {
                        *reinterpret_cast<uint4*>(f.vtcache + ((size_t)s0 * f.n_kv_heads * 128 + chan) * f.ctx_max + i0) = uint4{w4[0], w4[1], w4[2], w4[3]};
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int r = r0 + e;
                            if (r < p.M)
                                f.vtcache[((size_t)f.tok_slot[r] * f.n_kv_heads * 128 + chan) * f.ctx_max + f.tok_idx[r]] = (bf16_t)((w4[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
                        }
                    }
                }
                __syncthreads();                                 // the next pass overwrites the exchange buffer
                continue;
            }
            __syncthreads();
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int i = pass * 4 + ii;
                uint2 t4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int d0 = j * 16 + fg * 4;              // rotary pair index of this lane's first column (4 consecutive share an axis)
                    const int ax = d0 < f.sec0 ? 0 : (d0 < f.sec1 ? 1 : 2);
                    const size_t tb = (size_t)pos3v[ii][ax] * 64 + d0;
                    const uint2 c4 = *reinterpret_cast<const uint2*>(f.rope_cos + tb), s4 = *reinterpret_cast<const uint2*>(f.rope_sin + tb);
                    const uint2 oth = *reinterpret_cast<const uint2*>(ex + ((wave ^ 1) * 64 + ii * 16 + fr) * EX_RS + (j * 16 + fg * 4) * 2);
                    const float me[4] = {lo16(own[ii][j].x), hi16(own[ii][j].x), lo16(own[ii][j].y), hi16(own[ii][j].y)};
                    const float ot[4] = {lo16(oth.x), hi16(oth.x), lo16(oth.y), hi16(oth.y)};
                    const float cc[4] = {lo16(c4.x), hi16(c4.x), lo16(c4.y), hi16(c4.y)}, ss[4] = {lo16(s4.x), hi16(s4.x), lo16(s4.y), hi16(s4.y)};
                    float o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r)      // x1 holder: bf16(bf16(x1 c) + bf16(-x2 s));  x2 holder: bf16(bf16(x2 c) + bf16(x1 s))
                        o[r] = rbf(rbf(me[r] * cc[r]) + rbf((hi ? ot[r] : -ot[r]) * ss[r]));
                    t4[j] = uint2{pack2(o[0], o[1]), pack2(o[2], o[3])};
                }
                const int odd_ = fg & 1, half8_ = (fg >> 1) * 8;
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    const uint4 v = widen(t4[2 * jp], t4[2 * jp + 1]);
                    const int cw = (2 * jp + odd_) * 16 + half8_;            // column inside the wave's 64
                    if (!rok[i]) continue;
                    if (is_k) *reinterpret_cast<uint4*>(f.kcache + ((size_t)(slot[ii] * f.n_kv_heads + kvh) * f.ctx_max + cidx[ii]) * 128 + hi * 64 + cw) = v;
                    else *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)orow[i] * p.ldo + n0 + wn * 64 + cw) = v;
                }
            }
            __syncthreads();                                     // the next pass overwrites the exchange buffer
        }
        return;
    }
    const int odd = fg & 1, half8 = (fg >> 1) * 8;            // which tile of a pair this lane stores, and its 8-column half
    // EPI_RESID: ALL sixteen 16-byte residual loads of the lane go out before the first store.  The residual may alias the output (the
    // engine adds in place), so the compiler must keep a load behind every earlier store: left inside the row loop that made eight
    // load -> wait -> store round trips per tile (10-14 us of a 42-93 us ViT tile, tools/probe_gemm256_timeline.py).  Every 16 bytes
    // are read and later written by the same lane only, so reading them all first is the same computation; the operand fragments are
    // dead here and their 64 registers hold the values.
    uint4 rl[8][2];
    if constexpr (EPI == EPI_RESID) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int jp = 0; jp < 2; ++jp)
                rl[i][jp] = rok[i] ? *reinterpret_cast<const uint4*>(p.resid + (size_t)orow[i] * p.ldo + n0 + wn * 64 + (2 * jp + odd) * 16 + half8)
                                   : uint4{0, 0, 0, 0};
    }
    // The row loop is instantiated for the four (fp8 channel scale, bias) cases and picked by two block-uniform branches: left as run-time
    // conditions inside the loop they became a multiply / add and a v_cndmask per accumulator (512 instructions of the ~2200 of the SwiGLU
    // epilogue, which is VALU-bound: the two waves of a SIMD share it, tools/probe_gemm256_timeline.py).  Same arithmetic, same order.
    auto rows = [&](auto hs_, auto hb_) {
    constexpr bool HS = decltype(hs_)::value, HB = decltype(hb_)::value;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        // (no early exit on masked rows: the lane exchange needs every lane of the wave)
        if constexpr (EPI == EPI_SWIGLU) {
            uint2 t2[2];
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {                                  // tiles (2jp, 2jp+1) = (gate, up) of 16 columns
                const float gs[4] = {scl[2 * jp].x, scl[2 * jp].y, scl[2 * jp].z, scl[2 * jp].w};
                const float us[4] = {scl[2 * jp + 1].x, scl[2 * jp + 1].y, scl[2 * jp + 1].z, scl[2 * jp + 1].w};
                const float gb[4] = {lo16(bia[2 * jp].x), hi16(bia[2 * jp].x), lo16(bia[2 * jp].y), hi16(bia[2 * jp].y)};
                const float ub[4] = {lo16(bia[2 * jp + 1].x), hi16(bia[2 * jp + 1].x), lo16(bia[2 * jp + 1].y), hi16(bia[2 * jp + 1].y)};
                float o[4], uu[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float g = acc[i][2 * jp][r], u = acc[i][2 * jp + 1][r];
                    if constexpr (HS) { g *= gs[r]; u *= us[r]; }
                    if constexpr (HB) { g += gb[r]; u += ub[r]; }
                    rbf2(g, u);                                               // (two roundings per v_cvt_pk_bf16_f32)
                    o[r] = silu_f(g);
                    uu[r] = u;
                }
                rbf2(o[0], o[1]);
                rbf2(o[2], o[3]);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] *= uu[r];
                t2[jp] = uint2{pack2(o[0], o[1]), pack2(o[2], o[3])};
            }
            const uint4 v = widen(t2[0], t2[1]);
            const int no = (n0 + wn * 64) / 2 + odd * 16 + half8;
            if (rok[i]) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)orow[i] * p.ldo + no) = v;
        } else if constexpr (EPI == EPI_F32) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + j * 16 + fg * 4;
                float o[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if constexpr (HS) { o[0] *= scl[j].x; o[1] *= scl[j].y; o[2] *= scl[j].z; o[3] *= scl[j].w; }
                if constexpr (HB) { o[0] += bf2f(p.bias[n]); o[1] += bf2f(p.bias[n + 1]); o[2] += bf2f(p.bias[n + 2]); o[3] += bf2f(p.bias[n + 3]); }
                if (rok[i]) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (size_t)orow[i] * p.ldo + n) = float4{o[0], o[1], o[2], o[3]};
            }
        } else {
            uint2 rv[4], t4[4];
            if constexpr (EPI == EPI_RESID) {
                // the residual comes in with the same 16-byte accesses: load the 8 columns this lane will STORE, then the same lane
                // exchange hands every lane the 4 + 4 columns it accumulates (the swap is its own inverse on this layout)
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    const uint4 l4 = rl[i][jp];
                    const auto sx = __builtin_amdgcn_permlane16_swap(l4.x, l4.z, false, false);
                    const auto sy = __builtin_amdgcn_permlane16_swap(l4.y, l4.w, false, false);
                    rv[2 * jp] = uint2{sx[0], sy[0]};
                    rv[2 * jp + 1] = uint2{sx[1], sy[1]};
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float o[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if constexpr (HS) { o[0] *= scl[j].x; o[1] *= scl[j].y; o[2] *= scl[j].z; o[3] *= scl[j].w; }
                if constexpr (HB) { o[0] += lo16(bia[j].x); o[1] += hi16(bia[j].x); o[2] += lo16(bia[j].y); o[3] += hi16(bia[j].y); }
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
            for (int jp = 0; jp < 2; ++jp) {
                const uint4 v = widen(t4[2 * jp], t4[2 * jp + 1]);
                const int n = n0 + wn * 64 + (2 * jp + odd) * 16 + half8;
                if (rok[i]) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)orow[i] * p.ldo + n) = v;
            }
        }
    }
    };
    if (p.w_scale) { if (p.bias) rows(std::true_type{}, std::true_type{}); else rows(std::true_type{}, std::false_type{}); }
    else { if (p.bias) rows(std::false_type{}, std::true_type{}); else rows(std::false_type{}, std::false_type{}); }
#ifdef SR_G256_TIMING
    T256(3, wall_clock64());                                   // stores issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    T256(5, wall_clock64());                                   // stores acknowledged
#endif
}

template <int EPI, bool MX = false>
int launch256_t(hipStream_t s, const GemmArgs& a) {
    const int ntm = cdiv(a.M, BM2), ntn = a.N / BN2;
    constexpr int smem = 2 * BUF_BYTES + (MX ? 2048 : 0);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm256<EPI, MX>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (r != hipSuccess) return (int)r;
        attr_done = true;
    }
    const int group = sr_switches().g256_group;                // tuning hook (default 4)
    hipLaunchKernelGGL((k_gemm256<EPI, MX>), dim3(ntm * ntn), dim3(512), smem, s, a, ntm, ntn, group);
    SR_CHECK_LAUNCH();
    return 0;
}

}  // namespace

#ifdef SR_G256_TIMING
extern "C" __attribute__((visibility("default"))) int sr_dbg_g256_times(long long* host_out, int n_blocks) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_t256), (size_t)n_blocks * 6 * sizeof(long long), 0, hipMemcpyDeviceToHost);
}
extern "C" __attribute__((visibility("default"))) int sr_dbg_g256_phases(long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_p256), sizeof(long long) * (1024 * 2 * 6 + 2048), 0, hipMemcpyDeviceToHost);
}
#endif

// fused q/k/v epilogue: the q, k and v sections must start on 256-column tile boundaries (whole heads per tile, block-uniform role)
bool lmqkv_ok(const GemmArgs& a) {
    const QkvRope& f = a.rope;
    return f.pos3 && f.tok_slot && f.tok_idx && f.rope_cos && f.rope_sin && f.kcache && f.vtcache && (f.n_q_heads * 128) % BN2 == 0 &&
           (f.n_kv_heads * 128) % BN2 == 0 && a.N == (f.n_q_heads + 2 * f.n_kv_heads) * 128 && f.sec0 % 4 == 0 && f.sec1 % 4 == 0 && !a.rowmap;
}

// fused ViT qkv epilogue: sections on tile boundaries, heads made of whole 16-column tiles, paired q / k channel order (the caller's
// promise), V^T rows padded to the 8-row store granularity
bool vitqkv_ok(const GemmArgs& a) {
    const VitRope& f = a.vrope;
    return f.cos_t && f.sin_t && f.vt && f.C > 0 && f.hd > 0 && f.C % BN2 == 0 && a.N == 3 * f.C && f.hd % 16 == 0 && f.C % f.hd == 0 &&
           f.vt_stride % 8 == 0 && f.vt_stride >= (a.M + 7) / 8 * 8 && !a.rowmap && !a.w_scale && a.M % 4 == 0;
}

// shapes the 256-tile kernel takes: whole 256-column tiles, at least two 64-wide k-tiles (M is arbitrary: edge rows are clamped
// on load and masked on store)
bool gemm256_supports(const GemmArgs& a) { return a.N % BN2 == 0 && a.K % BK2 == 0 && a.K / BK2 >= 2 && a.M >= 1; }

// fp8 x fp8 on the block-scaled MFMA: A = fp8 bytes [M][lda], a_scale = [K/128][a_rows_pad][4] e8m0, W = tiled8 fp8 image, w_scale per channel
int launch_gemm256_mx(hipStream_t s, const GemmArgs& a, int epi) {
    if (a.N % BN2 != 0 || a.K % 128 != 0 || a.K / 128 < 2 || a.M < 1 || !a.a_scale || !a.w_tiled || a.a_rows_pad < cdiv(a.M, 256) * 256) return -22;
    switch (epi) {
        case EPI_STORE: return launch256_t<EPI_STORE, true>(s, a);
        case EPI_RESID: return launch256_t<EPI_RESID, true>(s, a);
        case EPI_SWIGLU: return launch256_t<EPI_SWIGLU, true>(s, a);
        case EPI_F32: return launch256_t<EPI_F32, true>(s, a);
        case EPI_LMQKV: return lmqkv_ok(a) ? launch256_t<EPI_LMQKV, true>(s, a) : -22;
    }
    return -22;
}

int launch_gemm256(hipStream_t s, const GemmArgs& a, int epi) {
    if (!gemm256_supports(a)) return -22;
    switch (epi) {
        case EPI_STORE: return launch256_t<EPI_STORE>(s, a);
        case EPI_RESID: return launch256_t<EPI_RESID>(s, a);
        case EPI_SWIGLU: return launch256_t<EPI_SWIGLU>(s, a);
        case EPI_GELU: return launch256_t<EPI_GELU>(s, a);
        case EPI_F32: return launch256_t<EPI_F32>(s, a);
        case EPI_LMQKV: return lmqkv_ok(a) ? launch256_t<EPI_LMQKV>(s, a) : -22;
        case EPI_VITQKV: return vitqkv_ok(a) ? launch256_t<EPI_VITQKV>(s, a) : -22;
    }
    return -22;
}
