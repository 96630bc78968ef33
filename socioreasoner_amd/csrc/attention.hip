// Attention kernels of the hot path, MFMA 16x16x32 bf16, exact HF eager semantics:
//   scores = bf16(q.k^T); scores = bf16(scores * scale); softmax in float32; P = bf16(softmax); O = bf16(P.V)
//   (hf: transformers/models/qwen2_5_vl/modeling_qwen2_5_vl.py:186-208 eager_attention_forward, 262-287, 641-689).
// Because HF rounds the *normalised* probabilities to bf16 before P.V, the kernels are two-pass: pass 1 computes the
// row max / sum, pass 2 recomputes the (bit-identical) scores, normalises, rounds and multiplies by V.  The second
// Q.K^T is ~1-2 % of a layer's FLOPs; keeping the reference's rounding points is worth more than that here.
//
// Everything is computed transposed so that no operand ever needs a transpose in LDS:
//   S^T[key][query] = K[key][:] . Q[query][:]      (A = K rows,  B = Q rows, both d-contiguous)
//   O^T[d][query]   = V^T[d][:] . P^T[:][query]    (A = V^T rows, key-contiguous; B = P from the S^T registers)
// V is therefore kept transposed in HBM ([d][key]: the ViT rope kernel and the LM KV-cache writer produce it).
// Measured and dropped (round 2): register-prefetched tiles (loads of tile t + 1 in flight while tile t is multiplied, issue early /
// write late) plus a single-tile path that reuses pass 1's scores: bit-identical, but the kernel grows from 44-56 to 116-160 VGPRs
// and loses to the plain version that lets 4+ blocks per CU cover each other's staging -- LM causal 215 vs 155 us, ViT full 1133 vs
// 906 us, ViT windows 112 vs 110 us (those read q, k, v exactly once: they sit on the HBM floor of the qkv buffer).
// Where the time goes (PMC, profiles/r02_pmc_attn_prefill.json, LM causal at 32 x 448 tokens, 159 us): waves wait 62 % of their cycles
// (SQ_WAIT_ANY), MFMA pipe 13 % busy, ~2950 VALU instructions per wave.  A 64-key tile of K and V^T is re-staged by every (64-query
// tile, head) block -- 688 MB per layer for 15 MB of unique K / V, 4.1 TB/s out of L2 -- and the staging is synchronous, so the kernel
// runs at (tiles in flight per CU) / (load latency): everything that lowers the block count per CU loses what it saves.  Measured
// against this version, same box, ms per 32-tile LM prefill (all bit-identical or within rounding noise, all dropped): pass-1 scores
// kept in registers for prompts <= 512 keys (no second Q.K^T, 112 VGPRs) 75.2 vs 73.5; 128-query blocks of 8 waves (half the staging)
// 75.8 vs 74.3; 32 queries per wave (each K / V^T fragment feeds two MFMAs, causal tiles skipped per wave, 168 + 72 registers) 79.1 vs
// 77.1; masks only on edge tiles + exp2/fma softmax + hoisted staging indices (fewer VALU instructions, 104 VGPRs) 72.6 vs 71.9;
// MFMA accumulators forced into VGPRs (-amdgpu-mfma-vgpr-form, no v_accvgpr moves) no change.  What would help is staging traffic cut
// at unchanged occupancy (one block for the 8 query heads of a kv head, <= 64 registers per wave) or asynchronous multi-stage staging.
#include "kernels.h"
#include <math.h>
#include <stdlib.h>

namespace {

// the four scores of an accumulator: s = bf16(bf16(acc) * scale), two roundings per v_cvt_pk_bf16_f32 (rbf2, common.h) instead of one each --
// the prefill attention kernels are bound by instruction issue; same values as rbf(rbf(x) * scale)
__device__ __forceinline__ void scaled_scores4(const f32x4& acc, float scale, float (&v)[4]) {
    float a0 = acc[0], a1 = acc[1], a2 = acc[2], a3 = acc[3];
    rbf2(a0, a1);
    rbf2(a2, a3);
    a0 *= scale; a1 *= scale; a2 *= scale; a3 *= scale;
    rbf2(a0, a1);
    rbf2(a2, a3);
    v[0] = a0; v[1] = a1; v[2] = a2; v[3] = a3;
}


constexpr int QT = 64;   // queries per block (4 waves x 16)
constexpr int KT = 64;   // keys per staged tile
constexpr int VT_RS = 136;  // V^T LDS row stride in bytes: 64 keys * 2 B + 8 B pad (conflict-free ds_read_b64)

template <int HD>
struct PrefillCfg {
    static constexpr int HDP = (HD + 31) / 32 * 32;   // K-dim of Q.K^T padded to the MFMA k-step
    static constexpr int KS = HDP / 32;
    static constexpr int DT = HD / 16;
    static constexpr int K_RS = HDP * 2 + 16;         // K LDS row stride (bytes), +16 B pad: conflict-free b128 reads
    static constexpr int K_BYTES = KT * K_RS;
    static constexpr int V_BYTES = HD * VT_RS;
};

template <int HD, bool CAUSAL>
__global__ __launch_bounds__(256) void k_attn_prefill(AttnArgs p) {
    using C = PrefillCfg<HD>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[C::K_BYTES + C::V_BYTES];
    unsigned char* ks = smem;
    unsigned char* vs = smem + C::K_BYTES;

    const AttnWork wk = p.work[blockIdx.x];
    const int h = blockIdx.y, kvh = h / p.group;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;

    const int q_local = wave * 16 + fr;                     // query index inside the tile
    const int qpos = wk.q_off + q_local;                    // position inside the sequence
    const int q_len = wk.q_len > 0 ? wk.q_len : wk.seq_len;      // (pooled queries: fewer queries than keys)
    const bool q_valid = qpos < q_len;
    const int q_rows_valid = min(QT, q_len - wk.q_off);
    const bf16_t* qptr = p.q + (size_t)(wk.q_row0 + min(q_local, q_rows_valid - 1)) * p.q_stride + h * HD;

    bf16x8 qf[C::KS];
#pragma unroll
    for (int kk = 0; kk < C::KS; ++kk) {
        const int d = kk * 32 + fg * 8;
        if (d < HD) qf[kk] = *reinterpret_cast<const bf16x8*>(qptr + d);
        else qf[kk] = __builtin_bit_cast(bf16x8, uint4{0, 0, 0, 0});
    }

    const bf16_t* kbase = p.k + (size_t)wk.k_row0 * p.k_stride + (size_t)kvh * p.k_head_stride;
    const bf16_t* vbase = p.vt + wk.vt_off + (size_t)kvh * p.vt_head_stride;
    const int kv_end = CAUSAL ? min(wk.seq_len, wk.q_off + QT) : wk.seq_len;

    auto stage_k = [&](int kv0) {
        constexpr int CH = C::HDP / 8;                       // 16-byte chunks per row (incl. zero padding)
        for (int c = tid; c < KT * CH; c += 256) {
            const int row = c / CH, ch = c % CH;
            uint4 v = uint4{0, 0, 0, 0};
            if (ch * 8 < HD) {
                const int j = min(kv0 + row, wk.seq_len - 1);
                v = *reinterpret_cast<const uint4*>(kbase + (size_t)j * p.k_stride + ch * 8);
            }
            *reinterpret_cast<uint4*>(ks + row * C::K_RS + ch * 16) = v;
        }
    };
    auto stage_v = [&](int kv0) {
        for (int c = tid; c < HD * (KT / 4); c += 256) {
            const int d = c / (KT / 4), g4 = c % (KT / 4);
            const int j0 = kv0 + g4 * 4;
            uint2 v = uint2{0, 0};
            if (j0 < wk.seq_len) {
                v = *reinterpret_cast<const uint2*>(vbase + (size_t)d * p.vt_stride + j0);
                // zero the keys past the end of the sequence (they hold other data or garbage)
                if (j0 + 1 >= wk.seq_len) v.x &= 0x0000ffffu;
                if (j0 + 2 >= wk.seq_len) v.y = 0;
                else if (j0 + 3 >= wk.seq_len) v.y &= 0x0000ffffu;
            }
            *reinterpret_cast<uint2*>(vs + d * VT_RS + g4 * 8) = v;
        }
    };
    // scores of one staged tile: s[t][r] = key kv0 + t*16 + fg*4 + r against query fr of this wave
    auto scores = [&](int kv0, float (&s)[4][4]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < C::KS; ++kk) {
                bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + (t * 16 + fr) * C::K_RS + (kk * 4 + fg) * 16);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kk], acc, 0, 0, 0);
            }
            float sv[4];
            scaled_scores4(acc, p.scale, sv);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kidx = kv0 + t * 16 + fg * 4 + r;
                const bool ok = kidx < wk.seq_len && (!CAUSAL || kidx <= qpos);
                s[t][r] = ok ? sv[r] : -INFINITY;
            }
        }
    };

    // ---------------- pass 1: row max m and sum l = sum exp(s - m)
    float m = -INFINITY, l = 0.f;
    for (int kv0 = 0; kv0 < kv_end; kv0 += KT) {
        __syncthreads();
        stage_k(kv0);
        __syncthreads();
        float s[4][4];
        scores(kv0, s);
        float tm = -INFINITY;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) tm = fmaxf(tm, s[t][r]);
        tm = fmaxf(tm, __shfl_xor(tm, 16, 64));
        tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
        const float mn = fmaxf(m, tm);
        float ts = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) ts += __expf(s[t][r] - mn);
        ts += __shfl_xor(ts, 16, 64);
        ts += __shfl_xor(ts, 32, 64);
        l = l * __expf(m - mn) + ts;
        m = mn;
    }
    const float inv_l = 1.0f / l;

    // ---------------- pass 2: P = bf16(exp(s - m) / l);  O^T += V^T . P^T
    f32x4 oacc[C::DT];
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kv0 = 0; kv0 < kv_end; kv0 += KT) {
        __syncthreads();
        stage_k(kv0);
        stage_v(kv0);
        __syncthreads();
        float s[4][4];
        scores(kv0, s);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            uint4 pv;
            pv.x = pack2(__expf(s[2 * kb][0] - m) * inv_l, __expf(s[2 * kb][1] - m) * inv_l);
            pv.y = pack2(__expf(s[2 * kb][2] - m) * inv_l, __expf(s[2 * kb][3] - m) * inv_l);
            pv.z = pack2(__expf(s[2 * kb + 1][0] - m) * inv_l, __expf(s[2 * kb + 1][1] - m) * inv_l);
            pv.w = pack2(__expf(s[2 * kb + 1][2] - m) * inv_l, __expf(s[2 * kb + 1][3] - m) * inv_l);
            const bf16x8 pf = __builtin_bit_cast(bf16x8, pv);
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) {
                const unsigned char* vr = vs + (dt * 16 + fr) * VT_RS + kb * 64 + fg * 8;
                uint2 v0 = *reinterpret_cast<const uint2*>(vr);        // keys kb*32 + fg*4 .. +3
                uint2 v1 = *reinterpret_cast<const uint2*>(vr + 32);   // keys kb*32 + 16 + fg*4 .. +3
                const bf16x8 vf = __builtin_bit_cast(bf16x8, uint4{v0.x, v0.y, v1.x, v1.y});
                oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, oacc[dt], 0, 0, 0);
            }
        }
    }
    if (q_valid) {
        bf16_t* optr = p.out + (size_t)(wk.q_row0 + q_local) * p.out_stride + h * HD + fg * 4;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            uint2 v = {pack2(oacc[dt][0], oacc[dt][1]), pack2(oacc[dt][2], oacc[dt][3])};
            *reinterpret_cast<uint2*>(optr + dt * 16) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------- 64-token windows, no LDS
// The ViT's window blocks (28 of its 32 layers): every window is exactly 64 tokens, so the K tile (64 x 80) and the V^T tile (80 x 64) of a
// (window, head) are 40 + 48 registers of MFMA fragments, loaded straight from global memory in the layouts the MFMAs want (the same row /
// key permutations k_attn_prefill reads from LDS).  ONE WAVE does a whole (window, head): no LDS, no barrier, every load issued before the
// first MFMA; k_attn_prefill staged the tiles synchronously through VGPRs into LDS twice (two passes) with two block barriers each.  The
// arithmetic per output is the same sequence (scores, max / sum over the lane's 16 keys then across the four lane groups, P = bf16(exp(s - m)
// / l), P.V over the two 32-key blocks in order): BIT-IDENTICAL to k_attn_prefill<80, false> (tests/test_gpu_round3.py).
template <int HD>
__global__ __launch_bounds__(256) void k_attn_win64(AttnArgs p) {
    using C = PrefillCfg<HD>;
    const AttnWork wk = p.work[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = blockIdx.y * 4 + wave, kvh = h / p.group;
    if (h >= p.n_heads) return;
    const int fr = lane & 15, fg = lane >> 4;
    const bf16_t* kbase = p.k + (size_t)wk.k_row0 * p.k_stride + (size_t)kvh * p.k_head_stride;
    const bf16_t* vbase = p.vt + wk.vt_off + (size_t)kvh * p.vt_head_stride;
    const bf16x8 zero8 = __builtin_bit_cast(bf16x8, uint4{0, 0, 0, 0});
    bf16x8 kf[4][C::KS];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int kk = 0; kk < C::KS; ++kk) {
            const int d = kk * 32 + fg * 8;
            kf[t][kk] = d < HD ? *reinterpret_cast<const bf16x8*>(kbase + (size_t)(t * 16 + fr) * p.k_stride + d) : zero8;
        }
    bf16x8 vf[C::DT][2];
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const bf16_t* vr = vbase + (size_t)(dt * 16 + fr) * p.vt_stride + kb * 32 + fg * 4;
            const uint2 v0 = *reinterpret_cast<const uint2*>(vr), v1 = *reinterpret_cast<const uint2*>(vr + 16);   // keys kb*32 + fg*4 .. +3 and + 16
            vf[dt][kb] = __builtin_bit_cast(bf16x8, uint4{v0.x, v0.y, v1.x, v1.y});
        }
    bf16x8 qf[4][C::KS];
#pragma unroll
    for (int qs = 0; qs < 4; ++qs)
#pragma unroll
        for (int kk = 0; kk < C::KS; ++kk) {
            const int d = kk * 32 + fg * 8;
            qf[qs][kk] = d < HD ? *reinterpret_cast<const bf16x8*>(p.q + (size_t)(wk.q_row0 + qs * 16 + fr) * p.q_stride + h * HD + d) : zero8;
        }
#pragma unroll
    for (int qs = 0; qs < 4; ++qs) {
        float s[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < C::KS; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t][kk], qf[qs][kk], acc, 0, 0, 0);
            scaled_scores4(acc, p.scale, s[t]);
        }
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, s[t][r]);
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) l += __expf(s[t][r] - m);
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv_l = 1.0f / l;
        f32x4 oacc[C::DT];
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            uint4 pv;
            pv.x = pack2(__expf(s[2 * kb][0] - m) * inv_l, __expf(s[2 * kb][1] - m) * inv_l);
            pv.y = pack2(__expf(s[2 * kb][2] - m) * inv_l, __expf(s[2 * kb][3] - m) * inv_l);
            pv.z = pack2(__expf(s[2 * kb + 1][0] - m) * inv_l, __expf(s[2 * kb + 1][1] - m) * inv_l);
            pv.w = pack2(__expf(s[2 * kb + 1][2] - m) * inv_l, __expf(s[2 * kb + 1][3] - m) * inv_l);
            const bf16x8 pf = __builtin_bit_cast(bf16x8, pv);
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[dt][kb], pf, oacc[dt], 0, 0, 0);
        }
        bf16_t* optr = p.out + (size_t)(wk.q_row0 + qs * 16 + fr) * p.out_stride + h * HD + fg * 4;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) *reinterpret_cast<uint2*>(optr + dt * 16) = uint2{pack2(oacc[dt][0], oacc[dt][1]), pack2(oacc[dt][2], oacc[dt][3])};
    }
}

// ---------------------------------------------------------------------------------------------- few queries, many keys
// SAM2's token-to-image attentions: <= 16 queries (one object's prompt tokens) against 4096 image keys per head.  k_attn_prefill gives such a
// work item one wave with live queries and walks the 2 x 64 key tiles one after the other (130 us of pure latency per launch, a third of the
// mask decoder).  Here ALL waves of the block serve the same <= 16 queries and split the key tiles (wave w: tiles w, w + NW, ...), each staging
// its own tiles in its own LDS region (no block barrier in the loops).  Same two passes and the same P = bf16(exp(s - m) / l) with the GLOBAL
// row max / sum, which the waves combine through LDS after pass 1 (flash-decoding's rescale, fixed order); the P.V partial sums of the waves
// are added in wave order.  Deterministic; differs from the one-wave walk only in the float32 association of l and of the P.V sum.
template <int HD, int NW>
__global__ __launch_bounds__(NW * 64) void k_attn_fewq(AttnArgs p) {
    using C = PrefillCfg<HD>;
    constexpr int WB = C::K_BYTES + C::V_BYTES;                 // per-wave staging
    static_assert(C::DT * 64 * 16 <= WB, "a wave's P.V partial sums reuse its staging region");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // NW * WB + NW * 16 * 8 bytes
    const AttnWork wk = p.work[blockIdx.x];
    const int h = blockIdx.y, kvh = h / p.group;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    unsigned char* ks = smem + wave * WB;
    unsigned char* vs = ks + C::K_BYTES;
    float* ml = reinterpret_cast<float*>(smem + NW * WB);       // [NW][16][2]: (m, l) of every wave and query
    const int q_len = wk.q_len > 0 ? wk.q_len : wk.seq_len;
    const int nq = min(16, q_len - wk.q_off);                   // live queries (<= 16 by contract: AttnArgs.q_tile == 16)
    const bool q_valid = fr < nq;
    const bf16_t* qptr = p.q + (size_t)(wk.q_row0 + min(fr, nq - 1)) * p.q_stride + h * HD;
    bf16x8 qf[C::KS];
#pragma unroll
    for (int kk = 0; kk < C::KS; ++kk) {
        const int d = kk * 32 + fg * 8;
        if (d < HD) qf[kk] = *reinterpret_cast<const bf16x8*>(qptr + d);
        else qf[kk] = __builtin_bit_cast(bf16x8, uint4{0, 0, 0, 0});
    }
    const bf16_t* kbase = p.k + (size_t)wk.k_row0 * p.k_stride + (size_t)kvh * p.k_head_stride;
    const bf16_t* vbase = p.vt + wk.vt_off + (size_t)kvh * p.vt_head_stride;
    const int ntile = (wk.seq_len + KT - 1) / KT;
    auto stage_k = [&](int kv0) {
        constexpr int CH = C::HDP / 8;
        for (int c = lane; c < KT * CH; c += 64) {
            const int row = c / CH, ch = c % CH;
            uint4 v = uint4{0, 0, 0, 0};
            if (ch * 8 < HD) v = *reinterpret_cast<const uint4*>(kbase + (size_t)min(kv0 + row, wk.seq_len - 1) * p.k_stride + ch * 8);
            *reinterpret_cast<uint4*>(ks + row * C::K_RS + ch * 16) = v;
        }
    };
    auto stage_v = [&](int kv0) {
        for (int c = lane; c < HD * (KT / 4); c += 64) {
            const int d = c / (KT / 4), g4 = c % (KT / 4);
            const int j0 = kv0 + g4 * 4;
            uint2 v = uint2{0, 0};
            if (j0 < wk.seq_len) {
                v = *reinterpret_cast<const uint2*>(vbase + (size_t)d * p.vt_stride + j0);
                if (j0 + 1 >= wk.seq_len) v.x &= 0x0000ffffu;
                if (j0 + 2 >= wk.seq_len) v.y = 0;
                else if (j0 + 3 >= wk.seq_len) v.y &= 0x0000ffffu;
            }
            *reinterpret_cast<uint2*>(vs + d * VT_RS + g4 * 8) = v;
        }
    };
    auto scores = [&](int kv0, float (&s)[4][4]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < C::KS; ++kk) {
                bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + (t * 16 + fr) * C::K_RS + (kk * 4 + fg) * 16);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kk], acc, 0, 0, 0);
            }
            float sv[4];
            scaled_scores4(acc, p.scale, sv);
#pragma unroll
            for (int r = 0; r < 4; ++r) s[t][r] = (kv0 + t * 16 + fg * 4 + r < wk.seq_len) ? sv[r] : -INFINITY;
        }
    };
    // ---- pass 1 over this wave's tiles
    float m = -INFINITY, l = 0.f;
    for (int ti = wave; ti < ntile; ti += NW) {
        stage_k(ti * KT);
        float s[4][4];
        scores(ti * KT, s);
        float tm = -INFINITY;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) tm = fmaxf(tm, s[t][r]);
        tm = fmaxf(tm, __shfl_xor(tm, 16, 64));
        tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
        const float mn = fmaxf(m, tm);
        float ts = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) ts += __expf(s[t][r] - mn);
        ts += __shfl_xor(ts, 16, 64);
        ts += __shfl_xor(ts, 32, 64);
        l = l * __expf(m - mn) + ts;
        m = mn;
    }
    if (fg == 0) { ml[(wave * 16 + fr) * 2] = m; ml[(wave * 16 + fr) * 2 + 1] = l; }
    __syncthreads();
    float gm = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) gm = fmaxf(gm, ml[(w * 16 + fr) * 2]);
    float gl = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const float mw = ml[(w * 16 + fr) * 2];
        if (mw > -INFINITY) gl += ml[(w * 16 + fr) * 2 + 1] * __expf(mw - gm);        // (a wave without tiles contributes nothing)
    }
    const float inv_l = 1.0f / gl;
    // ---- pass 2 over the same tiles
    f32x4 oacc[C::DT];
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ti = wave; ti < ntile; ti += NW) {
        stage_k(ti * KT);
        stage_v(ti * KT);
        float s[4][4];
        scores(ti * KT, s);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            uint4 pv;
            pv.x = pack2(__expf(s[2 * kb][0] - gm) * inv_l, __expf(s[2 * kb][1] - gm) * inv_l);
            pv.y = pack2(__expf(s[2 * kb][2] - gm) * inv_l, __expf(s[2 * kb][3] - gm) * inv_l);
            pv.z = pack2(__expf(s[2 * kb + 1][0] - gm) * inv_l, __expf(s[2 * kb + 1][1] - gm) * inv_l);
            pv.w = pack2(__expf(s[2 * kb + 1][2] - gm) * inv_l, __expf(s[2 * kb + 1][3] - gm) * inv_l);
            const bf16x8 pf = __builtin_bit_cast(bf16x8, pv);
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) {
                const unsigned char* vr = vs + (dt * 16 + fr) * VT_RS + kb * 64 + fg * 8;
                const uint2 v0 = *reinterpret_cast<const uint2*>(vr), v1 = *reinterpret_cast<const uint2*>(vr + 32);
                oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, uint4{v0.x, v0.y, v1.x, v1.y}), pf, oacc[dt], 0, 0, 0);
            }
        }
    }
    // the wave's partial sums go into its own (now dead) staging region
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) reinterpret_cast<f32x4*>(ks)[dt * 64 + lane] = oacc[dt];
    __syncthreads();
    if (wave == 0 && q_valid) {
        bf16_t* optr = p.out + (size_t)(wk.q_row0 + fr) * p.out_stride + h * HD + fg * 4;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            f32x4 o = reinterpret_cast<const f32x4*>(smem)[dt * 64 + lane];
#pragma unroll
            for (int w = 1; w < NW; ++w) {
                const f32x4 o2 = reinterpret_cast<const f32x4*>(smem + w * WB)[dt * 64 + lane];
                o[0] += o2[0]; o[1] += o2[1]; o[2] += o2[2]; o[3] += o2[3];
            }
            *reinterpret_cast<uint2*>(optr + dt * 16) = uint2{pack2(o[0], o[1]), pack2(o[2], o[3])};
        }
    }
}

// ---------------------------------------------------------------------------------------------- prefill, round 3
// k_attn_prefill2: the SAME per-wave arithmetic as k_attn_prefill (a wave = 16 queries of one head against 64-key tiles, two passes, same
// MFMA order, same softmax sums: results are bit-identical, tests/test_gpu_round3.py), but
//   * a block is 8 waves that share every staged K / V^T tile: QW query sub-tiles x 8 / QW heads of ONE kv head -- 32 queries x 4 of
//     the 8 query heads of a kv head for the LM (causal, GQA), 128 queries of one head for the ViT's full-attention blocks (MHA) -- so a
//     staged tile serves 128 (query, head) rows instead of 64 (PMC round 2: 688 MB of staging per LM layer for 15 MB of unique K / V,
//     waves parked 62 % of their cycles, MFMA pipe 13 % busy), two blocks per CU (<= 128 registers, <= 80 KB of LDS);
//   * tiles arrive by LDS-DMA (global_load_lds, 16 B per lane, no VGPR round trip) into an NS-deep ring; pass 1 and pass 2 are ONE
//     sequence of 2 x n_tiles steps, so the pipeline never drains between the passes; per step one counted `s_waitcnt vmcnt` + one raw
//     `s_barrier` (the structure of gemm.hip's small-M ring): the loads of step u + NS - 1 are issued right after the barrier that
//     proves step u - 1's buffer free, and stay in flight across barriers while step u is multiplied.  The DMA is issued from inline
//     asm: hipcc (ROCm 7.2) otherwise puts an `s_waitcnt vmcnt(0)` in front of the first V^T fragment read of every step (its LDS-DMA
//     alias rule cannot tell the ring's buffers apart), which drains the ring it is meant to keep full.  The waits are therefore
//     counted by hand: 2 DMA instructions per wave and pass-1 step (K), 4 per pass-2 step (K + V^T), nothing else in the loop
//     touches vmcnt (no spills: checked with -Rpass-analysis=kernel-resource-usage);
//   * LDS rows keep a power-of-two pitch and the 16-byte chunk index is XOR-swizzled on the SOURCE side (LDS-DMA writes linearly):
//     K rows 256 B, chunk ^ (row & 15) -> conflict-free ds_read_b128 fragments; V^T rows 128 B, chunk ^ vt_swz(row) -> conflict-free
//     ds_read_b64 pairs (see vt_swz).  head_dim 80 uses the same 256-B K pitch; its pad chunks (d 80..95 of the 96-wide MFMA k range) are zeroed
//     once and never written again (masked-off DMA lanes);
//   * blocks are dealt to XCDs in contiguous runs of (work item, head group, sub-tile), so the K / V of a sequence stay in ONE XCD's L2.
// Keys past the end of a sequence: K rows are clamped (their scores are masked); V^T columns are read as they are -- P is exactly 0
// there, and the caller guarantees FINITE contents (AttnArgs.v2_ok: the engine zeroes the KV cache and the ViT's V^T buffer when it is
// created, and only ever stores finite bf16 values into them).
typedef __attribute__((address_space(3))) void* lptr_t;

// V^T tile swizzle: 16-byte chunk (8 keys) c of row d is stored at chunk c ^ vt_swz(d): with 128-byte rows the 16 rows x 2 halves a
// 32-lane group reads as ds_read_b64 land in 32 distinct 8-byte bank slots only if all three chunk bits are swizzled.
// Measured alternative (round 3): swizzling bits 0 and 2 only keeps a lane's two 4-key runs at a constant 32 bytes, so the pair loads as
// ONE ds_read2_b64 into four consecutive registers and the 48 v_mov per step that assemble the MFMA operand disappear (VALU
// instructions 40 M -> 30 M per launch) -- but the reads are then 2-way conflicted, SQ_LDS_BANK_CONFLICT triples (7.3 M -> 22 M),
// SQ_WAIT_INST_LDS goes 6.9 M -> 31.7 M and the kernel 118 -> 140 us (LM), 610 -> 705 us (ViT full).  Conflict-free wins.
__device__ __forceinline__ int vt_swz(int d) { return (d >> 1) & 7; }

// one LDS-DMA instruction: 64 lanes x 16 B from gbase + voff (bytes, per lane) to LDS [lds_dst, lds_dst + 1 KB), invisible to hipcc's
// waitcnt bookkeeping (MI355X guide 5.7: M0 is written in the statement that uses it and restored)
__device__ __forceinline__ void dma16(const void* gbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void wait_vmcnt(int n) {      // immediate operand: one case per count
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    }
}

// V^T fragment reads of pass 2, issued by hand (round 5).  Written as two uint2 loads per d-tile, hipcc pairs the loads of NEIGHBOURING d-tiles (rows
// 16 apart = 2048 B, a multiple of the 512-B stride unit) into `ds_read2st64_b64`.  That instruction is serviced as two accesses of four CONTIGUOUS 16-lane
// groups with 32-bank addressing and half the bytes per clock of `ds_read_b64` (MI355X guide, LDS table) -- under that rule lanes fr and fr ^ 1 of a group
// (same chunk, rows 128 B apart) share a bank: 2-way conflicts on a layout that is conflict-free for `ds_read_b64` (two 32-lane groups, 64 banks).  This was
// the "unexplained" SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.32 (LM) / 0.255 (ViT) of profiles/r03_pmc_attn_prefill.json: per wave and 64-key tile
// 128 cycles of K reads + 128 of V^T reads + 128 of conflicts, against 128 + 64 + 0 for the instruction the layout was designed for.  The pairs also
// came back as (d-tile i, d-tile i + 1) tuples that had to be re-assembled into MFMA operands with v_mov.  Inline asm keeps `ds_read_b64`; the results are
// handed to the compiler through the operands of the counted `s_waitcnt lgkmcnt` that covers them (LDS operations return in order, so every count the
// compiler derives for its own K reads stays conservative).
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// group G = d-tiles 2G and 2G + 1: four reads, 8 registers
template <int DT, int G>
__device__ __forceinline__ void vt_issue(unsigned a0, unsigned a1, u32x2 (&v0)[DT], u32x2 (&v1)[DT]) {
    if constexpr (2 * G < DT) {
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v0[2 * G]) : "v"(a0), "n"(2 * G * 2048));
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v1[2 * G]) : "v"(a1), "n"(2 * G * 2048));
    }
    if constexpr (2 * G + 1 < DT) {
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v0[2 * G + 1]) : "v"(a0), "n"((2 * G + 1) * 2048));
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v1[2 * G + 1]) : "v"(a1), "n"((2 * G + 1) * 2048));
    }
}
// group G has landed once only the reads of group G + 1 (issued behind it) can still be outstanding; then its MFMAs
template <int DT, int G>
__device__ __forceinline__ void vt_consume(u32x2 (&v0)[DT], u32x2 (&v1)[DT], const bf16x8& pf, f32x4 (&oacc)[DT]) {
    if constexpr (2 * G < DT) {
        constexpr int LEFT = 2 * ((2 * G + 2 < DT) + (2 * G + 3 < DT));
        constexpr int I0 = 2 * G, I1 = 2 * G + 1 < DT ? 2 * G + 1 : 2 * G;
        // (a group of ONE d-tile -- head_dim 80 has five -- must not name its registers twice: the second copy of an in-out operand is made BEFORE the
        // wait, from a register the read has not reached yet)
        if constexpr (I1 != I0) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(v0[I0]), "+v"(v1[I0]), "+v"(v0[I1]), "+v"(v1[I1]) : "n"(LEFT));
        else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(v0[I0]), "+v"(v1[I0]) : "n"(LEFT));
        oacc[I0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, uint4{v0[I0].x, v0[I0].y, v1[I0].x, v1[I0].y}), pf, oacc[I0], 0, 0, 0);
        if constexpr (I1 != I0)
            oacc[I1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, uint4{v0[I1].x, v0[I1].y, v1[I1].x, v1[I1].y}), pf, oacc[I1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);      // the next group's reads are issued BEHIND these MFMAs (8 + 8 registers of fragments live, not 32)
    }
}

template <int HD, bool CAUSAL, int QW, int NS, bool VASM>
__global__ __launch_bounds__(512, 4) void k_attn_prefill2(AttnArgs p, int n_units, int Y, int subs) {
    using C = PrefillCfg<HD>;
    constexpr int NH = 8 / QW;                  // heads per block
    constexpr int QTB = QW * 16;                // queries per block
    constexpr int K_BYTES = KT * 256, V_BYTES = HD * 128, ST_BYTES = K_BYTES + V_BYTES;
    constexpr int VI = HD * 8 / 64;             // 1-KB LDS-DMA pieces per V^T tile (16 at hd 128, 10 at hd 80)
    static_assert(NS >= 2 && NS <= 4 && VI > 8 && VI <= 16, "ring depth / V^T pieces");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];

    // block -> (work item, head group, query sub-tile); XCD x (blocks x, x + 8, ...) takes a contiguous run of units
    int unit;
    {
        const int flat = blockIdx.x, q8 = n_units / 8, r8 = n_units % 8, xcd = flat % 8, idx = flat / 8;
        unit = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const AttnWork wk0 = p.work[unit / (Y * subs)];
    const int h0 = ((unit / subs) % Y) * NH, sub = unit % subs;
    const int q_off = wk0.q_off + sub * QTB;
    const int q_len = wk0.q_len > 0 ? wk0.q_len : wk0.seq_len;
    if (q_off >= q_len) return;                 // the work item's last sub-tiles may be empty
    const int q_row0 = wk0.q_row0 + sub * QTB, seq_len = wk0.seq_len;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int h = h0 + wave / QW, kvh = h0 / p.group;

    const int q_local = (wave % QW) * 16 + fr;
    const int qpos = q_off + q_local;
    const bool q_valid = qpos < q_len;
    const int q_rows_valid = min(QTB, q_len - q_off);
    const bf16_t* qptr = p.q + (size_t)(q_row0 + min(q_local, q_rows_valid - 1)) * p.q_stride + h * HD;
    bf16x8 qf[C::KS];
#pragma unroll
    for (int kk = 0; kk < C::KS; ++kk) {
        const int d = kk * 32 + fg * 8;
        if (d < HD) qf[kk] = *reinterpret_cast<const bf16x8*>(qptr + d);
        else qf[kk] = __builtin_bit_cast(bf16x8, uint4{0, 0, 0, 0});
    }

    // hipcc must see the Q fragments ARRIVE before the tile loop: it does not count the asm DMA below, so a wait for these loads placed
    // inside the loop (`vmcnt(0)` at their first use, repeated on every iteration) would drain the ring each step
#pragma unroll
    for (int kk = 0; kk < C::KS; ++kk) asm volatile("" : "+v"(qf[kk]));

    const bf16_t* kbase = p.k + (size_t)wk0.k_row0 * p.k_stride + (size_t)kvh * p.k_head_stride;
    const bf16_t* vbase = p.vt + wk0.vt_off + (size_t)kvh * p.vt_head_stride;
    const int kv_end = CAUSAL ? min(seq_len, q_off + QTB) : seq_len;
    const int nt = (kv_end + KT - 1) / KT, total = 2 * nt;

    // ---- LDS-DMA sources of this lane: wave-uniform base (SGPRs) + a 32-bit byte offset per lane.
    // K pieces wave and wave + 8: LDS 16-byte position piece * 64 + lane = (row, physical chunk); the lane fetches logical chunk kc.
    const int krow = wave * 4 + (lane >> 4), kc = ((lane & 15) ^ (krow & 15)) * 8;     // second piece: row + 32, same chunk (32 % 16 == 0)
    const bool k_on = kc < HD;
    // V^T pieces wave and wave + 8 (beyond VI: pieces 0.. once more -- same bytes to the same place, keeps every wave's DMA count equal)
    const int vp1 = wave + 8 < VI ? wave + 8 : wave + 8 - VI;
    const int vrow0 = wave * 8 + (lane >> 3), vrow1 = vp1 * 8 + (lane >> 3);
    const unsigned voff0 = (unsigned)(vrow0 * p.vt_stride + ((lane & 7) ^ vt_swz(vrow0)) * 8) * 2u;
    const unsigned voff1 = (unsigned)(vrow1 * p.vt_stride + ((lane & 7) ^ vt_swz(vrow1)) * 8) * 2u;
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem2;

    if constexpr (HD != 128) {          // pad chunks of the K rows: zero once (whole K regions), before any DMA lands
        for (int i = tid; i < NS * K_BYTES / 16; i += 512)
            *reinterpret_cast<uint4*>(smem2 + (i / (K_BYTES / 16)) * ST_BYTES + (i % (K_BYTES / 16)) * 16) = uint4{0, 0, 0, 0};
        __syncthreads();
    }
    auto issue = [&](int step, int buf) {
        const bool pass2 = step >= nt;
        const int kv0 = (pass2 ? step - nt : step) * KT;
        const unsigned st = lds0 + buf * ST_BYTES;
        const unsigned ko0 = (unsigned)(min(kv0 + krow, seq_len - 1) * p.k_stride + kc) * 2u;
        const unsigned ko1 = (unsigned)(min(kv0 + krow + 32, seq_len - 1) * p.k_stride + kc) * 2u;
        if (k_on) {
            dma16(kbase, ko0, st + wave * 1024);
            dma16(kbase, ko1, st + (wave + 8) * 1024);
        }
        if (pass2) {
            dma16(vbase, voff0 + (unsigned)kv0 * 2u, st + K_BYTES + wave * 1024);
            dma16(vbase, voff1 + (unsigned)kv0 * 2u, st + K_BYTES + vp1 * 1024);
        }
    };

    // scores of 16 keys of a staged tile: s[r] = key kv0 + t*16 + fg*4 + r against query fr of this wave (as k_attn_prefill).
    // `masked` is wave-uniform: false on tiles every query of the wave sees in full (all but the diagonal / last tile) -- the kernel is
    // instruction-issue bound (PMC round 3: ~3800 instructions per wave, 4 waves per SIMD, ACTIVE_INST_ANY x 4 = 113 % of the wave cycles),
    // and the two compares + select per score are 15 % of them
    auto scores16 = [&](const unsigned char* ks, int kv0, int t, float (&s)[4], bool masked) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < C::KS; ++kk) {
            bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + (t * 16 + fr) * 256 + (((kk * 4 + fg) ^ fr) << 4));
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kk], acc, 0, 0, 0);
        }
        if (masked) {
            float sv[4];
            scaled_scores4(acc, p.scale, sv);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kidx = kv0 + t * 16 + fg * 4 + r;
                const bool ok = kidx < seq_len && (!CAUSAL || kidx <= qpos);
                s[r] = ok ? sv[r] : -INFINITY;
            }
        } else {
            scaled_scores4(acc, p.scale, s);
        }
    };
    const int q_first = q_off + (wave % QW) * 16;       // first query position of this wave
    // VASM: byte offset of this lane's 8-byte V^T piece inside a stage for (key block kb, half j): row fr of a d-tile (d-tile i adds i * 2048 in the
    // instruction's offset field; vt_swz(i * 16 + fr) == vt_swz(fr))
    unsigned vo[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int j = 0; j < 2; ++j) vo[kb][j] = (unsigned)(K_BYTES + fr * 128 + (fg & 1) * 8 + (((kb * 4 + (fg >> 1) + 2 * j) ^ vt_swz(fr)) << 4));

    float m = -INFINITY, l = 0.f, inv_l = 0.f;
    f32x4 oacc[C::DT];
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int j = 0; j < NS - 1; ++j)
        if (j < total) issue(j, j);
    int buf = 0;
    for (int u = 0; u < total; ++u) {
        // DMA instructions of the steps issued after step u (u + 1 .. u + NS - 2): 2 per pass-1 step, 4 per pass-2 step
        int later = 0;
#pragma unroll
        for (int j = 1; j <= NS - 2; ++j)
            if (u + j < total) later += (u + j >= nt) ? 4 : 2;
        wait_vmcnt(later);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (u + NS - 1 < total) issue(u + NS - 1, buf == 0 ? NS - 1 : buf - 1);     // the buffer step u - 1 was computed from
        const unsigned char* ks = smem2 + buf * ST_BYTES;
        const unsigned char* vs = ks + K_BYTES;
        if (u < nt) {
            // ---------------- pass 1: row max m and sum l = sum exp(s - m)
            const int kv0 = u * KT;
            const bool masked = kv0 + KT > seq_len || (CAUSAL && kv0 + KT - 1 > q_first);
            float s[4][4];
#pragma unroll
            for (int t = 0; t < 4; ++t) scores16(ks, kv0, t, s[t], masked);
            float tm = -INFINITY;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) tm = fmaxf(tm, s[t][r]);
            tm = fmaxf(tm, __shfl_xor(tm, 16, 64));
            tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
            const float mn = fmaxf(m, tm);
            float ts = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) ts += __expf(s[t][r] - mn);
            ts += __shfl_xor(ts, 16, 64);
            ts += __shfl_xor(ts, 32, 64);
            l = l * __expf(m - mn) + ts;
            m = mn;
            if (u == nt - 1) inv_l = 1.0f / l;
        } else {
            // ---------------- pass 2: P = bf16(exp(s - m) / l);  O^T += V^T . P^T   (32 keys at a time: fewer live registers)
            const int kv0 = (u - nt) * KT;
            const bool masked = kv0 + KT > seq_len || (CAUSAL && kv0 + KT - 1 > q_first);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                float s0[4], s1[4];
                scores16(ks, kv0, 2 * kb, s0, masked);
                scores16(ks, kv0, 2 * kb + 1, s1, masked);
                uint4 pv;
                pv.x = pack2(__expf(s0[0] - m) * inv_l, __expf(s0[1] - m) * inv_l);
                pv.y = pack2(__expf(s0[2] - m) * inv_l, __expf(s0[3] - m) * inv_l);
                pv.z = pack2(__expf(s1[0] - m) * inv_l, __expf(s1[1] - m) * inv_l);
                pv.w = pack2(__expf(s1[2] - m) * inv_l, __expf(s1[3] - m) * inv_l);
                const bf16x8 pf = __builtin_bit_cast(bf16x8, pv);
                if constexpr (VASM) {
                    const unsigned a0 = lds0 + buf * ST_BYTES + vo[kb][0], a1 = lds0 + buf * ST_BYTES + vo[kb][1];
                    u32x2 v0[C::DT], v1[C::DT];
                    vt_issue<C::DT, 0>(a0, a1, v0, v1);
                    vt_issue<C::DT, 1>(a0, a1, v0, v1);
                    vt_consume<C::DT, 0>(v0, v1, pf, oacc);
                    vt_issue<C::DT, 2>(a0, a1, v0, v1);
                    vt_consume<C::DT, 1>(v0, v1, pf, oacc);
                    vt_issue<C::DT, 3>(a0, a1, v0, v1);
                    vt_consume<C::DT, 2>(v0, v1, pf, oacc);
                    vt_consume<C::DT, 3>(v0, v1, pf, oacc);
                    static_assert(C::DT <= 8, "four groups of two d-tiles");
                } else
#pragma unroll
                for (int dt = 0; dt < C::DT; ++dt) {
                    const int d = dt * 16 + fr;
                    const unsigned char* vr = vs + d * 128 + (fg & 1) * 8;
                    const int sw = vt_swz(d), c0 = kb * 4 + (fg >> 1);
                    uint2 v0 = *reinterpret_cast<const uint2*>(vr + ((c0 ^ sw) << 4));          // keys kb*32 + fg*4 .. +3
                    uint2 v1 = *reinterpret_cast<const uint2*>(vr + (((c0 + 2) ^ sw) << 4));    // keys kb*32 + 16 + fg*4 .. +3
                    const bf16x8 vf = __builtin_bit_cast(bf16x8, uint4{v0.x, v0.y, v1.x, v1.y});
                    oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, oacc[dt], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's LDS reads are done before it can reach the next barrier
        __builtin_amdgcn_sched_barrier(0);
        buf = buf + 1 == NS ? 0 : buf + 1;
    }
    if (q_valid) {
        bf16_t* optr = p.out + (size_t)(q_row0 + q_local) * p.out_stride + h * HD + fg * 4;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            uint2 v = {pack2(oacc[dt][0], oacc[dt][1]), pack2(oacc[dt][2], oacc[dt][3])};
            *reinterpret_cast<uint2*>(optr + dt * 16) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------- decode (q_len = 1)
// Two launches per layer.  A single CU sustains only ~50 GB/s, so the 295 KB of K / V^T of one (sequence, kv head)
// must be spread over many CUs (one block per (sequence, kv head) measured 13 us, 7 us of it waiting for 2 CUs to pull
// the cache).  Exact HF softmax needs the row max / sum over ALL keys before P is rounded, hence the split:
//   k_attn_dec_scores : grid (B, kvh, ctx/64)  mRoPE of the new q/k (bf16 ops, hf:557-599), KV-cache append, and the
//                       scaled + rounded scores of 64 keys -> scratch S[b][kvh][head][key] (bf16)
//   k_attn_dec_pv     : grid (B, kvh, 8)       softmax over all keys (float32, rounded to bf16), then one 16-wide d-tile
//                       of O^T = V^T . P^T
// The `group` query heads that share a kv head are the MFMA N dimension in both.
// Measured alternative (round 1): ONE launch with 8 blocks per (sequence, kv head), each recomputing all scores of its kv
// head and doing its d-slice of P.V (no scratch, no second launch, groups pinned to one XCD for L2 reuse of K) is bit-identical
// but SLOWER -- 11.3 vs 9.9 us per layer at batch 1, 17.0 vs 11.7 us at batch 32: 147 KB of K per block costs two dependent
// load rounds, more than the launch boundary it saves.
constexpr int DEC_HD = 128;
// -DSR_ATTN_TIMING (tools/probe_attn_decode_timeline.py builds its own library with it; never the product build): thread 0 of every block
// stamps the 100 MHz clock at the phase boundaries of the two decode attention kernels
#ifdef SR_ATTN_TIMING
__device__ long long g_tad[2][8192 * 5];
#define TAD(k, slot) do { if (threadIdx.x == 0) { const int b_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x; if (b_ < 8192) g_tad[k][b_ * 5 + (slot)] = wall_clock64(); } } while (0)
#else
#define TAD(k, slot)
#endif

__device__ __forceinline__ void rope_pair(float x1, float x2, float c, float s, float& o1, float& o2) {
    o1 = rbf(rbf(x1 * c) + rbf((-x2) * s));
    o2 = rbf(rbf(x2 * c) + rbf(x1 * s));
}

__global__ __launch_bounds__(256) void k_attn_dec_scores(DecodeAttnArgs p) {
    __shared__ __attribute__((aligned(16))) bf16_t q_s[16 * 136];
    __shared__ __attribute__((aligned(16))) bf16_t k_s[128];
    __shared__ float cs[64], sn[64];
    const int b = blockIdx.x, kvh = blockIdx.y, z = blockIdx.z;
    TAD(0, 0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int G = p.group, HQ = p.n_q_heads, HK = p.n_kv_heads;
    const bf16_t* row = p.qkv + (size_t)b * p.qkv_stride;
    // Everything that does not depend on the row's device-side state goes out FIRST, so that it travels together with the state loads
    // (ctx_len / slot / frozen) instead of one memory round trip behind them: the raw q / k / v of the new token and the rotary cos | sin
    // of its position (k_step wrote them for this step; without that buffer they hang off pos[b] like before).
    //   threads 0..127  : query head tid >> 3, 16-byte chunks (tid & 7) of both rotary halves
    //   threads 128..191: key element pair (d, d + 64)          (used by the block that owns the new token's cache row)
    //   threads 192..255: value element pair (d, d + 64)        (V^T append, no rotary, same block)
    const int qh = tid >> 3, qc = tid & 7;
    uint4 q1 = uint4{0, 0, 0, 0}, q2 = q1;
    float kx1 = 0.f, kx2 = 0.f;
    bf16_t vx1 = 0, vx2 = 0;
    if (tid < 128) {
        if (qh < G) {
            const bf16_t* q = row + (kvh * G + qh) * DEC_HD + qc * 8;
            q1 = *reinterpret_cast<const uint4*>(q);
            q2 = *reinterpret_cast<const uint4*>(q + 64);
        }
    } else if (tid < 192) {
        const bf16_t* k = row + (HQ + kvh) * DEC_HD + (tid - 128);
        kx1 = bf2f(k[0]);
        kx2 = bf2f(k[64]);
    } else {
        const bf16_t* v = row + (HQ + HK + kvh) * DEC_HD + (tid - 192);
        vx1 = v[0];
        vx2 = v[64];
    }
    float csv = 0.f;
    if (p.row_cs && tid < 128) csv = p.row_cs[(size_t)b * 128 + tid];
    const int nkeys = p.ctx_len[b];
    const int slot = p.slots ? p.slots[b] : b;
    const bool frozen = p.frozen && p.frozen[b];
    if (z * 64 >= nkeys) return;
    const int idx = nkeys - 1;                    // cache row of the new token
    bf16_t* kc = p.kcache + (size_t)(slot * HK + kvh) * p.ctx_max * DEC_HD;
    bf16_t* vc = p.vtcache + (size_t)(slot * HK + kvh) * DEC_HD * p.ctx_max;
    const int ntiles = (nkeys + 15) / 16;
    const int t = z * 4 + wave;
    // K prefetch (independent of the new token; its row is patched from LDS below): in flight while q / k are rotated
    bf16x8 kf[4];
    const int key = min(t * 16 + fr, nkeys - 1);
    if (t < ntiles) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) kf[kk] = *reinterpret_cast<const bf16x8*>(kc + (size_t)key * DEC_HD + kk * 32 + fg * 8);
    }
    const bool owner = z == (idx >> 6) && !frozen;
    if (owner && tid >= 192) {
        const int d = tid - 192;
        vc[(size_t)d * p.ctx_max + idx] = vx1;
        vc[(size_t)(d + 64) * p.ctx_max + idx] = vx2;
    }
    if (p.row_cs) {
        if (tid < 64) cs[tid] = csv;
        else if (tid < 128) sn[tid - 64] = csv;
    } else if (tid < 64) {
        const int pos = p.pos[b];
        cs[tid] = bf2f(p.rope_cos[(size_t)pos * 64 + tid]);
        sn[tid] = bf2f(p.rope_sin[(size_t)pos * 64 + tid]);
    }
    __syncthreads();
    TAD(0, 1);                                    // state, q / k / v and the rotary row have arrived
    if (tid < 128) {
        const float x1[8] = {lo16(q1.x), hi16(q1.x), lo16(q1.y), hi16(q1.y), lo16(q1.z), hi16(q1.z), lo16(q1.w), hi16(q1.w)};
        const float x2[8] = {lo16(q2.x), hi16(q2.x), lo16(q2.y), hi16(q2.y), lo16(q2.z), hi16(q2.z), lo16(q2.w), hi16(q2.w)};
        float o1[8], o2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) rope_pair(x1[e], x2[e], cs[qc * 8 + e], sn[qc * 8 + e], o1[e], o2[e]);
        *reinterpret_cast<uint4*>(q_s + qh * 136 + qc * 8) = uint4{pack2(o1[0], o1[1]), pack2(o1[2], o1[3]), pack2(o1[4], o1[5]), pack2(o1[6], o1[7])};
        *reinterpret_cast<uint4*>(q_s + qh * 136 + 64 + qc * 8) = uint4{pack2(o2[0], o2[1]), pack2(o2[2], o2[3]), pack2(o2[4], o2[5]), pack2(o2[6], o2[7])};
    } else if (owner && tid < 192) {              // the block that owns the new token's key appends K
        const int d = tid - 128;
        float o1, o2;
        rope_pair(kx1, kx2, cs[d], sn[d], o1, o2);
        const bf16_t b1 = f2bf(o1), b2 = f2bf(o2);
        kc[(size_t)idx * DEC_HD + d] = b1;
        kc[(size_t)idx * DEC_HD + d + 64] = b2;
        k_s[d] = b1;
        k_s[d + 64] = b2;
    }
    __syncthreads();
    TAD(0, 2);                                    // q / k rotated
    if (t >= ntiles) return;
    if (key == idx) {                             // rows of (or clamped to) the new token come from LDS
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) kf[kk] = *reinterpret_cast<const bf16x8*>(k_s + kk * 32 + fg * 8);
    }
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const bf16x8 qf = *reinterpret_cast<const bf16x8*>(q_s + fr * 136 + kk * 32 + fg * 8);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kk], qf, acc, 0, 0, 0);
    }
    TAD(0, 3);                                    // K fragments arrived, MFMAs issued
    if (fr < G) {
        uint2 v = {pack2(rbf(acc[0]) * p.scale, rbf(acc[1]) * p.scale), pack2(rbf(acc[2]) * p.scale, rbf(acc[3]) * p.scale)};
        *reinterpret_cast<uint2*>(p.scores + ((size_t)(b * HK + kvh) * G + fr) * p.ctx_max + t * 16 + fg * 4) = v;
    }
    TAD(0, 4);
}

// DT = 16-wide d-tiles per block: 1 at small batch (most blocks), 2 at batch >= 16 (the softmax of a (sequence, kv head) is then
// recomputed by 4 blocks instead of 8)
template <int DT>
__global__ __launch_bounds__(256) void k_attn_dec_pv(DecodeAttnArgs p, int s_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    // LDS: ored[DT][3][64] f32x4 | sb[group][s_stride] bf16
    f32x4* ored = reinterpret_cast<f32x4*>(dsm);
    bf16_t* sb = reinterpret_cast<bf16_t*>(ored + DT * 3 * 64);
    const int b = blockIdx.x, kvh = blockIdx.y, dt0 = blockIdx.z * DT;
    TAD(1, 0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int G = p.group, HK = p.n_kv_heads;
    // Short caches (ctx_max <= 1024): the score rows of this kv head are requested in full BEFORE the row's state (ctx_len, slot) is
    // known -- their address does not depend on it -- so that they arrive together with it and the softmax runs while V^T is still in
    // flight, instead of one round trip later.  Columns past the context hold stale values; they are masked below like the padding.
    const bf16_t* sg = p.scores + (size_t)(b * HK + kvh) * G * p.ctx_max;
    constexpr int SPEC = 4;
    const int nvm = p.ctx_max / 8;
    const bool spec = p.ctx_max <= 1024 && G * nvm <= 256 * SPEC;
    uint4 su[SPEC];
#pragma unroll
    for (int k = 0; k < SPEC; ++k) su[k] = uint4{0, 0, 0, 0};
    if (spec) {
#pragma unroll
        for (int k = 0; k < SPEC; ++k) {
            const int i = tid + k * 256;
            if (i < G * nvm) su[k] = *reinterpret_cast<const uint4*>(sg + (size_t)(i / nvm) * p.ctx_max + (i % nvm) * 8);
        }
    }
    const int slot = p.slots ? p.slots[b] : b;
    const int nkeys = p.ctx_len[b];
    const bf16_t* vc = p.vtcache + (size_t)(slot * HK + kvh) * DEC_HD * p.ctx_max;
    const int npad = (nkeys + 31) / 32 * 32, nkb = npad / 32, nvec = npad / 8;
    const uint4 z4 = uint4{0, 0, 0, 0};
    // early V^T prefetch: this wave's key blocks kb = wave, wave+4, ... of d-tile dt
    const bf16_t* vrow[DT];
    bf16x8 vf0[DT][8];
#pragma unroll
    for (int d = 0; d < DT; ++d) {
        vrow[d] = vc + (size_t)((dt0 + d) * 16 + fr) * p.ctx_max + fg * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (wave + 4 * i < nkb) vf0[d][i] = *reinterpret_cast<const bf16x8*>(vrow[d] + (wave + 4 * i) * 32);
    }
    // scores of all heads of this kv head -> LDS
    if (spec) {
#pragma unroll
        for (int k = 0; k < SPEC; ++k) {
            const int i = tid + k * 256;
            if (i < G * nvm && i % nvm < nvec) *reinterpret_cast<uint4*>(sb + (i / nvm) * s_stride + (i % nvm) * 8) = su[k];
        }
    } else {
        for (int i = tid; i < G * nvec; i += 256) {
            const int h = i / nvec, v = i % nvec;
            *reinterpret_cast<uint4*>(sb + h * s_stride + v * 8) = *reinterpret_cast<const uint4*>(sg + (size_t)h * p.ctx_max + v * 8);
        }
    }
    __syncthreads();
    TAD(1, 1);                                    // state + score rows arrived, scores in LDS
    // softmax per head (float32), probabilities rounded to bf16 in place; tail zero-filled
    for (int hh = wave; hh < G; hh += 4) {
        bf16_t* srow = sb + hh * s_stride;
        if (nvec <= 4 * 64) {
            float v[4][8];
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int vi = lane + i * 64;
                if (vi < nvec) {
                    const uint4 u = *reinterpret_cast<const uint4*>(srow + vi * 8);
                    const float t[8] = {lo16(u.x), hi16(u.x), lo16(u.y), hi16(u.y), lo16(u.z), hi16(u.z), lo16(u.w), hi16(u.w)};
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        v[i][e] = (vi * 8 + e < nkeys) ? t[e] : -INFINITY;
                        mx = fmaxf(mx, v[i][e]);
                    }
                }
            }
            mx = wave_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (lane + i * 64 < nvec) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { v[i][e] = __expf(v[i][e] - mx); sum += v[i][e]; }
                }
            sum = wave_sum(sum);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int vi = lane + i * 64;
                if (vi < nvec)
                    *reinterpret_cast<uint4*>(srow + vi * 8) = uint4{pack2(v[i][0] * inv, v[i][1] * inv), pack2(v[i][2] * inv, v[i][3] * inv),
                                                                      pack2(v[i][4] * inv, v[i][5] * inv), pack2(v[i][6] * inv, v[i][7] * inv)};
            }
        } else {                                                 // long contexts: three scalar passes
            float mx = -INFINITY;
            for (int j = lane; j < nkeys; j += 64) mx = fmaxf(mx, bf2f(srow[j]));
            mx = wave_max(mx);
            float sum = 0.f;
            for (int j = lane; j < nkeys; j += 64) sum += __expf(bf2f(srow[j]) - mx);
            sum = wave_sum(sum);
            const float inv = 1.0f / sum;
            for (int j = lane; j < npad; j += 64) srow[j] = (j < nkeys) ? f2bf(__expf(bf2f(srow[j]) - mx) * inv) : (bf16_t)0;
        }
    }
    __syncthreads();
    TAD(1, 2);                                    // softmax done
    // O^T[d][head] for the block's d-tiles; the 4 waves split the key blocks and reduce through LDS in fixed order
    f32x4 oacc[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) oacc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto pfrag = [&](int kb) {
        uint4 pv = z4;
        if (fr < G) pv = *reinterpret_cast<const uint4*>(sb + fr * s_stride + kb * 32 + fg * 8);
        return __builtin_bit_cast(bf16x8, pv);
    };
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (wave + 4 * i < nkb) {
            const bf16x8 pf = pfrag(wave + 4 * i);
#pragma unroll
            for (int d = 0; d < DT; ++d) oacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf0[d][i], pf, oacc[d], 0, 0, 0);
        }
    for (int kb0 = wave + 32; kb0 < nkb; kb0 += 32) {          // contexts beyond 1024 keys
#pragma unroll
        for (int d = 0; d < DT; ++d) {
            bf16x8 vf[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (kb0 + 4 * i < nkb) vf[i] = *reinterpret_cast<const bf16x8*>(vrow[d] + (kb0 + 4 * i) * 32);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (kb0 + 4 * i < nkb) oacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[i], pfrag(kb0 + 4 * i), oacc[d], 0, 0, 0);
        }
    }
    TAD(1, 3);                                    // V^T arrived, MFMAs issued
    if (wave > 0) {
#pragma unroll
        for (int d = 0; d < DT; ++d) ored[(d * 3 + wave - 1) * 64 + lane] = oacc[d];
    }
    __syncthreads();
    if (wave == 0 && fr < G) {
#pragma unroll
        for (int d = 0; d < DT; ++d) {
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                const f32x4 o2 = ored[(d * 3 + w) * 64 + lane];
                oacc[d][0] += o2[0]; oacc[d][1] += o2[1]; oacc[d][2] += o2[2]; oacc[d][3] += o2[3];
            }
            uint2 v = {pack2(oacc[d][0], oacc[d][1]), pack2(oacc[d][2], oacc[d][3])};
            const int col = (kvh * G + fr) * DEC_HD + (dt0 + d) * 16 + fg * 4;      // 4 consecutive columns stay contiguous in fragment order
            *reinterpret_cast<uint2*>(p.out + (p.out_tiled ? tiled_offset((size_t)b, (size_t)col, (size_t)p.out_stride) : (size_t)b * p.out_stride + col)) = v;
        }
    }
    TAD(1, 4);
}


// (the one-launch decode attention of round 3, k_attn_dec_one -- bit-identical and slower, 2.617 vs 2.545 ms per step -- is no longer compiled in:
// tools/experiments/README.md)


}  // namespace

#ifdef SR_ATTN_TIMING
extern "C" __attribute__((visibility("default"))) int sr_dbg_attn_dec_times(long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_tad), sizeof(long long) * 2 * 8192 * 5, 0, hipMemcpyDeviceToHost);
}
#endif

template <int HD, bool CAUSAL, int QW, int NS, bool VASM>
static int launch_prefill2(hipStream_t s, const AttnArgs& a) {
    constexpr int NH = 8 / QW, QTB = QW * 16;
    constexpr int smem = NS * (KT * 256 + HD * 128);
    static bool attr = false;
    if (!attr) {
        hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_prefill2<HD, CAUSAL, QW, NS, VASM>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (r != hipSuccess) return (int)r;
        attr = true;
    }
    const int Y = a.n_heads / NH, subs = a.q_tile / QTB, n_units = a.n_work * Y * subs;
    hipLaunchKernelGGL((k_attn_prefill2<HD, CAUSAL, QW, NS, VASM>), dim3(n_units), dim3(512), smem, s, a, n_units, Y, subs);
    SR_CHECK_LAUNCH();
    return 0;
}

// which kernel takes the launch: 2 = k_attn_prefill2 (8-wave blocks sharing LDS-DMA-staged tiles; needs a.v2_ok from the caller:
// 16-byte aligned V^T key runs with finite contents, work items cut at a.q_tile queries), 1 = k_attn_prefill.  SR_ATTN2=0 forces the
// round-2 kernel (the bit-identity tests flip it and call sr_switches_reload).
int attn_prefill_variant(const AttnArgs& a, int head_dim) {
    if (!sr_switches().attn2) return 1;
    if (!a.v2_ok) return 1;
    if (head_dim == 128 && a.causal && a.q_tile == 64 && a.group % 4 == 0 && a.n_heads % 4 == 0) return 2;
    if (head_dim == 80 && !a.causal && a.q_tile == 128 && a.group == 1) return 2;
    return 1;
}

int launch_attn_prefill(hipStream_t s, const AttnArgs& a, int head_dim) {
    if (a.n_work <= 0) return 0;
    if (attn_prefill_variant(a, head_dim) == 2) {
        const bool vasm = sr_switches().attn_vasm;          // hand-issued ds_read_b64 for V^T (bit-identical; SR_ATTN_VASM=0: the compiler's ds_read2st64_b64)
        if (head_dim == 128) return vasm ? launch_prefill2<128, true, 2, 2, true>(s, a) : launch_prefill2<128, true, 2, 2, false>(s, a);       // 32 queries x 4 heads, 2 x 32 KB ring
        return vasm ? launch_prefill2<80, false, 8, 3, true>(s, a) : launch_prefill2<80, false, 8, 3, false>(s, a);                            // 128 queries x 1 head, 3 x 26 KB ring
    }
    if (a.win64 && head_dim == 80 && !a.causal && a.n_heads % 4 == 0) {      // every item = one 64-token window starting at its first query (caller's promise)
        if (sr_switches().attn_win64) {
            hipLaunchKernelGGL((k_attn_win64<80>), dim3(a.n_work, a.n_heads / 4), dim3(256), 0, s, a);
            SR_CHECK_LAUNCH();
            return 0;
        }
    }
    if (a.q_tile == 16) {      // work items of <= 16 queries (SAM2 token-to-image attention): the waves of a block split the keys
        if (a.causal || head_dim != 16) return -22;
        // 16 waves per block: 4 key tiles per wave and pass at 4096 keys (8 waves: 8 tiles; mask decoder for one object 0.64 -> 0.60 ms)
        constexpr int NW = 16;
        constexpr int smem = NW * (PrefillCfg<16>::K_BYTES + PrefillCfg<16>::V_BYTES) + NW * 16 * 8;
        static bool attr = false;
        if (!attr) {
            hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_fewq<16, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            if (r != hipSuccess) return (int)r;
            attr = true;
        }
        hipLaunchKernelGGL((k_attn_fewq<16, NW>), dim3(a.n_work, a.n_heads), dim3(NW * 64), smem, s, a);
        SR_CHECK_LAUNCH();
        return 0;
    }
    if (a.q_tile != 0 && a.q_tile != 64) return -22;           // k_attn_prefill walks 64-query work items
    dim3 grid(a.n_work, a.n_heads), block(256);
    if (head_dim == 80 && !a.causal) hipLaunchKernelGGL((k_attn_prefill<80, false>), grid, block, 0, s, a);
    else if (head_dim == 128 && a.causal) hipLaunchKernelGGL((k_attn_prefill<128, true>), grid, block, 0, s, a);
    else if (head_dim == 128 && !a.causal) hipLaunchKernelGGL((k_attn_prefill<128, false>), grid, block, 0, s, a);
    else if (head_dim == 16 && !a.causal) hipLaunchKernelGGL((k_attn_prefill<16, false>), grid, block, 0, s, a);      // SAM2 mask decoder: cross attention
    else if (head_dim == 32 && !a.causal) hipLaunchKernelGGL((k_attn_prefill<32, false>), grid, block, 0, s, a);      //                    token self-attention
    else return -22;
    SR_CHECK_LAUNCH();
    return 0;
}

static size_t dec_smem(int ctx_max, int group, int dtiles = 2) { return (size_t)dtiles * 3 * 64 * 16 + (size_t)group * (ctx_max + 8) * sizeof(bf16_t); }

// raises the dynamic-LDS limit once, outside of any stream capture
int attn_decode_prepare(int ctx_max, int group) {
    const size_t smem = dec_smem(ctx_max, group);
    if (smem > 160 * 1024) return -22;
    int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_dec_pv<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (rc) return rc;
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_dec_pv<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
}

int launch_attn_decode(hipStream_t s, const DecodeAttnArgs& a) {
    if (a.B <= 0) return 0;
    if (a.group > 16 || a.ctx_max % 64 != 0 || !a.scores) return -22;
    hipLaunchKernelGGL(k_attn_dec_scores, dim3(a.B, a.n_kv_heads, a.ctx_max / 64), dim3(256), 0, s, a);
    if (a.B >= 16) hipLaunchKernelGGL(k_attn_dec_pv<2>, dim3(a.B, a.n_kv_heads, DEC_HD / 32), dim3(256), dec_smem(a.ctx_max, a.group, 2), s, a, a.ctx_max + 8);
    else hipLaunchKernelGGL(k_attn_dec_pv<1>, dim3(a.B, a.n_kv_heads, DEC_HD / 16), dim3(256), dec_smem(a.ctx_max, a.group, 1), s, a, a.ctx_max + 8);
    SR_CHECK_LAUNCH();
    return 0;
}
