// HBM-bound kernels of the hot path: RMSNorm,
// rotary embeddings (ViT 2-D float32 / LM mRoPE bf16), embedding gather + image scatter, patchify, argmax,
// decode bookkeeping and the synthetic-weight generator.  All bf16 traffic is 8- or 16-byte vectorised.
// Rounding points follow HF's bf16 eager path (hf: transformers/models/qwen2_5_vl/modeling_qwen2_5_vl.py).
#include "kernels.h"
#include "rownorm.h"
#include <math.h>

namespace {

// ----------------------------------------------------------------------------------------------- RMSNorm (hf:65-79)
// One wave per row.  Optional fused residual: h = bf16(x + bf16(sum_ks part[ks])) is written back to x first.
template <int MAXC>
__global__ __launch_bounds__(256) void k_rmsnorm(bf16_t* x, const bf16_t* xin, const float* part, int ksplit,
                                                 const bf16_t* w, bf16_t* out, int rows, int H, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int nch = H / 8;
    float v[MAXC][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            uint4 u = *reinterpret_cast<const uint4*>(xin + (size_t)row * H + c * 8);
            v[i][0] = lo16(u.x); v[i][1] = hi16(u.x); v[i][2] = lo16(u.y); v[i][3] = hi16(u.y);
            v[i][4] = lo16(u.z); v[i][5] = hi16(u.z); v[i][6] = lo16(u.w); v[i][7] = hi16(u.w);
            if (part) {
                float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                for (int ks = 0; ks < ksplit; ++ks) {
                    const float4* pp = reinterpret_cast<const float4*>(part + ((size_t)ks * rows + row) * H + c * 8);
                    float4 p0 = pp[0], p1 = pp[1];
                    a[0] += p0.x; a[1] += p0.y; a[2] += p0.z; a[3] += p0.w;
                    a[4] += p1.x; a[5] += p1.y; a[6] += p1.z; a[7] += p1.w;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[i][e] = rbf(v[i][e] + rbf(a[e]));
                uint4 o = {pack2(v[i][0], v[i][1]), pack2(v[i][2], v[i][3]), pack2(v[i][4], v[i][5]), pack2(v[i][6], v[i][7])};
                *reinterpret_cast<uint4*>(x + (size_t)row * H + c * 8) = o;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += v[i][e] * v[i][e];
        }
    }
    ss = wave_sum(ss);
    const float rs = 1.0f / sqrtf(ss / (float)H + eps);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            uint4 wu = *reinterpret_cast<const uint4*>(w + c * 8);
            float wv[8] = {lo16(wu.x), hi16(wu.x), lo16(wu.y), hi16(wu.y), lo16(wu.z), hi16(wu.z), lo16(wu.w), hi16(wu.w)};
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = wv[e] * rbf(v[i][e] * rs);
            uint4 ov = {pack2(o[0], o[1]), pack2(o[2], o[3]), pack2(o[4], o[5]), pack2(o[6], o[7])};
            *reinterpret_cast<uint4*>(out + (size_t)row * H + c * 8) = ov;
        }
    }
}

// Decode-sized variant (rows <= 128 = the engine's row cap): one block of 256 threads per row, every load issued up front, so the launch is
// one memory latency deep instead of a wave-serial chain (6.5 -> ~3.5 us at 32 rows).  H <= 2048, H % 8 == 0.
__global__ __launch_bounds__(256) void k_rmsnorm_row(bf16_t* x, const bf16_t* xin, const float* part, int ksplit,
                                                     const bf16_t* w, bf16_t* out, int rows, int H, float eps, int out_tiled) {
    __shared__ float wsum[4];
    rmsnorm_row_body(x, xin, part, ksplit, w, out, rows, blockIdx.x, H, eps, out_tiled, wsum);     // rownorm.h
}

// ----------------------------------------------------------------------------------------------- ViT 2-D RoPE (hf:160-171)
// float32 math, one rounding.  cos/sin tables [n_rows][hd/2] (the two halves of HF's emb are identical).
// one thread = 8 rotary pairs: 16-byte loads of x[d..d+7] and x[d+half..], two float4 each of cos / sin (hd/2 % 8 == 0)
__global__ __launch_bounds__(256) void k_vit_rope(bf16_t* qkv, int n_rows, int n_heads, int hd, const float* cos_t,
                                                  const float* sin_t, int paired) {
    const int half = hd / 2, C = n_heads * hd, nch = half / 8, per_row = 2 * n_heads * nch;
    const long long task = (long long)blockIdx.x * 256 + threadIdx.x;
    if (task >= (long long)n_rows * per_row) return;
    const int row = (int)(task / per_row), rem = (int)(task % per_row);
    const int sec = rem / (n_heads * nch), h = (rem / nch) % n_heads, c = rem % nch;
    // HF channel order: x1 = d, x2 = d + half.  Paired order (vit_qk_perm): 8 x1 then their 8 x2 in every group of 16
    bf16_t* p = qkv + (size_t)row * 3 * C + sec * C + h * hd + (paired ? c * 16 : c * 8);
    bf16_t* p2 = paired ? p + 8 : p + half;
    const uint4 u1 = *reinterpret_cast<const uint4*>(p), u2 = *reinterpret_cast<const uint4*>(p2);
    const float4* cp = reinterpret_cast<const float4*>(cos_t + (size_t)row * half + c * 8);
    const float4* sp = reinterpret_cast<const float4*>(sin_t + (size_t)row * half + c * 8);
    const float4 c0 = cp[0], c1 = cp[1], s0 = sp[0], s1 = sp[1];
    const float x1[8] = {lo16(u1.x), hi16(u1.x), lo16(u1.y), hi16(u1.y), lo16(u1.z), hi16(u1.z), lo16(u1.w), hi16(u1.w)};
    const float x2[8] = {lo16(u2.x), hi16(u2.x), lo16(u2.y), hi16(u2.y), lo16(u2.z), hi16(u2.z), lo16(u2.w), hi16(u2.w)};
    const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, ss[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    float o1[8], o2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {      // float32 mul, mul, add as HF's eager ops do (hf:124-135): no fused multiply-add
        o1[e] = __fadd_rn(__fmul_rn(x1[e], cc[e]), __fmul_rn(-x2[e], ss[e]));
        o2[e] = __fadd_rn(__fmul_rn(x2[e], cc[e]), __fmul_rn(x1[e], ss[e]));
    }
    *reinterpret_cast<uint4*>(p) = uint4{pack2(o1[0], o1[1]), pack2(o1[2], o1[3]), pack2(o1[4], o1[5]), pack2(o1[6], o1[7])};
    *reinterpret_cast<uint4*>(p2) = uint4{pack2(o2[0], o2[1]), pack2(o2[2], o2[3]), pack2(o2[4], o2[5]), pack2(o2[6], o2[7])};
}

// V^T for the attention kernel: vt[h][d][row] = qkv[row][2C + h*hd + d]; 64 rows x one head per block via LDS, 16-byte
// global loads and stores (hd % 8 == 0)
__global__ __launch_bounds__(256) void k_vit_vtranspose(const bf16_t* qkv, int n_rows, int n_heads, int hd, bf16_t* vt,
                                                        int vt_stride) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64][136];
    const int r0 = blockIdx.x * 64, h = blockIdx.y, C = n_heads * hd, nv = hd / 8;
    for (int i = threadIdx.x; i < 64 * nv; i += 256) {
        const int r = i / nv, v = i % nv;
        uint4 u = uint4{0, 0, 0, 0};
        if (r0 + r < n_rows) u = *reinterpret_cast<const uint4*>(qkv + (size_t)(r0 + r) * 3 * C + 2 * C + h * hd + v * 8);
        *reinterpret_cast<uint4*>(&tile[r][v * 8]) = u;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < hd * 8; i += 256) {
        const int d = i / 8, rv = i % 8;                      // 8 consecutive rows -> one 16-byte store
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (uint32_t)tile[rv * 8 + 2 * e][d] | ((uint32_t)tile[rv * 8 + 2 * e + 1][d] << 16);
        *reinterpret_cast<uint4*>(vt + ((size_t)h * hd + d) * vt_stride + r0 + rv * 8) = uint4{w[0], w[1], w[2], w[3]};
    }
}

// ----------------------------------------------------------------------------------------------- LM mRoPE (hf:486-599)
// cos/sin are bf16 (hf:538); q*cos, rot(q)*sin and their sum are three separate bf16 ops -> three roundings.
__device__ __forceinline__ void rope_pair_bf16(float x1, float x2, float c, float s, float& o1, float& o2) {
    o1 = rbf(rbf(x1 * c) + rbf((-x2) * s));
    o2 = rbf(rbf(x2 * c) + rbf(x1 * s));
}

// The precise cosf/sinf of the device library cost ~7 us per call chain at these argument sizes (measured), so the
// LM tables are built once per engine and the rope kernels only look them up.
__global__ __launch_bounds__(256) void k_rope_table(const float* inv_freq, int n_pos, bf16_t* cos_t, bf16_t* sin_t) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pos * 64) return;
    const float ang = (float)(i >> 6) * inv_freq[i & 63];
    cos_t[i] = f2bf(cosf(ang));
    sin_t[i] = f2bf(sinf(ang));
}

__global__ __launch_bounds__(256) void k_lm_rope_prefill(LmRopeArgs p) {
    __shared__ float cs[64], sn[64];
    const int t = blockIdx.x, tid = threadIdx.x;
    if (tid < 64) {
        const int axis = tid < p.sec0 ? 0 : (tid < p.sec1 ? 1 : 2);
        const int pos = p.pos3[(size_t)axis * p.n_tok + t];
        cs[tid] = bf2f(p.rope_cos[(size_t)pos * 64 + tid]);
        sn[tid] = bf2f(p.rope_sin[(size_t)pos * 64 + tid]);
    }
    __syncthreads();
    const int HQ = p.n_q_heads, HK = p.n_kv_heads;
    bf16_t* row = p.qkv + (size_t)t * (HQ + 2 * HK) * 128;
    const int slot = p.tok_slot[t], idx = p.tok_idx[t];
    for (int i = tid; i < (HQ + HK) * 64; i += 256) {
        const int h = i >> 6, d = i & 63;
        bf16_t* q = row + h * 128 + d;
        float o1, o2;
        rope_pair_bf16(bf2f(q[0]), bf2f(q[64]), cs[d], sn[d], o1, o2);
        if (h < HQ) {
            q[0] = f2bf(o1);
            q[64] = f2bf(o2);
        } else {
            bf16_t* kc = p.kcache + ((size_t)(slot * HK + (h - HQ)) * p.ctx_max + idx) * 128 + d;
            kc[0] = f2bf(o1);
            kc[64] = f2bf(o2);
        }
    }
    for (int i = tid; i < HK * 128; i += 256) {
        const int h = i >> 7, d = i & 127;
        p.vtcache[((size_t)(slot * HK + h) * 128 + d) * p.ctx_max + idx] = row[(HQ + HK) * 128 + i];
    }
}

// ----------------------------------------------------------------------------------------------- gathers
// src >= 0: token id -> embedding row; src < 0: image feature row -(src+1)  (hf:1210-1216 masked_scatter)
__global__ __launch_bounds__(256) void k_embed(const int* src, const bf16_t* table, const bf16_t* img, bf16_t* out, int H, int tiled) {
    const int t = blockIdx.x, s = src[t];
    for (int c = threadIdx.x; c < H / 8; c += blockDim.x) {
        const bf16_t* from = s < 0 ? img + (size_t)(-s - 1) * H + c * 8
                                   : table + (tiled ? tiled_offset(s, (size_t)c * 8, H) : (size_t)s * H + c * 8);
        reinterpret_cast<uint4*>(out + (size_t)t * H)[c] = *reinterpret_cast<const uint4*>(from);
    }
}
__global__ __launch_bounds__(256) void k_gather_rows(const bf16_t* in, const int* rows, bf16_t* out, int H) {
    const int t = blockIdx.x;
    const bf16_t* from = in + (size_t)rows[t] * H;
    for (int c = threadIdx.x; c < H / 8; c += blockDim.x)
        reinterpret_cast<uint4*>(out + (size_t)t * H)[c] = reinterpret_cast<const uint4*>(from)[c];
}

// uint8 HWC image -> normalised bf16 patches [N, ld_out], rows in (gh/m, gw/m, m, m) order, cols (C, T, p, p)
// (hf: models/qwen2_vl/image_processing_pil_qwen2_vl.py:153-185); lut[c*256+u] = bf16((u/255 - mean_c)/std_c)
__global__ __launch_bounds__(256) void k_patchify(const uint8_t* img, int h, int w, const bf16_t* lut, bf16_t* out,
                                                  int ld_out, int P, int mg, int T) {
    const int row = blockIdx.x;
    const int gw = w / P;
    const int mw = row % mg, mh = (row / mg) % mg, bw = (row / (mg * mg)) % (gw / mg), bh = row / (mg * mg * (gw / mg));
    const int y0 = (bh * mg + mh) * P, x0 = (bw * mg + mw) * P;
    const int ncol = 3 * T * P * P;
    for (int col = threadIdx.x; col < ld_out; col += blockDim.x) {
        bf16_t v = 0;
        if (col < ncol) {
            const int px = col % P, py = (col / P) % P, c = col / (T * P * P);
            v = lut[c * 256 + img[((size_t)(y0 + py) * w + x0 + px) * 3 + c]];
        }
        out[(size_t)row * ld_out + col] = v;
    }
}
__global__ __launch_bounds__(256) void k_f32_to_bf16_pad(const float* in, int cols, bf16_t* out, int ld_out) {
    const int row = blockIdx.x;
    for (int c = threadIdx.x; c < ld_out; c += blockDim.x)
        out[(size_t)row * ld_out + c] = c < cols ? f2bf(in[(size_t)row * cols + c]) : (bf16_t)0;
}

// ----------------------------------------------------------------------------------------------- greedy argmax
// lowest index among the maxima (torch.argmax on the reference's CPU path)
__global__ __launch_bounds__(1024) void k_argmax(const float* logits, int V, int* out_idx) {
    __shared__ float sv[16];
    __shared__ int si[16];
    const float* row = logits + (size_t)blockIdx.x * V;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < V; i += blockDim.x) {
        const float v = row[i];
        if (v > bv) { bv = v; bi = i; }   // strided ascending scan keeps the lowest index per thread
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < (int)(blockDim.x >> 6); ++k)
            if (sv[k] > bv || (sv[k] == bv && si[k] < bi)) { bv = sv[k]; bi = si[k]; }
        out_idx[blockIdx.x] = bi;
    }
}

// lowest index among the float32 maxima of row b's LM-head partials; valid in thread 0
__device__ __forceinline__ int reduce_amax_partials(const float* amax_val, const int* amax_idx, int n_part, int b, float* sv, int* si) {
    const int tid = threadIdx.x;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < n_part; i += 256) {
        const float v = amax_val[(size_t)b * n_part + i];
        const int ix = amax_idx[(size_t)b * n_part + i];
        if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { sv[tid >> 6] = bv; si[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0)
        for (int k = 1; k < 4; ++k)
            if (sv[k] > bv || (sv[k] == bv && si[k] < bi)) { bv = sv[k]; bi = si[k]; }
    return bi;
}

// greedy ids of the current logits as int64 (sr_decode_step's next_ids)
__global__ __launch_bounds__(256) void k_next_ids(const float* amax_val, const int* amax_idx, int n_part, long long* out) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int bi = reduce_amax_partials(amax_val, amax_idx, n_part, blockIdx.x, sv, si);
    if (threadIdx.x == 0) out[blockIdx.x] = bi;
}

// per-step bookkeeping on the device so that one captured graph replays for every step.  One block per sequence:
// finish the greedy argmax from the LM-head partials (lowest index among maxima), log the token, eos / teacher
// forcing, advance (ctx_len, pos, step) and gather the embedding row of the token that is fed back.
__global__ __launch_bounds__(256) void k_step(StepArgs a) {
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ int s_feed, s_pos;
    const int b = blockIdx.x, tid = threadIdx.x;
    int bi = reduce_amax_partials(a.amax_val, a.amax_idx, a.n_part, b, sv, si);
    if (tid == 0) {
        if (a.chosen) bi = (int)a.chosen[b];        // the caller sampled this step's token itself
        const int step = a.step[b];
        int tok = bi;
        const int fin = a.finished[b];
        if (fin) tok = a.pad_id;
        if (step < a.max_new) a.tokens_out[(size_t)b * a.max_new + step] = tok;
        if (!fin) a.n_gen[b] = step + 1;
        bool is_eos = false;
        for (int k = 0; k < a.n_eos; ++k) is_eos |= (tok == a.eos[k]);
        if (!fin && (is_eos || (a.row_limit && step + 1 >= a.row_limit[b]))) a.finished[b] = 1;
        a.cur_tok[b] = tok;
        int feed = tok;
        if (a.forced && step < a.max_new) feed = a.forced[(size_t)b * a.max_new + step];
        if (fin) feed = 0;
        int pos = a.pos[b];
        if (!fin && !is_eos) { a.ctx_len[b] += 1; pos += 1; a.pos[b] = pos; }
        a.step[b] = step + 1;
        s_feed = feed;
        s_pos = pos;
    }
    __syncthreads();
    if (a.row_cs && tid < 128) {       // rotary cos | sin of the position the coming forward pass uses, as the float32 values the attention kernel multiplies with
        const bf16_t* t = tid < 64 ? a.rope_cos : a.rope_sin;
        a.row_cs[(size_t)b * 128 + tid] = bf2f(t[(size_t)s_pos * 64 + (tid & 63)]);
    }
    for (int c = tid; c < a.H / 8; c += 256) {
        const bf16_t* from = a.table + (a.table_tiled ? tiled_offset(s_feed, (size_t)c * 8, a.H) : (size_t)s_feed * a.H + c * 8);
        reinterpret_cast<uint4*>(a.x + (size_t)b * a.H)[c] = *reinterpret_cast<const uint4*>(from);
    }
}

// counter-based synthetic weights; definition shared with oracle/weights.py (independent implementations)
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
__global__ __launch_bounds__(256) void k_synth_fill(bf16_t* out, long long n, uint32_t key, float base, float scale) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const uint32_t h = mix32((uint32_t)i * 0x9E3779B1u + key);
        const int s = (int)((h & 255u) + ((h >> 8) & 255u) + ((h >> 16) & 255u) + (h >> 24)) - 510;
        out[i] = f2bf(base + (float)s * scale);
    }
}

// weight loader: HF tensor [rows, cols] (bf16 or f32) -> engine layout.  mode 0: dst row = r + row_off;
// mode 1 / 2: gate / up rows interleaved in blocks of 16 (dst row = (r/16)*32 + r%16 (+16 for up));
// mode 3: ViT qkv [3C, .]: the q and k channels of every head go to their paired order (vit_qk_perm, kernels.h; row_off = head dim)
__global__ __launch_bounds__(256) void k_load2d(const void* src, int dtype, long long rows, long long cols, bf16_t* dst,
                                                long long dst_ld, int mode, long long row_off, int tiled) {
    const long long n = rows * cols;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long r = i / cols, c = i % cols;
        long long dr;
        if (mode == 0) dr = r + row_off;
        else if (mode == 3) {
            const long long C = rows / 3, hd = row_off;
            dr = r >= 2 * C ? r : r - (r % hd) + vit_qk_perm((int)(r % hd), (int)hd);      // (C % hd == 0: sections and heads start on multiples of hd)
        } else dr = (r / 16) * 32 + (r % 16) + (mode == 2 ? 16 : 0);
        const bf16_t v = dtype == 0 ? reinterpret_cast<const bf16_t*>(src)[i] : f2bf(reinterpret_cast<const float*>(src)[i]);
        dst[tiled ? (long long)tiled_offset(dr, c, dst_ld) : dr * dst_ld + c] = v;
    }
}



// continuous batching: one thread per admitted sequence installs its row state; the pending first token goes into the
// row's LM-head partials as a single (0, token) entry among -inf, which is what k_step reduces
__global__ void k_admit_rows(AdmitArgs a) {
    const int i = blockIdx.x;
    if (i >= a.n) return;
    const int r = a.rows[i];
    if (threadIdx.x == 0) {
        a.ctx_len[r] = a.ctx[i];
        a.d_pos[r] = a.pos[i];
        a.slots[r] = a.kv_slot ? a.kv_slot[i] : r;
        a.finished[r] = 0;
        a.step[r] = 0;
        a.n_gen[r] = 0;
        a.row_limit[r] = a.limit[i];
    }
    for (int k = threadIdx.x; k < a.n_part; k += blockDim.x) {
        a.amax_val[(size_t)r * a.n_part + k] = k == 0 ? 0.0f : -INFINITY;
        a.amax_idx[(size_t)r * a.n_part + k] = k == 0 ? a.first_tok[i] : 0x7fffffff;
    }
}

// continuous batching: rows named by the mask stop now (k_step then logs pads for them and leaves their cache slot alone)
__global__ void k_rows_abort(uint4 mask, int* finished) {
    // (selects, not an indexed local array: on this toolchain the indexed form read word 0 for every lane of the first wave -- rows 32..63 took
    // rows 0..31's bits; tools/probe_abort.py)
    const unsigned g = threadIdx.x >> 5;
    const unsigned word = g == 0 ? mask.x : g == 1 ? mask.y : g == 2 ? mask.z : mask.w;
    if (threadIdx.x < 128 && ((word >> (threadIdx.x & 31)) & 1u)) finished[threadIdx.x] = 1;
}

// fp8 weight quantisation, one block per 16-row tile of a fragment-ordered bf16 matrix (see kernels.h)
__global__ __launch_bounds__(256) void k_quant_f8(bf16_t* W, int K, unsigned char* W8, float* scale) {
    __shared__ float amax_s[16][17];
    __shared__ float sc_s[16];
    const int tile = blockIdx.x, tid = threadIdx.x;
    bf16_t* wt = W + (size_t)tile * 16 * K;                 // the tile's 16 x K elements are contiguous in tiled16x64
    // vector v of the tile: chunk = v >> 7, kstep = (v >> 6) & 1, lane = v & 63, row = lane & 15 = v & 15.  A thread's
    // vectors (v = tid + 256 j) all belong to row tid & 15.
    const int nvec = 16 * K / 8;                              // 16-byte vectors
    float mx = 0.f;
    for (int v = tid; v < nvec; v += 256) {
        const uint4 u = *reinterpret_cast<const uint4*>(wt + (size_t)v * 8);
        const float f[8] = {lo16(u.x), hi16(u.x), lo16(u.y), hi16(u.y), lo16(u.z), hi16(u.z), lo16(u.w), hi16(u.w)};
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(f[e]));
    }
    // thread tid owns row tid & 15; 16 threads per row
    amax_s[tid & 15][tid >> 4] = mx;
    __syncthreads();
    if (tid < 16) {
        float m = 0.f;
        for (int j = 0; j < 16; ++j) m = fmaxf(m, amax_s[tid][j]);
        const float sc = m > 0.f ? __fdiv_rn(m, 448.0f) : 1.0f;
        sc_s[tid] = sc;
        scale[tile * 16 + tid] = sc;
    }
    __syncthreads();
    const float sc = sc_s[tid & 15];
    unsigned char* w8t = W8 + (size_t)tile * 16 * K;          // tiled8: the tile's 16 x K bytes are contiguous
    for (int v = tid; v < nvec; v += 256) {
        const int chunk = v >> 7, kstep = (v >> 6) & 1, lane = v & 63;
        uint4 u = *reinterpret_cast<const uint4*>(wt + (size_t)v * 8);
        const float f[8] = {lo16(u.x), hi16(u.x), lo16(u.y), hi16(u.y), lo16(u.z), hi16(u.z), lo16(u.w), hi16(u.w)};
        int p0 = 0, p1 = 0;
        p0 = __builtin_amdgcn_cvt_pk_fp8_f32(__fdiv_rn(f[0], sc), __fdiv_rn(f[1], sc), p0, false);
        p0 = __builtin_amdgcn_cvt_pk_fp8_f32(__fdiv_rn(f[2], sc), __fdiv_rn(f[3], sc), p0, true);
        p1 = __builtin_amdgcn_cvt_pk_fp8_f32(__fdiv_rn(f[4], sc), __fdiv_rn(f[5], sc), p1, false);
        p1 = __builtin_amdgcn_cvt_pk_fp8_f32(__fdiv_rn(f[6], sc), __fdiv_rn(f[7], sc), p1, true);
        *reinterpret_cast<uint2*>(w8t + ((size_t)chunk * 64 + lane) * 16 + kstep * 8) = uint2{(uint32_t)p0, (uint32_t)p1};
        uint32_t a0, a1, b0, b1;
        f8x4_to_bf16((uint32_t)p0, a0, a1);
        f8x4_to_bf16((uint32_t)p1, b0, b1);
        *reinterpret_cast<uint4*>(wt + (size_t)v * 8) = uint4{a0, a1, b0, b1};
    }
}

// MX activation quantiser of the fp8 prefill path (BASELINE.json configs[4]; definition: oracle/model_ref.py mx_quantize, OCP
// Microscaling v1.0 with e4m3 elements): per row and 32-wide k-block, shared scale X = 2^(floor(log2(max|v|)) - 8) (e8m0, clamped
// to 2^-127..2^127; an all-zero block takes 2^-127), elements = fp8_e4m3(v / X) with round-to-nearest-even and saturation to +-448.
// One thread per block: 64 B in, 32 B out; the 4 scale bytes of a row's 128-wide k-tile are contiguous: scales[k/128][row][4].
__global__ __launch_bounds__(256) void k_quant_mx_act(const bf16_t* x, int ldx, int M, int K, unsigned char* q, unsigned char* scales, int rows_pad) {
    const int nb = K / 32;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)M * nb) return;
    const int m = (int)(t / nb), b = (int)(t % nb);
    const uint4* src = reinterpret_cast<const uint4*>(x + (size_t)m * ldx + b * 32);
    uint4 u[4] = {src[0], src[1], src[2], src[3]};
    float v[32];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i * 8 + 0] = lo16(u[i].x); v[i * 8 + 1] = hi16(u[i].x); v[i * 8 + 2] = lo16(u[i].y); v[i * 8 + 3] = hi16(u[i].y);
        v[i * 8 + 4] = lo16(u[i].z); v[i * 8 + 5] = hi16(u[i].z); v[i * 8 + 6] = lo16(u[i].w); v[i * 8 + 7] = hi16(u[i].w);
    }
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) amax = fmaxf(amax, fabsf(v[i]));
    // floor(log2(amax)) from the exponent field (bf16 inputs widened to float32 are normal or zero)
    int e = (amax > 0.f) ? (int)((__float_as_uint(amax) >> 23) & 255) - 127 - 8 : -127;
    e = max(-127, min(127, e));
    const float inv = __uint_as_float((uint32_t)(127 - e) << 23);          // 2^-e, exact (127 - e in 0..254; e = 127 -> 2^-127 is subnormal:
    const float inv2 = (e == 127) ? 5.877471754111438e-39f : inv;          //  written out, the shift above would give 0)
    int p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float a0 = fminf(fmaxf(v[4 * i + 0] * inv2, -448.f), 448.f), a1 = fminf(fmaxf(v[4 * i + 1] * inv2, -448.f), 448.f);
        float a2 = fminf(fmaxf(v[4 * i + 2] * inv2, -448.f), 448.f), a3 = fminf(fmaxf(v[4 * i + 3] * inv2, -448.f), 448.f);
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(a0, a1, w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(a2, a3, w, true);
        p[i] = w;
    }
    uint4* dst = reinterpret_cast<uint4*>(q + (size_t)m * K + b * 32);
    dst[0] = uint4{(uint32_t)p[0], (uint32_t)p[1], (uint32_t)p[2], (uint32_t)p[3]};
    dst[1] = uint4{(uint32_t)p[4], (uint32_t)p[5], (uint32_t)p[6], (uint32_t)p[7]};
    scales[((size_t)(b >> 2) * rows_pad + m) * 4 + (b & 3)] = (unsigned char)(e + 127);
}
}  // namespace

int launch_load2d(hipStream_t s, const void* src, int dtype, long long rows, long long cols, bf16_t* dst, long long dst_ld,
                  int mode, long long row_off, int tiled) {
    const long long n = rows * cols;
    if (n <= 0) return 0;
    long long blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_load2d, dim3((unsigned)blocks), dim3(256), 0, s, src, dtype, rows, cols, dst, dst_ld, mode, row_off, tiled);
    SR_CHECK_LAUNCH();
    return 0;
}

int launch_rmsnorm(hipStream_t s, const bf16_t* x, const bf16_t* w, bf16_t* out, int rows, int H, float eps, int out_tiled, int per_wave) {
    if (rows <= 0) return 0;
    if (H % 8 != 0 || H > 64 * 8 * 12) return -22;
    if (out_tiled && !(rows <= 128 && H <= 2048 && H % 64 == 0)) return -22;
    dim3 g(cdiv(rows, 4)), b(256);
    if (out_tiled && per_wave) return -22;
    if (!per_wave && rows <= 128 && H <= 2048) hipLaunchKernelGGL(k_rmsnorm_row, dim3(rows), b, 0, s, (bf16_t*)nullptr, x, (const float*)nullptr, 0, w, out, rows, H, eps, out_tiled);
    else if (H <= 2048) hipLaunchKernelGGL((k_rmsnorm<4>), g, b, 0, s, (bf16_t*)nullptr, x, (const float*)nullptr, 0, w, out, rows, H, eps);
    else hipLaunchKernelGGL((k_rmsnorm<12>), g, b, 0, s, (bf16_t*)nullptr, x, (const float*)nullptr, 0, w, out, rows, H, eps);
    SR_CHECK_LAUNCH();
    return 0;
}
int launch_resid_rmsnorm(hipStream_t s, bf16_t* x, const float* part, int ksplit, const bf16_t* w, bf16_t* out, int rows,
                         int H, float eps, int out_tiled, int per_wave) {
    if (rows <= 0) return 0;
    if (H % 8 != 0 || H > 2048) return -22;
    if (out_tiled && !(rows <= 128 && ksplit <= 4 && H % 64 == 0)) return -22;
    if (out_tiled && per_wave) return -22;
    if (!per_wave && rows <= 128 && ksplit <= 4) hipLaunchKernelGGL(k_rmsnorm_row, dim3(rows), dim3(256), 0, s, x, (const bf16_t*)x, part, ksplit, w, out, rows, H, eps, out_tiled);
    else hipLaunchKernelGGL((k_rmsnorm<4>), dim3(cdiv(rows, 4)), dim3(256), 0, s, x, (const bf16_t*)x, part, ksplit, w, out, rows, H, eps);
    SR_CHECK_LAUNCH();
    return 0;
}
int launch_vit_rope(hipStream_t s, bf16_t* qkv, int n_rows, int n_heads, int head_dim, const float* cos_t,
                    const float* sin_t, bf16_t* vt, int vt_stride, int paired) {
    if (n_rows <= 0) return 0;
    if (head_dim > 128) return -22;
    if (head_dim % 16 != 0 || vt_stride % 8 != 0) return -22;
    const long long tasks = (long long)n_rows * 2 * n_heads * (head_dim / 16);
    hipLaunchKernelGGL(k_vit_rope, dim3((unsigned)((tasks + 255) / 256)), dim3(256), 0, s, qkv, n_rows, n_heads, head_dim, cos_t, sin_t, paired);
    hipLaunchKernelGGL(k_vit_vtranspose, dim3(cdiv(n_rows, 64), n_heads), dim3(256), 0, s, (const bf16_t*)qkv, n_rows, n_heads,
                       head_dim, vt, vt_stride);
    SR_CHECK_LAUNCH();
    return 0;
}
int launch_rows_abort(hipStream_t s, const unsigned row_mask[4], int* finished) {
    if (!(row_mask[0] | row_mask[1] | row_mask[2] | row_mask[3])) return 0;
    const uint4 m = make_uint4(row_mask[0], row_mask[1], row_mask[2], row_mask[3]);      // (a braced initialiser inside the launch macro's argument list is split at its commas)
    hipLaunchKernelGGL(k_rows_abort, dim3(1), dim3(128), 0, s, m, finished);
    SR_CHECK_LAUNCH();
    return 0;
}
int launch_lm_rope_prefill(hipStream_t s, const LmRopeArgs& a) {
    if (a.n_tok <= 0) return 0;
    hipLaunchKernelGGL(k_lm_rope_prefill, dim3(a.n_tok), dim3(256), 0, s, a);
    SR_CHECK_LAUNCH();
    return 0;
}
int launch_rope_table(hipStream_t s, const float* inv_freq, int n_pos, bf16_t* cos_t, bf16_t* sin_t) {
    hipLaunchKernelGGL(k_rope_table, dim3(cdiv(n_pos * 64, 256)), dim3(256), 0, s, inv_freq, n_pos, cos_t, sin_t);
    SR_CHECK_LAUNCH();
    return 0;
}
int launch_embed(hipStream_t s, const int* src, const bf16_t* table, const bf16_t* image_embeds, bf16_t* out, int n_tok, int H,
                 int table_tiled) {
    if (n_tok <= 0) return 0;
    hipLaunchKernelGGL(k_embed, dim3(n_tok), dim3(256), 0, s, src, table, image_embeds, out, H, table_tiled);
    SR_CHECK_LAUNCH();
    return 0;
}
int launch_gather_rows(hipStream_t s, const bf16_t* in, const int* rows, bf16_t* out, int n, int H) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_gather_rows, dim3(n), dim3(256), 0, s, in, rows, out, H);
    SR_CHECK_LAUNCH();
    return 0;
}
int launch_patchify(hipStream_t s, const uint8_t* img, int h, int w, const bf16_t* lut, bf16_t* out, int ld_out, int patch,
                    int merge, int temporal) {
    const int n = (h / patch) * (w / patch);
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_patchify, dim3(n), dim3(256), 0, s, img, h, w, lut, out, ld_out, patch, merge, temporal);
    SR_CHECK_LAUNCH();
    return 0;
}
int launch_f32_to_bf16_pad(hipStream_t s, const float* in, int rows, int cols, bf16_t* out, int ld_out) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(k_f32_to_bf16_pad, dim3(rows), dim3(256), 0, s, in, cols, out, ld_out);
    SR_CHECK_LAUNCH();
    return 0;
}
int launch_argmax(hipStream_t s, const float* logits, int rows, int V, int* out_idx) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(k_argmax, dim3(rows), dim3(1024), 0, s, logits, V, out_idx);
    SR_CHECK_LAUNCH();
    return 0;
}
int launch_next_ids(hipStream_t s, const float* amax_val, const int* amax_idx, int n_part, int B, long long* out) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(k_next_ids, dim3(B), dim3(256), 0, s, amax_val, amax_idx, n_part, out);
    SR_CHECK_LAUNCH();
    return 0;
}
int launch_quant_f8(hipStream_t s, bf16_t* W_tiled, int N, int K, unsigned char* W8, float* scale) {
    if (N % 16 || K % 64) return -22;
    hipLaunchKernelGGL(k_quant_f8, dim3(N / 16), dim3(256), 0, s, W_tiled, K, W8, scale);
    SR_CHECK_LAUNCH();
    return 0;
}
int launch_admit_rows(hipStream_t s, const AdmitArgs& a) {
    if (a.n <= 0) return 0;
    hipLaunchKernelGGL(k_admit_rows, dim3(a.n), dim3(256), 0, s, a);
    SR_CHECK_LAUNCH();
    return 0;
}
int launch_step(hipStream_t s, const StepArgs& a) {
    if (a.B <= 0) return 0;
    hipLaunchKernelGGL(k_step, dim3(a.B), dim3(256), 0, s, a);
    SR_CHECK_LAUNCH();
    return 0;
}
int launch_synth_fill(hipStream_t s, bf16_t* out, long long n, uint32_t key, float base, float scale) {
    if (n <= 0) return 0;
    long long blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_synth_fill, dim3((unsigned)blocks), dim3(256), 0, s, out, n, key, base, scale);
    SR_CHECK_LAUNCH();
    return 0;
}
int launch_fill_zero(hipStream_t s, void* p, size_t bytes) { return (int)hipMemsetAsync(p, 0, bytes, s); }

int launch_quant_mx_act(hipStream_t s, const bf16_t* x, int ldx, int M, int K, unsigned char* q, unsigned char* scales, int rows_pad) {
    if (M <= 0) return 0;
    if (K % 128 != 0 || ldx % 8 != 0 || rows_pad < M) return -22;
    const long long n = (long long)M * (K / 32);
    hipLaunchKernelGGL(k_quant_mx_act, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, ldx, M, K, q, scales, rows_pad);
    SR_CHECK_LAUNCH();
    return 0;
}
