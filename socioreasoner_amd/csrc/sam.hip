// HBM-bound kernels of the SAM2 (Hiera-L) image path behind seg_infer (SURVEY.md "next" row N1, reference call site
// /root/reference/roll/distributed/strategy/seg_strategy.py:47-60; arithmetic: transformers/models/sam2/modeling_sam2.py, "hf:").
// The matrix work of that network runs on the MFMA kernels of gemm.hip / gemm256.hip and on the attention kernels of attention.hip;
// what is left are coalesced passes over bf16 token matrices [rows][ld] (channel-last, rows padded to whole 64-column k-tiles with
// zeros so that the next GEMM can consume them as they are), one rounding to bf16 per HF op boundary.
#include "kernels.h"
#include <math.h>

namespace {

// element access of the two storage modes: bf16 (bit patterns, one rounding per HF op boundary) and float32 (the reference's own
// precision for SAM2: /root/reference/roll/models/model_providers.py:540-548 builds the predictor in float32, no autocast)
__device__ __forceinline__ float ldf(const bf16_t* p) { return bf2f(*p); }
__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ void stf(bf16_t* p, float v) { *p = f2bf(v); }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }

// ---- predictor pre-processing: uint8 HWC [h][w][3] -> bf16 CHW [3][S][S]: /255, bilinear resize (align_corners = False, as
// torch.nn.functional.interpolate / torchvision Resize on an upscale), (x - mean) / std
template <typename T>
__global__ __launch_bounds__(256) void k_sam_preprocess(const uint8_t* img, int h, int w, T* out, int S) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= S * S) return;
    const int y = i / S, x = i % S;
    const float sy = fmaxf(((float)y + 0.5f) * ((float)h / (float)S) - 0.5f, 0.f), sx = fmaxf(((float)x + 0.5f) * ((float)w / (float)S) - 0.5f, 0.f);
    const int y0 = min((int)sy, h - 1), x0 = min((int)sx, w - 1), y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float fy = sy - (float)y0, fx = sx - (float)x0;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float p00 = img[((size_t)y0 * w + x0) * 3 + c] / 255.0f, p01 = img[((size_t)y0 * w + x1) * 3 + c] / 255.0f;
        const float p10 = img[((size_t)y1 * w + x0) * 3 + c] / 255.0f, p11 = img[((size_t)y1 * w + x1) * 3 + c] / 255.0f;
        const float top = p00 * (1.f - fx) + p01 * fx, bot = p10 * (1.f - fx) + p11 * fx;
        stf(out + (size_t)c * S * S + i, (top * (1.f - fy) + bot * fy - mean[c]) / stdv[c]);
    }
}

// ---- patch embedding as a GEMM operand: k x k / stride / pad convolution windows of a CHW image -> rows [ (ty, tx) ][ c*k*k + ky*k + kx ],
// zero beyond the image and in the pad columns; optional destination row map (window order)
template <typename E>
__global__ __launch_bounds__(256) void k_im2col(const E* chw, int S, int k, int stride, int pad, int T, E* out, int ld, const int* rowmap) {
    const int tok = blockIdx.x;
    const int ty = tok / T, tx = tok % T, kk = k * k;
    E* o = out + (size_t)(rowmap ? rowmap[tok] : tok) * ld;
    for (int j = threadIdx.x; j < ld; j += 256) {
        E v = 0;
        if (j < 3 * kk) {
            const int c = j / kk, ky = (j % kk) / k, kx = j % k;
            const int y = ty * stride - pad + ky, x = tx * stride - pad + kx;
            if (y >= 0 && y < S && x >= 0 && x < S) v = chw[(size_t)c * S * S + (size_t)y * S + x];
        }
        o[j] = v;
    }
}

// ---- LayerNorm with bias over C channels of every row (float32 statistics, one rounding), pad columns [C, ld_out) written as zeros.
// One wave per row (C <= 1152).
template <typename E>
__global__ __launch_bounds__(256) void k_layernorm(const E* x, int ldx, const E* w, const E* b, E* out, int ldo, int rows, int C, float eps) {
    constexpr int MAXN = 18;                                   // 64 channels per step: C, ldo <= 1152
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const E* xr = x + (size_t)row * ldx;
    float v[MAXN];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXN; ++i) {
        const int c = lane + i * 64;
        v[i] = c < C ? ldf(xr + c) : 0.f;
        s += v[i];
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXN; ++i) {
        const float d = (lane + i * 64) < C ? v[i] - mean : 0.f;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    E* o = out + (size_t)row * ldo;
#pragma unroll
    for (int i = 0; i < MAXN; ++i) {
        const int c = lane + i * 64;
        if (c < C) stf(o + c, (v[i] - mean) * rstd * ldf(w + c) + ldf(b + c));
        else if (c < ldo) o[c] = 0;
    }
}

// The same LayerNorm, 16 bytes per lane: LP lanes share a row (64 / LP rows per wave), each lane holds NV runs of 8 channels.  Same formula
// and rounding point as k_layernorm; the float32 statistics are summed in another order.  (k_layernorm moved 2 bytes per lane: 1.1 TB/s on
// the 524 288 x 144 rows of Hiera-L's first stage.)
template <int LP, int NV>
__global__ __launch_bounds__(256) void k_layernorm_v(const bf16_t* x, int ldx, const bf16_t* w, const bf16_t* b, bf16_t* out, int ldo, int rows, int C, float eps) {
    constexpr int RW = 64 / LP;
    const int lane = threadIdx.x & 63, l = lane % LP;
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RW + lane / LP;
    const bool live = row < rows;
    const bf16_t* xr = x + (size_t)(live ? row : 0) * ldx;
    float v[NV][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c0 = (l + i * LP) * 8;
        uint4 u = uint4{0, 0, 0, 0};
        if (live && c0 < ldx && c0 < C) u = *reinterpret_cast<const uint4*>(xr + c0);
        const float t[8] = {lo16(u.x), hi16(u.x), lo16(u.y), hi16(u.y), lo16(u.z), hi16(u.z), lo16(u.w), hi16(u.w)};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[i][e] = c0 + e < C ? t[e] : 0.f;
            s += v[i][e];
        }
    }
#pragma unroll
    for (int o = 1; o < LP; o <<= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = (l + i * LP) * 8 + e < C ? v[i][e] - mean : 0.f;
            q += d * d;
        }
#pragma unroll
    for (int o = 1; o < LP; o <<= 1) q += __shfl_xor(q, o, 64);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
    if (!live) return;
    bf16_t* orow = out + (size_t)row * ldo;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c0 = (l + i * LP) * 8;
        if (c0 >= ldo) continue;
        float o[8];
        if (c0 < C) {
            const uint4 wu = *reinterpret_cast<const uint4*>(w + c0), bu = *reinterpret_cast<const uint4*>(b + c0);      // (w, b padded to 8 by the caller's layout: C % 8 == 0)
            const float wf[8] = {lo16(wu.x), hi16(wu.x), lo16(wu.y), hi16(wu.y), lo16(wu.z), hi16(wu.z), lo16(wu.w), hi16(wu.w)};
            const float bf[8] = {lo16(bu.x), hi16(bu.x), lo16(bu.y), hi16(bu.y), lo16(bu.z), hi16(bu.z), lo16(bu.w), hi16(bu.w)};
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * wf[e] + bf[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = 0.f;
        }
        *reinterpret_cast<uint4*>(orow + c0) = uint4{pack2(o[0], o[1]), pack2(o[2], o[3]), pack2(o[4], o[5]), pack2(o[6], o[7])};
    }
}

// ---- 2 x 2 max pooling of tokens that are stored window by window ([n_win][ws * ws] rows -> [n_win][(ws/2)^2] rows; hf:290-298, 337-341)
template <typename E>
__global__ __launch_bounds__(256) void k_maxpool_win(const E* in, int ld_in, int C, int ws, E* out, int ld_out) {
    const int h2 = ws / 2, per = h2 * h2;
    const int orow = blockIdx.x, win = orow / per, py = (orow % per) / h2, px = orow % h2;
    const E* base = in + ((size_t)win * ws * ws + (size_t)(2 * py) * ws + 2 * px) * ld_in;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float a = fmaxf(fmaxf(ldf(base + c), ldf(base + ld_in + c)), fmaxf(ldf(base + (size_t)ws * ld_in + c), ldf(base + (size_t)(ws + 1) * ld_in + c)));
        stf(out + (size_t)orow * ld_out + c, a);
    }
}

// ---- elementwise, row-major [rows][ld] with C live columns: mode 0 out = a + b, 1 out = a + vec (row vector), 2 relu(a), 3 gelu(a) (erf form),
// 4 silu(a) * b (the LM's / ViT's gated MLP with the product's own silu_f: the float32 verification path, tests/f32_path.py)
template <typename E>
__global__ __launch_bounds__(256) void k_ew(const E* a, int lda, const E* b, int ldb, E* out, int ldo, int rows, int C, int mode) {
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= (long long)rows * C) return;
    const int r = (int)(i / C), c = (int)(i % C);
    const float x = ldf(a + (size_t)r * lda + c);
    float y;
    if (mode == 0) y = x + ldf(b + (size_t)r * ldb + c);
    else if (mode == 1) y = x + ldf(b + c);
    else if (mode == 2) y = fmaxf(x, 0.f);
    else if (mode == 4) y = silu_f(x) * ldf(b + (size_t)r * ldb + c);
    else y = gelu_f(x);
    stf(out + (size_t)r * ldo + c, y);
}

// ---- bf16 matrix transpose: out[c][r] = in[r][c]  (rows x cols), 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void k_transpose(const bf16_t* in, int ld_in, int rows, int cols, bf16_t* out, int ld_out) {
    __shared__ bf16_t t[32][33];
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        t[j][tx] = (r < rows && c < cols) ? in[(size_t)r * ld_in + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (c < cols && r < rows) out[(size_t)c * ld_out + r] = t[tx][j];
    }
}

// The same transpose with 16-byte accesses on both sides: 64 x 64 tiles, rows padded to 66 elements in LDS (a column walk then steps 33
// banks).  Needs cols, both leading dimensions and both base addresses in units of 8 elements; the row tail is stored element by element.
__global__ __launch_bounds__(256) void k_transpose_v(const bf16_t* in, int ld_in, int rows, int cols, bf16_t* out, int ld_out) {
    __shared__ __attribute__((aligned(16))) bf16_t t[64 * 66];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64, v = threadIdx.x & 7, q = threadIdx.x >> 3;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = r0 + q + 32 * j, c = c0 + v * 8;
        uint4 u = uint4{0, 0, 0, 0};
        if (r < rows && c < cols) u = *reinterpret_cast<const uint4*>(in + (size_t)r * ld_in + c);
        uint32_t* d = reinterpret_cast<uint32_t*>(t + (q + 32 * j) * 66 + v * 8);          // 4-byte aligned: 66 and 8 are even
        d[0] = u.x; d[1] = u.y; d[2] = u.z; d[3] = u.w;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int cl = q + 32 * j, c = c0 + cl, rl = v * 8, r = r0 + rl;
        if (c >= cols || r >= rows) continue;
        bf16_t e[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k] = t[(rl + k) * 66 + cl];
        bf16_t* o = out + (size_t)c * ld_out + r;
        if (r + 8 <= rows) {
            *reinterpret_cast<uint4*>(o) = uint4{(uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16),
                                                 (uint32_t)e[4] | ((uint32_t)e[5] << 16), (uint32_t)e[6] | ((uint32_t)e[7] << 16)};
        } else {
            for (int k = 0; k < 8 && r + k < rows; ++k) o[k] = e[k];
        }
    }
}

// ---- FPN top-down step: out[y][x] = bf16(lat[y][x] + top[y/2][x/2])  (nearest 2 x upsampling, hf:246-256)
template <typename E>
__global__ __launch_bounds__(256) void k_upsample2x_add(const E* lat, const E* top, E* out, int H2, int C, int ld) {
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= (long long)H2 * H2 * C) return;
    const int c = (int)(i % C), t = (int)(i / C), y = t / H2, x = t % H2;
    stf(out + (size_t)t * ld + c, ldf(lat + (size_t)t * ld + c) + ldf(top + ((size_t)(y / 2) * (H2 / 2) + x / 2) * ld + c));
}

// ---- transposed 2 x 2 / stride 2 convolution, second half: the GEMM wrote [H*W][co*4 + dy*2 + dx]; scatter to [(2y+dy)*(2W) + 2x+dx][co]
// and add the high-resolution feature (hf:1215-1221): out = bf16(conv + feat)
template <typename E>
__global__ __launch_bounds__(256) void k_pixel_shuffle_add(const E* g, int ldg, const E* feat, int ldfe, E* out, int ldo, int W, int Co) {
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    const long long total = 4ll * W * W * Co;
    if (i >= total) return;
    const int co = (int)(i % Co), t = (int)(i / Co), Y = t / (2 * W), X = t % (2 * W);
    const int y = Y >> 1, dy = Y & 1, x = X >> 1, dx = X & 1;
    stf(out + (size_t)t * ldo + co, ldf(g + ((size_t)y * W + x) * ldg + co * 4 + dy * 2 + dx) + ldf(feat + (size_t)t * ldfe + co));
}

// ---- predictor post-processing (one launch per object): bilinear resize (align_corners = False) of the BEST of n low-resolution mask
// logit maps (float32 [m*m][ld], column i = mask i; best = first arg-max of score[0..n)) to h x w, threshold at 0, OR into the
// running object union (seg_strategy.py:57-60); optionally the resized logits of all n masks (tests)
__global__ __launch_bounds__(256) void k_mask_resize_or(const float* low, int ld, int col0, int n, int m, const float* score, uint8_t* acc, float* logits_out, int h, int w) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= h * w) return;
    int best = 0;
    for (int k = 1; k < n; ++k)
        if (score[col0 + k] > score[col0 + best]) best = k;
    const int y = i / w, x = i % w;
    const float sy = fmaxf(((float)y + 0.5f) * ((float)m / (float)h) - 0.5f, 0.f), sx = fmaxf(((float)x + 0.5f) * ((float)m / (float)w) - 0.5f, 0.f);
    const int y0 = min((int)sy, m - 1), x0 = min((int)sx, m - 1), y1 = min(y0 + 1, m - 1), x1 = min(x0 + 1, m - 1);
    const float fy = sy - (float)y0, fx = sx - (float)x0;
    for (int k = 0; k < n; ++k) {
        if (!logits_out && k != best) continue;
        const int c = col0 + k;
        const float top = low[((size_t)y0 * m + x0) * ld + c] * (1.f - fx) + low[((size_t)y0 * m + x1) * ld + c] * fx;
        const float bot = low[((size_t)y1 * m + x0) * ld + c] * (1.f - fx) + low[((size_t)y1 * m + x1) * ld + c] * fx;
        const float v = top * (1.f - fy) + bot * fy;
        if (logits_out) logits_out[(size_t)k * h * w + i] = v;
        if (k == best && v > 0.f) acc[i] = 1;
    }
}

}  // namespace

#define LAUNCH_OK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; return 0; } while (0)

int launch_sam_preprocess(hipStream_t s, const uint8_t* img, int h, int w, bf16_t* out, int S) {
    hipLaunchKernelGGL(k_sam_preprocess<bf16_t>, dim3(cdiv(S * S, 256)), dim3(256), 0, s, img, h, w, out, S);
    LAUNCH_OK();
}
int launch_im2col(hipStream_t s, const bf16_t* chw, int S, int k, int stride, int pad, bf16_t* out, int ld, const int* rowmap) {
    const int T = (S + 2 * pad - k) / stride + 1;
    if (ld < 3 * k * k) return -22;
    hipLaunchKernelGGL(k_im2col<bf16_t>, dim3(T * T), dim3(256), 0, s, chw, S, k, stride, pad, T, out, ld, rowmap);
    LAUNCH_OK();
}
// ---- the float32 storage mode of the same passes (sam_f32.hip holds its GEMM and attention)
int launch_sam_preprocess_f32(hipStream_t s, const uint8_t* img, int h, int w, float* out, int S) {
    hipLaunchKernelGGL(k_sam_preprocess<float>, dim3(cdiv(S * S, 256)), dim3(256), 0, s, img, h, w, out, S);
    LAUNCH_OK();
}
int launch_im2col_f32(hipStream_t s, const float* chw, int S, int k, int stride, int pad, float* out, int ld, const int* rowmap) {
    const int T = (S + 2 * pad - k) / stride + 1;
    if (ld < 3 * k * k) return -22;
    hipLaunchKernelGGL(k_im2col<float>, dim3(T * T), dim3(256), 0, s, chw, S, k, stride, pad, T, out, ld, rowmap);
    LAUNCH_OK();
}
int launch_layernorm_f32(hipStream_t s, const float* x, int ldx, const float* w, const float* b, float* out, int ldo, int rows, int C, float eps) {
    if (rows <= 0) return 0;
    if (C > 1152 || ldo > 1152) return -22;
    hipLaunchKernelGGL(k_layernorm<float>, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, ldx, w, b, out, ldo, rows, C, eps);
    LAUNCH_OK();
}
int launch_maxpool_win_f32(hipStream_t s, const float* in, int ld_in, int C, int n_win, int ws, float* out, int ld_out) {
    if (ws % 2) return -22;
    hipLaunchKernelGGL(k_maxpool_win<float>, dim3(n_win * (ws / 2) * (ws / 2)), dim3(256), 0, s, in, ld_in, C, ws, out, ld_out);
    LAUNCH_OK();
}
int launch_ew_f32(hipStream_t s, const float* a, int lda, const float* b, int ldb, float* out, int ldo, int rows, int C, int mode) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(k_ew<float>, dim3((unsigned)(((long long)rows * C + 255) / 256)), dim3(256), 0, s, a, lda, b, ldb, out, ldo, rows, C, mode);
    LAUNCH_OK();
}
int launch_upsample2x_add_f32(hipStream_t s, const float* lat, const float* top, float* out, int H2, int C, int ld) {
    hipLaunchKernelGGL(k_upsample2x_add<float>, dim3((unsigned)(((long long)H2 * H2 * C + 255) / 256)), dim3(256), 0, s, lat, top, out, H2, C, ld);
    LAUNCH_OK();
}
int launch_pixel_shuffle_add_f32(hipStream_t s, const float* g, int ldg, const float* feat, int ldf, float* out, int ldo, int W, int Co) {
    hipLaunchKernelGGL(k_pixel_shuffle_add<float>, dim3((unsigned)((4ll * W * W * Co + 255) / 256)), dim3(256), 0, s, g, ldg, feat, ldf, out, ldo, W, Co);
    LAUNCH_OK();
}
int launch_layernorm(hipStream_t s, const bf16_t* x, int ldx, const bf16_t* w, const bf16_t* b, bf16_t* out, int ldo, int rows, int C, float eps) {
    if (rows <= 0) return 0;
    if (C > 1152 || ldo > 1152) return -22;
    const bool vec = C % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0 && ((uintptr_t)x | (uintptr_t)out | (uintptr_t)w | (uintptr_t)b) % 16 == 0 && ldo <= 1536;
    if (vec) {
        const int span = ldo > C ? ldo : C;                 // columns a row's lanes must cover (pad columns are written as zeros)
        if (span <= 256) hipLaunchKernelGGL((k_layernorm_v<32, 1>), dim3(cdiv(rows, 8)), dim3(256), 0, s, x, ldx, w, b, out, ldo, rows, C, eps);
        else if (span <= 512) hipLaunchKernelGGL((k_layernorm_v<64, 1>), dim3(cdiv(rows, 4)), dim3(256), 0, s, x, ldx, w, b, out, ldo, rows, C, eps);
        else if (span <= 1024) hipLaunchKernelGGL((k_layernorm_v<64, 2>), dim3(cdiv(rows, 4)), dim3(256), 0, s, x, ldx, w, b, out, ldo, rows, C, eps);
        else hipLaunchKernelGGL((k_layernorm_v<64, 3>), dim3(cdiv(rows, 4)), dim3(256), 0, s, x, ldx, w, b, out, ldo, rows, C, eps);
        LAUNCH_OK();
    }
    hipLaunchKernelGGL(k_layernorm<bf16_t>, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, ldx, w, b, out, ldo, rows, C, eps);
    LAUNCH_OK();
}
int launch_maxpool_win(hipStream_t s, const bf16_t* in, int ld_in, int C, int n_win, int ws, bf16_t* out, int ld_out) {
    if (ws % 2) return -22;
    hipLaunchKernelGGL(k_maxpool_win<bf16_t>, dim3(n_win * (ws / 2) * (ws / 2)), dim3(256), 0, s, in, ld_in, C, ws, out, ld_out);
    LAUNCH_OK();
}
int launch_ew(hipStream_t s, const bf16_t* a, int lda, const bf16_t* b, int ldb, bf16_t* out, int ldo, int rows, int C, int mode) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(k_ew<bf16_t>, dim3((unsigned)(((long long)rows * C + 255) / 256)), dim3(256), 0, s, a, lda, b, ldb, out, ldo, rows, C, mode);
    LAUNCH_OK();
}
int launch_transpose(hipStream_t s, const bf16_t* in, int ld_in, int rows, int cols, bf16_t* out, int ld_out) {
    if (cols % 8 == 0 && ld_in % 8 == 0 && ld_out % 8 == 0 && ((uintptr_t)in | (uintptr_t)out) % 16 == 0) {
        hipLaunchKernelGGL(k_transpose_v, dim3(cdiv(rows, 64), cdiv(cols, 64)), dim3(256), 0, s, in, ld_in, rows, cols, out, ld_out);
        LAUNCH_OK();
    }
    hipLaunchKernelGGL(k_transpose, dim3(cdiv(rows, 32), cdiv(cols, 32)), dim3(256), 0, s, in, ld_in, rows, cols, out, ld_out);
    LAUNCH_OK();
}
int launch_upsample2x_add(hipStream_t s, const bf16_t* lat, const bf16_t* top, bf16_t* out, int H2, int C, int ld) {
    hipLaunchKernelGGL(k_upsample2x_add<bf16_t>, dim3((unsigned)(((long long)H2 * H2 * C + 255) / 256)), dim3(256), 0, s, lat, top, out, H2, C, ld);
    LAUNCH_OK();
}
int launch_pixel_shuffle_add(hipStream_t s, const bf16_t* g, int ldg, const bf16_t* feat, int ldf, bf16_t* out, int ldo, int W, int Co) {
    hipLaunchKernelGGL(k_pixel_shuffle_add<bf16_t>, dim3((unsigned)((4ll * W * W * Co + 255) / 256)), dim3(256), 0, s, g, ldg, feat, ldf, out, ldo, W, Co);
    LAUNCH_OK();
}
int launch_mask_resize_or(hipStream_t s, const float* low, int ld, int col0, int n, int m, const float* score, uint8_t* acc, float* logits_out, int h, int w) {
    hipLaunchKernelGGL(k_mask_resize_or, dim3(cdiv(h * w, 256)), dim3(256), 0, s, low, ld, col0, n, m, score, acc, logits_out, h, w);
    LAUNCH_OK();
}
