// Float32 row operations of the VERIFICATION path (include/socior.h, "float32 VERIFICATION path"; tests/f32_path.py): the pieces of the Qwen2.5-VL
// forward that the float32 GEMM / attention of sam_f32.hip do not cover, without any bf16 rounding, so that the arithmetic this library implements can be
// held to 1e-3 on logits against HF run in float32.  HBM-bound row passes; speed is not a concern here.
#include "kernels.h"
#include <math.h>

namespace {

// out[row] = w * x[row] * rsqrt(mean(x[row]^2) + eps)      (Qwen2RMSNorm / Qwen2_5_VLRMSNorm, hf:65-79: float32 variance, then the weight)
__global__ __launch_bounds__(256) void k_rmsnorm_f32(const float* x, int ldx, const float* w, float* out, int ldo, int C, float eps) {
    __shared__ float wsum[4];
    const float* xr = x + (size_t)blockIdx.x * ldx;
    float ss = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) ss += xr[c] * xr[c];
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float rs = 1.0f / sqrtf((wsum[0] + wsum[1] + wsum[2] + wsum[3]) / (float)C + eps);
    float* o = out + (size_t)blockIdx.x * ldo;
    for (int c = threadIdx.x; c < C; c += 256) o[c] = w[c] * (xr[c] * rs);
}

// cos | sin of pos[i] * inv_freq[f]: the float32 values the engine's bf16 tables (k_rope_table, elementwise.hip) are rounded from -- same device cosf / sinf,
// same float32 product
__global__ __launch_bounds__(256) void k_rope_table_f32(const float* inv_freq, int n_freq, const int* pos, int n_pos, float* cos_t, float* sin_t) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pos * n_freq) return;
    const float ang = (float)pos[i / n_freq] * inv_freq[i % n_freq];
    cos_t[i] = cosf(ang);
    sin_t[i] = sinf(ang);
}

// x <- x * cos + rotate_half(x) * sin for n_heads heads of head_dim columns each (hf:100-123 apply_rotary_pos_emb_vision, hf:557-599 multimodal rotary after
// the section select): element d < hd / 2 pairs with d + hd / 2; cos / sin hold head_dim values per row
__global__ __launch_bounds__(256) void k_rope_f32(float* x, int ld, const float* cos_t, const float* sin_t, int ldc, int rows, int n_heads, int hd) {
    const int half = hd / 2;
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= (long long)rows * n_heads * half) return;
    const int d = (int)(i % half), h = (int)((i / half) % n_heads), r = (int)(i / ((long long)half * n_heads));
    float* p = x + (size_t)r * ld + h * hd;
    const float a = p[d], b = p[d + half];
    const float* c = cos_t + (size_t)r * ldc;
    const float* s = sin_t + (size_t)r * ldc;
    p[d] = a * c[d] - b * s[d];
    p[d + half] = b * c[d + half] + a * s[d + half];
}

}  // namespace

int launch_rmsnorm_f32(hipStream_t s, const float* x, int ldx, const float* w, float* out, int ldo, int rows, int C, float eps) {
    if (rows <= 0) return 0;
    if (C <= 0 || C > 8192) return -22;
    hipLaunchKernelGGL(k_rmsnorm_f32, dim3(rows), dim3(256), 0, s, x, ldx, w, out, ldo, C, eps);
    SR_CHECK_LAUNCH();
    return 0;
}
int launch_rope_table_f32(hipStream_t s, const float* inv_freq, int n_freq, const int* pos, int n_pos, float* cos_t, float* sin_t) {
    if (n_pos <= 0 || n_freq <= 0) return 0;
    hipLaunchKernelGGL(k_rope_table_f32, dim3(cdiv(n_pos * n_freq, 256)), dim3(256), 0, s, inv_freq, n_freq, pos, n_pos, cos_t, sin_t);
    SR_CHECK_LAUNCH();
    return 0;
}
int launch_rope_f32(hipStream_t s, float* x, int ld, const float* cos_t, const float* sin_t, int ldc, int rows, int n_heads, int head_dim) {
    if (rows <= 0 || n_heads <= 0) return 0;
    if (head_dim % 2 || ldc < head_dim) return -22;
    const long long n = (long long)rows * n_heads * (head_dim / 2);
    hipLaunchKernelGGL(k_rope_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, ld, cos_t, sin_t, ldc, rows, n_heads, head_dim);
    SR_CHECK_LAUNCH();
    return 0;
}
