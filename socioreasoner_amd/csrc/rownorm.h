// The decode-sized RMSNorm of ONE batch row by one 256-thread block (hf:65-79; H <= 2048, H % 8 == 0): the body of k_rmsnorm_row
// (elementwise.hip), one block per row -- plus the linear buffer descriptor the raw_buffer_* builtins want (sam_f32.hip).
// (Round 5 also ran this body INSIDE the decode GEMV launches -- as their last-arriving "tail" blocks or their first "head" blocks, with
// write-through stores, arrival tickets and sc1 loads -- bit-identical and 1.4 / 3.3 % slower per step than the launches: that code left the
// library in round 6, tools/experiments/gemv_tail_head_rmsnorm.patch restores it.)
#pragma once
#include "kernels.h"

// descriptor of a linear buffer for raw_buffer_* (gfx950 word 3 = 0x00020000: 32-bit data format, no swizzle); wave-uniform inputs only
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sr_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* p = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// row `row` of `rows`: v = xin (+ pending float32 slabs, written back to x), out = w * r(v * rsqrt(mean v^2 + eps)).  wsum: 4 floats of LDS.
__device__ __forceinline__ void rmsnorm_row_body(bf16_t* x, const bf16_t* xin, const float* part, int ksplit, const bf16_t* w,
                                                 bf16_t* out, int rows, int row, int H, float eps, int out_tiled, float* wsum) {
    const int c = threadIdx.x, nch = H / 8;
    const bool on = c < nch;
    uint4 u = uint4{0, 0, 0, 0}, wu = uint4{0, 0, 0, 0};
    float4 p0[4], p1[4];
    if (on) {
        u = *reinterpret_cast<const uint4*>(xin + (size_t)row * H + c * 8);
        wu = *reinterpret_cast<const uint4*>(w + c * 8);
        if (part) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                if (ks < ksplit) {
                    const float4* pp = reinterpret_cast<const float4*>(part + ((size_t)ks * rows + row) * H + c * 8);
                    p0[ks] = pp[0];
                    p1[ks] = pp[1];
                }
        }
    }
    float v[8] = {lo16(u.x), hi16(u.x), lo16(u.y), hi16(u.y), lo16(u.z), hi16(u.z), lo16(u.w), hi16(u.w)};
    if (on && part) {
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            if (ks < ksplit) {
                a[0] += p0[ks].x; a[1] += p0[ks].y; a[2] += p0[ks].z; a[3] += p0[ks].w;
                a[4] += p1[ks].x; a[5] += p1[ks].y; a[6] += p1[ks].z; a[7] += p1[ks].w;
            }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = rbf(v[e] + rbf(a[e]));
        *reinterpret_cast<uint4*>(x + (size_t)row * H + c * 8) = uint4{pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
    }
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ss += v[e] * v[e];
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float rs = 1.0f / sqrtf((wsum[0] + wsum[1] + wsum[2] + wsum[3]) / (float)H + eps);
    if (on) {
        const float wv[8] = {lo16(wu.x), hi16(wu.x), lo16(wu.y), hi16(wu.y), lo16(wu.z), hi16(wu.z), lo16(wu.w), hi16(wu.w)};
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = wv[e] * rbf(v[e] * rs);
        // fragment order: the 8 consecutive k of one row stay contiguous (16 B), see tiled_offset in common.h
        const size_t di = out_tiled ? tiled_offset((size_t)row, (size_t)c * 8, (size_t)H) : (size_t)row * H + c * 8;
        *reinterpret_cast<uint4*>(out + di) = uint4{pack2(o[0], o[1]), pack2(o[2], o[3]), pack2(o[4], o[5]), pack2(o[6], o[7])};
    }
}
