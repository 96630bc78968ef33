// The decode-sized RMSNorm of ONE batch row by one 256-thread block (hf:65-79; H <= 2048, H % 8 == 0), shared by
//   * k_rmsnorm_row (elementwise.hip): a launch of its own, one block per row, and
//   * the TAIL of the decode GEMVs that produce a layer's residual stream (gemv.hip, round 5): the last blocks of the o_proj / down-projection
//     launch to arrive do the residual add + RMSNorm of the 32 rows inside that launch, so the two RMSNorm launches of a batch > 4 decode
//     layer (9.6 % of all kernel time at 32 rows) disappear.  Same loads, same float32 association, same reduction order:
//     the two paths give the same bits (tests/test_gpu_round5.py).
// In-launch visibility (MI355X guide, Guideline 16 / the split-K counter recipe): the producing blocks store their slab / residual-stream
// pieces WRITE-THROUGH (sc1), every wave drains (`s_waitcnt vmcnt(0)`), the block takes an arrival ticket with a relaxed agent-scope
// fetch_add; a tail block polls the counter relaxed until every block has arrived and then reads what OTHER blocks wrote with sc1 loads
// (L1 bypass; the XCD's L2 never holds a dirty or stale copy of a line that was only ever written through).
#pragma once
#include "kernels.h"

typedef __attribute__((ext_vector_type(4))) unsigned int sr_u32x4;

// descriptor of a linear buffer for raw_buffer_* (gfx950 word 3 = 0x00020000: 32-bit data format, no swizzle); wave-uniform inputs only
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sr_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* p = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
enum { SR_AUX_SC1 = 16 };       // gfx940+ cache-policy bit of the buffer builtins: sc1 (write-through store / L1-bypassing load)
__device__ __forceinline__ sr_u32x4 ld16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, SR_AUX_SC1); }
__device__ __forceinline__ void st16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, sr_u32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)byte_off, 0, SR_AUX_SC1); }
__device__ __forceinline__ void st8_sc1(void* p, uint2 v) {     // global_store_dwordx2 ... sc1
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), ((unsigned long long)v.y << 32) | v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// row `row` of `rows`: v = xin (+ pending float32 slabs, written back to x), out = w * r(v * rsqrt(mean v^2 + eps)).
// COH: `part` (and xin when coh_x) were written by other blocks of this launch -- sc1 loads.  wsum: 4 floats of LDS.
template <bool COH>
__device__ __forceinline__ void rmsnorm_row_body(bf16_t* x, const bf16_t* xin, bool coh_x, const float* part, int ksplit, const bf16_t* w,
                                                 bf16_t* out, int rows, int row, int H, float eps, int out_tiled, float* wsum, bool wt_out = false) {
    const int c = threadIdx.x, nch = H / 8;
    const bool on = c < nch;
    uint4 u = uint4{0, 0, 0, 0}, wu = uint4{0, 0, 0, 0};
    float4 p0[4], p1[4];
    if (on) {
        if (COH && coh_x) {
            const sr_u32x4 t = ld16_sc1(sr_rsrc(xin, (unsigned)rows * H * 2), ((unsigned)row * H + c * 8) * 2);
            u = uint4{t[0], t[1], t[2], t[3]};
        } else u = *reinterpret_cast<const uint4*>(xin + (size_t)row * H + c * 8);
        wu = *reinterpret_cast<const uint4*>(w + c * 8);
        if (part) {
            if constexpr (COH) {
                const __amdgpu_buffer_rsrc_t r = sr_rsrc(part, (unsigned)ksplit * rows * H * 4);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    if (ks < ksplit) {
                        const unsigned off = (((unsigned)ks * rows + row) * H + c * 8) * 4;
                        p0[ks] = __builtin_bit_cast(float4, ld16_sc1(r, off));
                        p1[ks] = __builtin_bit_cast(float4, ld16_sc1(r, off + 16));
                    }
            } else {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    if (ks < ksplit) {
                        const float4* pp = reinterpret_cast<const float4*>(part + ((size_t)ks * rows + row) * H + c * 8);
                        p0[ks] = pp[0];
                        p1[ks] = pp[1];
                    }
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
        const uint4 ov = uint4{pack2(o[0], o[1]), pack2(o[2], o[3]), pack2(o[4], o[5]), pack2(o[6], o[7])};
        // wt_out: other blocks of THIS launch read the row (GemvHead): write-through.  (Fragment order pads the rows to a multiple of 16.)
        if (wt_out) st16_sc1(sr_rsrc(out, (unsigned)((rows + 15) / 16 * 16) * H * 2), (unsigned)di * 2, sr_u32x4{ov.x, ov.y, ov.z, ov.w});
        else *reinterpret_cast<uint4*>(out + di) = ov;
    }
}

// End of a 256-thread GEMV block whose launch carries a tail (t.counter != null).  EVERY thread of EVERY block of the launch calls it after its
// last (sc1) output store.  n_rows <= gridDim.x * gridDim.y (the launcher checks).  smem4: >= 8 floats of LDS nobody else touches any more.
// part / ksplit: the float32 slabs of a PARTIAL launch (null / 0 for RESID, whose rows are complete in t.x).
__device__ __forceinline__ void gemv_tail_rmsnorm(const GemvTail& t, const float* part, int ksplit, int n_rows, int H, float* smem4) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's write-through stores have left
    __syncthreads();
    unsigned* sh = reinterpret_cast<unsigned*>(smem4);
    const unsigned total = gridDim.x * gridDim.y;
    if (threadIdx.x == 0) sh[4] = __hip_atomic_fetch_add(t.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned ticket = sh[4];
    const unsigned first = total - (unsigned)n_rows;
    if (ticket < first) return;                            // (block-uniform)
    const int row = (int)(ticket - first);
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(t.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < total) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > (1u << 22)) { atomicAdd(t.timeout, 1u); break; }     // never hang the GPU: a wrong row is a failed test, a hang is a dead box
        }
    }
    __syncthreads();
    rmsnorm_row_body<true>(t.x, t.x, part == nullptr, part, ksplit, t.norm_w, t.xn, n_rows, row, H, t.eps, t.xn_tiled, smem4);
}
