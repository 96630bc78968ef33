// Research probe (not part of the product library): does launching a decode GEMV AHEAD of its producer -- weights
// prefetched into registers, then a device-side wait on the producer's completion counter -- beat plain stream order?
// Chain: per layer qkv (2560x2048) -> o (2048x2048) -> gate/up (22016x2048) -> down (2048x11008), true data dependencies,
// M = 1, fragment-ordered bf16 weights, 4-wave blocks with the K split over the waves, 16 x 1 KB loads in flight per wave.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }

struct Args {
    const bf16_t* W; int N, K;          // tiled16x64
    const bf16_t* xin; bf16_t* xout;    // [K], [N]
    const unsigned* wait_ctr; unsigned wait_target;   // 8 shards; null: no wait (plain stream order)
    unsigned* signal_ctr;               // 8 shards; null: no signal
    unsigned* err;
    float scale;
};

__global__ __launch_bounds__(256) void k_chain_gemv(Args p) {
    __shared__ __attribute__((aligned(16))) bf16_t xs[11008 + 64];
    __shared__ f32x4 red[3][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int tile = blockIdx.x;
    const int nchunks = p.K / 64, per = (nchunks + 3) / 4;
    const int c0 = min(wave * per, nchunks), cend = min(c0 + per, nchunks);
    constexpr int U = 8;
    u32x4 w[U][2];
    const bf16_t* wrow = p.W + (size_t)tile * 16 * p.K + lane * 8;
    auto fill = [&](int u, int c) {
        w[u][0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + (size_t)c * 1024));
        w[u][1] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + (size_t)c * 1024 + 512));
    };
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (c0 + u < cend) fill(u, c0 + u);
    // ---- wait for the producer (launch-ahead mode)
    if (p.wait_ctr) {
        if (tid == 0) {
            unsigned it = 0;
            for (;;) {
                unsigned s = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) s += __hip_atomic_load(p.wait_ctr + i * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (s >= p.wait_target) break;
                if (++it > 400000u) { *p.err = 1; break; }      // bounded: never hang the GPU
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
    // ---- x -> LDS (coherent 8-byte loads when the producer may still be running on another XCD)
    for (int i = tid; i < p.K / 4; i += 256) {
        unsigned long long v;
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(p.xin) + i;
        if (p.wait_ctr) v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else v = *src;
        reinterpret_cast<unsigned long long*>(xs)[i] = v;
    }
    __syncthreads();
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c = c0; c < cend; c += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (c + u < cend) {
                const bf16_t* xr = xs + (size_t)(c + u) * 64 + fg * 16;
                u32x4 x0 = u32x4{0, 0, 0, 0}, x1 = x0;
                if (fr == 0) { x0 = *reinterpret_cast<const u32x4*>(xr); x1 = *reinterpret_cast<const u32x4*>(xr + 8); }
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[u][0]), __builtin_bit_cast(bf16x8, x0), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[u][1]), __builtin_bit_cast(bf16x8, x1), acc, 0, 0, 0);
                if (c + U + u < cend) fill(u, c + U + u);
            }
        }
    }
    if (wave > 0) red[wave - 1][lane] = acc;
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { const f32x4 o = red[k][lane]; acc[0] += o[0]; acc[1] += o[1]; acc[2] += o[2]; acc[3] += o[3]; }
        if (fr == 0) {      // batch row 0: columns tile*16 + fg*4 .. +3
            const unsigned long long v = (unsigned long long)f2bf(acc[0] * p.scale) | ((unsigned long long)f2bf(acc[1] * p.scale) << 16) |
                                         ((unsigned long long)f2bf(acc[2] * p.scale) << 32) | ((unsigned long long)f2bf(acc[3] * p.scale) << 48);
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(p.xout + tile * 16 + fg * 4);
            if (p.signal_ctr) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *dst = v;
        }
        if (p.signal_ctr) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(p.signal_ctr + (blockIdx.x & 7) * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    const int H = 2048, QN = 2560, I = 11008, L = 36;
    const int Ns[4] = {QN, H, 2 * I, H}, Ks[4] = {H, H, H, I};
    std::vector<bf16_t*> W(4 * L);
    for (int l = 0; l < L; ++l)
        for (int j = 0; j < 4; ++j) {
            CK(hipMalloc(&W[l * 4 + j], (size_t)Ns[j] * Ks[j] * 2));
            CK(hipMemset(W[l * 4 + j], 0x11, (size_t)Ns[j] * Ks[j] * 2));     // tiny positive bf16 values
        }
    bf16_t* x[2];
    for (auto& p : x) { CK(hipMalloc(&p, 2 * I * 2 + 256)); CK(hipMemset(p, 0x3c, 2 * I * 2)); }
    unsigned *ctr, *err;
    const int NK = 4 * L;
    CK(hipMalloc(&ctr, (size_t)(NK + 1) * 8 * 16 * 4));
    CK(hipMalloc(&err, 4));
    hipStream_t s[2];
    CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    hipEvent_t e0, e1, ej;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    for (int mode = 0; mode < 3; ++mode) {       // 0: one stream, plain; 1: two streams + counters (launch-ahead); 2: one stream + counters (overhead check)
        float best = 1e9f;
        unsigned herr = 0;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipMemsetAsync(ctr, 0, (size_t)(NK + 1) * 8 * 16 * 4, s[0]));
            CK(hipMemsetAsync(err, 0, 4, s[0]));
            CK(hipStreamSynchronize(s[0]));
            CK(hipEventRecord(e0, s[0]));
            if (mode == 1) { CK(hipEventRecord(ej, s[0])); CK(hipStreamWaitEvent(s[1], ej, 0)); }
            for (int k = 0; k < NK; ++k) {
                const int j = k & 3;
                Args a{W[k], Ns[j], Ks[j], x[k & 1], x[(k + 1) & 1], nullptr, 0, nullptr, err, 1e-3f};
                if (mode >= 1) {
                    a.signal_ctr = ctr + (size_t)(k + 1) * 8 * 16;
                    if (k > 0) { a.wait_ctr = ctr + (size_t)k * 8 * 16; a.wait_target = Ns[(k - 1) & 3] / 16; }
                }
                hipStream_t q = mode == 1 ? s[k & 1] : s[0];
                hipLaunchKernelGGL(k_chain_gemv, dim3(Ns[j] / 16), dim3(256), 0, q, a);
            }
            if (mode == 1) { CK(hipEventRecord(ej, s[1])); CK(hipStreamWaitEvent(s[0], ej, 0)); }
            CK(hipEventRecord(e1, s[0]));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
            CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        }
        printf("mode %d: %.1f us per layer (4 GEMVs, 154 MB)  [%d launches %.3f ms]  timeout_flag=%u\n", mode, best * 1e3f / L, NK, best, herr);
    }
    return 0;
}
