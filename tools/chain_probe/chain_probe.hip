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


// ---------------------------------------------------------------------------------------------------------------------------
// mode 3: ONE persistent launch, one 4-wave block per CU.  Every block walks the same list of phases; a phase's work units
// (16-row tile x K range) are dealt round-robin over the blocks.  Before a block waits for the previous phase to complete it has
// already issued the first 16 loads of its own next unit, so the weight stream keeps running across the seam.
struct Phase { const bf16_t* W; int N, K, ksplit, n_units, offset; };     // units = (N/16) * ksplit
struct PArgs {
    const Phase* phases; int n_phases;
    bf16_t* y;              // [n_phases][22016] bf16 outputs (ksplit 1)
    float* slabs;           // [n_phases][2][2048] float partial outputs (ksplit 2)
    unsigned* ctr;          // [n_phases][8 shards x 16 words]
    unsigned* err;
    float scale;
};

__global__ __launch_bounds__(256) void k_chain_persistent(PArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);                              // [11008 + 64]
    f32x4* red = reinterpret_cast<f32x4*>(smem + (11008 + 64) * 2);            // [3][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int nb = gridDim.x, bid = blockIdx.x;
    constexpr int U = 8;
    u32x4 w[U][2];
    // unit iteration state: (phase ph, unit u).  first unit of this block in phase ph: smallest u >= 0 with (u + offset) % nb == bid
    auto first_unit = [&](const Phase& P) { return ((bid - P.offset) % nb + nb) % nb; };
    for (int ph = 0; ph < p.n_phases; ++ph) {
        const Phase P = p.phases[ph];
        const int nchunks_all = P.K / 64, per_split = nchunks_all / P.ksplit;
        int u = first_unit(P);
        const bool have = u < P.n_units;
        // ---- prefetch the first ring of my first unit of this phase (independent of x)
        int tile = 0, ks = 0, c0 = 0, cend = 0;
        const bf16_t* wrow = nullptr;
        auto setup = [&](int unit) {
            tile = unit / P.ksplit; ks = unit % P.ksplit;
            const int per = (per_split + 3) / 4;
            c0 = ks * per_split + min(wave * per, per_split);
            cend = ks * per_split + min((wave + 1) * per, per_split);
            wrow = P.W + (size_t)tile * 16 * P.K + lane * 8;
        };
        auto fill = [&](int uu, int c) {
            w[uu][0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + (size_t)c * 1024));
            w[uu][1] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + (size_t)c * 1024 + 512));
        };
        if (have) {
            setup(u);
#pragma unroll
            for (int uu = 0; uu < U; ++uu)
                if (c0 + uu < cend) fill(uu, c0 + uu);
        }
        // ---- wait for the previous phase (all of its units), then fetch x
        if (ph > 0) {
            const Phase Q = p.phases[ph - 1];
            if (tid == 0) {
                const unsigned* c = p.ctr + (size_t)(ph - 1) * 128;
                unsigned it = 0;
                for (;;) {
                    unsigned sum = 0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) sum += __hip_atomic_load(c + i * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (sum >= (unsigned)Q.n_units) break;
                    if (++it > 400000u) { *p.err = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            if (have) {
                if (Q.ksplit == 1) {
                    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(p.y + (size_t)(ph - 1) * 22016);
                    for (int i = tid; i < P.K / 4; i += 256)
                        reinterpret_cast<unsigned long long*>(xs)[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    const unsigned* s0 = reinterpret_cast<const unsigned*>(p.slabs + (size_t)(ph - 1) * 2 * 2048);
                    for (int i = tid; i < P.K; i += 256) {
                        const float a = __uint_as_float(__hip_atomic_load(s0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                        const float b = __uint_as_float(__hip_atomic_load(s0 + 2048 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                        xs[i] = f2bf((a + b) * p.scale);
                    }
                }
            }
        } else if (have) {
            for (int i = tid; i < P.K; i += 256) xs[i] = 0x3c00;
        }
        __syncthreads();
        // ---- my units of this phase
        for (; u < P.n_units; u += nb) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int c = c0; c < cend; c += U) {
#pragma unroll
                for (int uu = 0; uu < U; ++uu) {
                    if (c + uu < cend) {
                        const bf16_t* xr = xs + (size_t)(c + uu) * 64 + fg * 16;
                        u32x4 x0 = u32x4{0, 0, 0, 0}, x1 = x0;
                        if (fr == 0) { x0 = *reinterpret_cast<const u32x4*>(xr); x1 = *reinterpret_cast<const u32x4*>(xr + 8); }
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[uu][0]), __builtin_bit_cast(bf16x8, x0), acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[uu][1]), __builtin_bit_cast(bf16x8, x1), acc, 0, 0, 0);
                        if (c + U + uu < cend) fill(uu, c + U + uu);
                    }
                }
            }
            const int my_tile = tile, my_ks = ks;
            // next unit of this phase: start its stream before the reduction / store of the current one
            const int un = u + nb;
            if (un < P.n_units) {
                setup(un);
#pragma unroll
                for (int uu = 0; uu < U; ++uu)
                    if (c0 + uu < cend) fill(uu, c0 + uu);
            }
            if (wave > 0) red[(wave - 1) * 64 + lane] = acc;
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { const f32x4 o = red[k * 64 + lane]; acc[0] += o[0]; acc[1] += o[1]; acc[2] += o[2]; acc[3] += o[3]; }
                if (fr == 0) {
                    if (P.ksplit == 1) {
                        const unsigned long long v = (unsigned long long)f2bf(acc[0] * p.scale) | ((unsigned long long)f2bf(acc[1] * p.scale) << 16) |
                                                     ((unsigned long long)f2bf(acc[2] * p.scale) << 32) | ((unsigned long long)f2bf(acc[3] * p.scale) << 48);
                        __hip_atomic_store(reinterpret_cast<unsigned long long*>(p.y + (size_t)ph * 22016 + my_tile * 16 + fg * 4), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        unsigned* d = reinterpret_cast<unsigned*>(p.slabs + ((size_t)ph * 2 + my_ks) * 2048 + my_tile * 16 + fg * 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) __hip_atomic_store(d + r, __float_as_uint(acc[r]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(p.ctr + (size_t)ph * 128 + (bid & 7) * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
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
    {   // mode 3: persistent
        std::vector<Phase> hp(NK);
        int off = 0;
        for (int k = 0; k < NK; ++k) {
            const int j = k & 3, ksp = j == 3 ? 2 : 1;
            hp[k] = Phase{W[k], Ns[j], Ks[j], ksp, Ns[j] / 16 * ksp, off};
            off = (off + hp[k].n_units) % 256;
        }
        Phase* dp; bf16_t* y; float* slabs; unsigned* pctr;
        CK(hipMalloc(&dp, NK * sizeof(Phase)));
        CK(hipMemcpy(dp, hp.data(), NK * sizeof(Phase), hipMemcpyHostToDevice));
        CK(hipMalloc(&y, (size_t)NK * 22016 * 2));
        CK(hipMalloc(&slabs, (size_t)NK * 2 * 2048 * 4));
        CK(hipMalloc(&pctr, (size_t)NK * 128 * 4));
        const size_t smem = (11008 + 64) * 2 + 3 * 64 * 16 + 72 * 1024;     // padding forces one block per CU
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_persistent), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        float best = 1e9f; unsigned herr = 0;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipMemsetAsync(pctr, 0, (size_t)NK * 128 * 4, s[0]));
            CK(hipMemsetAsync(err, 0, 4, s[0]));
            CK(hipStreamSynchronize(s[0]));
            PArgs pa{dp, NK, y, slabs, pctr, err, 1e-3f};
            CK(hipEventRecord(e0, s[0]));
            hipLaunchKernelGGL(k_chain_persistent, dim3(256), dim3(256), smem, s[0], pa);
            CK(hipEventRecord(e1, s[0]));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
            unsigned h2 = 0; CK(hipMemcpy(&h2, err, 4, hipMemcpyDeviceToHost)); herr |= h2;
        }
        printf("mode 3 (persistent, 256 blocks): %.1f us per layer  [%.3f ms]  timeout_flag=%u\n", best * 1e3f / L, best, herr);
    }
    return 0;
}
