// Research probe (NOT part of the product library; round 6): what does an ALL-TO-ALL activation hand-off cost inside one
// persistent launch on MI355X at the edge sizes of THIS decode layer, next to the kernel boundary it would replace?
//
// The attention half of a 32-row decode layer is six launches (RMSNorm -> q/k/v -> scores -> P.V -> o_proj -> RMSNorm = 35.5 us for ~36 MB) whose every
// seam is all-to-all: each consumer block needs the whole activation (32 rows x 2048 x bf16 = 128 KB; 160 KB after q/k/v).  A persistent form replaces
// the five boundaries by five grid-wide hand-offs and may run the next phase's WEIGHT loads ahead of the hand-off (they do not depend on activations).
// This probe runs exactly that skeleton -- 256 persistent blocks (one per CU), per phase: [issue the phase's weight prefetch into registers] -> grid
// barrier -> gather the whole edge with sc1 loads (every 16-byte chunk carries the phase tag and is CHECKED: a stale read is counted) -> consume ->
// publish the block's slice of the next edge with write-through stores -> arrive -- and the same phases as separate graph-captured launches (plain
// loads / stores, the boundary does the hand-off).  Stamps of the 100 MHz wall clock per block and phase give the timeline.
//
//   hipcc --offload-arch=gfx950 -O3 -o handoff_probe handoff_probe.hip && ./handoff_probe            (prints one JSON line per configuration)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* p = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
enum { SC1 = 16 };
__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
    return v;
}

struct Bar {                    // every word on its own 128-byte line
    unsigned xcd_cnt[8 * 32];
    unsigned top[32];
    unsigned gen[8 * 32];
    unsigned census[8 * 32];
    unsigned flat[32];
    unsigned err[32];           // [0] spins that gave up, [1] stale chunks seen
};

struct Args {
    unsigned char* edge[2];     // ping-pong activation edges, edge_bytes each
    int edge_bytes;             // multiple of 256 blocks x 16 B
    const unsigned char* wts;   // weight stream: n_phases x grid x pf_bytes, every byte read once
    int pf_loads;               // 16-byte prefetch loads per thread and phase (0: no weight stream)
    int n_phases;
    int mode;                   // 0 xcd-hierarchical barrier, 1 flat counter
    Bar* bar;
    long long* stamps;          // [n_stamp_phases][grid][3]: phase start, barrier released, gathered
    int stamp_from;             // first stamped phase
    unsigned* sink;
};

__device__ __forceinline__ void spin_until(unsigned* p, unsigned target, unsigned* err) {
    unsigned it = 0;
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++it > (1u << 18)) { atomicAdd(err, 1u); break; }      // never hang the GPU
    }
}

__global__ __launch_bounds__(256) void k_persistent(Args p) {
    __shared__ unsigned s_n[2];
    const int tid = threadIdx.x, b = blockIdx.x, G = gridDim.x;
    const unsigned x = xcc_id();
    // census: blocks per XCD, then one flat barrier so that everybody knows the counts
    if (tid == 0) {
        __hip_atomic_fetch_add(&p.bar->census[x * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&p.bar->flat[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        spin_until(&p.bar->flat[0], (unsigned)G, &p.bar->err[0]);
        unsigned nx = 0;
        for (int i = 0; i < 8; ++i) nx += __hip_atomic_load(&p.bar->census[i * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0;
        s_n[0] = __hip_atomic_load(&p.bar->census[x * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_n[1] = nx;
    }
    __syncthreads();
    const unsigned n_mine = s_n[0], n_xcd = s_n[1];
    const int per_thread = p.edge_bytes / (256 * 16);            // 16-byte chunks of the edge per thread
    const int slice = p.edge_bytes / G;                          // bytes of the edge this block publishes
    unsigned acc = 0;
    for (int ph = 0; ph < p.n_phases; ++ph) {
        const long long t0 = wall_clock64();
        // (1) run-ahead weight prefetch of THIS phase: issued before the hand-off is known to be complete
        u32x4 w[16];
        const unsigned char* wp = p.wts + ((size_t)ph * G + b) * (size_t)p.pf_loads * 256 * 16 + tid * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < p.pf_loads) w[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + (size_t)i * 256 * 16));
        // (2) grid barrier: the edge written in phase ph - 1 is complete (phase 0: the host filled it)
        if (ph > 0) {
            if (tid == 0) {
                const unsigned g = (unsigned)ph;
                if (p.mode == 1) {
                    __hip_atomic_fetch_add(&p.bar->flat[16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    spin_until(&p.bar->flat[16], g * G, &p.bar->err[0]);
                } else {
                    const unsigned t = __hip_atomic_fetch_add(&p.bar->xcd_cnt[x * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
                    if (t == n_mine * g) {                       // this XCD's last arriver: report upstairs, wait for the others, release the XCD
                        __hip_atomic_fetch_add(&p.bar->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        spin_until(&p.bar->top[0], n_xcd * g, &p.bar->err[0]);
                        __hip_atomic_store(&p.bar->gen[x * 32], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else spin_until(&p.bar->gen[x * 32], g, &p.bar->err[0]);
                }
            }
            __syncthreads();
        }
        const long long t1 = wall_clock64();
        // (3) gather the whole edge (sc1: L1 bypass; the producers wrote through) and check every chunk's tag
        const __amdgpu_buffer_rsrc_t re = rsrc(p.edge[ph & 1], (unsigned)p.edge_bytes);
        unsigned stale = 0;
        for (int c0 = 0; c0 < per_thread; c0 += 8) {
            u32x4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (c0 + i < per_thread) v[i] = __builtin_amdgcn_raw_buffer_load_b128(re, ((c0 + i) * 256 + tid) * 16, 0, SC1);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (c0 + i < per_thread) { stale += v[i][0] != (unsigned)ph; acc += v[i][1] ^ v[i][3]; }
        }
        const long long t2 = wall_clock64();
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < p.pf_loads) acc += w[i][0] ^ w[i][2];
        if (stale) atomicAdd(&p.bar->err[1], stale);
        // (4) publish this block's slice of the next edge write-through, drain, (arrive at the top of the next phase)
        const __amdgpu_buffer_rsrc_t rn = rsrc(p.edge[(ph + 1) & 1], (unsigned)p.edge_bytes);
        for (int i = tid * 16; i < slice; i += 256 * 16)
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{(unsigned)(ph + 1), acc, (unsigned)b, acc}, rn, b * slice + i, 0, SC1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0 && ph >= p.stamp_from) {
            long long* s = p.stamps + ((size_t)(ph - p.stamp_from) * G + b) * 3;
            s[0] = t0; s[1] = t1; s[2] = t2;
        }
    }
    if (acc == 0x12345678u) p.sink[0] = acc;
}

// XCD-LOCAL variant (round 6): the edge is produced REDUNDANTLY inside every XCD (8 copies, one per XCD: what a per-XCD recomputation of an RMSNorm would
// be), so the hand-off never leaves the XCD's L2: plain stores (L1 is write-through) + vmcnt(0) + an L2 atomic (no sc bits: atomics always execute at L2) as the
// arrival, polled with an L2 atomic, then sc0 loads (L1 bypass, L2 hit) for the payload.  Every chunk's tag is checked like above.
enum { SC0 = 1 };
__global__ __launch_bounds__(256) void k_persistent_xcd(Args p, unsigned char* xedge /* [2][8][edge_bytes] */) {
    __shared__ unsigned s_n[2];
    const int tid = threadIdx.x, b = blockIdx.x, G = gridDim.x;
    const unsigned x = xcc_id();
    if (tid == 0) {
        const unsigned mine = __hip_atomic_fetch_add(&p.bar->census[x * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&p.bar->flat[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        spin_until(&p.bar->flat[0], (unsigned)G, &p.bar->err[0]);
        s_n[0] = __hip_atomic_load(&p.bar->census[x * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_n[1] = mine;                                            // this block's index inside its XCD
    }
    __syncthreads();
    const unsigned n_mine = s_n[0], idx = s_n[1];
    const int per_thread = p.edge_bytes / (256 * 16);
    const int slice = p.edge_bytes / (int)n_mine / 16 * 16;     // this block's share of ITS XCD's copy (the last block takes the remainder)
    const int s0 = (int)idx * slice, s1 = idx + 1 == n_mine ? p.edge_bytes : s0 + slice;
    unsigned acc = 0;
    for (int ph = 0; ph < p.n_phases; ++ph) {
        const long long t0 = wall_clock64();
        u32x4 w[16];
        const unsigned char* wp = p.wts + ((size_t)ph * G + b) * (size_t)p.pf_loads * 256 * 16 + tid * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < p.pf_loads) w[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + (size_t)i * 256 * 16));
        if (ph > 0) {
            if (tid == 0) {
                unsigned* c = &p.bar->xcd_cnt[x * 32];
                // (workgroup-scope atomics were tried first: the pollers never saw the other CUs' arrivals -- every spin gave up and every chunk read stale;
                // the counter needs agent scope even when all its users share one L2)
                __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned it = 0;
                while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n_mine * (unsigned)ph) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++it > (1u << 16)) { atomicAdd(&p.bar->err[0], 1u); break; }
                }
            }
            __syncthreads();
        }
        const long long t1 = wall_clock64();
        unsigned char* mine_r = xedge + ((size_t)(ph & 1) * 8 + x) * p.edge_bytes;
        const __amdgpu_buffer_rsrc_t re = rsrc(mine_r, (unsigned)p.edge_bytes);
        unsigned stale = 0;
        for (int c0 = 0; c0 < per_thread; c0 += 8) {
            u32x4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (c0 + i < per_thread) v[i] = __builtin_amdgcn_raw_buffer_load_b128(re, ((c0 + i) * 256 + tid) * 16, 0, SC0);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (c0 + i < per_thread) { stale += v[i][0] != (unsigned)ph; acc += v[i][1] ^ v[i][3]; }
        }
        const long long t2 = wall_clock64();
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < p.pf_loads) acc += w[i][0] ^ w[i][2];
        if (stale) atomicAdd(&p.bar->err[1], stale);
        u32x4* o = reinterpret_cast<u32x4*>(xedge + ((size_t)((ph + 1) & 1) * 8 + x) * p.edge_bytes);
        for (int i = s0 / 16 + tid; i < s1 / 16; i += 256) o[i] = u32x4{(unsigned)(ph + 1), acc, (unsigned)b, acc};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0 && ph >= p.stamp_from) {
            long long* s = p.stamps + ((size_t)(ph - p.stamp_from) * G + b) * 3;
            s[0] = t0; s[1] = t1; s[2] = t2;
        }
    }
    if (acc == 0x12345678u) p.sink[0] = acc;
}

// the same phase as a launch of its own: the kernel boundary is the hand-off (plain loads and stores)
__global__ __launch_bounds__(256) void k_phase(Args p, int ph) {
    const int tid = threadIdx.x, b = blockIdx.x, G = gridDim.x;
    const int per_thread = p.edge_bytes / (256 * 16), slice = p.edge_bytes / G;
    unsigned acc = 0, stale = 0;
    u32x4 w[16];
    const unsigned char* wp = p.wts + ((size_t)ph * G + b) * (size_t)p.pf_loads * 256 * 16 + tid * 16;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i < p.pf_loads) w[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + (size_t)i * 256 * 16));
    const u32x4* e = reinterpret_cast<const u32x4*>(p.edge[ph & 1]);
    for (int c0 = 0; c0 < per_thread; c0 += 8) {
        u32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (c0 + i < per_thread) v[i] = e[(c0 + i) * 256 + tid];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (c0 + i < per_thread) { stale += v[i][0] != (unsigned)ph; acc += v[i][1] ^ v[i][3]; }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i < p.pf_loads) acc += w[i][0] ^ w[i][2];
    if (stale) atomicAdd(&p.bar->err[1], stale);
    u32x4* o = reinterpret_cast<u32x4*>(p.edge[(ph + 1) & 1] + (size_t)b * slice);
    for (int i = tid; i < slice / 16; i += 256) o[i] = u32x4{(unsigned)(ph + 1), acc, (unsigned)b, acc};
    if (acc == 0x12345678u) p.sink[0] = acc;
}

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0 : v[v.size() / 2]; }

int main(int argc, char** argv) {
    const int G = 256, LAYERS = 36, PH_PER_LAYER = 6, NPH = LAYERS * PH_PER_LAYER, STAMP = 4 * PH_PER_LAYER;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const size_t wbytes = (size_t)NPH * G * 16 * 256 * 16;         // 16 loads x 256 threads x 16 B = 64 KB per block and phase at most
    unsigned char *wts, *e0, *e1;
    Bar* bar; long long* stamps; unsigned* sink;
    CK(hipMalloc(&wts, wbytes)); CK(hipMemset(wts, 1, wbytes));
    CK(hipMalloc(&e0, 1 << 20)); CK(hipMalloc(&e1, 1 << 20));
    CK(hipMalloc(&bar, sizeof(Bar))); CK(hipMalloc(&stamps, (size_t)STAMP * G * 3 * 8)); CK(hipMalloc(&sink, 64));
    std::vector<long long> hs((size_t)STAMP * G * 3);
    // edge sizes: 4 KB (the guide's batch-1 edges), 32 KB, 128 KB (32 rows x 2048 bf16: this layer), 256 KB;  weight prefetch per block and phase:
    // 0 / 40 KB (q/k/v: 10.5 MB over 256 CUs) per block
    for (int edge_kb : {4, 32, 128, 256}) for (int pf : {0, 10}) for (int mode : {0, 1}) {
        Args a{};
        a.edge[0] = e0; a.edge[1] = e1; a.edge_bytes = edge_kb * 1024; a.wts = wts; a.pf_loads = pf; a.n_phases = NPH; a.mode = mode; a.bar = bar;
        a.stamps = stamps; a.stamp_from = NPH - STAMP; a.sink = sink;
        double best = 1e30;
        unsigned herr[2] = {0, 0};
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(bar, 0, sizeof(Bar), s));
            CK(hipMemsetAsync(e0, 0, 1 << 20, s));               // phase 0 expects tag 0
            hipEvent_t a0, a1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1));
            CK(hipEventRecord(a0, s));
            hipLaunchKernelGGL(k_persistent, dim3(G), dim3(256), 0, s, a);
            CK(hipEventRecord(a1, s));
            CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, a0, a1));
            best = std::min(best, (double)ms);
            unsigned e2[2];
            CK(hipMemcpy(e2, bar->err, 8, hipMemcpyDeviceToHost));
            herr[0] += e2[0]; herr[1] += e2[1];
        }
        CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
        // per stamped phase: barrier = (last block released) - (last block arrived = its t0 ... approximated by the max t0); gather = median (t2 - t1)
        std::vector<double> wait_med, wait_last, gather, span;
        for (int ph = 1; ph < STAMP; ++ph) {
            long long first0 = 1ll << 62, last0 = 0, last1 = 0, last2 = 0;
            std::vector<double> w_, g_;
            for (int b = 0; b < G; ++b) {
                const long long* q = &hs[((size_t)ph * G + b) * 3];
                first0 = std::min(first0, q[0]); last0 = std::max(last0, q[0]); last1 = std::max(last1, q[1]); last2 = std::max(last2, q[2]);
                w_.push_back((q[1] - q[0]) * 0.01); g_.push_back((q[2] - q[1]) * 0.01);
            }
            wait_med.push_back(med(w_)); wait_last.push_back((last1 - last0) * 0.01); gather.push_back(med(g_));
            const long long* n0 = &hs[((size_t)(ph - 1) * G) * 3];
            long long prev_first0 = 1ll << 62;
            for (int b = 0; b < G; ++b) prev_first0 = std::min(prev_first0, n0[b * 3]);
            span.push_back((first0 - prev_first0) * 0.01);
        }
        // XCD-local hand-off (8 redundant copies of the edge, nothing leaves an XCD's L2)
        double xcd_us = 0, xcd_wait = 0, xcd_gather = 0;
        unsigned xcd_err[2] = {0, 0};
        if (mode == 0) {
            static unsigned char* xedge = nullptr;
            if (!xedge) CK(hipMalloc(&xedge, (size_t)2 * 8 * (1 << 20)));
            double bx = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemsetAsync(bar, 0, sizeof(Bar), s));
                CK(hipMemsetAsync(xedge, 0, (size_t)2 * 8 * (1 << 20), s));
                hipEvent_t a0, a1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1));
                CK(hipEventRecord(a0, s));
                hipLaunchKernelGGL(k_persistent_xcd, dim3(G), dim3(256), 0, s, a, xedge);
                CK(hipEventRecord(a1, s));
                CK(hipStreamSynchronize(s));
                float ms; CK(hipEventElapsedTime(&ms, a0, a1));
                bx = std::min(bx, (double)ms);
                unsigned e3[2];
                CK(hipMemcpy(e3, bar->err, 8, hipMemcpyDeviceToHost));
                xcd_err[0] += e3[0]; xcd_err[1] += e3[1];
            }
            xcd_us = bx * 1e3 / NPH;
            CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> w_, g_;
            for (int ph = 1; ph < STAMP; ++ph)
                for (int b = 0; b < G; ++b) {
                    const long long* q = &hs[((size_t)ph * G + b) * 3];
                    w_.push_back((q[1] - q[0]) * 0.01); g_.push_back((q[2] - q[1]) * 0.01);
                }
            xcd_wait = med(w_); xcd_gather = med(g_);
        }
        // the same phases as graph-captured launches
        double launches_us = 0;
        if (mode == 0) {
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipMemsetAsync(bar, 0, sizeof(Bar), s));
            CK(hipMemsetAsync(e0, 0, 1 << 20, s));
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            for (int ph = 0; ph < NPH; ++ph) hipLaunchKernelGGL(k_phase, dim3(G), dim3(256), 0, s, a, ph);
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            double bestl = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemsetAsync(e0, 0, 1 << 20, s));
                hipEvent_t a0, a1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1));
                CK(hipEventRecord(a0, s));
                CK(hipGraphLaunch(ge, s));
                CK(hipEventRecord(a1, s));
                CK(hipStreamSynchronize(s));
                float ms; CK(hipEventElapsedTime(&ms, a0, a1));
                bestl = std::min(bestl, (double)ms);
            }
            launches_us = bestl * 1e3 / NPH;
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
        unsigned e2[2];
        CK(hipMemcpy(e2, bar->err, 8, hipMemcpyDeviceToHost));
        printf("{\"edge_KB\": %d, \"weight_prefetch_KB_per_block_and_phase\": %d, \"barrier\": \"%s\", \"persistent_us_per_phase\": %.2f, "
               "\"phase_span_us_median\": %.2f, \"wait_us_median_block\": %.2f, \"last_arrival_to_last_release_us\": %.2f, \"gather_us_median_block\": %.2f, "
               "\"launches_us_per_phase\": %.2f, \"spins_given_up\": %u, \"stale_chunks_persistent\": %u, \"stale_chunks_launches\": %u, "
               "\"xcd_local_us_per_phase\": %.2f, \"xcd_local_wait_us\": %.2f, \"xcd_local_gather_us\": %.2f, \"xcd_local_spins_given_up\": %u, \"xcd_local_stale_chunks\": %u}\n",
               edge_kb, pf * 4, mode ? "flat counter" : "xcd-hierarchical", best * 1e3 / NPH, med(span), med(wait_med), med(wait_last), med(gather),
               launches_us, herr[0], herr[1], e2[1], xcd_us, xcd_wait, xcd_gather, xcd_err[0], xcd_err[1]);
        fflush(stdout);
    }
    return 0;
}
