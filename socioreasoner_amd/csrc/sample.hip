// Token choice by sampling, on the device (SURVEY.md 2.3 K17 "... or top-k / top-p / temperature"; the reference hands
// these to vLLM's sampler, roll/distributed/strategy/vllm_strategy.py:289-309, and its shipped YAML uses temperature 1,
// top_k 100, top_p 0.8).  One 1024-thread block per sequence works on the float32 logits row the LM head left in HBM:
//   1. repetition penalty on tokens already seen (l > 0 ? l / rp : l * rp), applied on the fly in every pass;
//   2. exact top-k (k <= 1024) by a 3-level radix select on the order-preserving integer image of the floats
//      (12 + 12 + 8 bits, LDS histograms), ties at the threshold resolved towards the LOWEST token id;
//   3. the k candidates are sorted (value descending, id ascending) with a bitonic network in LDS;
//   4. softmax with temperature over the candidates, top-p: the smallest prefix of the sorted list whose tail mass
//      exceeds 1 - top_p (the most likely token always stays), renormalised;
//   5. one categorical draw by inverse CDF with a counter-based uniform u(seed, row, step).
// The chosen id goes to the buffer k_step reads as the caller-chosen token, so a sampled decode step is still one
// captured graph.  vLLM's random stream cannot be reproduced: parity for this path is distributional (tests compare the
// empirical frequencies with softmax over the filtered set) plus the deterministic limits (top_k = 1 == greedy).
#include "kernels.h"

namespace {

constexpr int NT = 1024, KMAX = 1024;

__device__ __forceinline__ uint32_t fkey(float f) {        // larger float -> larger unsigned
    const uint32_t b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float unkey(uint32_t k) {
    const uint32_t b = (k >> 31) ? (k ^ 0x80000000u) : ~k;
    return __uint_as_float(b);
}
__device__ __forceinline__ uint32_t hmix(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

// finds, scanning bins from the top, the bin in which the running count reaches `need`; returns the bin and, through
// `above`, the number of elements in higher bins.  All threads get the result.  hist has nb bins (nb <= 4096).
__device__ int find_bin(const int* hist, int nb, int need, int* above, int* s_tmp) {
    const int tid = threadIdx.x;
    if (tid < 64) {
        const int per = nb / 64;                                   // bins per lane, lane 0 = TOP bins
        const int hi = nb - tid * per;                             // exclusive upper bin of this lane
        int sum = 0;
        for (int b = hi - 1; b >= hi - per; --b) sum += hist[b];
        int incl = sum;                                            // inclusive scan over lanes 0..tid
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if (tid >= o) incl += v;
        }
        const int excl = incl - sum;
        if (excl < need && incl >= need) {                         // the crossing happens inside this lane's bins
            int run = excl, b = hi - 1;
            for (; b >= hi - per; --b) {
                if (run + hist[b] >= need) break;
                run += hist[b];
            }
            s_tmp[0] = b;
            s_tmp[1] = run;
        }
    }
    __syncthreads();
    *above = s_tmp[1];
    return s_tmp[0];
}

__global__ __launch_bounds__(NT) void k_sample(SampleArgs a) {
    __shared__ int hist[4096];
    __shared__ unsigned long long cand[KMAX];      // (sortable value << 32) | (0xffffffff - id): descending order = value desc, id asc
    __shared__ float prob[KMAX];
    __shared__ int s_tmp[4];
    __shared__ int s_cnt;
    __shared__ float s_red[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = a.logits + (size_t)b * a.V;
    const unsigned* seen = a.seen ? a.seen + (size_t)b * a.seen_words : nullptr;
    const int K = a.top_k;
    // ---- candidate reduction through the LM head's per-block maxima (exact): an element of the global top-K is >= the K-th
    // largest element, which is >= the K-th largest BLOCK maximum, so it lives in one of the blocks whose maximum reaches
    // that value.  With K = 100 that leaves ~100 x 64 of the 151 936 logits to look at.  Not usable with a repetition
    // penalty (it changes the values the maxima were taken from) or when ties make the block set too large.
    __shared__ int sel_blk[KMAX];
    __shared__ int s_nsel;
    int n_items = a.V;                      // number of candidate positions; position -> token id through id_of
    const int R = a.blk_rows;
    bool filtered = false;
    if (a.blk_max && !seen && a.n_blk > K && R > 0) {
        const float* bm = a.blk_max + (size_t)b * a.n_blk;
        uint32_t pfx = 0;
        int need_b = K;
#pragma unroll
        for (int lvl = 0; lvl < 3; ++lvl) {
            const int sh = lvl == 0 ? 20 : (lvl == 1 ? 8 : 0), nb = lvl == 2 ? 256 : 4096;
            const uint32_t himask = lvl == 0 ? 0u : (lvl == 1 ? 0xfff00000u : 0xffffff00u);
            for (int i = tid; i < nb; i += NT) hist[i] = 0;
            __syncthreads();
            for (int i = tid; i < a.n_blk; i += NT) {
                const uint32_t k = fkey(bm[i]);
                if ((k & himask) == pfx) atomicAdd(&hist[(k >> sh) & (nb - 1)], 1);
            }
            __syncthreads();
            int above;
            const int bin = find_bin(hist, nb, need_b, &above, s_tmp);
            need_b -= above;
            pfx |= (uint32_t)bin << sh;
            __syncthreads();
        }
        if (tid == 0) s_nsel = 0;
        __syncthreads();
        for (int i = tid; i < a.n_blk; i += NT)
            if (fkey(bm[i]) >= pfx) {                      // every block whose maximum reaches the K-th largest maximum
                const int p = atomicAdd(&s_nsel, 1);
                if (p < KMAX) sel_blk[p] = i;
            }
        __syncthreads();
        if (s_nsel <= KMAX && s_nsel * R >= K) {
            filtered = true;
            n_items = s_nsel * R;
        }
        __syncthreads();
    }
    auto id_of = [&](int i) -> int { return filtered ? sel_blk[i / R] * R + i % R : i; };
    auto adj = [&](int id) -> float {
        if (id >= a.V) return -INFINITY;
        float l = row[id];
        if (seen && ((seen[id >> 5] >> (id & 31)) & 1u)) l = l > 0.f ? l / a.rep_penalty : l * a.rep_penalty;
        return l;
    };
    // ---- radix select of the K-th largest key
    uint32_t prefix = 0;
    int need = K;
    const int shifts[3] = {20, 8, 0}, widths[3] = {12, 12, 8};
#pragma unroll
    for (int lvl = 0; lvl < 3; ++lvl) {
        const int nb = 1 << widths[lvl];
        for (int i = tid; i < nb; i += NT) hist[i] = 0;
        __syncthreads();
        const uint32_t himask = lvl == 0 ? 0u : (lvl == 1 ? 0xfff00000u : 0xffffff00u);
        for (int i = tid; i < n_items; i += NT) {
            const uint32_t k = fkey(adj(id_of(i)));
            if ((k & himask) == prefix) atomicAdd(&hist[(k >> shifts[lvl]) & (nb - 1)], 1);
        }
        __syncthreads();
        int above;
        const int bin = find_bin(hist, nb, need, &above, s_tmp);
        need -= above;
        prefix |= (uint32_t)bin << shifts[lvl];
        __syncthreads();
    }
    const uint32_t thr = prefix;                    // K-th largest key; `need` of the elements equal to it are wanted
    const int n_eq = hist[thr & 0xff];              // level-3 histogram: elements equal to thr
    // ---- collect: everything above the threshold, then `need` ties (lowest ids first)
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    for (int i = tid; i < n_items; i += NT) {
        const int id = id_of(i);
        const uint32_t k = fkey(adj(id));
        if (id < a.V && (k > thr || (k == thr && n_eq == need))) {
            const int p = atomicAdd(&s_cnt, 1);
            cand[p] = ((unsigned long long)k << 32) | (0xffffffffu - (uint32_t)id);
        }
    }
    __syncthreads();
    if (n_eq != need) {                              // more ties than wanted: ordered pass over the vocabulary, ids ascending
        // (with the block filter every element equal to thr is inside a selected block, because its block maximum is
        // >= thr >= the block threshold; scanning the whole row in id order finds exactly those)
        int taken = 0;                               // uniform across the block
        for (int i0 = 0; i0 < a.V && taken < need; i0 += NT) {
            const int i = i0 + tid;
            const bool eq = i < a.V && fkey(adj(i)) == thr;
            const unsigned long long bal = __ballot(eq);
            if (lane == 0) hist[wave] = __popcll(bal);
            __syncthreads();
            int before = 0, total = 0;
            for (int w = 0; w < NT / 64; ++w) {
                if (w < wave) before += hist[w];
                total += hist[w];
            }
            const int rank = taken + before + __popcll(bal & ((1ull << lane) - 1ull));
            if (eq && rank < need) {
                const int p = atomicAdd(&s_cnt, 1);
                cand[p] = ((unsigned long long)thr << 32) | (0xffffffffu - (uint32_t)i);
            }
            taken += total;
            __syncthreads();
        }
    }
    __syncthreads();
    // ---- bitonic sort, descending, padded with zeros to the next power of two
    int n2 = 1;
    while (n2 < K) n2 <<= 1;
    for (int i = K + tid; i < n2; i += NT) cand[i] = 0ull;
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n2; i += NT) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long x = cand[i], y = cand[l];
                    const bool desc = (i & k) == 0;
                    if (desc ? (x < y) : (x > y)) { cand[i] = y; cand[l] = x; }
                }
            }
            __syncthreads();
        }
    }
    // ---- softmax with temperature over the K candidates
    const float vmax = unkey((uint32_t)(cand[0] >> 32));
    float e = 0.f;
    if (tid < K) e = __expf((unkey((uint32_t)(cand[tid] >> 32)) - vmax) * a.inv_temp);
    float ssum = wave_sum(e);
    if (lane == 0) s_red[wave] = ssum;
    __syncthreads();
    float tot = 0.f;
    for (int w = 0; w < NT / 64; ++w) tot += s_red[w];
    const float p = tid < K ? e / tot : 0.f;
    // inclusive prefix sum of p over the sorted order: wave scan + fixed-order wave offsets
    float incl = p;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    __syncthreads();
    if (lane == 63) s_red[wave] = incl;
    __syncthreads();
    float base = 0.f;
    for (int w = 0; w < wave; ++w) base += s_red[w];
    incl += base;
    prob[tid] = incl;                               // c_j
    __syncthreads();
    // ---- top-p: keep j while the tail mass 1 - c_{j-1} exceeds 1 - top_p, i.e. c_{j-1} < top_p (j = 0 always)
    if (tid == 0) { s_tmp[0] = 1; s_tmp[1] = 0; }
    __syncthreads();
    if (tid >= 1 && tid < K && prob[tid - 1] < a.top_p) atomicMax(&s_tmp[0], tid + 1);      // prefix property: max index + 1 = count
    __syncthreads();
    const int n_keep = s_tmp[0];
    const float mass = prob[n_keep - 1];
    const int stp = a.step ? a.step[b] : 0;
    const uint32_t h = hmix(a.seed ^ hmix((uint32_t)b * 0x9E3779B1u + (uint32_t)stp * 0x85EBCA77u + 0x1234567u));
    const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
    const float target = u * mass;
    if (tid < n_keep && prob[tid] <= target) atomicAdd(&s_tmp[1], 1);                       // number of c_j <= target
    __syncthreads();
    if (tid == 0) {
        int j = s_tmp[1];
        if (j > n_keep - 1) j = n_keep - 1;
        a.out[b] = (long long)(0xffffffffu - (uint32_t)(cand[j] & 0xffffffffull));
    }
}

// continuous marking of emitted tokens + the prompt, for the repetition penalty
__global__ void k_mark_prompt(const int* src, const int* lastrow, unsigned* seen, int seen_words) {
    const int b = blockIdx.x;
    const int t0 = b == 0 ? 0 : lastrow[b - 1] + 1, t1 = lastrow[b] + 1;
    unsigned* row = seen + (size_t)b * seen_words;
    for (int i = threadIdx.x; i < seen_words; i += blockDim.x) row[i] = 0u;
    __syncthreads();
    for (int t = t0 + threadIdx.x; t < t1; t += blockDim.x) {
        const int id = src[t];
        if (id >= 0) atomicOr(&row[id >> 5], 1u << (id & 31));
    }
}
__global__ void k_mark_chosen(const long long* chosen, unsigned* seen, int seen_words, int B) {
    const int b = threadIdx.x;
    if (b < B) {
        const long long id = chosen[b];
        atomicOr(&seen[(size_t)b * seen_words + (id >> 5)], 1u << (id & 31));
    }
}

__global__ void k_scatter_rows(const int* rows, const long long* src, long long* dst, int n) {
    const int i = threadIdx.x;
    if (i < n) dst[rows[i]] = src[i];
}

}  // namespace

int launch_scatter_rows(hipStream_t s, const int* rows, const long long* src, long long* dst, int n) {
    hipLaunchKernelGGL(k_scatter_rows, dim3(1), dim3(64), 0, s, rows, src, dst, n);
    SR_CHECK_LAUNCH();
    return 0;
}
int launch_sample(hipStream_t s, const SampleArgs& a) {
    if (a.B <= 0) return 0;
    if (a.top_k < 1 || a.top_k > KMAX || a.top_k > a.V || !(a.inv_temp > 0.f) || !(a.top_p > 0.f)) return -22;
    hipLaunchKernelGGL(k_sample, dim3(a.B), dim3(NT), 0, s, a);
    SR_CHECK_LAUNCH();
    return 0;
}
int launch_mark_prompt(hipStream_t s, const int* src, const int* lastrow, unsigned* seen, int seen_words, int B) {
    hipLaunchKernelGGL(k_mark_prompt, dim3(B), dim3(256), 0, s, src, lastrow, seen, seen_words);
    SR_CHECK_LAUNCH();
    return 0;
}
int launch_mark_chosen(hipStream_t s, const long long* chosen, unsigned* seen, int seen_words, int B) {
    hipLaunchKernelGGL(k_mark_chosen, dim3(1), dim3(64), 0, s, chosen, seen, seen_words, B);
    SR_CHECK_LAUNCH();
    return 0;
}
