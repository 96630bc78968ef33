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

// ---------------------------------------------------------------------------------------------------------------------------------
// top_k <= 0: NO top-k bound (vLLM's top_k = -1, which /root/reference/roll/distributed/strategy/vllm_strategy.py:289-309 passes through when the
// YAML does not set one) -- round 5; rounds 1-4 ran this case as a host loop over sr_decode_step.  Nucleus sampling over the whole vocabulary
// without sorting it: with p_i = softmax(l_i / T), the kept set of the sorted definition above ("keep j while the mass in front of it is < top_p") is
//     { i : M(key_i) < top_p },   M(t) = mass of the tokens with a key STRICTLY above t,
// plus, among the tokens that tie at the smallest kept key tau, the lowest ids while the mass in front stays < top_p.  tau is found by the same
// 12 + 12 + 8-bit radix descent as the top-k select, with histograms of MASS instead of counts -- in 2^-40 fixed point and 64-bit LDS atomics,
// so the sums (and with them every decision) are independent of the order in which threads arrive.  The draw is one inverse-CDF pass in TOKEN-ID
// order over the kept set (the distribution does not depend on the enumeration order): per-thread sums over contiguous id ranges, a fixed-order
// scan over the 1024 threads, then the owning thread walks its range.
constexpr int FX_SHIFT = 40;


// exclusive prefix sum of one value per thread over the block (NT threads) in a FIXED association: Hillis-Steele inside each wave (shuffles), then the wave
// totals in wave order.  sw: >= NT / 64 floats of LDS nobody else uses across the call.  total = the block's sum (same association for every thread).
__device__ __forceinline__ float block_excl_scan(int tid, float v, float* sw, float& total) {
    const int lane = tid & 63, wave = tid >> 6;
    float inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __syncthreads();                 // (sw may still be read by an earlier call)
    if (lane == 63) sw[wave] = inc;
    __syncthreads();
    float base = 0.f, tot = 0.f;
    for (int w = 0; w < NT / 64; ++w) {
        if (w < wave) base += sw[w];
        tot += sw[w];
    }
    total = tot;
    const float excl = __shfl_up(inc, 1, 64);         // the wave-inclusive sum of the lane before (every lane executes the shuffle)
    return base + (lane ? excl : 0.f);
}
__global__ __launch_bounds__(NT) void k_sample_full(SampleArgs a) {
    __shared__ unsigned long long mh[4096];        // mass histogram (fixed point)
    __shared__ int ch[256];                        // count histogram of the last level (ties at tau)
    __shared__ float s_red[16];
    __shared__ unsigned long long s_u64[2];
    __shared__ int s_i[4];
    __shared__ float s_scan[NT];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = a.logits + (size_t)b * a.V;
    const unsigned* seen = a.seen ? a.seen + (size_t)b * a.seen_words : nullptr;
    auto adj = [&](int id) -> float {
        float l = row[id];
        if (seen && ((seen[id >> 5] >> (id & 31)) & 1u)) l = l > 0.f ? l / a.rep_penalty : l * a.rep_penalty;
        return l;
    };
    // ---- max and partition function (fixed-order block reductions)
    float mx = -INFINITY;
    for (int i = tid; i < a.V; i += NT) mx = fmaxf(mx, adj(i));
    mx = wave_max(mx);
    if (lane == 0) s_red[wave] = mx;
    __syncthreads();
    mx = s_red[0];
    for (int w = 1; w < NT / 64; ++w) mx = fmaxf(mx, s_red[w]);
    __syncthreads();
    float z = 0.f;
    for (int i = tid; i < a.V; i += NT) z += __expf((adj(i) - mx) * a.inv_temp);
    z = wave_sum(z);
    if (lane == 0) s_red[wave] = z;
    __syncthreads();
    float Z = 0.f;
    for (int w = 0; w < NT / 64; ++w) Z += s_red[w];
    const float inv_z = 1.0f / Z;
    auto prob_fx = [&](float l) -> unsigned long long { return (unsigned long long)(__expf((l - mx) * a.inv_temp) * inv_z * (float)(1ull << FX_SHIFT)); };
    const unsigned long long want = a.top_p >= 1.0f ? ~0ull : (unsigned long long)((double)a.top_p * (double)(1ull << FX_SHIFT));
    // ---- radix descent on mass: tau = the smallest key whose strictly-above mass is < top_p
    uint32_t prefix = 0;
    unsigned long long above = 0;                   // mass of the keys above the current prefix range
    const int shifts[3] = {20, 8, 0}, widths[3] = {12, 12, 8};
#pragma unroll
    for (int lvl = 0; lvl < 3; ++lvl) {
        const int nb = 1 << widths[lvl];
        for (int i = tid; i < nb; i += NT) mh[i] = 0ull;
        if (lvl == 2) for (int i = tid; i < 256; i += NT) ch[i] = 0;
        __syncthreads();
        const uint32_t himask = lvl == 0 ? 0u : (lvl == 1 ? 0xfff00000u : 0xffffff00u);
        for (int i = tid; i < a.V; i += NT) {
            const float l = adj(i);
            const uint32_t k = fkey(l);
            if ((k & himask) == prefix) {
                atomicAdd(&mh[(k >> shifts[lvl]) & (nb - 1)], prob_fx(l));
                if (lvl == 2) atomicAdd(&ch[k & 0xff], 1);
            }
        }
        __syncthreads();
        if (tid == 0) {      // bins from the top: the first bin whose inclusive mass reaches top_p holds tau (if none does -- rounding, top_p = 1 -- the lowest non-empty one)
            unsigned long long run = above;
            int bin = -1, last = 0;
            for (int bb = nb - 1; bb >= 0; --bb) {
                if (mh[bb] == 0ull && !(lvl == 2 && ch[bb])) continue;
                last = bb;
                if (run + mh[bb] >= want) { bin = bb; break; }
                run += mh[bb];
            }
            if (bin < 0) { bin = last; run -= mh[last]; }
            s_i[0] = bin;
            s_u64[0] = run;
        }
        __syncthreads();
        prefix |= (uint32_t)s_i[0] << shifts[lvl];
        above = s_u64[0];
        __syncthreads();
    }
    const uint32_t tau = prefix;
    const int n_eq = ch[tau & 0xff];
    // ties at tau: keep the r lowest ids, r = the smallest count with above + r p_tau >= top_p (at least one, at most all)
    const unsigned long long p_tau = prob_fx(unkey(tau));
    int r_keep = n_eq;
    if (want != ~0ull && p_tau > 0ull && above < want) {
        const unsigned long long need = want - above;
        const unsigned long long r = (need + p_tau - 1) / p_tau;
        if (r < (unsigned long long)n_eq) r_keep = (int)(r < 1 ? 1 : r);
    }
    // ---- inverse CDF in id order over the kept set.  Thread t owns ids [t * per, (t + 1) * per)
    const int per = (a.V + NT - 1) / NT, i0 = tid * per, i1 = min(a.V, i0 + per);
    float msum = 0.f;
    int ties = 0;
    for (int i = i0; i < i1; ++i) {
        const uint32_t k = fkey(adj(i));
        ties += k == tau;
    }
    // exclusive scan of the tie counts -> which ties of this range are among the first r_keep.  Round 6: a wave-level shuffle scan + a
    // 16-entry cross-wave pass in fixed order instead of every thread walking s_scan[0 .. tid) (O(NT^2) LDS reads inside the captured decode step)
    float tie_total;
    const int ties_before = (int)block_excl_scan(tid, (float)ties, s_scan, tie_total);        // (counts <= 149: exact in float32)
    int seen_ties = ties_before;
    for (int i = i0; i < i1; ++i) {
        const float l = adj(i);
        const uint32_t k = fkey(l);
        const bool keep = k > tau || (k == tau && seen_ties++ < r_keep);
        if (keep) msum += __expf((l - mx) * a.inv_temp) * inv_z;
    }
    float total;
    const float before = block_excl_scan(tid, msum, s_scan, total);          // fixed association: lanes inside a wave, then the waves in order -> run-to-run identical
    const int stp = a.step ? a.step[b] : 0;
    const uint32_t h = hmix(a.seed ^ hmix((uint32_t)b * 0x9E3779B1u + (uint32_t)stp * 0x85EBCA77u + 0x1234567u));
    const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
    const float target = u * total;
    if (tid == 0) s_i[1] = -1;
    __syncthreads();
    // the owner: the thread whose [before, before + msum) contains the target (the last thread with kept mass takes a target that rounding pushed past the end)
    const bool owner = msum > 0.f && target >= before && (target < before + msum);
    if (owner) atomicMax(&s_i[1], tid);
    __syncthreads();
    if (s_i[1] < 0) {            // rounding: nobody's interval held the target -> the last thread with mass
        if (msum > 0.f) atomicMax(&s_i[1], tid);
        __syncthreads();
    }
    if (tid == s_i[1]) {
        float run = before;
        int st = ties_before, pick = -1, lastk = -1;
        for (int i = i0; i < i1; ++i) {
            const float l = adj(i);
            const uint32_t k = fkey(l);
            const bool keep = k > tau || (k == tau && st++ < r_keep);
            if (!keep) continue;
            lastk = i;
            run += __expf((l - mx) * a.inv_temp) * inv_z;
            if (target < run) { pick = i; break; }
        }
        a.out[b] = (long long)(pick >= 0 ? pick : lastk);
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
    if (!(a.inv_temp > 0.f) || !(a.top_p > 0.f)) return -22;
    if (a.top_k <= 0 || a.top_k >= a.V) {            // no top-k bound (vLLM's -1): nucleus sampling over the whole vocabulary
        hipLaunchKernelGGL(k_sample_full, dim3(a.B), dim3(NT), 0, s, a);
        SR_CHECK_LAUNCH();
        return 0;
    }
    if (a.top_k > KMAX) return -22;
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
