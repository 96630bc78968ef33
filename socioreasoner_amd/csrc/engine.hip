// libsocior engine: the C ABI of include/socior.h and the host-side orchestration of the hot path
// (ViT -> merger -> LM prefill -> greedy decode), one engine per GPU/process.
//
// What it replaces in the reference (paths relative to /root/reference): the engine behind
// VllmStrategy.generate (roll/distributed/strategy/vllm_strategy.py:114-141), i.e. vLLM's Qwen2.5-VL executor; the
// arithmetic follows HF's eager implementation (`hf:` = transformers/models/qwen2_5_vl/modeling_qwen2_5_vl.py) which
// is the reference's CPU/eager path (roll/distributed/strategy/hf_strategy.py:49-94).
//
// Memory: one caller-provided workspace carved by a bump allocator: weights (engine layout), activations sized by the
// capacities in sr_config, KV cache (K [layer][slot][kvh][ctx][128], V^T [layer][slot][kvh][128][ctx]), control arrays.
// Decode runs from device-resident state (ctx_len, positions, current token, step) so that ONE captured hipGraph
// replays for every step (launch-bound inner loop -> graph, per the MI355X guide).
#include "../../include/socior.h"
#include "kernels.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

namespace {

thread_local char g_err[512] = "ok";

constexpr int MAXB = 128;     // batch rows an engine can be configured with (the reference's request-level mode keeps up to 128 requests
                              // in flight per worker: /root/reference/roll/distributed/scheduler/generate_scheduler.py:57)
constexpr int SPLITK_KS = 4, SPLITK_ROWS = 1024;      // split-K of the residual GEMMs of small prefills (prefill_splitk below)
struct VitBlockW { bf16_t *norm1, *qkv_w, *qkv_b, *proj_w, *proj_b, *norm2, *gu_w, *gu_b, *down_w, *down_b; };
struct LmLayerW {
    bf16_t *ln1, *qkv_w, *qkv_b, *o_w, *ln2, *gu_w, *down_w;
    // lm_weight_dtype = 1: fp8 images (tiled8) + per-output-channel scales of the four linears (null in bf16 mode)
    unsigned char *qkv_w8 = nullptr, *o_w8 = nullptr, *gu_w8 = nullptr, *down_w8 = nullptr;
    float *qkv_s = nullptr, *o_s = nullptr, *gu_s = nullptr, *down_s = nullptr;
    // HF tensors (re)loaded into each fused matrix since its last quantisation: qkv {q,k,v} = 7, o = 1, gate/up = 3, down = 1
    unsigned char dirty[4] = {0, 0, 0, 0};
};

struct Arena {
    char* base = nullptr;
    size_t off = 0, cap = 0;
    template <class T>
    T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

inline int rup(int a, int b) { return (a + b - 1) / b * b; }

uint32_t fnv1a32(const char* s) {
    uint32_t h = 2166136261u;
    for (; *s; ++s) { h ^= (unsigned char)*s; h *= 16777619u; }
    return h;
}
uint32_t mix32h(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
bf16_t host_f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

}  // namespace

// ---- SR_* switches: the environment is read here and nowhere else in the library
static SrSwitches g_sw;
static bool g_sw_loaded = false;
static int env_int(const char* name, int dflt) { const char* v = getenv(name); return (v && *v) ? atoi(v) : dflt; }
static void load_switches() {
    g_sw.splitk = env_int("SR_SPLITK", 0);
    g_sw.gemm_bm = env_int("SR_GEMM_BM", 0);
    g_sw.gemm_ring = env_int("SR_GEMM_RING", 1);
    g_sw.gemm256 = env_int("SR_GEMM256", 1);
    g_sw.fuse_qkv = env_int("SR_FUSE_QKV", 1);
    g_sw.g256_group = env_int("SR_G256_GROUP", 4);
    if (g_sw.g256_group < 1) g_sw.g256_group = 4;
    g_sw.attn2 = env_int("SR_ATTN2", 1);
    g_sw.attn_win64 = env_int("SR_ATTN_WIN64", 1);
    g_sw.attn_vasm = env_int("SR_ATTN_VASM", 1);
    g_sw.sam_f32_split = env_int("SR_SAM_F32_SPLIT", 1);
    g_sw.gemv_counted = env_int("SR_GEMV_COUNTED", 1);
    g_sw.gemv_xlds = env_int("SR_GEMV_XLDS", 11);
    g_sw_loaded = true;
}
const SrSwitches& sr_switches() {
    if (!g_sw_loaded) load_switches();
    return g_sw;
}

struct sr_engine {
    sr_config c;
    // derived geometry
    int v_hd, v_pd, v_pd_pad, v_inter_pad, v_mh, t_inter_pad, t_qn, t_group;
    int ks_down;                 // cross-block split-K (float32 slabs) of the decode down-projection
    Arena ar;
    size_t weights_begin = 0, weights_end = 0;
    // ---- weights
    bf16_t* patch_w;
    std::vector<VitBlockW> vb;
    bf16_t *ln_q, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
    bf16_t *embed, *final_norm;
    std::vector<LmLayerW> ll;
    bf16_t* lut;       // [3*256] normalise LUT
    float* inv_freq;   // [64]
    bf16_t *rope_cos, *rope_sin;   // [max_ctx + 1][64]
    // ---- ViT activations
    bf16_t *v_pix, *v_x, *v_xn, *v_qkv, *v_attn, *v_act, *v_vt, *v_m1;
    int v_vt_stride;
    bool v_paired = false;       // ViT q / k channels stored in the paired order (head_dim / 2 % 8 == 0): see vit_qk_perm
    float *v_cos, *v_sin;
    int *v_rowmap_embed, *v_rowmap_merge;
    AttnWork *v_work_win, *v_work_full, *v_work_full128;     // full attention: 64-query items (k_attn_prefill) and 128-query items (k_attn_prefill2)
    int v_nwork_full128; bool v_full_aligned;
    bool v_win_all64;              // every window of the cached grids is a full 8 x 8 merge-unit window (64 tokens): k_attn_win64 applies
    int v_nwork_win = 0, v_nwork_full = 0;
    std::vector<int64_t> v_grid_cached;
    // ---- LM activations
    bf16_t *t_x, *t_xn, *t_qkv, *t_attn, *t_act;
    float* t_slabs = nullptr;    // [SPLITK_KS][SPLITK_ROWS][hidden] float32: split-K partial products of the residual GEMMs of a SMALL prefill
    int n_slots = 0;                         // KV-cache slots (>= max_batch): spare slots take admissions prefilled UNDER the running rows' decode
    int staged_n = 0, staged_off = 0;        // sequences of a staged admission that still wait for rows / already installed
    bf16_t *d_xadm = nullptr, *d_xadm_n = nullptr; int* d_adm_slots = nullptr; int h_rows[MAXB];
    unsigned char *t_q8 = nullptr, *t_qs = nullptr; int t_rows_pad = 0;     // lm_weight_dtype 2: MX-quantised GEMM input of the prefill
    int *t_src, *t_pos3, *t_slot, *t_idx, *t_lastrow;
    AttnWork* t_work;
    // ---- decode state (device)
    bf16_t *d_xa, *d_xb, *d_xn, *d_qkv, *d_attn, *d_act, *d_scores;    // d_xa / d_xb: residual stream ping-pong
    float *d_logits, *d_slabs, *d_amax_val;
    float* d_row_cs = nullptr;               // [MAXB][128] rotary cos | sin of every row's current position (k_step -> decode attention)
    unsigned* d_px = nullptr;                // [256] x-stationary GEMVs: 8 ticket shards + done count (k_gemv_px; zero between launches), word 160 = CU limit (sr_rows_set_cus)
    int *d_amax_idx, *d_cur_tok, *d_ctx_len, *d_pos, *d_finished, *d_step, *d_slots, *d_eos, *d_tokens;
    int n_part = 0;      // LM-head blocks = partial argmax entries per row
    // continuous batching (sr_rows_*): admission scratch so that a prefill never touches the pending tokens of running rows
    float *d_logits_adm = nullptr, *d_amax_val_adm = nullptr;
    int *d_amax_idx_adm = nullptr, *d_row_limit = nullptr, *d_ngen = nullptr, *d_adm = nullptr;   // d_adm: [rows | ctx | pos | limit | first_tok] x MAXB
    bool rows_mode = false;
    bool finalized = false;        // sr_finalize_weights ran (fp8 mode: the LM linears are quantised)
    bf16_t *kcache, *vtcache;
    size_t kv_layer_elems;
    // ---- host staging (pinned) + its device mirror
    char* h_stage = nullptr;
    size_t stage_bytes = 0;
    char* d_stage = nullptr;
    hipEvent_t ev_copy = nullptr;
    bool copy_pending = false;
    // ---- decode graph cache
    hipGraphExec_t graph[2] = {nullptr, nullptr};     // [px_mode]: the decode step on the whole chip / on a CU-limited stream (x-stationary GEMVs, sr_rows_set_cus)
    int px_mode = 0;
    int device = 0;                     // the GPU the workspace lives on: every C-ABI entry makes it current (per-thread HIP state)
    hipStream_t cap_stream = nullptr;   // used only to CAPTURE the decode step (the caller's stream may be the null stream)
    int graph_B[2] = {-1, -1}, graph_neos[2] = {-1, -1}, graph_pad[2] = {0, 0};
    hipGraphExec_t step_graph[2] = {nullptr, nullptr};   // sr_decode_step: [0] engine-greedy token, [1] caller-chosen token
    int step_graph_B[2] = {-1, -1}, step_graph_px[2] = {0, 0};
    long long *d_chosen = nullptr, *d_next = nullptr, *d_sampled = nullptr;
    unsigned* d_seen = nullptr;    // [MAXB][seen_words] token bitmask for the repetition penalty
    int seen_words = 0;
    // continuous batching with sampling (sr_rows_sampling): parameters shared by all rows; 0 temperature = greedy
    float rows_temp = 0.f, rows_topp = 1.f; int rows_topk = 0; unsigned rows_seed = 0, adm_count = 0;
    long long* d_adm_pick = nullptr;
    struct RowsGraph { hipGraphExec_t g = nullptr; int neos = -1, pad = 0, topk = 0; float it = 0.f, topp = 0.f; unsigned seed = 0; } rgraph[2];      // [px_mode]
    hipGraphExec_t sgraph = nullptr;      // sampled decode step: bookkeeping + forward + k_sample
    int sg_B = -1, sg_neos = -1, sg_pad = 0, sg_topk = 0, sg_px = 0; float sg_it = 0.f, sg_topp = 0.f, sg_rp = 0.f; unsigned sg_seed = 0;
    int prefilled_B = 0;
    int h_ctx_hi = 0;              // longest context any slot can have reached (prefill length + decode steps issued)
    // ---- bookkeeping
    std::map<std::string, bool> loaded;
    char err[512];
};

namespace {

// HIP's current device is per-thread state (default 0): make the engine's device current at every entry, so that a
// caller on a fresh thread (the request-level server loop) launches, copies and captures on the right GPU
inline void enter(const sr_engine* e) { if (e) (void)hipSetDevice(e->device); }

int fail(sr_engine* e, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    snprintf(g_err, sizeof g_err, "%s", buf);
    if (e) snprintf(e->err, sizeof e->err, "%s", buf);
    return code;
}
#define SR_TRY(expr)                                                                                          \
    do {                                                                                                      \
        int rc_ = (expr);                                                                                     \
        if (rc_ != 0) return fail(e, rc_, "%s failed with %d (%s) at %s:%d", #expr, rc_,                       \
                                  rc_ > 0 ? hipGetErrorString((hipError_t)rc_) : "invalid argument", __FILE__, __LINE__); \
    } while (0)

const char* validate(const sr_config& c) {
    if (c.v_hidden <= 0 || c.v_heads <= 0 || c.v_hidden % c.v_heads) return "v_hidden/v_heads";
    const int hd = c.v_hidden / c.v_heads;
    if (hd != 80 && hd != 128) return "vision head_dim must be 80 or 128";
    if (c.v_hidden % 64) return "v_hidden must be a multiple of 64";
    if (c.v_merge != 2 || c.v_temporal < 1 || c.v_patch < 1) return "v_merge must be 2";
    if (c.v_out_hidden != c.t_hidden) return "v_out_hidden must equal t_hidden";
    if (c.t_head_dim != 128) return "t_head_dim must be 128";
    if (c.t_hidden % 64 || c.t_vocab % 16) return "t_hidden % 64, t_vocab % 16";
    if (c.t_hidden > 2048) return "t_hidden must be <= 2048 (the decode RMSNorm and fragment-ordered activation kernels are sized for SocioReasoner-3B)";
    if (c.t_heads % c.t_kv_heads || c.t_heads / c.t_kv_heads > 16) return "GQA group must divide and be <= 16";
    if (c.mrope_section[0] + c.mrope_section[1] + c.mrope_section[2] != 64) return "mrope_section must sum to 64";
    if (c.max_batch < 1 || c.max_batch > MAXB) return "max_batch in 1..128";
    if (c.kv_slots != 0 && (c.kv_slots < c.max_batch || c.kv_slots > 2 * MAXB)) return "kv_slots 0 (= max_batch) or max_batch..256";
    if (c.max_ctx < 64 || c.max_ctx % 64) return "max_ctx multiple of 64";
    if (c.lm_weight_dtype < 0 || c.lm_weight_dtype > 2) return "lm_weight_dtype 0 (bf16), 1 (fp8 e4m3 weights, per-channel scale) or 2 (1 + MX fp8 activations in prefill)";
    if (c.lm_weight_dtype == 2 && (c.t_hidden % 256 || ((c.t_heads + 2 * c.t_kv_heads) * 128) % 256 || c.t_hidden / 128 < 2))
        return "lm_weight_dtype 2: the block-scaled fp8 GEMM needs hidden and q/k/v widths in multiples of 256";
    if (c.lm_weight_dtype == 2 && ((c.t_inter + 63) / 64 * 64) % 128)
        return "lm_weight_dtype 2: the MX activation quantiser needs the padded intermediate size (t_inter rounded up to 64) in multiples of 128";
    if (c.max_patches < 4 || c.max_patches % 4 || c.max_prefill_tokens < 1 || c.max_new_tokens < 1) return "capacities";
    if (c.v_n_fullatt < 0 || c.v_n_fullatt > 16 || c.v_depth < 1 || c.t_layers < 1) return "depths";
    return nullptr;
}

// Carves every buffer.  With ar.base == nullptr this is a dry run that only measures.
void carve(sr_engine* e) {
    const sr_config& c = e->c;
    Arena& ar = e->ar;
    e->v_hd = c.v_hidden / c.v_heads;
    e->v_paired = (e->v_hd / 2) % 8 == 0 && e->v_hd % 16 == 0;
    e->v_pd = c.v_in_ch * c.v_temporal * c.v_patch * c.v_patch;
    e->v_pd_pad = rup(e->v_pd, 64);
    e->v_inter_pad = rup(c.v_inter, 64);
    e->v_mh = c.v_hidden * c.v_merge * c.v_merge;
    e->t_inter_pad = rup(c.t_inter, 64);
    e->t_qn = (c.t_heads + 2 * c.t_kv_heads) * 128;
    e->t_group = c.t_heads / c.t_kv_heads;
    e->ks_down = (e->t_inter_pad / 64 >= 16) ? 2 : 1;          // batches above 16 use 4 slabs (32-row GEMV variant)
    e->n_part = gemv_f32_blocks(c.t_vocab, 1, c.t_hidden, 1);   // upper bound (16-row tiles); per-launch count below
    const int C = c.v_hidden, H = c.t_hidden;

    e->weights_begin = (ar.off + 255) & ~(size_t)255;
    e->patch_w = ar.take<bf16_t>((size_t)C * e->v_pd_pad);
    e->vb.resize(c.v_depth);
    for (auto& b : e->vb) {
        b.norm1 = ar.take<bf16_t>(C);
        b.qkv_w = ar.take<bf16_t>((size_t)3 * C * C);
        b.qkv_b = ar.take<bf16_t>(3 * C);
        b.proj_w = ar.take<bf16_t>((size_t)C * C);
        b.proj_b = ar.take<bf16_t>(C);
        b.norm2 = ar.take<bf16_t>(C);
        b.gu_w = ar.take<bf16_t>((size_t)2 * e->v_inter_pad * C);
        b.gu_b = ar.take<bf16_t>(2 * e->v_inter_pad);
        b.down_w = ar.take<bf16_t>((size_t)C * e->v_inter_pad);
        b.down_b = ar.take<bf16_t>(C);
    }
    e->ln_q = ar.take<bf16_t>(C);
    e->fc1_w = ar.take<bf16_t>((size_t)e->v_mh * e->v_mh);
    e->fc1_b = ar.take<bf16_t>(e->v_mh);
    e->fc2_w = ar.take<bf16_t>((size_t)c.v_out_hidden * e->v_mh);
    e->fc2_b = ar.take<bf16_t>(c.v_out_hidden);
    e->embed = ar.take<bf16_t>((size_t)c.t_vocab * H);
    e->final_norm = ar.take<bf16_t>(H);
    e->ll.resize(c.t_layers);
    for (auto& l : e->ll) {
        l.ln1 = ar.take<bf16_t>(H);
        l.qkv_w = ar.take<bf16_t>((size_t)e->t_qn * H);
        l.qkv_b = ar.take<bf16_t>(e->t_qn);
        l.o_w = ar.take<bf16_t>((size_t)H * c.t_heads * 128);
        l.ln2 = ar.take<bf16_t>(H);
        l.gu_w = ar.take<bf16_t>((size_t)2 * e->t_inter_pad * H);
        l.down_w = ar.take<bf16_t>((size_t)H * e->t_inter_pad);
        if (c.lm_weight_dtype >= 1) {
            l.qkv_w8 = ar.take<unsigned char>((size_t)e->t_qn * H);
            l.o_w8 = ar.take<unsigned char>((size_t)H * c.t_heads * 128);
            l.gu_w8 = ar.take<unsigned char>((size_t)2 * e->t_inter_pad * H);
            l.down_w8 = ar.take<unsigned char>((size_t)H * e->t_inter_pad);
            l.qkv_s = ar.take<float>(e->t_qn);
            l.o_s = ar.take<float>(H);
            l.gu_s = ar.take<float>(2 * e->t_inter_pad);
            l.down_s = ar.take<float>(H);
        }
    }
    e->lut = ar.take<bf16_t>(3 * 256);
    e->inv_freq = ar.take<float>(64);
    e->rope_cos = ar.take<bf16_t>((size_t)(c.max_ctx + 1) * 64);
    e->rope_sin = ar.take<bf16_t>((size_t)(c.max_ctx + 1) * 64);
    e->weights_end = ar.off;

    // ViT activations
    const size_t NP = c.max_patches;
    e->v_pix = ar.take<bf16_t>(NP * e->v_pd_pad);
    e->v_x = ar.take<bf16_t>(NP * C);
    e->v_xn = ar.take<bf16_t>(NP * C);
    e->v_qkv = ar.take<bf16_t>(NP * 3 * C);
    e->v_attn = ar.take<bf16_t>(NP * C);
    e->v_act = ar.take<bf16_t>(NP * e->v_inter_pad);
    e->v_vt_stride = rup((int)NP, 64) + 64;
    e->v_vt = ar.take<bf16_t>((size_t)C * e->v_vt_stride);
    e->v_m1 = ar.take<bf16_t>(NP / 4 * e->v_mh);
    e->v_cos = ar.take<float>(NP * (e->v_hd / 2));
    e->v_sin = ar.take<float>(NP * (e->v_hd / 2));
    e->v_rowmap_embed = ar.take<int>(NP);
    e->v_rowmap_merge = ar.take<int>(NP / 4);
    e->v_work_win = ar.take<AttnWork>(NP / 4 + 64);
    e->v_work_full = ar.take<AttnWork>(NP / 64 + 64);
    e->v_work_full128 = ar.take<AttnWork>(NP / 128 + 64);

    // LM prefill activations
    const size_t TP = c.max_prefill_tokens;
    e->t_x = ar.take<bf16_t>(TP * H);
    e->t_xn = ar.take<bf16_t>(TP * H);
    e->t_qkv = ar.take<bf16_t>(TP * e->t_qn);
    e->t_attn = ar.take<bf16_t>(TP * c.t_heads * 128);
    e->t_act = ar.take<bf16_t>(TP * e->t_inter_pad);
    if (c.lm_weight_dtype == 2) {      // MX activations of the fp8 x fp8 prefill GEMMs: element bytes + e8m0 block scales [K/128][rows_pad][4]
        e->t_rows_pad = (int)((TP + 255) / 256 * 256);
        const size_t kmax = std::max((size_t)e->t_inter_pad, std::max((size_t)H, (size_t)c.t_heads * 128));
        e->t_q8 = ar.take<unsigned char>(TP * kmax);
        e->t_qs = ar.take<unsigned char>((kmax / 128 + 1) * (size_t)e->t_rows_pad * 4);
    }
    e->t_src = ar.take<int>(TP);
    e->t_pos3 = ar.take<int>(3 * TP);
    e->t_slot = ar.take<int>(TP);
    e->t_idx = ar.take<int>(TP);
    e->t_lastrow = ar.take<int>(MAXB);
    e->t_work = ar.take<AttnWork>(TP / 64 + 64);
    e->t_slabs = ar.take<float>((size_t)SPLITK_KS * std::min<size_t>(TP, SPLITK_ROWS) * c.t_hidden);

    // decode
    const size_t B = c.max_batch;
    e->d_xa = ar.take<bf16_t>(B * H);
    e->d_xb = ar.take<bf16_t>(B * H);
    const size_t Bp = (B + 15) / 16 * 16;           // the fragment-ordered x buffers of the batch > 4 decode path hold whole 16-row groups
    e->d_xn = ar.take<bf16_t>(Bp * H);
    e->d_qkv = ar.take<bf16_t>(B * e->t_qn);
    e->d_attn = ar.take<bf16_t>(Bp * c.t_heads * 128);
    e->d_act = ar.take<bf16_t>(Bp * e->t_inter_pad);
    e->d_scores = ar.take<bf16_t>(B * c.t_heads * (size_t)c.max_ctx);
    e->d_logits = ar.take<float>(B * c.t_vocab);
    e->d_slabs = ar.take<float>(4 * B * H);
    e->d_amax_val = ar.take<float>(B * e->n_part);
    e->d_amax_idx = ar.take<int>(B * e->n_part);
    e->d_logits_adm = ar.take<float>(B * c.t_vocab);
    e->d_amax_val_adm = ar.take<float>(B * e->n_part);
    e->d_amax_idx_adm = ar.take<int>(B * e->n_part);
    e->d_row_limit = ar.take<int>(MAXB);
    e->d_row_cs = ar.take<float>(MAXB * 128);
    e->d_px = ar.take<unsigned>(256);
    e->d_ngen = ar.take<int>(MAXB);
    e->d_adm = ar.take<int>(5 * MAXB);
    e->d_adm_slots = ar.take<int>(MAXB);
    e->d_xadm = ar.take<bf16_t>(B * H);           // admission scratch of its own: an admission may run on another stream while rows decode
    e->d_xadm_n = ar.take<bf16_t>((size_t)std::max<size_t>(32, Bp) * H);      // (the admission's 32-row LM-head GEMV addresses two 16-row groups whatever the batch)
    e->d_sampled = ar.take<long long>(MAXB);
    e->d_adm_pick = ar.take<long long>(MAXB);
    e->seen_words = (c.t_vocab + 31) / 32;
    e->d_seen = ar.take<unsigned>((size_t)MAXB * e->seen_words);
    e->d_chosen = ar.take<long long>(MAXB);
    e->d_next = ar.take<long long>(MAXB);
    e->d_cur_tok = ar.take<int>(MAXB);
    e->d_ctx_len = ar.take<int>(MAXB);
    e->d_pos = ar.take<int>(MAXB);
    e->d_finished = ar.take<int>(MAXB);
    e->d_step = ar.take<int>(MAXB);
    e->d_slots = ar.take<int>(MAXB);
    e->d_eos = ar.take<int>(32);
    e->d_tokens = ar.take<int>(B * c.max_new_tokens);
    e->n_slots = c.kv_slots ? c.kv_slots : c.max_batch;
    e->kv_layer_elems = (size_t)e->n_slots * c.t_kv_heads * (size_t)c.max_ctx * 128;
    e->kcache = ar.take<bf16_t>(e->kv_layer_elems * c.t_layers);
    e->vtcache = ar.take<bf16_t>(e->kv_layer_elems * c.t_layers);

    // control staging: ViT needs NP*(8 + 4*hd) + work lists; prefill needs ~24 B per token + work lists
    e->stage_bytes = NP * (8 + 4 * (size_t)e->v_hd) + (NP / 4 + NP / 64 + NP / 128 + 192) * sizeof(AttnWork) + TP * 28 +
                     (TP / 64 + 64) * sizeof(AttnWork) + 4096 + 5 * MAXB * sizeof(int);
    e->d_stage = ar.take<char>(e->stage_bytes);
}

void register_expected(sr_engine* e) {
    const sr_config& c = e->c;
    char n[160];
    auto add = [&](const char* s) { e->loaded[s] = false; };
    add("visual.patch_embed.proj.weight");
    for (int i = 0; i < c.v_depth; ++i)
        for (const char* s : {"norm1.weight", "norm2.weight", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight",
                              "attn.proj.bias", "mlp.gate_proj.weight", "mlp.gate_proj.bias", "mlp.up_proj.weight",
                              "mlp.up_proj.bias", "mlp.down_proj.weight", "mlp.down_proj.bias"}) {
            snprintf(n, sizeof n, "visual.blocks.%d.%s", i, s);
            add(n);
        }
    for (const char* s : {"visual.merger.ln_q.weight", "visual.merger.mlp.0.weight", "visual.merger.mlp.0.bias",
                          "visual.merger.mlp.2.weight", "visual.merger.mlp.2.bias", "model.embed_tokens.weight",
                          "model.norm.weight"})
        add(s);
    for (int i = 0; i < c.t_layers; ++i)
        for (const char* s : {"input_layernorm.weight", "post_attention_layernorm.weight", "self_attn.q_proj.weight",
                              "self_attn.q_proj.bias", "self_attn.k_proj.weight", "self_attn.k_proj.bias",
                              "self_attn.v_proj.weight", "self_attn.v_proj.bias", "self_attn.o_proj.weight",
                              "mlp.gate_proj.weight", "mlp.up_proj.weight", "mlp.down_proj.weight"}) {
            snprintf(n, sizeof n, "model.layers.%d.%s", i, s);
            add(n);
        }
}

// waits until the previous control upload has left the pinned staging buffer
int stage_begin(sr_engine* e) {
    if (e->copy_pending) {
        hipError_t r = hipEventSynchronize(e->ev_copy);
        if (r != hipSuccess) return (int)r;
        e->copy_pending = false;
    }
    return 0;
}
int stage_upload(sr_engine* e, size_t off, size_t bytes, hipStream_t s) {
    hipError_t r = hipMemcpyAsync(e->d_stage + off, e->h_stage + off, bytes, hipMemcpyHostToDevice, s);
    if (r != hipSuccess) return (int)r;
    r = hipEventRecord(e->ev_copy, s);
    if (r != hipSuccess) return (int)r;
    e->copy_pending = true;
    return 0;
}

// ------------------------------------------------------------------------------------------------ ViT control data
// window index / cu_seqlens / rotary tables: hf vision_utils.get_vision_window_index (transformers/vision_utils.py:130-188),
// get_vision_position_ids, Qwen2_5_VisionRotaryEmbedding (hf:125-134), written from the definitions.
int vit_prepare(sr_engine* e, const int64_t* grid, int n_img, hipStream_t s, int* n_rows_out) {
    const sr_config& c = e->c;
    int N = 0;
    for (int i = 0; i < n_img; ++i) {
        if (grid[3 * i] != 1) return fail(e, -22, "grid t must be 1 (images only)");
        const int h = (int)grid[3 * i + 1], w = (int)grid[3 * i + 2];
        if (h <= 0 || w <= 0 || h % c.v_merge || w % c.v_merge) return fail(e, -22, "grid h,w must be multiples of merge");
        N += h * w;
    }
    if (N > c.max_patches) return fail(e, -22, "%d patches exceed max_patches %d", N, c.max_patches);
    *n_rows_out = N;
    std::vector<int64_t> key(grid, grid + 3 * n_img);
    if (key == e->v_grid_cached) return 0;

    SR_TRY(stage_begin(e));
    const int half = e->v_hd / 2, nf = half / 2;        // 40 rotary values per row = 20 h-freqs + 20 w-freqs
    const int mg = c.v_merge, unit = mg * mg;
    const int ws = c.v_window / mg / c.v_patch;          // window side in merged units (4)
    std::vector<float> inv(nf);
    for (int i = 0; i < nf; ++i) inv[i] = (float)(1.0 / pow(10000.0, (double)(2 * i) / (double)half));

    char* hp = e->h_stage;
    float* h_cos = reinterpret_cast<float*>(hp);
    float* h_sin = h_cos + (size_t)N * half;
    int* h_rm_embed = reinterpret_cast<int*>(h_sin + (size_t)N * half);
    int* h_rm_merge = h_rm_embed + N;
    AttnWork* h_win = reinterpret_cast<AttnWork*>(((uintptr_t)(h_rm_merge + N / unit) + 15) & ~(uintptr_t)15);
    int n_win = 0;
    std::vector<AttnWork> full, full128;
    bool aligned = true;         // every image starts on a multiple of 8 patches: 16-byte aligned V^T key runs (k_attn_prefill2's LDS-DMA)
    bool all64 = true;           // every window holds exactly 64 tokens (k_attn_win64)

    int unit_base = 0, row_base = 0, new_unit = 0;
    for (int im = 0; im < n_img; ++im) {
        const int gh = (int)grid[3 * im + 1], gw = (int)grid[3 * im + 2];
        const int lh = gh / mg, lw = gw / mg;
        const int nh = (lh + (ws - lh % ws)) / ws, nw = (lw + (ws - lw % ws)) / ws;   // HF pads a full window when aligned
        for (int wy = 0; wy < nh; ++wy)
            for (int wx = 0; wx < nw; ++wx) {
                const int start_unit = new_unit;
                for (int iy = 0; iy < ws; ++iy)
                    for (int ix = 0; ix < ws; ++ix) {
                        const int uy = wy * ws + iy, ux = wx * ws + ix;
                        if (uy >= lh || ux >= lw) continue;
                        const int old_unit = unit_base + uy * lw + ux;
                        h_rm_merge[new_unit] = old_unit;                      // merger output row of window-order unit
                        for (int k = 0; k < unit; ++k) {
                            const int old_row = old_unit * unit + k, new_row = new_unit * unit + k;
                            h_rm_embed[old_row] = new_row;
                            const int py = uy * mg + k / mg, px = ux * mg + k % mg;     // patch (h, w) position
                            for (int f = 0; f < nf; ++f) {
                                const float ah = (float)py * inv[f], aw = (float)px * inv[f];
                                h_cos[(size_t)new_row * half + f] = cosf(ah);
                                h_sin[(size_t)new_row * half + f] = sinf(ah);
                                h_cos[(size_t)new_row * half + nf + f] = cosf(aw);
                                h_sin[(size_t)new_row * half + nf + f] = sinf(aw);
                            }
                        }
                        ++new_unit;
                    }
                const int len = (new_unit - start_unit) * unit;
                // (HF pads one whole extra window row / column when the grid IS a multiple of the window: those windows are empty and make no work item)
                if (len != 0 && (len != 64 || (start_unit * unit) % 4)) all64 = false;
                for (int q0 = 0; q0 < len; q0 += 64)
                    h_win[n_win++] = AttnWork{start_unit * unit + q0, len, q0, start_unit * unit, (long long)start_unit * unit};
            }
        const int len = gh * gw;
        for (int q0 = 0; q0 < len; q0 += 64) full.push_back(AttnWork{row_base + q0, len, q0, row_base, (long long)row_base});
        for (int q0 = 0; q0 < len; q0 += 128) full128.push_back(AttnWork{row_base + q0, len, q0, row_base, (long long)row_base});
        if (row_base % 8) aligned = false;
        unit_base += lh * lw;
        row_base += len;
    }
    AttnWork* h_full = h_win + n_win;
    memcpy(h_full, full.data(), full.size() * sizeof(AttnWork));
    AttnWork* h_full128 = h_full + full.size();
    memcpy(h_full128, full128.data(), full128.size() * sizeof(AttnWork));
    const size_t total = reinterpret_cast<char*>(h_full128 + full128.size()) - hp;
    if (total > e->stage_bytes) return fail(e, -12, "control staging too small");
    SR_TRY(stage_upload(e, 0, total, s));
    // device-side views into the mirror; copied out so that later prefill uploads cannot clobber them
    auto dmirror = [&](const void* hptr) { return e->d_stage + (reinterpret_cast<const char*>(hptr) - hp); };
    SR_TRY((int)hipMemcpyAsync(e->v_cos, dmirror(h_cos), (size_t)N * half * 4, hipMemcpyDeviceToDevice, s));
    SR_TRY((int)hipMemcpyAsync(e->v_sin, dmirror(h_sin), (size_t)N * half * 4, hipMemcpyDeviceToDevice, s));
    SR_TRY((int)hipMemcpyAsync(e->v_rowmap_embed, dmirror(h_rm_embed), (size_t)N * 4, hipMemcpyDeviceToDevice, s));
    SR_TRY((int)hipMemcpyAsync(e->v_rowmap_merge, dmirror(h_rm_merge), (size_t)N / unit * 4, hipMemcpyDeviceToDevice, s));
    SR_TRY((int)hipMemcpyAsync(e->v_work_win, dmirror(h_win), n_win * sizeof(AttnWork), hipMemcpyDeviceToDevice, s));
    SR_TRY((int)hipMemcpyAsync(e->v_work_full, dmirror(h_full), full.size() * sizeof(AttnWork), hipMemcpyDeviceToDevice, s));
    e->v_nwork_win = n_win;
    SR_TRY((int)hipMemcpyAsync(e->v_work_full128, dmirror(h_full128), full128.size() * sizeof(AttnWork), hipMemcpyDeviceToDevice, s));
    e->v_nwork_full = (int)full.size();
    e->v_nwork_full128 = (int)full128.size();
    e->v_full_aligned = aligned;
    e->v_win_all64 = all64;
    e->v_grid_cached = key;
    return 0;
}

bool is_fullatt(const sr_config& c, int blk) {
    for (int i = 0; i < c.v_n_fullatt; ++i) if (c.v_fullatt[i] == blk) return true;
    return false;
}

int gemm(sr_engine* e, hipStream_t s, const bf16_t* A, int lda, const bf16_t* W, int M, int N, int K, void* out, int ldo,
         const bf16_t* bias, const bf16_t* resid, const int* rowmap, int epi, int w_tiled = 0, const float* w_scale = nullptr) {
    GemmArgs a{A, lda, W, M, N, K, out, ldo, bias, resid, rowmap, w_tiled, w_scale, 0};
    SR_TRY(launch_gemm(s, a, epi));
    return 0;
}

// Small prefills (one or two tiles: M = 448 .. 1024 rows): the residual GEMMs (o_proj N = 2048 / K = 2048, down N = 2048 / K = 11008) have
// only ceil(M / 64) * 16 = 112 .. 256 output tiles for 256 CUs and the down-projection walks 172 k-tiles in each of them (140 us of the
// 323 us layer at batch 1).  Split over K (gridDim.y = 4, float32 slabs; the RMSNorm launch that follows anyway sums the slabs into the
// residual stream, launch_resid_rmsnorm: no extra launch, same rounding points) the batch-1 prefill takes 8.0 instead of 11.7 ms.
// OPT-IN (SR_SPLITK=1), static prefill only: the split changes the association of the float32 sums, so a prompt prefilled alone would no
// longer get bit for bit the logits it gets inside a larger batch (the 128- and 256-tile kernels accumulate in the same order), and
// through 36 layers of bf16 rounding that is a noise-floor-sized difference (rms 0.03 on the logits, test_small_prefill_split_k_*):
// correct, but it gives up the batch invariance that the continuous-batching and data-parallel tests assert with torch.equal.
bool prefill_splitk(const sr_engine* e, int n_tok) {
    if (sr_switches().splitk != 1) return false;
    return e->c.lm_weight_dtype != 2 && n_tok <= SPLITK_ROWS && e->c.t_hidden <= 2048 && e->c.t_hidden % 512 == 0;
}

// one linear of the LM prefill.  lm_weight_dtype 0 / 1: MFMA GEMM on bf16 operands (mode 1 multiplies the bf16 image of the fp8-quantised
// weights and scales in the epilogue).  Mode 2 (BASELINE.json configs[4] "CDNA4 fp8 MFMA"): the activations are MX-quantised
// (k_quant_mx_act) and multiplied with the fp8 weight image by the block-scaled K = 128 MFMA -- ALWAYS, whatever the row count, so
// that a sequence's result does not depend on what else is in the batch.
int lm_gemm(sr_engine* e, hipStream_t s, const bf16_t* A, int lda, const bf16_t* W, const unsigned char* W8, const float* w_scale, int M,
            int N, int K, void* out, int ldo, const bf16_t* bias, const bf16_t* resid, int epi, const QkvRope* rope = nullptr, bool* fused = nullptr,
            int ksplit = 0) {
    if (e->c.lm_weight_dtype != 2) {
        GemmArgs a{A, lda, W, M, N, K, out, ldo, bias, resid, nullptr, 1, w_scale, 0};
        a.ksplit = ksplit;
        if (rope) {
            a.rope = *rope;
            if (gemm_fuses_lmqkv(a)) { epi = EPI_LMQKV; *fused = true; }
        }
        SR_TRY(launch_gemm(s, a, epi));
        return 0;
    }
    SR_TRY(launch_quant_mx_act(s, A, lda, M, K, e->t_q8, e->t_qs, e->t_rows_pad));
    GemmArgs a{reinterpret_cast<const bf16_t*>(e->t_q8), K, reinterpret_cast<const bf16_t*>(W8), M, N, K, out, ldo, bias, resid, nullptr, 1, w_scale, 256,
               e->t_qs, e->t_rows_pad};
    if (rope) {
        a.rope = *rope;
        if (gemm_fuses_lmqkv(a)) { epi = EPI_LMQKV; *fused = true; }
    }
    SR_TRY(launch_gemm256_mx(s, a, epi));
    return 0;
}

GemvArgs gv(const bf16_t* x, int ldx, const bf16_t* W, int M, int N, int K, void* out, int ldo) {
    GemvArgs a{};
    a.x = x; a.ldx = ldx; a.W = W; a.M = M; a.N = N; a.K = K; a.out = out; a.ldo = ldo; a.ksplit = 1;
    a.w_tiled = 1;      // every LM matrix of the engine is stored fragment-ordered
    return a;
}

// RMSNorms (and the pending residual add) live in the prologue of the consuming GEMV for batches <= 4 (a few KB per
// block); for larger batches every block would have to ingest B rows (+ float32 slabs) before it can start, which was
// measured slower than two small RMSNorm launches spread over the chip
bool fused_norms(const sr_engine* e, int B) { return B <= 4 && e->c.t_hidden % 512 == 0; }
// fragment-ordered activations between the launches of the batch > 4 decode layer (x_tiled): needs whole 64-wide k chunks
bool x_tiled_ok(const sr_engine* e) {
    return e->c.t_hidden % 64 == 0 && (e->c.t_heads * 128) % 64 == 0 && e->t_inter_pad % 64 == 0 && e->c.t_hidden <= 2048;
}
int ks_down(const sr_engine* e, int B) { return (B > 16 && e->t_inter_pad / 64 >= 32) ? 4 : e->ks_down; }

// LM head on B rows of `x` (+ optional pending slabs): float32 logits + per-block argmax partials.
// Small batches fuse the final RMSNorm (and the pending residual) into the GEMV prologue.
int enqueue_lm_head(sr_engine* e, int B, bf16_t* x, bf16_t* x_alt, bool pending, hipStream_t s, bool admission = false) {
    const sr_config& c = e->c;
    const int H = c.t_hidden;
    GemvArgs g = gv(x, H, e->embed, B, c.t_vocab, H, admission ? e->d_logits_adm : e->d_logits, c.t_vocab);
    g.amax_val = admission ? e->d_amax_val_adm : e->d_amax_val; g.amax_idx = admission ? e->d_amax_idx_adm : e->d_amax_idx;
    if (fused_norms(e, B)) {
        g.norm_w = e->final_norm; g.eps = c.t_rms_eps;
        if (pending) { g.slabs = e->d_slabs; g.n_slabs = ks_down(e, B); g.x_out = x_alt; }
    } else {
        // batches > 4: the normalised rows are written fragment-ordered, so the GEMV's x loads are 1 KB contiguous
        const int xt = x_tiled_ok(e) ? 1 : 0;
        if (pending) SR_TRY(launch_resid_rmsnorm(s, x, e->d_slabs, ks_down(e, B), e->final_norm, e->d_xn, B, H, c.t_rms_eps, xt));
        else SR_TRY(launch_rmsnorm(s, x, e->final_norm, e->d_xn, B, H, c.t_rms_eps, xt));
        g.x = e->d_xn; g.x_tiled = xt;
        // on a CU-limited stream the decode step's head keeps x in LDS (k_gemv32_hpx: same bits; the admission's own head stays the streaming kernel)
        if (!admission && (e->px_mode || (sr_switches().gemv_xlds & 4))) g.px_counter = e->d_px;
    }
    SR_TRY(launch_gemv(s, g, GV_F32));
    return 0;
}

// one decode forward pass for rows 0..B-1 (device state decides tokens / positions / context lengths).
// 6 launches per layer at batch <= 4 (qkv, attention scores, attention softmax+PV, o_proj, gate/up, down): norms and
// residual adds live in GEMV prologues / epilogues; larger batches add the two RMSNorm launches.
int enqueue_decode_forward(sr_engine* e, int B, hipStream_t s) {
    const sr_config& c = e->c;
    const int H = c.t_hidden, QD = c.t_heads * 128;
    const bool fused = fused_norms(e, B);
    const int xt = (!fused && x_tiled_ok(e)) ? 1 : 0;      // activations handed from launch to launch in fragment order
    // the x-stationary gate/up and down GEMVs (17..32 rows): on a CU-limited stream (sr_rows_set_cus) -- on the whole chip the streaming kernels are 1 % faster per step;
    // bit 2 of SR_GEMV_XLDS: everywhere (A/B and test hook)
    const bool px = e->px_mode || (sr_switches().gemv_xlds & 4);
    bf16_t *x = e->d_xa, *x_alt = e->d_xb;      // k_step gathered the input embedding into d_xa
    bool pending = false;                       // down-projection slabs not yet added to the residual stream
    const float scale = (float)(1.0 / sqrt(128.0));
    for (int l = 0; l < c.t_layers; ++l) {
        const LmLayerW& w = e->ll[l];
        bf16_t* kc = e->kcache + (size_t)l * e->kv_layer_elems;
        bf16_t* vc = e->vtcache + (size_t)l * e->kv_layer_elems;
        GemvArgs gq = gv(x, H, w.qkv_w, B, e->t_qn, H, e->d_qkv, e->t_qn);
        gq.bias = w.qkv_b; gq.W8 = w.qkv_w8; gq.w_scale = w.qkv_s;
        if (fused) {
            gq.norm_w = w.ln1; gq.eps = c.t_rms_eps;
            if (pending) { gq.slabs = e->d_slabs; gq.n_slabs = ks_down(e, B); gq.x_out = x_alt; }
        } else {
            if (pending) SR_TRY(launch_resid_rmsnorm(s, x, e->d_slabs, ks_down(e, B), w.ln1, e->d_xn, B, H, c.t_rms_eps, xt));
            else SR_TRY(launch_rmsnorm(s, x, w.ln1, e->d_xn, B, H, c.t_rms_eps, xt));
            gq.x = e->d_xn; gq.x_tiled = xt;
        }
        SR_TRY(launch_gemv(s, gq, GV_BIAS));
        if (fused && pending) { bf16_t* t = x; x = x_alt; x_alt = t; }     // block 0 wrote the updated stream there
        pending = false;
        DecodeAttnArgs da{e->d_qkv, e->t_qn, e->d_pos, e->d_ctx_len, e->d_slots, e->rope_cos, e->rope_sin, kc, vc, e->d_attn, QD,
                          B, c.t_heads, c.t_kv_heads, e->t_group, c.max_ctx, scale, e->d_scores, xt, e->d_finished, e->d_row_cs};
        SR_TRY(launch_attn_decode(s, da));
        GemvArgs go = gv(e->d_attn, QD, w.o_w, B, H, QD, x, H);
        go.W8 = w.o_w8; go.w_scale = w.o_s; go.x_tiled = xt;
        SR_TRY(launch_gemv(s, go, GV_RESID));
        GemvArgs gg = gv(x, H, w.gu_w, B, 2 * e->t_inter_pad, H, e->d_act, e->t_inter_pad);
        gg.W8 = w.gu_w8; gg.w_scale = w.gu_s;
        if (fused) { gg.norm_w = w.ln2; gg.eps = c.t_rms_eps; }
        else if (B > 64 && !w.gu_w8) {
            // more than 64 rows: gate/up as an LDS-tiled MFMA GEMM (gemm.hip, 64 x 128 tiles: every block stages its x tile ONCE for its four
            // waves).  The row-group GEMV reads G x the weight bytes of x from L2 per wave (43 us at 128 rows; decode step 4.50 -> 4.19 ms with
            // the tile kernel); the other launches keep the GEMV (their narrow matrices would give the tile kernel 32-40 blocks), and at 64 rows
            // the GEMV is as fast (3.24 vs 3.29 ms per step).  Row-major x in, fragment-ordered activation out (the down GEMV's x).
            SR_TRY(launch_rmsnorm(s, x, w.ln2, e->d_xn, B, H, c.t_rms_eps, 0));
            GemmArgs ga{e->d_xn, H, w.gu_w, B, 2 * e->t_inter_pad, H, e->d_act, e->t_inter_pad, nullptr, nullptr, nullptr, 1, nullptr, 128};
            ga.out_tiled = xt;
            SR_TRY(launch_gemm(s, ga, EPI_SWIGLU));
            gg.M = 0;       // (done)
        }
        else {
            SR_TRY(launch_rmsnorm(s, x, w.ln2, e->d_xn, B, H, c.t_rms_eps, xt));
            gg.x = e->d_xn; gg.x_tiled = xt; gg.out_tiled = xt;
            gg.px_counter = px ? e->d_px : nullptr;        // 17..32 rows: launch_gemv may take the persistent x-resident kernel (same bits)
        }
        if (gg.M > 0) SR_TRY(launch_gemv(s, gg, GV_SWIGLU));
        GemvArgs gd = gv(e->d_act, e->t_inter_pad, w.down_w, B, H, e->t_inter_pad, e->d_slabs, H);
        gd.ksplit = ks_down(e, B);
        gd.W8 = w.down_w8; gd.w_scale = w.down_s; gd.x_tiled = xt;
        gd.px_counter = px ? e->d_px : nullptr;            // 17..32 rows, bf16: the x-stationary split-K kernel (same bits)
        SR_TRY(launch_gemv(s, gd, GV_PARTIAL));
        pending = true;
    }
    return enqueue_lm_head(e, B, x, x_alt, pending, s, false);
}

int enqueue_step(sr_engine* e, int B, int n_eos, int pad_id, const int* forced, hipStream_t s, const long long* chosen = nullptr) {
    StepArgs a{e->d_amax_val, e->d_amax_idx, gemv_f32_blocks(e->c.t_vocab, B, e->c.t_hidden, fused_norms(e, B) ? 1 : 0), e->d_cur_tok, e->d_ctx_len, e->d_pos, e->d_step, e->d_finished,
               e->d_tokens, e->c.max_new_tokens, e->d_eos, n_eos, pad_id, B, forced, e->embed, e->d_xa, e->c.t_hidden, 1, chosen, e->d_row_limit, e->d_ngen,
               e->rope_cos, e->rope_sin, e->d_row_cs};
    SR_TRY(launch_step(s, a));
    return 0;
}

}  // namespace

// counter block of the x-stationary GEMVs for the op-level entry points (engines own theirs): 8 ticket shards + done count + the CU limit, 64 B apart
static unsigned* g_op_px = nullptr;
static int g_op_px_mode = 0;         // as sr_engine::px_mode: the x-stationary forms after sr_op_gemv_set_cus(n > 0) (or bit 2 of SR_GEMV_XLDS)
static int op_px() {
    if (g_op_px) return 0;
    if (hipMalloc(&g_op_px, 256 * sizeof(unsigned)) != hipSuccess || hipMemset(g_op_px, 0, 256 * sizeof(unsigned)) != hipSuccess) return -12;
    return 0;
}

// =================================================================================================== C ABI
extern "C" {

int sr_version(void) { return 1; }

int sr_switches_reload(void) { load_switches(); return 0; }

const char* sr_last_error(const sr_engine* e) { return e ? e->err : g_err; }

size_t sr_workspace_bytes(const sr_config* cfg) {
    if (!cfg || validate(*cfg)) return 0;
    sr_engine tmp;
    tmp.c = *cfg;
    carve(&tmp);
    return tmp.ar.off + 4096;
}

int sr_engine_create(const sr_config* cfg, void* workspace, size_t workspace_bytes, sr_engine** out) {
    if (!cfg || !workspace || !out) return fail(nullptr, -22, "null argument");
    if (const char* why = validate(*cfg)) return fail(nullptr, -22, "invalid sr_config: %s", why);
    if ((uintptr_t)workspace & 255) return fail(nullptr, -22, "workspace must be 256-byte aligned");
    sr_engine* e = new sr_engine();
    e->c = *cfg;
    load_switches();                            // the environment is read here (and by sr_switches_reload), never in per-call dispatch
    snprintf(e->err, sizeof e->err, "ok");
    {   // the engine belongs to the device that owns the workspace, whatever the calling thread's current device is
        hipPointerAttribute_t pa{};
        int cur = 0;
        (void)hipGetDevice(&cur);
        e->device = (hipPointerGetAttributes(&pa, workspace) == hipSuccess) ? pa.device : cur;
        (void)hipGetLastError();
        (void)hipSetDevice(e->device);
    }
    e->ar.base = static_cast<char*>(workspace);
    e->ar.cap = workspace_bytes;
    carve(e);
    if (e->ar.off > workspace_bytes) {
        size_t need = e->ar.off;
        delete e;
        return fail(nullptr, -12, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    }
    register_expected(e);
    hipError_t r = hipHostMalloc(reinterpret_cast<void**>(&e->h_stage), e->stage_bytes, hipHostMallocDefault);
    if (r != hipSuccess) { delete e; return fail(nullptr, (int)r, "hipHostMalloc: %s", hipGetErrorString(r)); }
    r = hipEventCreateWithFlags(&e->ev_copy, hipEventDisableTiming);
    if (r != hipSuccess) { (void)hipHostFree(e->h_stage); delete e; return fail(nullptr, (int)r, "hipEventCreate"); }
    r = hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking);
    if (r != hipSuccess) { (void)hipEventDestroy(e->ev_copy); (void)hipHostFree(e->h_stage); delete e; return fail(nullptr, (int)r, "hipStreamCreate"); }
    // zero weights (padding must be 0), KV cache (0 * stale must stay finite) and decode state
    r = hipMemset(e->ar.base + e->weights_begin, 0, e->weights_end - e->weights_begin);
    if (r == hipSuccess) r = hipMemset(e->kcache, 0, e->kv_layer_elems * e->c.t_layers * sizeof(bf16_t));
    if (r == hipSuccess) r = hipMemset(e->vtcache, 0, e->kv_layer_elems * e->c.t_layers * sizeof(bf16_t));
    if (r == hipSuccess) r = hipMemset(e->d_cur_tok, 0, (char*)e->d_tokens - (char*)e->d_cur_tok);
    if (r == hipSuccess) r = hipMemset(e->d_px, 0, 256 * sizeof(unsigned));
    if (r == hipSuccess) r = hipMemset(e->v_vt, 0, (size_t)e->c.v_hidden * e->v_vt_stride * sizeof(bf16_t));
    // normalise LUT (hf image_transforms.py:89-124, 384-440) and rotary inverse frequencies (hf:506-523)
    std::vector<bf16_t> lut(768);
    const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f}, stdv[3] = {0.26862954f, 0.26130258f, 0.27577711f};
    for (int ch = 0; ch < 3; ++ch)
        for (int u = 0; u < 256; ++u) {
            const float x = (float)((double)u * (1.0 / 255.0));
            lut[ch * 256 + u] = host_f2bf((x - mean[ch]) / stdv[ch]);
        }
    std::vector<float> inv(64);
    for (int i = 0; i < 64; ++i) inv[i] = (float)(1.0 / pow((double)e->c.t_rope_theta, (double)(2 * i) / 128.0));
    if (r == hipSuccess) r = hipMemcpy(e->lut, lut.data(), 768 * sizeof(bf16_t), hipMemcpyHostToDevice);
    if (r == hipSuccess) r = hipMemcpy(e->inv_freq, inv.data(), 64 * sizeof(float), hipMemcpyHostToDevice);
    if (r == hipSuccess) r = (hipError_t)launch_rope_table(nullptr, e->inv_freq, e->c.max_ctx + 1, e->rope_cos, e->rope_sin);
    if (r == hipSuccess) r = hipDeviceSynchronize();
    if (r == hipSuccess) r = (hipError_t)attn_decode_prepare(e->c.max_ctx, e->t_group);
    if (r == hipSuccess && e->c.max_batch > 64) r = (hipError_t)gemm_prepare_decode();
    if (r == hipSuccess) r = (hipError_t)gemv_prepare_px();
    if (r != hipSuccess) {
        int rc = fail(nullptr, (int)r, "engine init: %s", hipGetErrorString(r));
        sr_engine_destroy(e);
        return rc;
    }
    *out = e;
    return 0;
}

int sr_engine_destroy(sr_engine* e) {
    enter(e);
    if (!e) return 0;
    for (auto g : e->graph) if (g) (void)hipGraphExecDestroy(g);
    for (auto g : e->step_graph) if (g) (void)hipGraphExecDestroy(g);
    if (e->sgraph) (void)hipGraphExecDestroy(e->sgraph);
    for (auto& r : e->rgraph) if (r.g) (void)hipGraphExecDestroy(r.g);
    if (e->cap_stream) (void)hipStreamDestroy(e->cap_stream);
    if (e->ev_copy) (void)hipEventDestroy(e->ev_copy);
    if (e->h_stage) (void)hipHostFree(e->h_stage);
    delete e;
    return 0;
}

int sr_weights_missing(const sr_engine* e, char* first_missing, size_t cap) {
    int n = 0;
    for (const auto& kv : e->loaded)
        if (!kv.second) {
            if (n == 0 && first_missing && cap) snprintf(first_missing, cap, "%s", kv.first.c_str());
            ++n;
        }
    return n;
}

int sr_synth_fill(void* dev_out_bf16, int64_t n, const char* hf_name, uint32_t seed, float base, void* stream) {
    const uint32_t key = mix32h(fnv1a32(hf_name) ^ mix32h(seed + 0x9E3779B9u));
    int rc = launch_synth_fill((hipStream_t)stream, (bf16_t*)dev_out_bf16, n, key, base, 1.3531647e-4f);
    if (rc) return fail(nullptr, rc, "synth_fill launch failed (%d)", rc);
    return 0;
}

int sr_load_weight(sr_engine* e, const char* hf_name, const void* p, int dtype, const int64_t* shape, int ndim, void* stream) {
    enter(e);
    if (!e || !hf_name || !p || !shape || ndim < 1) return fail(e, -22, "sr_load_weight: null argument");
    if (dtype != SR_DTYPE_BF16 && dtype != SR_DTYPE_F32) return fail(e, -22, "sr_load_weight: dtype");
    hipStream_t s = (hipStream_t)stream;
    const sr_config& c = e->c;
    std::string name = hf_name;
    if (name.rfind("model.visual.", 0) == 0) name = name.substr(6);
    else if (name.rfind("model.language_model.", 0) == 0) name = "model." + name.substr(21);
    if (name == "lm_head.weight") return 0;   // tied to model.embed_tokens.weight
    long long rows = shape[0], cols = 1;
    for (int i = 1; i < ndim; ++i) cols *= shape[i];
    const int C = c.v_hidden, H = c.t_hidden;
    bf16_t* dst = nullptr;
    long long ld = cols, exp_rows = -1, exp_cols = -1, row_off = 0;
    int mode = 0, tiled = 0;
    int idx = -1;
    char sub[96] = "";
    if (name == "visual.patch_embed.proj.weight") { dst = e->patch_w; ld = e->v_pd_pad; exp_rows = C; exp_cols = e->v_pd; }
    else if (sscanf(name.c_str(), "visual.blocks.%d.%95s", &idx, sub) == 2 && idx >= 0 && idx < c.v_depth) {
        VitBlockW& b = e->vb[idx];
        const std::string t = sub;
        if (t == "norm1.weight") { dst = b.norm1; exp_rows = C; exp_cols = 1; }
        else if (t == "norm2.weight") { dst = b.norm2; exp_rows = C; exp_cols = 1; }
        // q / k channels of every head in paired order (vit_qk_perm): rotary partners 8 apart inside one 16-column tile
        else if (t == "attn.qkv.weight") { dst = b.qkv_w; exp_rows = 3 * C; exp_cols = C; if (e->v_paired) { mode = 3; row_off = e->v_hd; } }
        else if (t == "attn.qkv.bias") { dst = b.qkv_b; exp_rows = 3 * C; exp_cols = 1; if (e->v_paired) { mode = 3; row_off = e->v_hd; } }
        else if (t == "attn.proj.weight") { dst = b.proj_w; exp_rows = C; exp_cols = C; }
        else if (t == "attn.proj.bias") { dst = b.proj_b; exp_rows = C; exp_cols = 1; }
        else if (t == "mlp.gate_proj.weight") { dst = b.gu_w; exp_rows = c.v_inter; exp_cols = C; mode = 1; }
        else if (t == "mlp.gate_proj.bias") { dst = b.gu_b; exp_rows = c.v_inter; exp_cols = 1; mode = 1; }
        else if (t == "mlp.up_proj.weight") { dst = b.gu_w; exp_rows = c.v_inter; exp_cols = C; mode = 2; }
        else if (t == "mlp.up_proj.bias") { dst = b.gu_b; exp_rows = c.v_inter; exp_cols = 1; mode = 2; }
        else if (t == "mlp.down_proj.weight") { dst = b.down_w; exp_rows = C; exp_cols = c.v_inter; ld = e->v_inter_pad; }
        else if (t == "mlp.down_proj.bias") { dst = b.down_b; exp_rows = C; exp_cols = 1; }
    }
    else if (name == "visual.merger.ln_q.weight") { dst = e->ln_q; exp_rows = C; exp_cols = 1; }
    else if (name == "visual.merger.mlp.0.weight") { dst = e->fc1_w; exp_rows = e->v_mh; exp_cols = e->v_mh; }
    else if (name == "visual.merger.mlp.0.bias") { dst = e->fc1_b; exp_rows = e->v_mh; exp_cols = 1; }
    else if (name == "visual.merger.mlp.2.weight") { dst = e->fc2_w; exp_rows = c.v_out_hidden; exp_cols = e->v_mh; }
    else if (name == "visual.merger.mlp.2.bias") { dst = e->fc2_b; exp_rows = c.v_out_hidden; exp_cols = 1; }
    else if (name == "model.embed_tokens.weight") { dst = e->embed; exp_rows = c.t_vocab; exp_cols = H; tiled = 1; }
    else if (name == "model.norm.weight") { dst = e->final_norm; exp_rows = H; exp_cols = 1; }
    else if (sscanf(name.c_str(), "model.layers.%d.%95s", &idx, sub) == 2 && idx >= 0 && idx < c.t_layers) {
        LmLayerW& l = e->ll[idx];
        const std::string t = sub;
        const int QD = c.t_heads * 128, KD = c.t_kv_heads * 128;
        int dm = -1, dbit = 0;          // fused matrix + constituent bit of an LM linear weight (fp8 re-quantisation bookkeeping)
        if (t == "input_layernorm.weight") { dst = l.ln1; exp_rows = H; exp_cols = 1; }
        else if (t == "post_attention_layernorm.weight") { dst = l.ln2; exp_rows = H; exp_cols = 1; }
        else if (t == "self_attn.q_proj.weight") { tiled = 1; dst = l.qkv_w; exp_rows = QD; exp_cols = H; dm = 0; dbit = 1; }
        else if (t == "self_attn.q_proj.bias") { dst = l.qkv_b; exp_rows = QD; exp_cols = 1; }
        else if (t == "self_attn.k_proj.weight") { tiled = 1; dst = l.qkv_w; exp_rows = KD; exp_cols = H; row_off = QD; dm = 0; dbit = 2; }
        else if (t == "self_attn.k_proj.bias") { dst = l.qkv_b; exp_rows = KD; exp_cols = 1; row_off = QD; }
        else if (t == "self_attn.v_proj.weight") { tiled = 1; dst = l.qkv_w; exp_rows = KD; exp_cols = H; row_off = QD + KD; dm = 0; dbit = 4; }
        else if (t == "self_attn.v_proj.bias") { dst = l.qkv_b; exp_rows = KD; exp_cols = 1; row_off = QD + KD; }
        else if (t == "self_attn.o_proj.weight") { tiled = 1; dst = l.o_w; exp_rows = H; exp_cols = QD; dm = 1; dbit = 1; }
        else if (t == "mlp.gate_proj.weight") { tiled = 1; dst = l.gu_w; exp_rows = c.t_inter; exp_cols = H; mode = 1; dm = 2; dbit = 1; }
        else if (t == "mlp.up_proj.weight") { tiled = 1; dst = l.gu_w; exp_rows = c.t_inter; exp_cols = H; mode = 2; dm = 2; dbit = 2; }
        else if (t == "mlp.down_proj.weight") { tiled = 1; dst = l.down_w; exp_rows = H; exp_cols = c.t_inter; ld = e->t_inter_pad; dm = 3; dbit = 1; }
        if (dm >= 0 && rows == exp_rows && cols == exp_cols) l.dirty[dm] |= (unsigned char)dbit;
    }
    if (!dst) return fail(e, -2, "sr_load_weight: unknown parameter '%s'", hf_name);
    if (rows != exp_rows || cols != exp_cols)
        return fail(e, -22, "sr_load_weight: '%s' has shape [%lld, %lld], expected [%lld, %lld]", hf_name, rows, cols, exp_rows, exp_cols);
    SR_TRY(launch_load2d(s, p, dtype, rows, cols, dst, ld, mode, row_off, tiled));
    e->loaded[name] = true;
    e->finalized = false;     // fp8 mode: the matrix has to be (re)quantised before the next forward
    return 0;
}

int sr_vit_plan(const sr_engine* e, int32_t* out4) {
    if (!e || !out4) return -22;
    out4[0] = e->v_nwork_win; out4[1] = e->v_nwork_full; out4[2] = e->v_win_all64 ? 1 : 0; out4[3] = e->v_full_aligned ? 1 : 0;
    return 0;
}
int sr_pixel_ld(const sr_engine* e) { return e->v_pd_pad; }

int sr_patchify_u8(sr_engine* e, const uint8_t* img, int h, int w, void* out, void* stream) {
    enter(e);
    const sr_config& c = e->c;
    const int f = c.v_patch * c.v_merge;
    if (c.v_in_ch != 3 || h % f || w % f || h <= 0 || w <= 0) return fail(e, -22, "sr_patchify_u8: h,w must be multiples of %d", f);
    SR_TRY(launch_patchify((hipStream_t)stream, img, h, w, e->lut, (bf16_t*)out, e->v_pd_pad, c.v_patch, c.v_merge, c.v_temporal));
    return 0;
}

int sr_vit_forward(sr_engine* e, const void* pixels, int pixels_dtype, const int64_t* grid, int n_img, void* out, void* stream) {
    enter(e);
    if (!e || !pixels || !grid || !out || n_img < 1) return fail(e, -22, "sr_vit_forward: null argument");
    char miss[160];
    if (sr_weights_missing(e, miss, sizeof miss)) return fail(e, -61, "weights missing, e.g. '%s'", miss);
    hipStream_t s = (hipStream_t)stream;
    const sr_config& c = e->c;
    int N = 0;
    if (int rc = vit_prepare(e, grid, n_img, s, &N)) return rc;
    const int C = c.v_hidden, hd = e->v_hd;
    const bf16_t* pix = static_cast<const bf16_t*>(pixels);
    if (pixels_dtype == SR_DTYPE_F32) {
        SR_TRY(launch_f32_to_bf16_pad(s, static_cast<const float*>(pixels), N, e->v_pd, e->v_pix, e->v_pd_pad));
        pix = e->v_pix;
    } else if (pixels_dtype != SR_DTYPE_BF16) return fail(e, -22, "pixels dtype");
    // patch embed (Conv3d k=s, no bias, hf:99-122) with the window permutation fused into the store (hf:433-441)
    if (int rc = gemm(e, s, pix, e->v_pd_pad, e->patch_w, N, C, e->v_pd_pad, e->v_x, C, nullptr, nullptr, e->v_rowmap_embed, EPI_STORE)) return rc;
    const float scale = (float)(1.0 / sqrt((double)hd));
    for (int blk = 0; blk < c.v_depth; ++blk) {
        const VitBlockW& w = e->vb[blk];
        SR_TRY(launch_rmsnorm(s, e->v_x, w.norm1, e->v_xn, N, C, 1e-6f, 0, 1));
        // qkv Linear; the 2-D rotary embedding and the V transpose ride in the GEMM epilogue when the 256-tile kernel takes the shape
        // (large batches; needs the paired q / k channel order the loader established), else they are the two launches of round 1
        {
            GemmArgs ga{e->v_xn, C, w.qkv_w, N, 3 * C, C, e->v_qkv, 3 * C, w.qkv_b, nullptr, nullptr, 0, nullptr, 0};
            ga.vrope = VitRope{e->v_cos, e->v_sin, e->v_vt, e->v_vt_stride, C, hd};
            const bool fused = e->v_paired && gemm_fuses_vitqkv(ga);
            SR_TRY(launch_gemm(s, ga, fused ? EPI_VITQKV : EPI_STORE));
            if (!fused) SR_TRY(launch_vit_rope(s, e->v_qkv, N, c.v_heads, hd, e->v_cos, e->v_sin, e->v_vt, e->v_vt_stride, e->v_paired ? 1 : 0));
        }
        const bool full = is_fullatt(c, blk);
        AttnArgs a{e->v_qkv, 3 * C, e->v_qkv + C, 3 * C, hd, e->v_vt, e->v_vt_stride, (long long)hd * e->v_vt_stride,
                   e->v_attn, C, full ? e->v_work_full : e->v_work_win, full ? e->v_nwork_full : e->v_nwork_win,
                   c.v_heads, 1, scale, 0, 64, 0};
        if (!full && e->v_win_all64 && e->v_vt_stride % 4 == 0) a.win64 = 1;     // one wave per (window, head), operands straight into registers
        if (full && e->v_full_aligned) {      // full attention: 128-query blocks that share LDS-DMA-staged tiles (k_attn_prefill2) where it applies
            AttnArgs a2 = a;
            a2.work = e->v_work_full128; a2.n_work = e->v_nwork_full128; a2.q_tile = 128; a2.v2_ok = 1;
            if (attn_prefill_variant(a2, hd) == 2) a = a2;
        }
        SR_TRY(launch_attn_prefill(s, a, hd));
        if (int rc = gemm(e, s, e->v_attn, C, w.proj_w, N, C, C, e->v_x, C, w.proj_b, e->v_x, nullptr, EPI_RESID)) return rc;
        SR_TRY(launch_rmsnorm(s, e->v_x, w.norm2, e->v_xn, N, C, 1e-6f, 0, 1));
        if (int rc = gemm(e, s, e->v_xn, C, w.gu_w, N, 2 * e->v_inter_pad, C, e->v_act, e->v_inter_pad, w.gu_b, nullptr, nullptr, EPI_SWIGLU)) return rc;
        if (int rc = gemm(e, s, e->v_act, e->v_inter_pad, w.down_w, N, C, e->v_inter_pad, e->v_x, C, w.down_b, e->v_x, nullptr, EPI_RESID)) return rc;
    }
    // merger (hf:137-150) with the inverse window permutation fused into the last store (hf:463-465)
    SR_TRY(launch_rmsnorm(s, e->v_x, e->ln_q, e->v_xn, N, C, 1e-6f, 0, 1));
    const int T = N / (c.v_merge * c.v_merge);
    if (int rc = gemm(e, s, e->v_xn, e->v_mh, e->fc1_w, T, e->v_mh, e->v_mh, e->v_m1, e->v_mh, e->fc1_b, nullptr, nullptr, EPI_GELU)) return rc;
    if (int rc = gemm(e, s, e->v_m1, e->v_mh, e->fc2_w, T, c.v_out_hidden, e->v_mh, out, c.v_out_hidden, e->fc2_b, nullptr, e->v_rowmap_merge, EPI_STORE)) return rc;
    return 0;
}

int sr_finalize_weights(sr_engine* e, void* stream) {
    enter(e);
    if (!e) return fail(e, -22, "sr_finalize_weights: null engine");
    if (e->finalized) return 0;
    {
        char first[160] = "";
        const int missing = sr_weights_missing(e, first, sizeof first);
        if (missing) return fail(e, -22, "sr_finalize_weights: %d parameters missing, e.g. %s", missing, first);
    }
    hipStream_t s = (hipStream_t)stream;
    if (e->c.lm_weight_dtype >= 1) {
        const int H = e->c.t_hidden, QD = e->c.t_heads * 128;
        // a fused matrix is re-quantised when ALL of its HF tensors were reloaded (a trainer -> engine weight sync sends every
        // parameter); a partial reload would mix fresh bf16 rows with rows that already hold quantised values
        static const unsigned char full[4] = {7, 1, 3, 1};
        for (size_t i = 0; i < e->ll.size(); ++i) {
            LmLayerW& l = e->ll[i];
            for (int m = 0; m < 4; ++m)
                if (l.dirty[m] != 0 && l.dirty[m] != full[m])
                    return fail(e, -22, "fp8 weights: layer %zu matrix %d was only partly reloaded (mask %d of %d)", i, m, l.dirty[m], full[m]);
            if (l.dirty[0]) SR_TRY(launch_quant_f8(s, l.qkv_w, e->t_qn, H, l.qkv_w8, l.qkv_s));
            if (l.dirty[1]) SR_TRY(launch_quant_f8(s, l.o_w, H, QD, l.o_w8, l.o_s));
            if (l.dirty[2]) SR_TRY(launch_quant_f8(s, l.gu_w, 2 * e->t_inter_pad, H, l.gu_w8, l.gu_s));
            if (l.dirty[3]) SR_TRY(launch_quant_f8(s, l.down_w, H, e->t_inter_pad, l.down_w8, l.down_s));
            l.dirty[0] = l.dirty[1] = l.dirty[2] = l.dirty[3] = 0;
        }
        SR_TRY((int)hipStreamSynchronize(s));
    }
    e->finalized = true;
    return 0;
}

// installs the staged admission's sequences into batch rows (row state + pending first token); runs on the DECODE stream, between steps
static int commit_admission(sr_engine* e, const int32_t* rows, int n, hipStream_t s) {
    const sr_config& c = e->c;
    // the staged sequences may be committed in several portions (rows free up one by one): `off` = how many are installed already
    const int off = e->staged_off;
    for (int i = 0; i < n; ++i) e->h_rows[off + i] = rows[i];
    SR_TRY((int)hipMemcpyAsync(e->d_adm + off, e->h_rows + off, (size_t)n * 4, hipMemcpyHostToDevice, s));
    const int MB = c.max_batch;
    AdmitArgs aa{e->d_adm + off, e->d_adm + MAXB + off, e->d_adm + 2 * MAXB + off, e->d_adm + 3 * MAXB + off, e->d_adm + 4 * MAXB + off, n,
                 e->d_ctx_len, e->d_pos, e->d_slots, e->d_finished, e->d_step, e->d_row_limit, e->d_ngen,
                 e->d_amax_val, e->d_amax_idx, gemv_f32_blocks(c.t_vocab, MB, c.t_hidden, fused_norms(e, MB) ? 1 : 0), e->d_adm_slots + off};
    SR_TRY(launch_admit_rows(s, aa));
    if (e->rows_temp > 0.f) SR_TRY(launch_scatter_rows(s, e->d_adm + off, e->d_adm_pick + off, e->d_sampled, n));   // the drawn first tokens
    e->staged_n -= n;
    e->staged_off = e->staged_n ? off + n : 0;
    return 0;
}

// `commit`: admission only -- install the rows right away (rows = slots: the plain sr_admit); false = stage, sr_admit_commit installs later
static int prefill_impl(sr_engine* e, const int64_t* ids, const int64_t* pos3, const int32_t* seq_lens, const int32_t* slots, int B,
                        const void* image_embeds, int n_image_rows, float* logits_out, void* stream, const int32_t* limits,
                        float* all_logits_out = nullptr, bool commit = true) {
    enter(e);
    if (!e || !ids || !pos3 || !seq_lens || !slots) return fail(e, -22, "sr_prefill: null argument");
    char miss[160];
    if (sr_weights_missing(e, miss, sizeof miss)) return fail(e, -61, "weights missing, e.g. '%s'", miss);
    const sr_config& c = e->c;
    if (c.lm_weight_dtype >= 1 && !e->finalized) return fail(e, -22, "fp8 weights: call sr_finalize_weights after loading");
    if (B < 1 || B > c.max_batch) return fail(e, -22, "sr_prefill: B=%d outside 1..%d", B, c.max_batch);
    hipStream_t s = (hipStream_t)stream;
    int n_tok = 0;
    for (int b = 0; b < B; ++b) {
        if (seq_lens[b] < 1 || seq_lens[b] + 1 > c.max_ctx)
            return fail(e, -22, "sequence %d length %d does not fit max_ctx %d", b, seq_lens[b], c.max_ctx);
        if (slots[b] < 0 || slots[b] >= e->n_slots) return fail(e, -22, "KV slot %d out of range 0..%d", slots[b], e->n_slots - 1);
        n_tok += seq_lens[b];
    }
    if (n_tok > c.max_prefill_tokens) return fail(e, -22, "%d prompt tokens exceed max_prefill_tokens %d", n_tok, c.max_prefill_tokens);

    // ---- control arrays (host) -> device
    SR_TRY(stage_begin(e));
    int* h_src = reinterpret_cast<int*>(e->h_stage);
    int* h_pos = h_src + n_tok;
    int* h_slot = h_pos + 3 * n_tok;
    int* h_idx = h_slot + n_tok;
    int* h_last = h_idx + n_tok;
    int* h_state = h_last + MAXB;           // [ctx_len | pos | slots | limit] x MAXB
    for (int i = 0; i < 4 * MAXB; ++i) h_state[i] = 0;
    AttnWork* h_work = reinterpret_cast<AttnWork*>(((uintptr_t)(h_state + 4 * MAXB) + 15) & ~(uintptr_t)15);
    int n_work = 0, img_row = 0, t0 = 0;
    const int KVH = c.t_kv_heads;
    for (int b = 0; b < B; ++b) {
        const int S = seq_lens[b];
        long long maxpos = 0;
        for (int i = 0; i < S; ++i) {
            const int t = t0 + i;
            const int64_t id = ids[t];
            if (id < 0 || id >= c.t_vocab) return fail(e, -22, "token id %lld out of range", (long long)id);
            // image placeholders consume the image feature rows in order (hf:1210-1216 masked_scatter)
            h_src[t] = (image_embeds && id == c.image_token_id) ? -(img_row++) - 1 : (int)id;
            for (int a = 0; a < 3; ++a) {
                const int64_t p = pos3[(size_t)a * n_tok + t];
                h_pos[(size_t)a * n_tok + t] = (int)p;
                if (p < 0 || p > c.max_ctx) return fail(e, -22, "position id %lld outside 0..max_ctx (%d)", (long long)p, c.max_ctx);
                if (p > maxpos) maxpos = p;
            }
            h_slot[t] = slots[b];
            h_idx[t] = i;
        }
        for (int q0 = 0; q0 < S; q0 += 64)
            h_work[n_work++] = AttnWork{t0 + q0, S, q0, slots[b] * KVH * c.max_ctx, (long long)slots[b] * KVH * 128 * c.max_ctx};
        h_last[b] = t0 + S - 1;
        h_state[b] = S;                     // keys in the cache; k_step_advance adds the new token before each forward
        h_state[MAXB + b] = (int)maxpos;      // decode positions continue at max+1 on all three axes
                                            // (reference rule: roll/utils/functionals.py:816-818)
        h_state[2 * MAXB + b] = slots[b];
        h_state[3 * MAXB + b] = limits ? limits[b] : 0x7f7f7f7f;
        t0 += S;
    }
    if (image_embeds && img_row != n_image_rows)
        return fail(e, -22, "image features and image tokens do not match: %d tokens, %d feature rows", img_row, n_image_rows);
    const size_t total = reinterpret_cast<char*>(h_work + n_work) - e->h_stage;
    if (total > e->stage_bytes) return fail(e, -12, "control staging too small");
    SR_TRY(stage_upload(e, 0, total, s));
    auto dmirror = [&](const void* hptr) { return e->d_stage + (reinterpret_cast<const char*>(hptr) - e->h_stage); };
    SR_TRY((int)hipMemcpyAsync(e->t_src, dmirror(h_src), (size_t)n_tok * 4, hipMemcpyDeviceToDevice, s));
    SR_TRY((int)hipMemcpyAsync(e->t_pos3, dmirror(h_pos), (size_t)n_tok * 12, hipMemcpyDeviceToDevice, s));
    SR_TRY((int)hipMemcpyAsync(e->t_slot, dmirror(h_slot), (size_t)n_tok * 4, hipMemcpyDeviceToDevice, s));
    SR_TRY((int)hipMemcpyAsync(e->t_idx, dmirror(h_idx), (size_t)n_tok * 4, hipMemcpyDeviceToDevice, s));
    SR_TRY((int)hipMemcpyAsync(e->t_lastrow, dmirror(h_last), MAXB * 4, hipMemcpyDeviceToDevice, s));
    SR_TRY((int)hipMemcpyAsync(e->t_work, dmirror(h_work), n_work * sizeof(AttnWork), hipMemcpyDeviceToDevice, s));
    if (!limits) {       // static batch: rows 0..B-1 are (re)initialised, all other rows are dropped
        SR_TRY((int)hipMemcpyAsync(e->d_ctx_len, dmirror(h_state), MAXB * 4, hipMemcpyDeviceToDevice, s));
        SR_TRY((int)hipMemcpyAsync(e->d_pos, dmirror(h_state + MAXB), MAXB * 4, hipMemcpyDeviceToDevice, s));
        SR_TRY((int)hipMemcpyAsync(e->d_slots, dmirror(h_state + 2 * MAXB), MAXB * 4, hipMemcpyDeviceToDevice, s));
        SR_TRY((int)hipMemsetAsync(e->d_finished, 0, MAXB * 4, s));
        SR_TRY((int)hipMemsetAsync(e->d_step, 0, MAXB * 4, s));
        SR_TRY((int)hipMemsetAsync(e->d_row_limit, 0x7f, MAXB * 4, s));
        SR_TRY((int)hipMemsetAsync(e->d_ngen, 0, MAXB * 4, s));
    } else {             // admission: [rows | ctx | pos | limit] for the new rows only; installed after the LM head below
        SR_TRY((int)hipMemcpyAsync(e->d_adm_slots, dmirror(h_state + 2 * MAXB), MAXB * 4, hipMemcpyDeviceToDevice, s));
        SR_TRY((int)hipMemcpyAsync(e->d_adm + MAXB, dmirror(h_state), MAXB * 4, hipMemcpyDeviceToDevice, s));
        SR_TRY((int)hipMemcpyAsync(e->d_adm + 2 * MAXB, dmirror(h_state + MAXB), MAXB * 4, hipMemcpyDeviceToDevice, s));
        SR_TRY((int)hipMemcpyAsync(e->d_adm + 3 * MAXB, dmirror(h_state + 3 * MAXB), MAXB * 4, hipMemcpyDeviceToDevice, s));
    }

    // ---- forward over the packed tokens
    const int H = c.t_hidden, QD = c.t_heads * 128;
    SR_TRY(launch_embed(s, e->t_src, e->embed, static_cast<const bf16_t*>(image_embeds), e->t_x, n_tok, H, 1));
    const float scale = (float)(1.0 / sqrt(128.0));
    const bool splitk = !limits && prefill_splitk(e, n_tok);
    bool pending = false;        // split-K: the down-projection's slabs have not been added to t_x yet
    for (int l = 0; l < c.t_layers; ++l) {
        const LmLayerW& w = e->ll[l];
        bf16_t* kc = e->kcache + (size_t)l * e->kv_layer_elems;
        bf16_t* vc = e->vtcache + (size_t)l * e->kv_layer_elems;
        if (pending) SR_TRY(launch_resid_rmsnorm(s, e->t_x, e->t_slabs, SPLITK_KS, w.ln1, e->t_xn, n_tok, H, c.t_rms_eps, 0, 1));
        else SR_TRY(launch_rmsnorm(s, e->t_x, w.ln1, e->t_xn, n_tok, H, c.t_rms_eps, 0, 1));
        // q/k/v Linear; mRoPE and the KV-cache write ride in the GEMM epilogue when the 256-tile kernel takes the shape (large batches),
        // otherwise they are the separate launch of round 1 (bit-identical results either way)
        const QkvRope qr{e->t_pos3, e->t_slot, e->t_idx, e->rope_cos, e->rope_sin, kc, vc, n_tok, c.t_heads, c.t_kv_heads,
                         c.mrope_section[0], c.mrope_section[0] + c.mrope_section[1], c.max_ctx};
        bool fused_qkv = false;
        if (int rc = lm_gemm(e, s, e->t_xn, H, w.qkv_w, w.qkv_w8, w.qkv_s, n_tok, e->t_qn, H, e->t_qkv, e->t_qn, w.qkv_b, nullptr, EPI_STORE, &qr, &fused_qkv)) return rc;
        if (!fused_qkv) {
            LmRopeArgs ra{e->t_qkv, n_tok, c.t_heads, c.t_kv_heads, e->t_pos3, e->t_slot, e->t_idx, e->rope_cos, e->rope_sin,
                          c.mrope_section[0], c.mrope_section[0] + c.mrope_section[1], kc, vc, c.max_ctx};
            SR_TRY(launch_lm_rope_prefill(s, ra));
        }
        AttnArgs a{e->t_qkv, e->t_qn, kc, 128, (long long)c.max_ctx * 128, vc, c.max_ctx, (long long)128 * c.max_ctx,
                   e->t_attn, QD, e->t_work, n_work, c.t_heads, e->t_group, scale, 1, 64, 1};      // max_ctx % 64 == 0: aligned, readable key runs
        SR_TRY(launch_attn_prefill(s, a, 128));
        if (splitk) {
            if (int rc = lm_gemm(e, s, e->t_attn, QD, w.o_w, w.o_w8, w.o_s, n_tok, H, QD, e->t_slabs, H, nullptr, nullptr, EPI_F32, nullptr, nullptr, SPLITK_KS)) return rc;
            SR_TRY(launch_resid_rmsnorm(s, e->t_x, e->t_slabs, SPLITK_KS, w.ln2, e->t_xn, n_tok, H, c.t_rms_eps, 0, 1));
        } else {
            if (int rc = lm_gemm(e, s, e->t_attn, QD, w.o_w, w.o_w8, w.o_s, n_tok, H, QD, e->t_x, H, nullptr, e->t_x, EPI_RESID)) return rc;
            SR_TRY(launch_rmsnorm(s, e->t_x, w.ln2, e->t_xn, n_tok, H, c.t_rms_eps, 0, 1));
        }
        if (int rc = lm_gemm(e, s, e->t_xn, H, w.gu_w, w.gu_w8, w.gu_s, n_tok, 2 * e->t_inter_pad, H, e->t_act, e->t_inter_pad, nullptr, nullptr, EPI_SWIGLU)) return rc;
        if (splitk) {
            if (int rc = lm_gemm(e, s, e->t_act, e->t_inter_pad, w.down_w, w.down_w8, w.down_s, n_tok, H, e->t_inter_pad, e->t_slabs, H, nullptr, nullptr, EPI_F32, nullptr, nullptr, SPLITK_KS)) return rc;
            pending = true;
        } else if (int rc = lm_gemm(e, s, e->t_act, e->t_inter_pad, w.down_w, w.down_w8, w.down_s, n_tok, H, e->t_inter_pad, e->t_x, H, nullptr, e->t_x, EPI_RESID)) return rc;
    }
    // last position of every sequence -> final norm -> tied LM head (hf:1386-1387) -> greedy token
    // (split-K: the last down-projection's slabs still have to reach t_x; the norm output is only used by the all-positions path)
    if (pending) SR_TRY(launch_resid_rmsnorm(s, e->t_x, e->t_slabs, SPLITK_KS, e->final_norm, e->t_xn, n_tok, H, c.t_rms_eps, 0, 1));
    if (all_logits_out) {       // every position: final norm over all rows, tied LM head as an MFMA GEMM with float32 output
        if (!pending) SR_TRY(launch_rmsnorm(s, e->t_x, e->final_norm, e->t_xn, n_tok, H, c.t_rms_eps, 0, 1));
        if (int rc = gemm(e, s, e->t_xn, H, e->embed, n_tok, c.t_vocab, H, all_logits_out, c.t_vocab, nullptr, nullptr, nullptr, EPI_F32, 1)) return rc;
    }
    // (an admission keeps to scratch of its own -- d_xadm / d_xadm_n, the *_adm logits and partials -- because it may run on another
    // stream while the rows decode)
    bf16_t* xl = limits ? e->d_xadm : e->d_xa;
    SR_TRY(launch_gather_rows(s, e->t_x, e->t_lastrow, xl, B, H));
    if (!limits) {
        if (int rc = enqueue_lm_head(e, B, xl, e->d_xb, false, s)) return rc;
        if (logits_out) SR_TRY((int)hipMemcpyAsync(logits_out, e->d_logits, (size_t)B * c.t_vocab * 4, hipMemcpyDeviceToDevice, s));
        e->h_ctx_hi = 0;
        for (int b = 0; b < B; ++b) e->h_ctx_hi = std::max(e->h_ctx_hi, (int)seq_lens[b]);
        e->prefilled_B = B;
        e->rows_mode = false;
        return 0;
    }
    // admission: final norm + LM head into the admission scratch, greedy first token, then install the rows.  ONE kernel path whatever the
    // number of sequences admitted together (block-per-row norm, 32-row MFMA GEMV -- what a static batch of more than 16 runs), so that a
    // request's first token does not depend on how the scheduler grouped it
    {
        GemvArgs g = gv(xl, H, e->embed, B, c.t_vocab, H, e->d_logits_adm, c.t_vocab);
        g.amax_val = e->d_amax_val_adm; g.amax_idx = e->d_amax_idx_adm;
        const int xt = x_tiled_ok(e) ? 1 : 0;
        SR_TRY(launch_rmsnorm(s, xl, e->final_norm, e->d_xadm_n, B, H, c.t_rms_eps, xt));
        g.x = e->d_xadm_n; g.x_tiled = xt; g.force32 = 1;
        SR_TRY(launch_gemv(s, g, GV_F32));
    }
    if (logits_out) SR_TRY((int)hipMemcpyAsync(logits_out, e->d_logits_adm, (size_t)B * c.t_vocab * 4, hipMemcpyDeviceToDevice, s));
    SR_TRY(launch_argmax(s, e->d_logits_adm, B, c.t_vocab, e->d_adm + 4 * MAXB));
    if (e->rows_temp > 0.f) {      // sampling mode: the first token of the new rows is drawn from the admission logits
        SampleArgs sa{e->d_logits_adm, c.t_vocab, B, 1.0f / e->rows_temp, e->rows_topk, e->rows_topp, 1.0f, nullptr, e->seen_words,
                      e->rows_seed ^ (0x9E3779B9u * ++e->adm_count), nullptr, e->d_adm_pick, nullptr, 0, 0};
        SR_TRY(launch_sample(s, sa));
    }
    e->staged_n = B;
    e->staged_off = 0;
    if (commit) return commit_admission(e, slots, B, s);
    return 0;
}

int sr_prefill(sr_engine* e, const int64_t* ids, const int64_t* pos3, const int32_t* seq_lens, const int32_t* slots, int B,
               const void* image_embeds, int n_image_rows, float* logits_out, void* stream) {
    return prefill_impl(e, ids, pos3, seq_lens, slots, B, image_embeds, n_image_rows, logits_out, stream, nullptr);
}

int sr_forward_logits(sr_engine* e, const int64_t* ids, const int64_t* pos3, const int32_t* seq_lens, int B, const void* image_embeds,
                      int n_image_rows, float* all_logits_out, void* stream) {
    if (!all_logits_out) return fail(e, -22, "sr_forward_logits: null output");
    if (!e || B < 1 || B > MAXB) return fail(e, -22, "sr_forward_logits: bad batch");
    int32_t slots[MAXB];
    for (int b = 0; b < B; ++b) slots[b] = b;
    return prefill_impl(e, ids, pos3, seq_lens, slots, B, image_embeds, n_image_rows, nullptr, stream, nullptr, all_logits_out);
}

// ---- continuous batching: batch rows with independent lifecycles (row index = KV slot); a decode step always runs all
// max_batch rows, rows that are free or finished are masked by their `finished` flag
int sr_rows_begin(sr_engine* e, void* stream) {
    enter(e);
    if (!e) return fail(e, -22, "sr_rows_begin: null engine");
    hipStream_t s = (hipStream_t)stream;
    std::vector<int> st(5 * MAXB, 0);
    for (int i = 0; i < MAXB; ++i) { st[i] = 1; st[2 * MAXB + i] = 1; st[3 * MAXB + i] = 0; st[4 * MAXB + i] = i; }   // ctx 1 | pos 0 | finished 1 | step 0 | slots
    SR_TRY((int)hipMemcpyAsync(e->d_ctx_len, st.data(), MAXB * 4, hipMemcpyHostToDevice, s));
    SR_TRY((int)hipMemcpyAsync(e->d_pos, st.data() + MAXB, MAXB * 4, hipMemcpyHostToDevice, s));
    SR_TRY((int)hipMemcpyAsync(e->d_finished, st.data() + 2 * MAXB, MAXB * 4, hipMemcpyHostToDevice, s));
    SR_TRY((int)hipMemcpyAsync(e->d_step, st.data() + 3 * MAXB, MAXB * 4, hipMemcpyHostToDevice, s));
    SR_TRY((int)hipMemcpyAsync(e->d_slots, st.data() + 4 * MAXB, MAXB * 4, hipMemcpyHostToDevice, s));
    SR_TRY((int)hipMemsetAsync(e->d_row_limit, 0, MAXB * 4, s));
    SR_TRY((int)hipMemsetAsync(e->d_ngen, 0, MAXB * 4, s));
    SR_TRY((int)hipStreamSynchronize(s));
    e->rows_mode = true;
    e->staged_n = e->staged_off = 0;
    e->prefilled_B = e->c.max_batch;
    e->h_ctx_hi = 0;
    e->rows_temp = 0.f;
    return 0;
}

int sr_rows_sampling(sr_engine* e, float temperature, int top_k, float top_p, uint32_t seed) {
    if (!e || !e->rows_mode) return fail(e, -22, "sr_rows_sampling: call sr_rows_begin first");
    if (temperature > 0.f && (top_k > 1024 || !(top_p > 0.f) || top_p > 1.f))
        return fail(e, -22, "sr_rows_sampling: top_k <= 1024 (<= 0: no top-k bound) and 0 < top_p <= 1 required");
    e->rows_temp = temperature > 0.f ? temperature : 0.f;
    e->rows_topk = top_k; e->rows_topp = top_p; e->rows_seed = seed; e->adm_count = 0;
    return 0;
}

int sr_admit(sr_engine* e, const int64_t* ids, const int64_t* pos3, const int32_t* seq_lens, const int32_t* rows, const int32_t* max_new,
             int n, const void* image_embeds, int n_image_rows, float* logits_out, void* stream) {
    if (!e || !rows || !max_new) return fail(e, -22, "sr_admit: null argument");
    if (!e->rows_mode) return fail(e, -22, "sr_admit: call sr_rows_begin first");
    for (int i = 0; i < n; ++i) {
        if (max_new[i] < 1 || max_new[i] > e->c.max_new_tokens || seq_lens[i] + max_new[i] > e->c.max_ctx)
            return fail(e, -22, "sr_admit: sequence %d (len %d, max_new %d) does not fit max_new_tokens %d / max_ctx %d", i, seq_lens[i],
                        max_new[i], e->c.max_new_tokens, e->c.max_ctx);
        if (rows[i] < 0 || rows[i] >= e->c.max_batch) return fail(e, -22, "sr_admit: row %d out of range", rows[i]);
        for (int j = 0; j < i; ++j)
            if (rows[i] == rows[j]) return fail(e, -22, "sr_admit: row %d given twice", rows[i]);
    }
    if (e->staged_n) return fail(e, -22, "sr_admit: %d staged sequences are waiting for sr_admit_commit", e->staged_n);
    return prefill_impl(e, ids, pos3, seq_lens, rows, n, image_embeds, n_image_rows, logits_out, stream, max_new);      // row r uses KV slot r
}

// Admission in two halves, so that the expensive half can run UNDER the decode steps of the running rows on another (CU-masked)
// stream: sr_admit_stage = ViT-fed prefill of n sequences into SPARE KV slots + LM head + first-token choice (touches no row state and
// only admission-owned scratch); sr_admit_commit = install them into free batch rows (row -> KV slot indirection), on the decode
// stream, between two steps, once the staging stream's work is known to be complete (the caller orders the streams with an event).
int sr_admit_stage(sr_engine* e, const int64_t* ids, const int64_t* pos3, const int32_t* seq_lens, const int32_t* kv_slots, const int32_t* max_new,
                   int n, const void* image_embeds, int n_image_rows, float* logits_out, void* stream) {
    if (!e || !kv_slots || !max_new) return fail(e, -22, "sr_admit_stage: null argument");
    if (!e->rows_mode) return fail(e, -22, "sr_admit_stage: call sr_rows_begin first");
    if (e->staged_n) return fail(e, -22, "sr_admit_stage: %d staged sequences are waiting for sr_admit_commit", e->staged_n);
    for (int i = 0; i < n; ++i) {
        if (max_new[i] < 1 || max_new[i] > e->c.max_new_tokens || seq_lens[i] + max_new[i] > e->c.max_ctx)
            return fail(e, -22, "sr_admit_stage: sequence %d (len %d, max_new %d) does not fit max_new_tokens %d / max_ctx %d", i, seq_lens[i],
                        max_new[i], e->c.max_new_tokens, e->c.max_ctx);
        for (int j = 0; j < i; ++j)
            if (kv_slots[i] == kv_slots[j]) return fail(e, -22, "sr_admit_stage: KV slot %d given twice", kv_slots[i]);
    }
    return prefill_impl(e, ids, pos3, seq_lens, kv_slots, n, image_embeds, n_image_rows, logits_out, stream, max_new, nullptr, false);
}

int sr_admit_commit(sr_engine* e, const int32_t* rows, int n, void* stream) {
    enter(e);
    if (!e || !rows) return fail(e, -22, "sr_admit_commit: null argument");
    if (!e->rows_mode || n > e->staged_n || n < 1) return fail(e, -22, "sr_admit_commit: %d rows given, %d sequences staged", n, e ? e->staged_n : 0);
    for (int i = 0; i < n; ++i) {
        if (rows[i] < 0 || rows[i] >= e->c.max_batch) return fail(e, -22, "sr_admit_commit: row %d out of range", rows[i]);
        for (int j = 0; j < i; ++j)
            if (rows[i] == rows[j]) return fail(e, -22, "sr_admit_commit: row %d given twice", rows[i]);
    }
    return commit_admission(e, rows, n, (hipStream_t)stream);
}

// one decode step (bookkeeping kernel + forward) captured once per (B, eos count, pad) and replayed
static int ensure_decode_graph(sr_engine* e, int B, int n_eos, int pad_id) {
    const int m = e->px_mode;
    if (e->graph[m] != nullptr && e->graph_B[m] == B && e->graph_neos[m] == n_eos && e->graph_pad[m] == pad_id) return 0;
    if (e->graph[m]) { (void)hipGraphExecDestroy(e->graph[m]); e->graph[m] = nullptr; }
    hipGraph_t g = nullptr;
    SR_TRY((int)hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeRelaxed));
    int rc = enqueue_step(e, B, n_eos, pad_id, nullptr, e->cap_stream);
    if (!rc) rc = enqueue_decode_forward(e, B, e->cap_stream);
    hipError_t er = hipStreamEndCapture(e->cap_stream, &g);
    if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
    SR_TRY((int)er);
    er = hipGraphInstantiate(&e->graph[m], g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    SR_TRY((int)er);
    e->graph_B[m] = B; e->graph_neos[m] = n_eos; e->graph_pad[m] = pad_id;
    return 0;
}

int sr_decode(sr_engine* e, const int32_t* host_slots, int B, int max_new, const int32_t* host_eos, int n_eos, int32_t pad_id,
              int32_t* tokens_out, float* logits_trace, const int32_t* forced, int use_graph, void* stream, int* steps_done) {
    enter(e);
    if (!e || !tokens_out) return fail(e, -22, "sr_decode: null argument");
    const sr_config& c = e->c;
    if (B < 1 || B > c.max_batch || max_new < 1 || max_new > c.max_new_tokens || n_eos < 0 || n_eos > 32)
        return fail(e, -22, "sr_decode: B=%d max_new=%d n_eos=%d out of range", B, max_new, n_eos);
    (void)host_slots;   // slots were fixed by sr_prefill (kept in the signature for the continuous-batching scheduler)
    if (e->rows_mode) return fail(e, -22, "sr_decode: the engine is in continuous-batching mode (use sr_rows_step)");
    if (B != e->prefilled_B) return fail(e, -22, "sr_decode: B=%d but %d sequences were prefilled", B, e->prefilled_B);
    if (e->h_ctx_hi + max_new > c.max_ctx)
        return fail(e, -22, "sr_decode: context %d + %d new tokens exceeds max_ctx %d", e->h_ctx_hi, max_new, c.max_ctx);
    e->h_ctx_hi += max_new;
    hipStream_t s = (hipStream_t)stream;
    if (n_eos) SR_TRY((int)hipMemcpyAsync(e->d_eos, host_eos, n_eos * 4, hipMemcpyHostToDevice, s));
    const size_t V = c.t_vocab;
    if (logits_trace) SR_TRY((int)hipMemcpyAsync(logits_trace, e->d_logits, B * V * 4, hipMemcpyDeviceToDevice, s));
    // forced tokens are laid out [B][max_new] by the caller; the device log uses the engine's row stride
    const int* forced_dev = nullptr;
    if (forced) {
        if (max_new != c.max_new_tokens) return fail(e, -22, "teacher forcing requires max_new == max_new_tokens (%d)", c.max_new_tokens);
        forced_dev = forced;
    }
    const bool graph_ok = use_graph && !logits_trace && !forced;
    if (graph_ok) if (int rc = ensure_decode_graph(e, B, n_eos, pad_id)) return rc;
    int done = 0;
    std::vector<int> fin(B);
    for (int i = 0; i < max_new; ++i) {
        if (i == max_new - 1) {   // the last token only needs to be recorded
            if (int rc = enqueue_step(e, B, n_eos, pad_id, forced_dev, s)) return rc;
            done = max_new;
            break;
        }
        if (graph_ok) SR_TRY((int)hipGraphLaunch(e->graph[e->px_mode], s));
        else {
            if (int rc = enqueue_step(e, B, n_eos, pad_id, forced_dev, s)) return rc;
            if (int rc = enqueue_decode_forward(e, B, s)) return rc;
            if (logits_trace)
                SR_TRY((int)hipMemcpyAsync(logits_trace + (size_t)(i + 1) * B * V, e->d_logits, B * V * 4, hipMemcpyDeviceToDevice, s));
        }
        done = i + 1;
        if (n_eos && (i % 16) == 15) {   // early exit once every sequence has emitted an eos token
            SR_TRY((int)hipMemcpyAsync(fin.data(), e->d_finished, B * 4, hipMemcpyDeviceToHost, s));
            SR_TRY((int)hipStreamSynchronize(s));
            bool all = true;
            for (int b = 0; b < B; ++b) all &= fin[b] != 0;
            if (all) break;
        }
    }
    // token log [B][max_new_tokens] -> caller layout [B][max_new]; unwritten positions read as pad
    if (done < max_new) {
        // fill the tail of the device log with pad before copying out
        std::vector<int> padrow(max_new, pad_id);
        for (int b = 0; b < B; ++b)
            SR_TRY((int)hipMemcpyAsync(e->d_tokens + (size_t)b * c.max_new_tokens + done, padrow.data(), (max_new - done) * 4, hipMemcpyHostToDevice, s));
        SR_TRY((int)hipStreamSynchronize(s));
    }
    SR_TRY((int)hipMemcpy2DAsync(tokens_out, (size_t)max_new * 4, e->d_tokens, (size_t)c.max_new_tokens * 4, (size_t)max_new * 4, B,
                                 hipMemcpyDeviceToDevice, s));
    SR_TRY((int)hipStreamSynchronize(s));
    if (steps_done) *steps_done = done;
    return 0;
}

// Sampled decode (temperature / top-k / top-p / repetition penalty of vllm_strategy.py:289-309), entirely on the device:
// the token of every step is drawn by k_sample from the logits the LM head left in HBM and fed back through k_step.
int sr_decode_sample(sr_engine* e, int B, int max_new, const int32_t* host_eos, int n_eos, int32_t pad_id, float temperature, int top_k,
                     float top_p, float rep_penalty, uint32_t seed, int32_t* tokens_out, int use_graph, void* stream, int* steps_done) {
    enter(e);
    if (!e || !tokens_out) return fail(e, -22, "sr_decode_sample: null argument");
    const sr_config& c = e->c;
    if (e->rows_mode) return fail(e, -22, "sr_decode_sample: the engine is in continuous-batching mode");
    if (B < 1 || B != e->prefilled_B || max_new < 1 || max_new > c.max_new_tokens || n_eos < 0 || n_eos > 32)
        return fail(e, -22, "sr_decode_sample: B=%d (prefilled %d) max_new=%d n_eos=%d out of range", B, e->prefilled_B, max_new, n_eos);
    if (!(temperature > 0.f) || top_k > 1024 || !(top_p > 0.f) || top_p > 1.f || !(rep_penalty > 0.f))
        return fail(e, -22, "sr_decode_sample: temperature > 0, top_k <= 1024 (<= 0: no top-k bound), 0 < top_p <= 1, repetition_penalty > 0 required");
    if (e->h_ctx_hi + max_new > c.max_ctx)
        return fail(e, -22, "sr_decode_sample: context %d + %d new tokens exceeds max_ctx %d", e->h_ctx_hi, max_new, c.max_ctx);
    e->h_ctx_hi += max_new;
    hipStream_t s = (hipStream_t)stream;
    if (n_eos) SR_TRY((int)hipMemcpyAsync(e->d_eos, host_eos, n_eos * 4, hipMemcpyHostToDevice, s));
    const bool rp = rep_penalty != 1.0f;
    const float it = 1.0f / temperature;
    const int hn = fused_norms(e, B) ? 1 : 0;
    SampleArgs sa{e->d_logits, c.t_vocab, B, it, top_k, top_p, rep_penalty, rp ? e->d_seen : nullptr, e->seen_words, seed, e->d_step, e->d_sampled,
                  e->d_amax_val, gemv_f32_blocks(c.t_vocab, B, c.t_hidden, hn), gemv_f32_block_rows(c.t_vocab, B, c.t_hidden, hn)};
    if (rp) SR_TRY(launch_mark_prompt(s, e->t_src, e->t_lastrow, e->d_seen, e->seen_words, B));
    SR_TRY(launch_sample(s, sa));                  // first token, from the prefill logits
    auto enqueue = [&](hipStream_t q) -> int {     // one step: consume the drawn token, forward, draw the next
        if (int rc = enqueue_step(e, B, n_eos, pad_id, nullptr, q, e->d_sampled)) return rc;
        if (rp) SR_TRY(launch_mark_chosen(q, e->d_sampled, e->d_seen, e->seen_words, B));
        if (int rc = enqueue_decode_forward(e, B, q)) return rc;
        return launch_sample(q, sa);
    };
    if (use_graph && (e->sgraph == nullptr || e->sg_B != B || e->sg_neos != n_eos || e->sg_pad != pad_id || e->sg_topk != top_k ||
                      e->sg_it != it || e->sg_topp != top_p || e->sg_rp != rep_penalty || e->sg_seed != seed || e->sg_px != e->px_mode)) {
        if (e->sgraph) { (void)hipGraphExecDestroy(e->sgraph); e->sgraph = nullptr; }
        hipGraph_t g = nullptr;
        SR_TRY((int)hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeRelaxed));
        int rc = enqueue(e->cap_stream);
        hipError_t er = hipStreamEndCapture(e->cap_stream, &g);
        if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
        SR_TRY((int)er);
        er = hipGraphInstantiate(&e->sgraph, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        SR_TRY((int)er);
        e->sg_B = B; e->sg_neos = n_eos; e->sg_pad = pad_id; e->sg_topk = top_k; e->sg_it = it; e->sg_topp = top_p; e->sg_rp = rep_penalty; e->sg_seed = seed; e->sg_px = e->px_mode;
    }
    int done = 0;
    std::vector<int> fin(B);
    for (int i = 0; i < max_new; ++i) {
        if (i == max_new - 1) {                    // the last token only needs to be recorded
            if (int rc = enqueue_step(e, B, n_eos, pad_id, nullptr, s, e->d_sampled)) return rc;
            done = max_new;
            break;
        }
        if (use_graph) SR_TRY((int)hipGraphLaunch(e->sgraph, s));
        else if (int rc = enqueue(s)) return rc;
        done = i + 1;
        if (n_eos && (i % 16) == 15) {
            SR_TRY((int)hipMemcpyAsync(fin.data(), e->d_finished, B * 4, hipMemcpyDeviceToHost, s));
            SR_TRY((int)hipStreamSynchronize(s));
            bool all = true;
            for (int b = 0; b < B; ++b) all &= fin[b] != 0;
            if (all) break;
        }
    }
    if (done < max_new) {
        std::vector<int> padrow(max_new, pad_id);
        for (int b = 0; b < B; ++b)
            SR_TRY((int)hipMemcpyAsync(e->d_tokens + (size_t)b * c.max_new_tokens + done, padrow.data(), (max_new - done) * 4, hipMemcpyHostToDevice, s));
        SR_TRY((int)hipStreamSynchronize(s));
    }
    SR_TRY((int)hipMemcpy2DAsync(tokens_out, (size_t)max_new * 4, e->d_tokens, (size_t)c.max_new_tokens * 4, (size_t)max_new * 4, B,
                                 hipMemcpyDeviceToDevice, s));
    SR_TRY((int)hipStreamSynchronize(s));
    if (steps_done) *steps_done = done;
    return 0;
}

int sr_rows_step(sr_engine* e, int n_steps, const int32_t* host_eos, int n_eos, int32_t pad_id, void* stream) {
    enter(e);
    if (!e || !e->rows_mode) return fail(e, -22, "sr_rows_step: call sr_rows_begin first");
    if (n_steps < 1 || n_eos < 0 || n_eos > 32) return fail(e, -22, "sr_rows_step: n_steps=%d n_eos=%d out of range", n_steps, n_eos);
    hipStream_t s = (hipStream_t)stream;
    if (n_eos) SR_TRY((int)hipMemcpyAsync(e->d_eos, host_eos, n_eos * 4, hipMemcpyHostToDevice, s));
    const int B = e->c.max_batch;
    if (e->rows_temp > 0.f) {      // sampled step: consume the drawn tokens, forward, draw the next ones for every row
        const sr_config& c = e->c;
        const float it = 1.0f / e->rows_temp;
        auto& rg = e->rgraph[e->px_mode];
        if (rg.g == nullptr || rg.neos != n_eos || rg.pad != pad_id || rg.topk != e->rows_topk || rg.it != it || rg.topp != e->rows_topp || rg.seed != e->rows_seed) {
            if (rg.g) { (void)hipGraphExecDestroy(rg.g); rg.g = nullptr; }
            const int hn = fused_norms(e, B) ? 1 : 0;
            SampleArgs sa{e->d_logits, c.t_vocab, B, it, e->rows_topk, e->rows_topp, 1.0f, nullptr, e->seen_words, e->rows_seed, e->d_step, e->d_sampled,
                          e->d_amax_val, gemv_f32_blocks(c.t_vocab, B, c.t_hidden, hn), gemv_f32_block_rows(c.t_vocab, B, c.t_hidden, hn)};
            hipGraph_t g = nullptr;
            SR_TRY((int)hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeRelaxed));
            int rc = enqueue_step(e, B, n_eos, pad_id, nullptr, e->cap_stream, e->d_sampled);
            if (!rc) rc = enqueue_decode_forward(e, B, e->cap_stream);
            if (!rc) rc = launch_sample(e->cap_stream, sa);
            hipError_t er = hipStreamEndCapture(e->cap_stream, &g);
            if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
            SR_TRY((int)er);
            er = hipGraphInstantiate(&rg.g, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            SR_TRY((int)er);
            rg.neos = n_eos; rg.pad = pad_id; rg.topk = e->rows_topk; rg.it = it; rg.topp = e->rows_topp; rg.seed = e->rows_seed;
        }
        for (int i = 0; i < n_steps; ++i) SR_TRY((int)hipGraphLaunch(rg.g, s));
        return 0;
    }
    if (int rc = ensure_decode_graph(e, B, n_eos, pad_id)) return rc;
    for (int i = 0; i < n_steps; ++i) SR_TRY((int)hipGraphLaunch(e->graph[e->px_mode], s));
    return 0;
}

int sr_rows_poll(sr_engine* e, int32_t* host_finished, int32_t* host_steps, void* stream) {
    enter(e);
    if (!e || !e->rows_mode || !host_finished || !host_steps) return fail(e, -22, "sr_rows_poll: bad argument / not in rows mode");
    hipStream_t s = (hipStream_t)stream;
    SR_TRY((int)hipMemcpyAsync(host_finished, e->d_finished, e->c.max_batch * 4, hipMemcpyDeviceToHost, s));
    SR_TRY((int)hipMemcpyAsync(host_steps, e->d_ngen, e->c.max_batch * 4, hipMemcpyDeviceToHost, s));
    SR_TRY((int)hipStreamSynchronize(s));
    return 0;
}

int sr_rows_read(sr_engine* e, int row, int32_t* dev_tokens_out, int n, void* stream) {
    enter(e);
    if (!e || !e->rows_mode || row < 0 || row >= e->c.max_batch || n < 0 || n > e->c.max_new_tokens)
        return fail(e, -22, "sr_rows_read: bad argument / not in rows mode");
    SR_TRY((int)hipMemcpyAsync(dev_tokens_out, e->d_tokens + (size_t)row * e->c.max_new_tokens, (size_t)n * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

int sr_rows_set_cus(sr_engine* e, int n_cus, void* stream) {
    enter(e);
    if (!e || n_cus < 0) return fail(e, -22, "sr_rows_set_cus: bad argument");
    hipError_t r = hipMemsetD32Async((hipDeviceptr_t)(e->d_px + 10 * 16), n_cus, 1, (hipStream_t)stream);
    if (r != hipSuccess) return fail(e, (int)r, "sr_rows_set_cus: %s", hipGetErrorString(r));
    e->px_mode = n_cus > 0 ? 1 : 0;         // the decode step has two captured forms: the next steps replay the one for this stream
    return 0;
}

int sr_rows_abort(sr_engine* e, const int32_t* host_rows, int n, void* stream) {
    enter(e);
    if (!e || !e->rows_mode || !host_rows || n < 0) return fail(e, -22, "sr_rows_abort: bad argument / not in rows mode");
    unsigned mask[MAXB / 32] = {};
    for (int i = 0; i < n; ++i) {
        if (host_rows[i] < 0 || host_rows[i] >= e->c.max_batch || host_rows[i] >= MAXB) return fail(e, -22, "sr_rows_abort: row %d out of range", host_rows[i]);
        mask[host_rows[i] >> 5] |= 1u << (host_rows[i] & 31);
    }
    SR_TRY(launch_rows_abort((hipStream_t)stream, mask, e->d_finished));
    return 0;
}

int sr_decode_step(sr_engine* e, const int64_t* dev_last_ids, int B, float* dev_logits_out, int64_t* dev_next_ids, void* stream) {
    enter(e);
    if (!e) return fail(e, -22, "sr_decode_step: null engine");
    const sr_config& c = e->c;
    if (e->rows_mode) return fail(e, -22, "sr_decode_step: the engine is in continuous-batching mode");
    if (B < 1 || B != e->prefilled_B) return fail(e, -22, "sr_decode_step: B=%d but %d sequences were prefilled", B, e->prefilled_B);
    if (e->h_ctx_hi + 1 > c.max_ctx) return fail(e, -22, "sr_decode_step: context %d would exceed max_ctx %d", e->h_ctx_hi + 1, c.max_ctx);
    e->h_ctx_hi += 1;
    hipStream_t s = (hipStream_t)stream;
    // eos handling stays with the caller (n_eos = 0): a finished row is simply ignored by whoever samples.
    // One captured graph per (token source, B): step bookkeeping + forward + greedy ids, replayed from engine-owned buffers.
    const int kind = dev_last_ids ? 1 : 0;
    const int n_part = gemv_f32_blocks(c.t_vocab, B, c.t_hidden, fused_norms(e, B) ? 1 : 0);
    if (e->step_graph[kind] == nullptr || e->step_graph_B[kind] != B || e->step_graph_px[kind] != e->px_mode) {
        if (e->step_graph[kind]) { (void)hipGraphExecDestroy(e->step_graph[kind]); e->step_graph[kind] = nullptr; }
        hipGraph_t g = nullptr;
        SR_TRY((int)hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeRelaxed));
        int rc = enqueue_step(e, B, 0, 0, nullptr, e->cap_stream, kind ? e->d_chosen : nullptr);
        if (!rc) rc = enqueue_decode_forward(e, B, e->cap_stream);
        if (!rc) rc = launch_next_ids(e->cap_stream, e->d_amax_val, e->d_amax_idx, n_part, B, e->d_next);
        hipError_t er = hipStreamEndCapture(e->cap_stream, &g);
        if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
        SR_TRY((int)er);
        er = hipGraphInstantiate(&e->step_graph[kind], g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        SR_TRY((int)er);
        e->step_graph_B[kind] = B; e->step_graph_px[kind] = e->px_mode;
    }
    if (kind) SR_TRY((int)hipMemcpyAsync(e->d_chosen, dev_last_ids, (size_t)B * 8, hipMemcpyDeviceToDevice, s));
    SR_TRY((int)hipGraphLaunch(e->step_graph[kind], s));
    if (dev_logits_out) SR_TRY((int)hipMemcpyAsync(dev_logits_out, e->d_logits, (size_t)B * c.t_vocab * 4, hipMemcpyDeviceToDevice, s));
    if (dev_next_ids) SR_TRY((int)hipMemcpyAsync(dev_next_ids, e->d_next, (size_t)B * 8, hipMemcpyDeviceToDevice, s));
    return 0;
}

// ---------------------------------------------------------------------------------------------------- raster + ops
#define SR_WRAP(call) do { int rc_ = (call); if (rc_) return fail(nullptr, rc_, #call " failed with %d", rc_); return 0; } while (0)

int sr_mask_union(uint8_t* acc, const uint8_t* m, size_t n, void* stream) { SR_WRAP(launch_mask_union((hipStream_t)stream, acc, m, n)); }
int sr_resize_nearest_u8(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw, void* stream) {
    SR_WRAP(launch_resize_nearest_u8((hipStream_t)stream, src, sh, sw, dst, dh, dw));
}
int sr_iou_counts(const uint8_t* p, const uint8_t* g, size_t n, int64_t* out2, void* stream) {
    SR_WRAP(launch_iou_counts((hipStream_t)stream, p, g, n, reinterpret_cast<long long*>(out2)));
}
int sr_iou_counts_batched(const uint8_t* p, const uint8_t* g, size_t n, int n_items, int64_t* out, void* stream) {
    SR_WRAP(launch_iou_counts_batched((hipStream_t)stream, p, g, n, n_items, reinterpret_cast<long long*>(out)));
}
int sr_render_overlay(uint8_t* img, int h, int w, const uint8_t* mask, int mh, int mw, const int32_t* boxes, int nb, void* stream) {
    SR_WRAP(launch_render_overlay((hipStream_t)stream, img, h, w, mask, mh, mw, boxes, nb));
}
int sr_op_gemm(const void* A, int lda, const void* W, int M, int N, int K, void* out, int ldo, const void* bias, const void* resid,
               const int32_t* rowmap, int epilogue, void* stream) {
    GemmArgs a{(const bf16_t*)A, lda, (const bf16_t*)W, M, N, K, out, ldo, (const bf16_t*)bias, (const bf16_t*)resid, rowmap,
               (epilogue & 0x100) ? 1 : 0, nullptr,      // bit 8 of `epilogue`: W is fragment-ordered (tiled16x64)
               (epilogue & 0x200) ? 256 : (epilogue & 0x400) ? 128 : 0};      // bits 9 / 10: force the 256- / 128-tile kernel
    a.gelu_fast = (epilogue & 0x1000) ? 2 : (epilogue & 0x800) ? 1 : 0;      // bit 11: EPI_GELU through gelu_fast_f; bit 12: ReLU instead of GELU
    SR_WRAP(launch_gemm((hipStream_t)stream, a, epilogue & 0xff));
}
// ---- SAM2 image path (socioreasoner_amd/sam2.py drives these; shapes and layouts: sam.hip, attention.hip)
int sr_op_attention(const void* q, int q_stride, const void* k, int k_stride, long long k_head_stride, const void* vt, int vt_stride,
                    long long vt_head_stride, void* out, int out_stride, const void* dev_work, int n_work, int n_heads, int group, float scale,
                    int causal, int head_dim, int q_tile, int v2_ok, void* stream) {
    AttnArgs a{(const bf16_t*)q, q_stride, (const bf16_t*)k, k_stride, k_head_stride, (const bf16_t*)vt, vt_stride, vt_head_stride,
               (bf16_t*)out, out_stride, (const AttnWork*)dev_work, n_work, n_heads, group, scale, causal, q_tile, v2_ok};
    SR_WRAP(launch_attn_prefill((hipStream_t)stream, a, head_dim));
}
int sr_op_sam_preprocess(const uint8_t* img, int h, int w, void* out_chw, int S, void* stream) {
    SR_WRAP(launch_sam_preprocess((hipStream_t)stream, img, h, w, (bf16_t*)out_chw, S));
}
int sr_op_im2col(const void* chw, int S, int k, int stride, int pad, void* out, int ld, const int32_t* rowmap, void* stream) {
    SR_WRAP(launch_im2col((hipStream_t)stream, (const bf16_t*)chw, S, k, stride, pad, (bf16_t*)out, ld, rowmap));
}
int sr_op_layernorm(const void* x, int ldx, const void* w, const void* b, void* out, int ldo, int rows, int C, float eps, void* stream) {
    SR_WRAP(launch_layernorm((hipStream_t)stream, (const bf16_t*)x, ldx, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)out, ldo, rows, C, eps));
}
int sr_op_maxpool_win(const void* in, int ld_in, int C, int n_win, int ws, void* out, int ld_out, void* stream) {
    SR_WRAP(launch_maxpool_win((hipStream_t)stream, (const bf16_t*)in, ld_in, C, n_win, ws, (bf16_t*)out, ld_out));
}
int sr_op_ew(const void* a, int lda, const void* b, int ldb, void* out, int ldo, int rows, int C, int mode, void* stream) {
    SR_WRAP(launch_ew((hipStream_t)stream, (const bf16_t*)a, lda, (const bf16_t*)b, ldb, (bf16_t*)out, ldo, rows, C, mode));
}
int sr_op_transpose(const void* in, int ld_in, int rows, int cols, void* out, int ld_out, void* stream) {
    SR_WRAP(launch_transpose((hipStream_t)stream, (const bf16_t*)in, ld_in, rows, cols, (bf16_t*)out, ld_out));
}
int sr_op_upsample2x_add(const void* lat, const void* top, void* out, int H2, int C, int ld, void* stream) {
    SR_WRAP(launch_upsample2x_add((hipStream_t)stream, (const bf16_t*)lat, (const bf16_t*)top, (bf16_t*)out, H2, C, ld));
}
int sr_op_pixel_shuffle_add(const void* g, int ldg, const void* feat, int ldf, void* out, int ldo, int W, int Co, void* stream) {
    SR_WRAP(launch_pixel_shuffle_add((hipStream_t)stream, (const bf16_t*)g, ldg, (const bf16_t*)feat, ldf, (bf16_t*)out, ldo, W, Co));
}
int sr_op_mask_resize_or(const float* low, int ld, int col0, int n, int m, const float* score, uint8_t* acc, float* logits_out, int h, int w, void* stream) {
    SR_WRAP(launch_mask_resize_or((hipStream_t)stream, low, ld, col0, n, m, score, acc, logits_out, h, w));
}
int sr_op_gather_rows(const void* in, const int32_t* rows, void* out, int n, int H, void* stream) {
    SR_WRAP(launch_gather_rows((hipStream_t)stream, (const bf16_t*)in, rows, (bf16_t*)out, n, H));
}
// ---- SAM2 at the reference's precision (float32 storage and arithmetic: sam_f32.hip and sam.hip's float instantiations)
int sr_op_gemm_f32(const float* A, int lda, const float* W, int M, int N, int K, float* out, int ldo, const float* bias, const float* resid,
                   const int32_t* rowmap, int epilogue, void* stream) {
    const int e = epilogue & 0xff;
    if (e != EPI_STORE && e != EPI_RESID && e != EPI_GELU && e != EPI_F32) return -22;
    GemmF32Args a{A, lda, W, M, N, K, out, ldo, bias, e == EPI_RESID ? resid : nullptr, rowmap, e == EPI_GELU ? ((epilogue & 0x1000) ? 2 : 1) : 0};
    if (epilogue & 0x2000) return -22;      // (round 5's pre-split weight planes: measured slower, left the library in round 6)
    SR_WRAP(launch_gemm_f32((hipStream_t)stream, a));
}
int sr_op_attention_f32(const float* q, int q_stride, const float* k, int k_stride, const float* v, int v_stride, float* out, int out_stride,
                        const void* dev_work, int n_work, int n_heads, float scale, int head_dim, void* stream) {
    AttnF32Args a{q, q_stride, k, k_stride, v, v_stride, out, out_stride, (const AttnWork*)dev_work, n_work, n_heads, scale, 0};
    SR_WRAP(launch_attn_f32((hipStream_t)stream, a, head_dim));
}
int sr_op_attention_f32_causal(const float* q, int q_stride, const float* k, int k_stride, const float* v, int v_stride, float* out, int out_stride,
                               const void* dev_work, int n_work, int n_heads, float scale, int head_dim, void* stream) {
    AttnF32Args a{q, q_stride, k, k_stride, v, v_stride, out, out_stride, (const AttnWork*)dev_work, n_work, n_heads, scale, 1};
    SR_WRAP(launch_attn_f32((hipStream_t)stream, a, head_dim));
}
int sr_op_rmsnorm_f32(const float* x, int ldx, const float* w, float* out, int ldo, int rows, int C, float eps, void* stream) {
    SR_WRAP(launch_rmsnorm_f32((hipStream_t)stream, x, ldx, w, out, ldo, rows, C, eps));
}
int sr_op_rope_f32(float* x, int ld, const float* cos_t, const float* sin_t, int ldc, int rows, int n_heads, int head_dim, void* stream) {
    SR_WRAP(launch_rope_f32((hipStream_t)stream, x, ld, cos_t, sin_t, ldc, rows, n_heads, head_dim));
}
int sr_op_rope_table_f32(const float* inv_freq, int n_freq, const int32_t* pos, int n_pos, float* cos_t, float* sin_t, void* stream) {
    SR_WRAP(launch_rope_table_f32((hipStream_t)stream, inv_freq, n_freq, pos, n_pos, cos_t, sin_t));
}
int sr_op_sam_preprocess_f32(const uint8_t* img, int h, int w, float* out_chw, int S, void* stream) {
    SR_WRAP(launch_sam_preprocess_f32((hipStream_t)stream, img, h, w, out_chw, S));
}
int sr_op_im2col_f32(const float* chw, int S, int k, int stride, int pad, float* out, int ld, const int32_t* rowmap, void* stream) {
    SR_WRAP(launch_im2col_f32((hipStream_t)stream, chw, S, k, stride, pad, out, ld, rowmap));
}
int sr_op_layernorm_f32(const float* x, int ldx, const float* w, const float* b, float* out, int ldo, int rows, int C, float eps, void* stream) {
    SR_WRAP(launch_layernorm_f32((hipStream_t)stream, x, ldx, w, b, out, ldo, rows, C, eps));
}
int sr_op_maxpool_win_f32(const float* in, int ld_in, int C, int n_win, int ws, float* out, int ld_out, void* stream) {
    SR_WRAP(launch_maxpool_win_f32((hipStream_t)stream, in, ld_in, C, n_win, ws, out, ld_out));
}
int sr_op_ew_f32(const float* a, int lda, const float* b, int ldb, float* out, int ldo, int rows, int C, int mode, void* stream) {
    SR_WRAP(launch_ew_f32((hipStream_t)stream, a, lda, b, ldb, out, ldo, rows, C, mode));
}
int sr_op_upsample2x_add_f32(const float* lat, const float* top, float* out, int H2, int C, int ld, void* stream) {
    SR_WRAP(launch_upsample2x_add_f32((hipStream_t)stream, lat, top, out, H2, C, ld));
}
int sr_op_pixel_shuffle_add_f32(const float* g, int ldg, const float* feat, int ldf, float* out, int ldo, int W, int Co, void* stream) {
    SR_WRAP(launch_pixel_shuffle_add_f32((hipStream_t)stream, g, ldg, feat, ldf, out, ldo, W, Co));
}
int sr_op_quant_mx(const void* x, int ldx, int M, int K, void* q, void* scales, int rows_pad, void* stream) {
    SR_WRAP(launch_quant_mx_act((hipStream_t)stream, (const bf16_t*)x, ldx, M, K, (unsigned char*)q, (unsigned char*)scales, rows_pad));
}
int sr_op_gemm_mx(const void* A8, int lda, const void* a_scale, int a_rows_pad, const void* W8, const float* w_scale, int M, int N, int K, void* out,
                  int ldo, const void* bias, const void* resid, int epilogue, void* stream) {
    GemmArgs a{(const bf16_t*)A8, lda, (const bf16_t*)W8, M, N, K, out, ldo, (const bf16_t*)bias, (const bf16_t*)resid, nullptr, 1, w_scale, 256,
               (const unsigned char*)a_scale, a_rows_pad};
    SR_WRAP(launch_gemm256_mx((hipStream_t)stream, a, epilogue & 0xff));
}
int sr_op_gemv(const void* x, int ldx, const void* W, int M, int N, int K, void* out, int ksplit, int mode, void* stream) {
    GemvArgs a = gv((const bf16_t*)x, ldx, (const bf16_t*)W, M, N, K, out, (mode & 0xff) == GV_SWIGLU ? N / 2 : N);
    a.ksplit = ksplit;
    a.w_tiled = (mode & 0x100) ? 1 : 0;          // bit 8 of `mode`: W is fragment-ordered (tiled16x64)
    a.x_tiled = (mode & 0x800) ? 1 : 0;          // bit 11: x is fragment-ordered (tiled16x64 of [ceil16(M), K])
    if (M > 16) { if (int rc = op_px()) return rc; }
    a.px_counter = (g_op_px_mode || (sr_switches().gemv_xlds & 4)) ? g_op_px : nullptr;
    SR_WRAP(launch_gemv((hipStream_t)stream, a, mode & 0xff));
}
int sr_op_gemv_fused(const void* x, int ldx, const void* W, int M, int N, int K, void* out, int ldo, int mode, const void* bias,
                     const void* norm_w, float eps, const float* slabs, int n_slabs, void* x_out, float* amax_val,
                     int32_t* amax_idx, void* stream) {
    GemvArgs a = gv((const bf16_t*)x, ldx, (const bf16_t*)W, M, N, K, out, ldo);
    a.bias = (const bf16_t*)bias; a.norm_w = (const bf16_t*)norm_w; a.eps = eps; a.slabs = slabs; a.n_slabs = n_slabs;
    a.x_out = (bf16_t*)x_out; a.amax_val = amax_val; a.amax_idx = amax_idx;
    a.w_tiled = (mode & 0x100) ? 1 : 0;
    a.x_tiled = (mode & 0x800) ? 1 : 0;          // bit 11: x fragment-ordered; bit 12: SWIGLU output fragment-ordered
    a.out_tiled = (mode & 0x1000) ? 1 : 0;
    // the op-level callers (tests, probes; one stream at a time) share one process-wide counter block for the x-stationary kernels
    if (M > 16) { if (int rc = op_px()) return rc; }
    a.px_counter = (g_op_px_mode || (sr_switches().gemv_xlds & 4)) ? g_op_px : nullptr;
    SR_WRAP(launch_gemv((hipStream_t)stream, a, mode & 0xff));
}
int sr_op_gemv_set_cus(int n_cus, void* stream) {
    if (n_cus < 0) return fail(nullptr, -22, "sr_op_gemv_set_cus: negative CU count");
    if (int rc = op_px()) return rc;
    g_op_px_mode = n_cus > 0 ? 1 : 0;
    SR_WRAP((int)hipMemsetD32Async((hipDeviceptr_t)(g_op_px + 10 * 16), n_cus, 1, (hipStream_t)stream));
}
int sr_op_gemv_f32_blocks(int N, int M, int K, int has_norm) { return gemv_f32_blocks(N, M, K, has_norm); }
int sr_op_attn_decode(const void* qkv, int qkv_stride, const int32_t* pos, const int32_t* ctx_len, const void* rope_cos,
                      const void* rope_sin, void* kcache, void* vtcache, void* out, int out_stride, int B, int n_q_heads, int n_kv_heads, int ctx_max,
                      float scale, void* scores_scratch, void* stream) {
    int rc = attn_decode_prepare(ctx_max, n_q_heads / n_kv_heads);
    if (rc) return fail(nullptr, rc, "attn_decode_prepare failed with %d", rc);
    DecodeAttnArgs a{(const bf16_t*)qkv, qkv_stride, pos, ctx_len, nullptr, (const bf16_t*)rope_cos, (const bf16_t*)rope_sin, (bf16_t*)kcache, (bf16_t*)vtcache, (bf16_t*)out,
                     out_stride, B, n_q_heads, n_kv_heads, n_q_heads / n_kv_heads, ctx_max, scale, (bf16_t*)scores_scratch, 0, nullptr};
    SR_WRAP(launch_attn_decode((hipStream_t)stream, a));
}
int sr_op_rmsnorm(const void* x, const void* w, void* out, int rows, int H, float eps, void* stream) {
    SR_WRAP(launch_rmsnorm((hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)out, rows, H, eps));
}
int sr_op_resid_rmsnorm(void* x, const float* partials, int ksplit, const void* w, void* out, int rows, int H, float eps, void* stream) {
    SR_WRAP(launch_resid_rmsnorm((hipStream_t)stream, (bf16_t*)x, partials, ksplit, (const bf16_t*)w, (bf16_t*)out, rows, H, eps));
}
int sr_op_quant_f8(void* w_tiled, int N, int K, void* w8, float* scale, void* stream) {
    SR_WRAP(launch_quant_f8((hipStream_t)stream, (bf16_t*)w_tiled, N, K, (unsigned char*)w8, scale));
}
int sr_op_gemv_f8(const void* x, int ldx, const void* w8, const float* w_scale, int M, int N, int K, void* out, int ldo, int mode,
                  const void* bias, const void* norm_w, float eps, int ksplit, void* stream) {
    GemvArgs a{};
    a.x = (const bf16_t*)x; a.ldx = ldx; a.W = nullptr; a.W8 = (const unsigned char*)w8; a.w_scale = w_scale; a.w_tiled = 1;
    a.M = M; a.N = N; a.K = K; a.out = out; a.ldo = ldo; a.ksplit = ksplit > 0 ? ksplit : 1;
    a.bias = (const bf16_t*)bias; a.norm_w = (const bf16_t*)norm_w; a.eps = eps;
    a.x_tiled = (mode & 0x800) ? 1 : 0;          // bit 11: x fragment-ordered; bit 12: SWIGLU output fragment-ordered (as sr_op_gemv_fused)
    a.out_tiled = (mode & 0x1000) ? 1 : 0;
    if (M > 16) { if (int rc = op_px()) return rc; }
    a.px_counter = (g_op_px_mode || (sr_switches().gemv_xlds & 4)) ? g_op_px : nullptr;
    SR_WRAP(launch_gemv((hipStream_t)stream, a, mode & 0xff));
}
int sr_op_sample(const float* logits, int B, int V, float temperature, int top_k, float top_p, float rep_penalty, const uint32_t* seen,
                 uint32_t seed, const int32_t* step, int64_t* out, const float* blk_max, int n_blk, int blk_rows, void* stream) {
    if (!(temperature > 0.f)) return fail(nullptr, -22, "sr_op_sample: temperature must be > 0");
    SampleArgs a{logits, V, B, 1.0f / temperature, top_k, top_p, rep_penalty, seen, (V + 31) / 32, seed, step, reinterpret_cast<long long*>(out),
                 blk_max, n_blk, blk_rows};
    SR_WRAP(launch_sample((hipStream_t)stream, a));
}
int sr_op_argmax(const float* logits, int rows, int V, int32_t* out_idx, void* stream) {
    SR_WRAP(launch_argmax((hipStream_t)stream, logits, rows, V, out_idx));
}

}  // extern "C"
