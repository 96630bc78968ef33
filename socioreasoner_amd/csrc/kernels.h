// Host-side launchers of every device kernel of the hot path (one .hip file per family).
// All take the caller's stream, launch asynchronously, and return 0 or a hipError_t value.
#pragma once
#include "common.h"

// ------------------------------------------------------------------ switches (engine.hip)
// Every SR_* tuning / test switch of the library, read from the environment ONCE (first use, and again at every sr_engine_create and
// sr_switches_reload) -- never in per-call dispatch (hygiene).  Defaults = the shipped behaviour.
struct SrSwitches {
    int splitk;         // SR_SPLITK      1: split-K residual GEMMs of a SMALL static prefill (gives up bit-exact batch invariance; default 0)
    int gemm_bm;        // SR_GEMM_BM     64 / 1288: force a tile shape of the 128-tile GEMM (tools/bench_gemm.py); 0 = pick
    int gemm_ring;      // SR_GEMM_RING   small-M 6-stage ring: 0 never, 1 when it pays (default), 2 whenever K allows
    int gemm256;        // SR_GEMM256     256-tile GEMM: 0 never, 1 when it pays (default), 2 whenever the shape is supported
    int fuse_qkv;       // SR_FUSE_QKV    0: separate rotary + cache-write launches instead of the fused q/k/v epilogues (default 1)
    int g256_group;     // SR_G256_GROUP  m-tiles per W panel of the 256-tile GEMM (default 4)
    int attn2;          // SR_ATTN2       0: round-2 prefill attention kernel (default 1)
    int attn_win64;     // SR_ATTN_WIN64  0: no 64-token window kernel (default 1)
    int attn_vasm;      // SR_ATTN_VASM   0: V^T fragment reads of k_attn_prefill2 left to the compiler (ds_read2st64_b64, 2-way bank conflicts; default 1: hand-issued ds_read_b64)
    int sam_f32_split;  // SR_SAM_F32_SPLIT 0: SAM2's float32 GEMM on the f32-input MFMA (round 4) instead of the three-term bf16 split on the bf16 pipe (default 1)
    int gemv_xlds;      // SR_GEMV_XLDS   bit 0: the 17..32-row gate/up GEMV keeps its activations in registers in a persistent launch (k_gemv_px), bit 1: so does the split-K
                        //                 down-projection (k_gemv32_px), bit 3: the LM head keeps them in LDS (k_gemv32_hpx) -- in the engine's decode step on a CU-limited stream
                        //                 only (sr_rows_set_cus), bit 2: on the whole chip too; 0: all re-read x from L2 in every wave (rounds 2-5).  Round 6, default 11;
                        //                 bit-identical: A/B + test hook
    int gemv_counted;   // SR_GEMV_COUNTED 0: the <= 32-row decode GEMVs use the conditional-refill ring loops of rounds 1-4 (vmcnt(0) every round) instead of the
                        //                 unconditional refills with counted vmcnt waits (default 1; bit-identical: A/B + test hook)
};
const SrSwitches& sr_switches();

// ------------------------------------------------------------------ gemm.hip (MFMA, M large)
// out[M,N] = A[M,K] . W[N,K]^T, bf16 operands, float32 accumulate.  K % 64 == 0, N % 16 == 0.
enum { EPI_STORE = 0, EPI_RESID = 1, EPI_SWIGLU = 2, EPI_GELU = 3, EPI_F32 = 4,
       EPI_LMQKV = 5,     // LM q/k/v Linear + mRoPE + KV-cache write in the epilogue (gemm256.hip only; GemmArgs.rope)
       EPI_VITQKV = 6 };  // ViT qkv Linear + 2-D rotary + V^T write in the epilogue (gemm256.hip only; GemmArgs.vrope)
// fused ViT qkv epilogue (replaces k_vit_rope + k_vit_vtranspose).  Needs the q / k output channels of every head in the engine's
// PAIRED order (vit_qk_perm below): rotary partners 8 columns apart inside one 16-column MFMA tile.
struct VitRope {
    const float* cos_t; const float* sin_t;   // [row][head_dim / 2] float32 (hf:124-135 angles)
    bf16_t* vt; int vt_stride;                // V^T [C][vt_stride]
    int C, hd;                                // hidden size (= heads * hd), head dim
};
// paired channel order of a ViT q / k head (head_dim hd, hd / 2 % 8 == 0): HF channel d = part * hd/2 + j  (part 0: x1, 1: x2 of rotary
// pair j) sits at (j / 8) * 16 + part * 8 + j % 8.  Dot products q.k are unchanged (same order on both sides).
__host__ __device__ inline int vit_qk_perm(int d, int hd) {
    const int half = hd / 2, part = d / half, j = d % half;
    return (j / 8) * 16 + part * 8 + (j % 8);
}
// what the fused LM q/k/v epilogue needs beyond the GEMM operands (the arguments of k_lm_rope_prefill, which it replaces)
struct QkvRope {
    const int* pos3;            // [3][n_tok] mRoPE position ids
    const int* tok_slot;        // [n_tok] cache slot
    const int* tok_idx;         // [n_tok] index of the token inside its sequence (cache row)
    const bf16_t* rope_cos;     // [max_pos+1][64] bf16 tables
    const bf16_t* rope_sin;
    bf16_t* kcache; bf16_t* vtcache;
    int n_tok, n_q_heads, n_kv_heads, sec0, sec1, ctx_max;
};
struct GemmArgs {
    const bf16_t* A; int lda;
    const bf16_t* W;            // [N,K] row-major (nn.Linear layout); SWIGLU: rows interleaved 16 gate / 16 up
    int M, N, K;
    void* out; int ldo;         // bf16 (float for EPI_F32); SWIGLU writes N/2 columns
    const bf16_t* bias;         // [N] or null
    const bf16_t* resid;        // EPI_RESID: [M, ldo] (may alias out)
    const int* rowmap;          // optional destination row per source row
    int w_tiled;                // W stored fragment-ordered (tiled16x64, see common.h) instead of row-major
    const float* w_scale;       // optional [N]: per-output-channel scale applied to the accumulator (fp8-quantised W)
    int force_tile;             // 0: launch_gemm picks the kernel; 256 / 128: force gemm256.hip / gemm.hip (tests, tuning)
    const unsigned char* a_scale;   // MX fp8 path (launch_gemm256_mx): e8m0 block scales of A, [K/128][a_rows_pad][4]
    int a_rows_pad;
    QkvRope rope;               // EPI_LMQKV only
    VitRope vrope;              // EPI_VITQKV only
    int ksplit;                 // > 1 (EPI_F32, gemm.hip kernel only, bias must be null): K is split over gridDim.y blocks, block y writes its float32 partial
                                // products to out + y * M * ldo (slabs the consumer sums: launch_resid_rmsnorm)
    int gelu_fast;              // EPI_GELU: 1 = gelu_fast_f (common.h) instead of the erff form, 2 = ReLU (the activation epilogue's third function)
    int out_tiled;              // EPI_SWIGLU (gemm.hip kernel): write the activation fragment-ordered (tiled16x64 of [ceil16(M)][ldo], common.h)
};
bool gemm_fuses_vitqkv(const GemmArgs& a);
// true when launch_gemm would run `a` with the fused q/k/v epilogue (same predicate as its dispatch to gemm256.hip)
bool gemm_fuses_lmqkv(const GemmArgs& a);
int launch_gemm(hipStream_t s, const GemmArgs& a, int epi);
int gemm_prepare_decode();      // attribute calls of the tile kernels a decode step launches under stream capture (engines with more than 64 rows)
// gemm256.hip: 256 x 256 x 64 tile, 8-phase ping-pong schedule (large M); launch_gemm dispatches to it
bool gemm256_supports(const GemmArgs& a);
int launch_gemm256(hipStream_t s, const GemmArgs& a, int epi);
int launch_gemm256_mx(hipStream_t s, const GemmArgs& a, int epi);
// MX activation quantiser (OCP MX, e4m3 elements, 32-wide blocks): x bf16 [M][ldx] -> q fp8 [M][K] + e8m0 scales [K/128][rows_pad][4]
int launch_quant_mx_act(hipStream_t s, const bf16_t* x, int ldx, int M, int K, unsigned char* q, unsigned char* scales, int rows_pad);

// ------------------------------------------------------------------ gemv.hip (weight streaming, M <= 32)
enum { GV_PARTIAL = 0, GV_SWIGLU = 1, GV_F32 = 2, GV_BIAS = 3, GV_RESID = 4 };

struct GemvArgs {
    const bf16_t* x; int ldx;   // [M, K] B operand; with norm_w: the residual stream the RMSNorm prologue reads
    const bf16_t* W;            // [N, K]
    int M, N, K;
    void* out; int ldo;         // PARTIAL: float [ksplit][M][N]; SWIGLU: bf16 [M][N/2]; F32: float [M][N];
                                // BIAS: bf16 [M][ldo]; RESID: bf16 [M][ldo] updated in place
    int ksplit;                 // PARTIAL only (gridDim.y)
    const bf16_t* bias;         // BIAS
    const bf16_t* norm_w; float eps;        // non-null: fused RMSNorm prologue (BIAS / SWIGLU / F32)
    const float* slabs; int n_slabs;        // pending residual [n_slabs][M][K] added before the norm
    bf16_t* x_out;                          // with slabs: block 0 stores h = bf16(x + bf16(sum slabs)) here (ld = ldx)
    float* amax_val; int* amax_idx;         // F32: per-block (max, lowest index) [M][gridDim.x] (null: skip)
    int w_tiled;                            // W stored fragment-ordered (tiled16x64) instead of row-major
    const unsigned char* W8;                // non-null: stream THIS fp8 image (tiled8, common.h) instead of W ...
    const float* w_scale;                   // ... and scale output channel n by w_scale[n]
    int x_tiled;                            // x is stored fragment-ordered (tiled16x64 of a [ceil16(M), K] matrix, common.h): a wave's
                                            // x load is 1 KB contiguous instead of 16 rows x 64 B (batches > 4, no fused norm)
    int out_tiled;                          // SWIGLU: write the activation fragment-ordered (it is the next GEMV's x)
    int force32;                            // always the 32-row MFMA variant (whatever M): a row's result then does not depend on how many rows share the launch
    int counted;                            // set by launch_gemv from SR_GEMV_COUNTED: un-staged launches with fragment-ordered x use the loop whose refills are all unconditional (counted vmcnt waits)
    unsigned* px_counter;                   // round 6, non-null: 11 words 64 bytes apart (8 ticket shards + blocks done: zero between launches; word 160 = CU limit of the static
                                            // deal, 0 = the grid) -- SWIGLU / PARTIAL at 17..32 rows with fragment-ordered x and weights may run as the PERSISTENT kernels that
                                            // keep x in registers (k_gemv_px / k_gemv32_px: same per-tile arithmetic, same bits)
};
int launch_gemv(hipStream_t s, const GemvArgs& a, int mode);
int gemv_prepare_px();     // attribute call of the persistent x-resident kernel (outside stream capture: sr_engine_create)
int gemv_f32_blocks(int N, int M, int K, int has_norm);
int gemv_f32_block_rows(int N, int M, int K, int has_norm);   // vocabulary rows per block of that launch    // gridDim.x of the F32 launch (length of the amax rows)

// ------------------------------------------------------------------ attention.hip
// one 64-query tile of one sequence.  K element (kvh, key j, d) = k[(k_row0 + j)*k_stride + kvh*k_head_stride + d];
// V^T element (kvh, d, key j) = vt[vt_off + kvh*vt_head_stride + d*vt_stride + j]
struct AttnWork { int q_row0; int seq_len; int q_off; int k_row0; long long vt_off;
                  int q_len; int pad_; };   // q_len: queries of the sequence when they are fewer than its keys (Hiera's pooled queries); 0 = seq_len
struct AttnArgs {
    const bf16_t* q; int q_stride;        // element (row, h*HD + d) at q[row*q_stride + h*HD + d]
    const bf16_t* k; int k_stride; long long k_head_stride;
    const bf16_t* vt; int vt_stride; long long vt_head_stride;
    bf16_t* out; int out_stride;          // out[row*out_stride + h*HD + d]
    const AttnWork* work; int n_work;
    int n_heads, group;                   // kv head = h / group
    float scale;
    int causal;
    int q_tile;                           // queries per work item: 0 / 64 (k_attn_prefill), or 128 (k_attn_prefill2, MHA full attention)
    int v2_ok;                            // caller's promise for k_attn_prefill2: vt + vt_off + 64 * j is 16-byte aligned for every work item,
                                          // the V^T rows are readable (and finite) up to the end of the last 64-key tile
    int win64;                            // caller's promise for k_attn_win64: EVERY work item is one whole window of exactly 64 tokens (q_off 0,
                                          // seq_len 64, no pooled queries) whose V^T key run starts 8-byte aligned
};
int launch_attn_prefill(hipStream_t s, const AttnArgs& a, int head_dim);
int attn_prefill_variant(const AttnArgs& a, int head_dim);   // 2: k_attn_prefill2 takes it, 1: k_attn_prefill

// decode: q/k/v rows (after the Linear bias, before rope) -> mRoPE -> KV-cache append -> attention (two launches)
struct DecodeAttnArgs {
    const bf16_t* qkv; int qkv_stride;    // [B, (Hq + 2 Hkv)*128] bf16
    const int* pos;                       // [B] rotary position of the new token (all three mRoPE axes equal)
    const int* ctx_len;                   // [B] keys in the cache INCLUDING the new token
    const int* slots;                     // [B] cache slot of each row (null: identity)
    const bf16_t* rope_cos;               // [max_pos+1][64] bf16 cos(pos * inv_freq) (hf:486-539), sin likewise
    const bf16_t* rope_sin;
    bf16_t* kcache;                       // [slot][kvh][ctx_max][128]
    bf16_t* vtcache;                      // [slot][kvh][128][ctx_max]
    bf16_t* out; int out_stride;          // [B, Hq*128]
    int B, n_q_heads, n_kv_heads, group, ctx_max;
    float scale;
    bf16_t* scores;                       // scratch [B][kvh][group][ctx_max] bf16 between the two decode launches
    int out_tiled;                        // write `out` fragment-ordered (tiled16x64, K = out_stride): it is the o_proj GEMV's x
    const int* frozen;                    // [B] or null: rows whose flag is set append nothing to the cache (finished rows: their
                                          // KV slot may already be staged for the next sequence -- sr_admit_stage)
    const float* row_cs;                  // [B][128] or null: cos | sin of pos[b] as float32 (written by k_step); null: looked up from the tables
};
int launch_attn_decode(hipStream_t s, const DecodeAttnArgs& a);
int attn_decode_prepare(int ctx_max, int group);

// ------------------------------------------------------------------ elementwise.hip
// out_tiled (decode rows only): `out` is written fragment-ordered (tiled16x64 of [ceil16(rows), H]) for the consuming GEMV
// per_wave: always the wave-per-row kernel.  The block-per-row kernel (rows <= 64: decode) sums the squares in another order, and a
// prefill must give a prompt the same bits whether 47 or 4700 rows are normalised with it.
int launch_rmsnorm(hipStream_t s, const bf16_t* x, const bf16_t* w, bf16_t* out, int rows, int H, float eps, int out_tiled = 0, int per_wave = 0);
// h = r(x + r(sum_ks part[ks] + bias)) written back to x; out = rmsnorm(h) * w.   part may be null (plain norm).
int launch_resid_rmsnorm(hipStream_t s, bf16_t* x, const float* part, int ksplit, const bf16_t* w, bf16_t* out,
                         int rows, int H, float eps, int out_tiled = 0, int per_wave = 0);
int launch_vit_rope(hipStream_t s, bf16_t* qkv, int n_rows, int n_heads, int head_dim, const float* cos_t,
                    const float* sin_t, bf16_t* vt, int vt_stride, int paired = 0);
// prefill: rope q,k in place in qkv [T, (Hq+2Hkv)*128]; write K / V^T into the cache at (slot, pos_in_seq)
struct LmRopeArgs {
    bf16_t* qkv; int n_tok; int n_q_heads, n_kv_heads;
    const int* pos3;            // [3][n_tok] mRoPE position ids
    const int* tok_slot;        // [n_tok] cache slot
    const int* tok_idx;         // [n_tok] index of the token inside its sequence (cache row)
    const bf16_t* rope_cos;     // [max_pos+1][64] bf16 tables
    const bf16_t* rope_sin;
    int sec0, sec1;             // mrope_section boundaries in rotary pairs (16, 40)
    bf16_t* kcache; bf16_t* vtcache; int ctx_max;
};
int launch_lm_rope_prefill(hipStream_t s, const LmRopeArgs& a);
// cos/sin tables of the LM rotary embedding: [n_pos][64] bf16, angle = float(pos) * inv_freq[f] in float32 (hf:526-539)
int launch_rope_table(hipStream_t s, const float* inv_freq, int n_pos, bf16_t* cos_t, bf16_t* sin_t);
int launch_embed(hipStream_t s, const int* src, const bf16_t* table, const bf16_t* image_embeds, bf16_t* out,
                 int n_tok, int H, int table_tiled);
int launch_gather_rows(hipStream_t s, const bf16_t* in, const int* rows, bf16_t* out, int n, int H);
int launch_patchify(hipStream_t s, const uint8_t* img, int h, int w, const bf16_t* lut, bf16_t* out, int ld_out,
                    int patch, int merge, int temporal);
int launch_f32_to_bf16_pad(hipStream_t s, const float* in, int rows, int cols, bf16_t* out, int ld_out);
int launch_argmax(hipStream_t s, const float* logits, int rows, int V, int* out_idx);
// per-step bookkeeping on the device (one captured graph replays for every step): finishes the greedy argmax from the
// LM-head partials, logs the token, handles eos / teacher forcing, gathers the next input embedding, advances state
struct StepArgs {
    const float* amax_val; const int* amax_idx; int n_part;     // [B][n_part]
    int* cur_tok; int* ctx_len; int* pos; int* step; int* finished; int* tokens_out;
    int max_new; const int* eos; int n_eos; int pad_id; int B;
    const int* forced;          // optional [B][max_new]: token fed back instead of the greedy one (teacher forcing)
    const bf16_t* table; bf16_t* x; int H; int table_tiled;     // embedding gather of the token fed back
    const long long* chosen;    // optional [B]: this step's token picked by the caller (replaces the greedy argmax)
    const int* row_limit;       // optional [B]: a row finishes after this many generated tokens (continuous batching)
    int* n_gen;                 // [B]: tokens generated while the row was live (pads written after eos are not counted)
    const bf16_t* rope_cos; const bf16_t* rope_sin;     // optional LM rotary tables [pos][64] ...
    float* row_cs;              // ... and [B][128]: cos | sin of every row's NEW position, for the decode attention of this step (which then
                                // does not have to chase pos[b] -> table row through two dependent loads in every layer)
};
// fp8 quantisation of a fragment-ordered bf16 matrix [N, K]: scale[n] = amax_n / 448 (1 if the row is zero),
// q = fp8(W / scale) -> W8 (tiled8); W itself is overwritten with q as bf16 (what the prefill GEMM multiplies, scaled in its epilogue)
int launch_quant_f8(hipStream_t s, bf16_t* W_tiled, int N, int K, unsigned char* W8, float* scale);
int launch_step(hipStream_t s, const StepArgs& a);
// continuous batching: install n freshly prefilled sequences into batch rows (state + pending first token)
struct AdmitArgs {
    const int* rows; const int* ctx; const int* pos; const int* limit; const int* first_tok; int n;
    int* ctx_len; int* d_pos; int* slots; int* finished; int* step; int* row_limit; int* n_gen;
    float* amax_val; int* amax_idx; int n_part;
    const int* kv_slot;         // KV-cache slot of each admitted sequence (null: slot = row)
};
int launch_admit_rows(hipStream_t s, const AdmitArgs& a);
int launch_rows_abort(hipStream_t s, const unsigned row_mask[4], int* finished);      // finished[b] = 1 for every bit b of the 128-bit row mask
// ------------------------------------------------------------------ sample.hip
struct SampleArgs {
    const float* logits; int V; int B;      // float32 [B, V]
    float inv_temp; int top_k; float top_p; float rep_penalty;
    const unsigned* seen; int seen_words;   // optional bitmask [B][seen_words] of tokens already in prompt / output
    unsigned seed; const int* step;         // RNG stream: (seed, row, step[row])
    long long* out;                         // [B] chosen token
    const float* blk_max; int n_blk, blk_rows;   // optional: the LM head's per-block maxima [B][n_blk], block j = ids [j*blk_rows, (j+1)*blk_rows)
};
int launch_sample(hipStream_t s, const SampleArgs& a);
int launch_scatter_rows(hipStream_t s, const int* rows, const long long* src, long long* dst, int n);   // dst[rows[i]] = src[i]
int launch_mark_prompt(hipStream_t s, const int* src, const int* lastrow, unsigned* seen, int seen_words, int B);
int launch_mark_chosen(hipStream_t s, const long long* chosen, unsigned* seen, int seen_words, int B);
int launch_next_ids(hipStream_t s, const float* amax_val, const int* amax_idx, int n_part, int B, long long* out);
int launch_synth_fill(hipStream_t s, bf16_t* out, long long n, uint32_t key, float base, float scale);
int launch_fill_zero(hipStream_t s, void* p, size_t bytes);
int launch_load2d(hipStream_t s, const void* src, int dtype, long long rows, long long cols, bf16_t* dst, long long dst_ld,
                  int mode, long long row_off, int tiled);

// ------------------------------------------------------------------ sam.hip (SAM2 image path: the passes between its GEMMs and attention launches)
int launch_sam_preprocess(hipStream_t s, const uint8_t* img, int h, int w, bf16_t* out_chw, int S);
int launch_im2col(hipStream_t s, const bf16_t* chw, int S, int k, int stride, int pad, bf16_t* out, int ld, const int* rowmap);
int launch_layernorm(hipStream_t s, const bf16_t* x, int ldx, const bf16_t* w, const bf16_t* b, bf16_t* out, int ldo, int rows, int C, float eps);
int launch_maxpool_win(hipStream_t s, const bf16_t* in, int ld_in, int C, int n_win, int ws, bf16_t* out, int ld_out);
int launch_ew(hipStream_t s, const bf16_t* a, int lda, const bf16_t* b, int ldb, bf16_t* out, int ldo, int rows, int C, int mode);
int launch_transpose(hipStream_t s, const bf16_t* in, int ld_in, int rows, int cols, bf16_t* out, int ld_out);
int launch_upsample2x_add(hipStream_t s, const bf16_t* lat, const bf16_t* top, bf16_t* out, int H2, int C, int ld);
int launch_pixel_shuffle_add(hipStream_t s, const bf16_t* g, int ldg, const bf16_t* feat, int ldf, bf16_t* out, int ldo, int W, int Co);
int launch_mask_resize_or(hipStream_t s, const float* low, int ld, int col0, int n, int m, const float* score, uint8_t* acc, float* logits_out, int h, int w);

// float32 storage mode of the same passes (the reference runs SAM2 in float32) + the float32 GEMM / attention of sam_f32.hip
int launch_sam_preprocess_f32(hipStream_t s, const uint8_t* img, int h, int w, float* out_chw, int S);
int launch_im2col_f32(hipStream_t s, const float* chw, int S, int k, int stride, int pad, float* out, int ld, const int* rowmap);
int launch_layernorm_f32(hipStream_t s, const float* x, int ldx, const float* w, const float* b, float* out, int ldo, int rows, int C, float eps);
int launch_maxpool_win_f32(hipStream_t s, const float* in, int ld_in, int C, int n_win, int ws, float* out, int ld_out);
int launch_ew_f32(hipStream_t s, const float* a, int lda, const float* b, int ldb, float* out, int ldo, int rows, int C, int mode);
int launch_upsample2x_add_f32(hipStream_t s, const float* lat, const float* top, float* out, int H2, int C, int ld);
int launch_pixel_shuffle_add_f32(hipStream_t s, const float* g, int ldg, const float* feat, int ldf, float* out, int ldo, int W, int Co);
// out[M][N] = act(A . W^T + bias) (+ resid), everything float32 row-major; K % 16 == 0, N / lda / ldo % 4 == 0, 16-byte aligned bases
struct GemmF32Args {
    const float* A; int lda;
    const float* W;             // [N][K]
    int M, N, K;
    float* out; int ldo;
    const float* bias;          // [N] or null
    const float* resid;         // [*, ldo] indexed by DESTINATION row (may alias out) or null; added after the activation
    const int* rowmap;          // optional destination row per source row
    int act;                    // 0 none, 1 GELU (erf form), 2 ReLU
};
int launch_gemm_f32(hipStream_t s, const GemmF32Args& a);
// float32 attention over the same work items as launch_attn_prefill (q_tile 64); V is ROW-major here ([key][head * hd + d])
struct AttnF32Args {
    const float* q; int q_stride;
    const float* k; int k_stride;
    const float* v; int v_stride;
    float* out; int out_stride;
    const AttnWork* work; int n_work;
    int n_heads;
    float scale;
    int causal;                 // 1: query i of an item sees keys 0 .. (seq_len - queries of the sequence + q_off + i)  (the LM; head_dim 128)
};
// float32 row ops of the verification path (f32_ops.hip): RMSNorm (hf:65-79 without the bf16 roundings) and rotate-half rotary with per-row cos | sin
int launch_rmsnorm_f32(hipStream_t s, const float* x, int ldx, const float* w, float* out, int ldo, int rows, int C, float eps);
int launch_rope_f32(hipStream_t s, float* x, int ld, const float* cos_t, const float* sin_t, int ldc, int rows, int n_heads, int head_dim);
int launch_rope_table_f32(hipStream_t s, const float* inv_freq, int n_freq, const int* pos, int n_pos, float* cos_t, float* sin_t);
int launch_attn_f32(hipStream_t s, const AttnF32Args& a, int head_dim);

// ------------------------------------------------------------------ raster.hip
int launch_mask_union(hipStream_t s, uint8_t* acc, const uint8_t* m, size_t n);
int launch_resize_nearest_u8(hipStream_t s, const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw);
int launch_iou_counts(hipStream_t s, const uint8_t* p, const uint8_t* g, size_t n, long long* out2);
int launch_iou_counts_batched(hipStream_t s, const uint8_t* p, const uint8_t* g, size_t n, int n_items, long long* out);     // out [n_items][2], zeroed here
int launch_render_overlay(hipStream_t s, uint8_t* img, int h, int w, const uint8_t* mask, int mh, int mw,
                          const int* boxes, int nb);
