// Internal launcher declarations shared by the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define UW_F32 0
#define UW_BF16 1

static inline size_t uw_elem_size(int dtype) { return dtype == UW_F32 ? 4 : 2; }

// ---- uw_gemm.hip ------------------------------------------------------------------------------
struct GemmParams {
    const void* A = nullptr;  // [M][K], row stride lda (elements of T)
    const void* B = nullptr;  // [N][K], row stride ldb
    void* C = nullptr;        // [M][N], row stride ldc (elements of TC)
    const float* bias = nullptr;
    const float* residual = nullptr;  // f32 [M][N], row stride ldr; may alias C
    long lda = 0, ldb = 0, ldc = 0, ldr = 0;
    long sA = 0, sB = 0, sC = 0, sR = 0;  // batch strides (elements)
    int M = 0, N = 0, K = 0, batch = 1;
    int bias_mode = 0;  // 0 none, 1 bias[n], 2 bias[m]
    int act = 0;        // 0 none, 1 GELU(erf)
    int ab_packed = 0, c_packed = 0;  // fragment-major operands / output (skinny form only)
    int c_xlayout = 0;                // skinny form: f32 C and residual are in the step's x layout (x_elem_offset)
    // skinny form, LayerNorm fused: A is the f32 residual stream in the x layout, B the layer's weights, gamma [K] the
    // LayerNorm scale, bias holds bias + W.beta, wsum[n] = sum_k gamma[k] B[n][k];
    // C = rstd * ((A * gamma).B^T - mean * wsum) + bias
    int ln_fused = 0;
    const float* wsum = nullptr;
    const float* gamma = nullptr;
    // Split activations (bf16 compute only).  A value v is carried as hi = bf16(v) and lo = bf16(v - hi), which keeps
    // about 16 mantissa bits; a product against W then runs as hi.W + lo.W into the same f32 accumulator.
    //   tile kernels: the operand row is [hi | lo] (2K elements); the OTHER operand repeats after kwrap elements of K
    //                 (kwrap_b: B repeats, kwrap_a: A repeats), so the weights are not duplicated in memory.
    //                 c_split: C (bf16) is written as [hi | lo], lo at column n + N (ldc >= 2N).
    //   skinny form:  A_lo / C_lo are the lo planes of fragment-major A / C.
    int kwrap_a = 0, kwrap_b = 0, c_split = 0;
    // K-blocked operands (two-workgroup tile kernel only).  A [R][K] matrix stored as [K / KT][R][KT], KT = 64 bytes of K:
    // the 16 rows x 64 bytes one LDS-DMA instruction fetches are then 1 KiB of consecutive bytes (whole cache lines)
    // instead of sixteen half lines, which nearly doubles what a CU takes in.  kstep_x = R * KT (elements between
    // K-steps) with ldx = KT marks operand x as K-blocked; c_kblock = rows of C writes the output in the same form.
    long kstep_a = 0, kstep_b = 0, c_kblock = 0;
    int panel_n = 0;   // set by the launcher: column tiles per panel of the panel tile order (0: plain order)
    // second, TRANSPOSED copy of the output columns n >= c2_col0 (two-workgroup tile kernel, compute-dtype output):
    // C2[(n - c2_col0) * ldc2 + m], batch item b at C2 + b * sC2 -- V^T beside the head-contiguous K|V of a projection
    void* C2 = nullptr;
    long ldc2 = 0, sC2 = 0;
    int c2_col0 = 0;
    const void* A_lo = nullptr;
    void* C_lo = nullptr;
    long c_group = 0;   // 0: C[m][n] at m*ldc + n.  Else columns are stored in groups of 64 (one attention head):
                        // C[m][n] at (n / 64) * c_group + m * ldc + (n % 64), ldc = 64 for a head-contiguous cache
    // Skinny form: fragment-major weights [pf_n][pf_k] of the product that FOLLOWS this one in the decode step (or null).
    // The launch then carries extra workgroups that only request those weights, each the block that the workgroup of the
    // following launch scheduled on the same XCD will read, so that product does not start by waiting for HBM.  (The
    // requested lines are found again on the memory side -- a product's L2 fetch count is the same with warm and with
    // cold weights -- and fastest when the request came from the XCD that reads them.)  pf_cols: output columns per
    // workgroup of the following launch (uw_gemm_skinny_block_columns).  pf_first / pf_shift / pf_never are filled in
    // by the launcher.
    const void* pf = nullptr;
    int pf_n = 0, pf_k = 0, pf_cols = 0;
    int pf_first = 0x7fffffff, pf_shift = 0, pf_never = 0;   // pf_first: workgroups from this x index on compute nothing
};
// output columns per workgroup that uw_gemm_skinny picks for a product of M rows and N columns (hilo: split A planes)
int uw_gemm_skinny_block_columns(int dtype, int M, int N, int ln_fused, int hilo);
// dtype: element type of A and B.  out_f32: C is f32 even when dtype is bf16 (always f32 for UW_F32).
int uw_gemm(int dtype, int out_f32, const GemmParams& p, hipStream_t s);
// M <= 128, batch 1: weight-streaming form for the decode step.  With p.ab_packed both A and B are in the
// fragment-major layout (frag_elem_offset in uw_common.h; rows padded to 16, lda/ldb unused); with p.c_packed
// C (compute dtype only) is written in that layout too, ready to be the next product's A.
int uw_gemm_skinny(int dtype, int out_f32, const GemmParams& p, hipStream_t s);
// bf16, M <= 128: C (f32) [M][N] = (A_hi + A_lo) . B^T with A rows stored [hi | lo] (lo starts K elements in, lda >= 2K),
// B [N][K]; the tied output projection of a decode step.  Uses A, B, C, lda, ldb, ldc, M, N, K of the parameter block.
int uw_gemm_rows_hilo(const GemmParams& p, hipStream_t s);
// Tied output projection of a decode step for at most 16 windows, bf16 compute, in ONE launch: final LayerNorm of the
// rows (held in LDS as hi + lo MFMA operand blocks), product with the [V][d] embedding streamed from memory once, and
// per row the arg-max candidate of every 256-column workgroup, so that the greedy step need not scan the logits.
struct LogitsRows16Params {
    const float* x = nullptr;      // residual stream rows in the step's x layout (x_elem_offset<bf16_t>), one 16-row band
    const float* gamma = nullptr;  // final LayerNorm scale and shift [d]
    const float* beta = nullptr;
    const void* W = nullptr;       // [V][d] bf16, fragment-major (uw_pack_frag): a wave fetches an operand block as 1 KiB of consecutive bytes
    float* logits = nullptr;       // [M][ldl] or null (candidates only)
    long ldl = 0;
    float* cand_val = nullptr;     // [16][uw_logits_rows16_blocks(V)] row maximum per workgroup (NaN if the block holds one) or null
    int32_t* cand_idx = nullptr;   // ... and its first index
    int M = 0, V = 0, d = 0;
};
int uw_logits_rows16_blocks(int V);
int uw_logits_rows16(const LogitsRows16Params& p, hipStream_t s);
// row-major [R][K] (row stride ld) -> fragment-major, rows zero-padded to a multiple of 16
int uw_pack_frag(int dtype, const void* src, long ld, void* dst, int R, int K, hipStream_t s);
// row-major [R][K] (dtype) -> K-blocked [K / KT][R][KT], KT = 64 bytes of K (see GemmParams::kstep_a)
int uw_pack_kblock(int dtype, const void* src, void* dst, long R, int K, hipStream_t s);

// ---- uw_elementwise.hip -----------------------------------------------------------------------
// y[r][:] = LayerNorm(x[r][:]) * gamma + beta, eps 1e-5.  y in dtype; y32 (optional) also gets f32.
// packed = 1: y is written fragment-major (frag_elem_offset), the A operand of the decode-step GEMMs
// packed = 2: y is written K-blocked ([d / KT][rows][KT], see GemmParams), the operand form of the encoder's tile GEMMs
// split (bf16 only): row r of y is [hi | lo], 2d elements (see GemmParams)
int uw_layernorm(int dtype, const float* x, void* y, float* y32, const float* gamma, const float* beta, long rows,
                 int d, hipStream_t s, int packed = 0, int x_layout = 0, int split = 0);
// Operands of a LayerNorm-fused product (rows of w): dst[n][k] = T(w[n][k] * scale), the weights as every other kernel
// holds them; bias_out[n] = scale * bias_in[n] + sum_k float(dst[n][k]) * beta[k]; wsum[n] = sum_k float(dst[n][k]) * gamma[k]
int uw_fold_layernorm(int dtype, const float* w, const float* bias_in, const float* gamma, const float* beta, float scale,
                      void* dst, float* bias_out, float* wsum, int N, int K, hipStream_t s);
// dst (dtype) = src (f32) * scale, n elements
int uw_convert(int dtype, const float* src, void* dst, long n, float scale, hipStream_t s);
// conv weight [co][ci][3] f32 -> [co][kpad] dtype with k index = tap * ci_count + ci, zero padded to kpad
int uw_pack_conv_weight(int dtype, const float* w, void* dst, int co, int ci, int kpad, hipStream_t s);
// mel f32 [B][n_mels][3000] -> time-major zero-padded [B][3002 + slack][n_mels] dtype (row t+1 = frame t)
int uw_mel_to_time_major(int dtype, const float* mel, void* dst, int B, int n_mels, int n_frames, long dst_stride,
                         hipStream_t s);
// x[b][t][:] += pos[t][:]  (f32, in place); rows = B*T
int uw_add_positions(float* x, const float* pos, int B, int T, int d, hipStream_t s);
// decoder input embedding: x[b][j][:] = tok_emb[ids[b*ld_ids + j]] + pos_emb[pos0[b] + j], j < T
int uw_embed_tokens(float* x, const float* tok_emb, const float* pos_emb, const int32_t* ids, int ld_ids,
                    const int32_t* pos0, int B, int T, int d, int vocab, int n_ctx, hipStream_t s);
// first-max argmax over logits rows: out[b] = argmax_n logits[b][n]
int uw_argmax_rows(const float* logits, long ld, int B, int V, int32_t* out, hipStream_t s);

// ---- uw_attention.hip -------------------------------------------------------------------------
struct AttnParams {
    const void* Q = nullptr;   // [B][Sq][ldq], head h at column h*64, already scaled by 1/8
    const void* K = nullptr;   // [B][Sk][ldk]
    const void* VT = nullptr;  // [B][H*64][ldvt]  (V transposed: row = h*64 + hd, column = key)
    void* O = nullptr;         // [B][Sq][ldo]
    long ldq = 0, ldk = 0, ldvt = 0, ldo = 0;
    long sKh = 64;             // head stride of K (elements): 64 when heads are columns of a row, Sk*64 when head-contiguous
    long sQ = 0, sK = 0, sVT = 0, sO = 0;  // per-window strides (elements)
    int B = 0, H = 0, Sq = 0, Sk = 0;
    int causal = 0;            // query i sees keys <= i + (Sk - Sq)
    const int32_t* sq_len = nullptr;  // optional per-window valid query count (<= Sq); rows beyond are skipped
    int o_split = 0;           // bf16 only: O rows are [hi | lo], lo at column H*64 + c (ldo >= 2*H*64)
    long o_kblock = 0;         // > 0: O is written K-blocked (see GemmParams) as a [o_kblock][H*64] matrix, window b at rows b*Sq
};
int uw_flash_attention(int dtype, const AttnParams& p, hipStream_t s);

struct ProbParams {
    const void* Q = nullptr;  // [B][Sq][ldq]
    const void* K = nullptr;  // [B][Sk][ldk]
    float* P = nullptr;       // rows[b][sel][n][Sk]  (n = query index - q_first[b])
    long ldq = 0, ldk = 0, sQ = 0, sK = 0;
    long sKh = 64;              // head stride of K (elements)
    long sP_b = 0, sP_sel = 0;  // strides of P in floats; row stride is Sk
    int B = 0, Sq = 0, Sk = 0;
    int n_heads_sel = 0;
    int heads[32];              // head index of each selected head in this layer
    int sel_slot[32];           // its slot in the [sel] axis of P
    const int32_t* q_first = nullptr;  // per window: first query row to emit
    const int32_t* q_count = nullptr;  // per window: number of rows to emit
};
int uw_attention_probs(int dtype, const ProbParams& p, hipStream_t s);

// ---- uw_decode.hip ----------------------------------------------------------------------------
struct SelfDecodeParams {
    const void* qkv = nullptr;  // [B][3d] (q scaled | k | v) of the current token
    void* kcache = nullptr;     // this layer's cache: element (b, h, r, e) at b*cache_sB + h*cache_sH + r*ld_cache + e
    void* vcache = nullptr;
    void* out = nullptr;        // [B][d]
    void* out_lo = nullptr;     // bf16, fragment-major only: lo plane of the split output (see GemmParams)
    int out_packed = 0;         // out in the fragment-major layout
    const int32_t* pos = nullptr;     // per window position of the current token
    const int32_t* active = nullptr;  // per window: 0 = finished (skip)
    long ld_cache = 0;          // row stride of the caches (elements)
    long cache_sB = 0, cache_sH = 0;  // window and head strides of the caches (elements)
    int B = 0, H = 0, d = 0, n_ctx = 0;
    int max_rows = 0;           // host-side upper bound of pos + 1 over the windows (0: n_ctx); sizes the bf16 kernel's rounds
};
int uw_self_attn_decode(int dtype, const SelfDecodeParams& p, hipStream_t s);

struct CrossDecodeParams {
    const void* q = nullptr;   // [B][d] scaled
    const void* K = nullptr;   // this layer: element (b, h, r, e) at b*kv_sB + h*kv_sH + r*ld_kv + e
    const void* V = nullptr;
    void* out = nullptr;       // [B][d]
    void* out_lo = nullptr;    // bf16, fragment-major only: lo plane of the split output (see GemmParams)
    int out_packed = 0;        // out in the fragment-major layout
    float* rows = nullptr;     // attention rows [B][n_sel][n_rows_max][S] or null
    long rows_sB = 0, rows_sSel = 0;
    const int32_t* row_index = nullptr;  // per window: row slot to write (text tokens emitted so far)
    const int32_t* active = nullptr;
    int sel_slot[32];          // per head: slot in [n_sel] or -1
    long ld_kv = 0;            // row stride of K and V (elements)
    int n_split = 1;           // > 1: the S rows of a (window, head) are cut over n_split workgroups; the last one to finish merges
    float* split_ws = nullptr; // [B][H][n_split][66] partial (max, sum, 64 accumulators), needed when n_split > 1
    int32_t* split_ticket = nullptr;  // [B][H], zero before the launch and zero again after it, needed when n_split > 1
    long kv_sB = 0, kv_sH = 0; // window and head strides of K and V (elements)
    int B = 0, H = 0, d = 0, S = 0, n_rows_max = 0;
};
int uw_cross_attn_decode(int dtype, const CrossDecodeParams& p, hipStream_t s);

struct GreedyState {
    int32_t* tokens = nullptr;     // [B][ld_tok]  prompt + generated
    int ld_tok = 0;
    int32_t* pos = nullptr;        // [B] position of the token fed to the current step
    int32_t* n_new = nullptr;      // [B] generated tokens so far
    int32_t* n_text = nullptr;     // [B] text tokens among them (= attention row slot)
    int32_t* active = nullptr;     // [B]
    int32_t* dirty_from = nullptr; // [B] n_text when the first non-text token was generated, else -1
    int32_t* max_new = nullptr;    // [B]
    int32_t* n_active = nullptr;   // [1] windows still decoding
    int32_t* text_ids = nullptr;   // [B][ld_tok] generated text tokens
    int32_t* failed = nullptr;     // [B] 1: decoding stopped at a logits row that held a NaN (tokens before it are kept)
};
// end of a greedy step: argmax of logits[b][0..V) per window, append / stop on eot / budget, classify text tokens,
// advance pos, and write the chosen token's embedding as row b of the next step's residual stream x (x layout)
// cand_val / cand_idx [B][n_cand] (optional): per row the arg-max candidates of its column blocks (uw_logits_rows16); the
// logits are then not read
int uw_greedy_next(int dtype, const GreedyState& st, const float* logits, long ld, int B, int V, int eot, int sot, int ts_begin,
                   int ts_end, float* x, const float* tok_emb, const float* pos_emb, int d, hipStream_t s,
                   const float* cand_val = nullptr, const int32_t* cand_idx = nullptr, int n_cand = 0);

// ---- uw_mel.hip -------------------------------------------------------------------------------
struct MelTables;  // device tables owned by the context
int uw_mel_tables_create(int n_mels, const float* filters_host /*[n_mels][201]*/, MelTables** out);
void uw_mel_tables_destroy(MelTables* t);
// audio [B][stride] f32 with n_samples[b] valid samples -> mel [B][n_mels][3000]
int uw_mel_run(const MelTables* t, const float* audio, long stride, const int32_t* n_samples, int B, float* mel,
               float* frame_max_ws /*[B]*/, hipStream_t s);

// ---- uw_align.hip -----------------------------------------------------------------------------
// rows [B][n_sel][n_rows_max][S] f32 -> cost [B][n_rows_max][S] f64 (first n_tok[b] rows written)
int uw_cost_matrix_run(const float* rows, long sB, long sSel, int n_sel, const int32_t* n_tok, int B, int n_rows_max,
                       int S, double* cost, hipStream_t s);
// cost [B][n_rows_max][S] f64 -> boundaries [B][n_rows_max+1], optional path
int uw_dtw_run(const double* cost, const int32_t* n_tok, int B, int n_rows_max, int S, uint8_t* dir_ws,
               int32_t* boundaries, int32_t* path_i, int32_t* path_j, int32_t* path_len, hipStream_t s);
// the same with the direction matrix in a fixed device array (no workspace argument, no allocation)
int uw_dtw_static_run(const double* cost, const int32_t* n_tok, int B, int n_rows_max, int S, int32_t* boundaries,
                      int32_t* path_i, int32_t* path_j, int32_t* path_len, hipStream_t s);
