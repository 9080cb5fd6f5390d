/* C ABI of libuw_hip.so: the MI355X (gfx950) implementation of urro_whisper's transcription hot path.
 *
 * The reference (urroxyz/whisper, pure Python) reaches its numerics through four seams; each entry point
 * below replaces one of them and cites it.  A Python binding over ctypes is in whisper_amd/lib.py.
 *
 * Conventions: every function returns 0 on success, non-zero otherwise (uw_last_error() then holds a
 * message for the calling thread); nothing throws across the boundary.  Unless stated otherwise pointers are
 * DEVICE pointers owned by the caller, work is enqueued on `stream` (a hipStream_t passed as void*, NULL =
 * default stream) and is asynchronous.  A context belongs to one GPU and one caller thread at a time.
 */
#ifndef UW_HIP_H
#define UW_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct uw_ctx uw_ctx;

#define UW_DTYPE_F32 0   /* exact-fp32 MFMA path: parity runs                     */
#define UW_DTYPE_BF16 1  /* bf16 operands, fp32 accumulate and residual stream    */
#define UW_MAX_ALIGN_HEADS 32

typedef struct uw_config {
    int32_t d_model, n_heads, enc_layers, dec_layers, vocab, n_mels;
    int32_t n_audio_ctx;   /* 1500 */
    int32_t n_text_ctx;    /* 448  */
    int32_t dtype;         /* UW_DTYPE_*                                            */
    int32_t max_batch;     /* windows in flight; sizes the workspace                */
    int32_t n_align_heads; /* alignment heads, (layer, head) pairs below            */
    int32_t align_layer[UW_MAX_ALIGN_HEADS];
    int32_t align_head[UW_MAX_ALIGN_HEADS];
} uw_config;

const char* uw_last_error(void);
int uw_abi_version(void);

/* Model loader seam: urro_whisper/model/load.py:13-98 builds two onnxruntime sessions from the encoder
 * and decoder graphs.  Here the graphs' initialisers arrive as `n_weights` fp32 device tensors in the
 * order of whisper_amd.weights.tensor_plan (HF WhisperForConditionalGeneration naming); the context keeps
 * its own packed copies (compute dtype, fused QKV, q pre-scaled by 1/8, conv kernels tap-major) plus the
 * whole workspace, so the caller may free the inputs afterwards.  `mel_filters` is a HOST pointer to the
 * [n_mels][201] fp32 Slaney filterbank.  Synchronous. */
int uw_ctx_create(const uw_config* cfg, const void* const* weights, int n_weights, const float* mel_filters,
                  uw_ctx** out);
int uw_ctx_destroy(uw_ctx* ctx);
/* bytes of device memory the context holds */
int64_t uw_ctx_bytes(const uw_ctx* ctx);

/* Mel seam: calculate_mel_for_segment (urro_whisper/audio/load.py:67-121), i.e.
 * whisper.log_mel_spectrogram(audio, n_mels, padding=480000)[:, :3000].
 * audio: f32 [B][stride] 16 kHz PCM, n_samples: i32 [B] valid samples per window (<= stride),
 * mel_out: f32 [B][n_mels][3000]. */
int uw_mel(uw_ctx* ctx, const float* audio, int64_t stride, const int32_t* n_samples, int B, float* mel_out,
           void* stream);

/* Encoder seam: encoder_sess.run(None, {input_features: mel}) (urro_whisper/core.py:824-830).
 * mel: f32 [B][n_mels][3000] -> enc_out: f32 [B][1500][d_model] (may be NULL: result stays in the context). */
int uw_encode(uw_ctx* ctx, const float* mel, int B, float* enc_out, void* stream);

/* The reference re-projects the encoder states to cross-attention K/V inside every decoder run
 * (core.py:942-949); this does it once per window batch.  enc_states: f32 [B][1500][d_model], or NULL to use
 * the output of the last uw_encode. */
int uw_cross_kv(uw_ctx* ctx, const float* enc_states, int B, void* stream);

/* Decoder seam, cache-less form: decoder_sess.run(["logits"] | cross_attentions.*, {input_ids,
 * encoder_hidden_states}) (core.py:941-949, align.py:608-618).  Uses the K/V of the last uw_cross_kv.
 * ids: i32 [B][ld_ids], every window T tokens.  Optional outputs (NULL to skip):
 *   logits_all   f32 [B][T][vocab]
 *   logits_last  f32 [B][vocab]           (position T-1)
 *   cross_probs  f32 [dec_layers][B][n_heads][T][1500]   softmax probabilities. */
int uw_decoder_forward(uw_ctx* ctx, const int32_t* ids, int ld_ids, int B, int T, float* logits_all,
                       float* logits_last, float* cross_probs, void* stream);

/* Decoder seam, KV-cached form: one decoder_sess.run of the greedy loop (core.py:941-949) for the newest token
 * only.  The reference's decoder declares past_key_values.* inputs but always feeds them empty (core.py:185-192,
 * 943-946) and so re-runs the whole prefix; here the self-attention K/V of earlier positions stay in the context
 * (written by uw_decoder_forward or by earlier steps), so window b's positions 0..pos[b]-1 must have been
 * processed first.  ids, pos: i32 [B] device arrays; logits: f32 [B][vocab] for the token at pos[b].  A pos
 * outside 0..n_text_ctx-1 skips that window.  Invalidates the state of a uw_decode_greedy in progress. */
int uw_decode_step(uw_ctx* ctx, const int32_t* ids, const int32_t* pos, int B, float* logits, void* stream);

/* Greedy loop seam: core.py:938-968 plus the text filter of core.py:991, KV-cached.  prompt_ids: i32
 * [B][ld_prompt], prompt_len: i32 [B] HOST array, max_tokens: total sequence budget (prompt included).
 * Outputs (device): new_tokens i32 [B][n_text_ctx], n_new i32 [B], text_ids i32 [B][n_text_ctx], n_text i32 [B].
 * While decoding, the alignment heads' cross-attention rows are written into the context's rows buffer;
 * windows whose text continues after a generated non-text token get their rows from one teacher-forced
 * pass over prompt + text[:-1] (identical to the reference's per-token replay, align.py:603-630).
 * Synchronises the stream (the number of steps depends on the tokens). */
int uw_decode_greedy(uw_ctx* ctx, const int32_t* prompt_ids, int ld_prompt, const int32_t* prompt_len_host, int B,
                     int max_tokens, int eot, int sot, int ts_begin, int ts_end, int32_t* new_tokens, int32_t* n_new,
                     int32_t* text_ids, int32_t* n_text, void* stream);

/* Per-window outcome of the last uw_decode_greedy / uw_transcribe_windows on this context (both synchronise, so the
 * answer is complete when they return).  status: i32 [B] HOST array; 0 = decoded normally; 1 = the decoder produced a
 * logits row holding a NaN and the window stopped there, keeping the tokens generated before it -- the reference's
 * decoder-run failure inside the generation loop (core.py:948-956, 978-980: error stored, partial tokens kept). */
int uw_window_status(uw_ctx* ctx, int B, int32_t* status_host);

/* Alignment rows for a given text (forced alignment, align.py:446-471 / 603-630): one teacher-forced pass over
 * prompt + text[:-1]; rows land in the context's rows buffer.  All arrays device except the *_host lengths. */
int uw_attn_rows_forced(uw_ctx* ctx, const int32_t* prompt_ids, int ld_prompt, const int32_t* prompt_len_host,
                        const int32_t* text_ids, int ld_text, const int32_t* n_text_host, int B, void* stream);

/* Copies the context's rows buffer out: rows f32 [B][n_align_heads][n_rows][1500]. */
int uw_get_attn_rows(uw_ctx* ctx, int B, int n_rows, float* rows, void* stream);

/* perform_word_alignment's numeric front half (align.py:758-811,818): head mean, median filter 7 ('reflect'),
 * softmax, negate, f64.  rows f32 [B][n_sel][n_rows_max][S]; n_tok i32 [B] (device); cost f64 [B][n_rows_max][S]. */
int uw_cost_matrix(const float* rows, int n_sel, const int32_t* n_tok, int B, int n_rows_max, int S, double* cost,
                   void* stream);

/* dtw.dtw(cost, step_pattern=symmetric1) + boundary extraction (align.py:818-853).  cost f64 [B][n_rows_max][S];
 * n_tok i32 [B]; dir_ws: scratch of uw_dtw_workspace_bytes(); boundaries i32 [B][n_rows_max+1] (n_tok[b]+1 used);
 * optional path_i/path_j i32 [B][n_rows_max+S] and path_len i32 [B].
 * A window without a warping path (a NaN or infinite cost leaves cells without a predecessor; the reference returns
 * no timestamps when dtw raises, align.py:823-827) gets boundaries[b][0] = -1 and path_len[b] < 0; the call itself
 * still returns 0 because the other windows of the batch are valid. */
int64_t uw_dtw_workspace_bytes(int B, int n_rows_max, int S);
int uw_dtw(const double* cost, const int32_t* n_tok, int B, int n_rows_max, int S, void* dir_ws, int32_t* boundaries,
           int32_t* path_i, int32_t* path_j, int32_t* path_len, void* stream);

/* uw_dtw for callers that cannot allocate or hand over a workspace (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same inputs, outputs and results as
 * uw_dtw; the direction matrix lives in a fixed 16 MiB device array inside the library (one per GPU, held until the
 * process ends), so B * n_rows_max * ceil(S / 16) * 4 bytes must not exceed 16 MiB (e.g. 95 windows x 448 rows x 1500
 * frames) -- larger problems are rejected with an error, use uw_dtw.  The call allocates nothing.  All calls on one GPU
 * share the array: calls that overlap on one GPU (different streams or threads) race.  The array of the CURRENT device is
 * used, so the device that owns `cost`, the outputs and `stream` must be current when the call is made (a cost matrix on
 * another device is rejected). */
#define UW_DTW_STATIC_BYTES (16 << 20)
int uw_dtw_static(const double* cost, const int32_t* n_tok, int B, int n_rows_max, int S, int32_t* boundaries,
                  int32_t* path_i, int32_t* path_j, int32_t* path_len, void* stream);

/* cost matrix + DTW on the context's rows buffer: boundaries i32 [B][n_text_ctx+1]. n_text i32 [B] device.
 * boundaries[b][0] = -1 marks a window for which no path exists (see uw_dtw). */
int uw_align(uw_ctx* ctx, const int32_t* n_text, int B, int32_t* boundaries, void* stream);

/* The whole path for a batch of windows: mel -> encoder -> cross K/V -> greedy decode with row capture ->
 * cost matrix -> DTW (core.py:798-1030 numerics).  Same outputs as uw_decode_greedy plus boundaries.
 * A context created without alignment heads (the reference's "no alignment" case, align.py:81-84,595-597) skips the
 * cost matrix and DTW and fills boundaries with -1; boundaries[b][0] = -1 always means "window b has no timestamps". */
int uw_transcribe_windows(uw_ctx* ctx, const float* audio, int64_t stride, const int32_t* n_samples,
                          const int32_t* prompt_ids, int ld_prompt, const int32_t* prompt_len_host, int B, int max_tokens,
                          int eot, int sot, int ts_begin, int ts_end, int32_t* new_tokens, int32_t* n_new,
                          int32_t* text_ids, int32_t* n_text, int32_t* boundaries, void* stream);

/* Measurement aid (no reference counterpart): brackets every launch of the cross-attention decode kernel --
 * the kernel that streams the cross K/V cache and bounds the path -- with HIP events on the caller's stream.
 * uw_profile_end synchronises the device and returns how many launches ran, how many were timed (the event
 * pool holds max_launches pairs), how many windows those launches covered in total (a launch streams the K/V
 * of one decoder layer for its windows) and the summed duration of the timed ones. */
int uw_profile_begin(uw_ctx* ctx, int max_launches);
int uw_profile_end(uw_ctx* ctx, int64_t* launches_total, int64_t* launches_timed, int64_t* windows_total,
                   double* timed_ms);
/* Between the same begin / end, four more spans are bracketed with events, sparsely: kind 1 = launches of the
 * self-attention decode kernel (one in eight timed), 2 = the encoder pass of uw_transcribe_windows, 3 = the DTW launch
 * of uw_align, 4 = whole decode steps of the greedy loop (all launches of one step; one step in eight timed).  Read after
 * uw_profile_end: how many ran, how many were timed, the summed duration of the timed ones. */
int uw_profile_span(uw_ctx* ctx, int kind, int64_t* launches_total, int64_t* launches_timed, double* timed_ms);

#ifdef __cplusplus
}
#endif
#endif
