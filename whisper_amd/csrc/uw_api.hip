// Context, weight packing and the launch sequences behind the C ABI (include/uw_hip.h).
#include "../../include/uw_hip.h"
#include "uw_common.h"
#include "uw_kernels.h"
#include <stdarg.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

static thread_local char g_err[512] = "";
void uw_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
size_t uw_dtw_dir_words(int S);

namespace {

struct AttnW {            // one attention block
    void* wqkv = nullptr; // self: [3d][d] (q*1/8 | k | v); cross: unused
    float* bqkv = nullptr;
    void* wq = nullptr;   // cross / encoder: [d][d] q*1/8 ; encoder: [2d][d] (q | k)
    float* bq = nullptr;
    void* wkv = nullptr;  // cross: [2d][d] (k | v)
    float* bkv = nullptr;
    void* wv = nullptr;   // [d][d] for the transposed-V product
    float* bv = nullptr;
    void* wo = nullptr;
    float* bo = nullptr;
    float *ln_g = nullptr, *ln_b = nullptr;
    void *wqkv_p = nullptr, *wq_p = nullptr, *wo_p = nullptr;   // fragment-major copies for the decode step
    float *bln = nullptr, *wsum = nullptr;                        // LayerNorm folded into wqkv_p / wq_p: bias and row sums
    const float *src_wq = nullptr, *src_wk = nullptr, *src_wv = nullptr, *src_bq = nullptr, *src_bv = nullptr;   // caller's f32 tensors (valid during load)
};
struct LayerW {
    AttnW self, cross;
    void *w1 = nullptr, *w2 = nullptr;
    void *w1_p = nullptr, *w2_p = nullptr;
    float *b1ln = nullptr, *w1sum = nullptr;
    const float *src_w1 = nullptr, *src_b1 = nullptr;
    float *b1 = nullptr, *b2 = nullptr, *ln2_g = nullptr, *ln2_b = nullptr;
};

}  // namespace

struct uw_ctx {
    uw_config cfg;
    int device = 0;
    int dt = UW_BF16;
    size_t es = 2;             // element size of the compute dtype
    int d = 0, H = 0, S = 0, Spad = 0, Tm = 0, V = 0, Bm = 0, k1pad = 0;
    long melT_stride = 0;      // elements per window of the time-major mel buffer
    int64_t bytes = 0;
    std::vector<void*> allocs;
    MelTables* mel_tables = nullptr;
    // encoder weights
    void *conv1_w = nullptr, *conv2_w = nullptr;
    float *conv1_b = nullptr, *conv2_b = nullptr, *enc_pos = nullptr, *enc_ln_g = nullptr, *enc_ln_b = nullptr;
    std::vector<LayerW> enc, dec;
    // decoder weights
    float *tok_emb32 = nullptr, *dec_pos = nullptr, *dec_ln_g = nullptr, *dec_ln_b = nullptr;
    void* tok_emb = nullptr;   // [V][d] compute dtype (tied output projection)
    void* tok_emb_p = nullptr; // bf16 contexts of up to 16 windows: the same, fragment-major (the one-launch logits form streams it)
    // alignment heads
    int n_sel = 0;
    // workspace: encoder
    float* mel_max = nullptr;
    void *melT = nullptr, *h1p = nullptr, *e_ln = nullptr, *e_qk = nullptr, *e_vt = nullptr, *e_attn = nullptr, *e_mlp = nullptr,
         *enc_T = nullptr;
    float *e_x = nullptr, *enc_out32 = nullptr;
    // workspace: decoder
    void *cross_kv = nullptr, *cross_vt = nullptr, *self_kv = nullptr;
    void *d_ln = nullptr, *d_qkv = nullptr, *d_vt = nullptr, *d_attn = nullptr, *d_mlp = nullptr;
    float *d_x = nullptr, *d_last = nullptr, *logits = nullptr, *rows = nullptr;
    long logits_ld = 0;
    double* cost = nullptr;
    uint8_t* dtw_ws = nullptr;
    int32_t *st_tokens = nullptr, *st_pos = nullptr, *st_n_new = nullptr, *st_n_text = nullptr, *st_active = nullptr,
            *st_dirty = nullptr, *st_max_new = nullptr, *st_n_active = nullptr, *st_text = nullptr, *st_next = nullptr,
            *st_qfirst = nullptr, *st_qcount = nullptr, *st_zero = nullptr, *st_ids = nullptr, *st_failed = nullptr;
    std::vector<int32_t> window_status;   // host copy of st_failed after the last greedy decode
    int32_t* host_pin = nullptr;   // pinned scratch for small read-backs
    int cross_batch = 0;           // windows covered by the current cross K/V
    int step_rows = 0;             // upper bound of (position + 1) of the next decode step over its windows (0: unknown)
    bool enc_valid = false;
    long enc_rows = 0;         // rows (windows x frames) of the K-blocked encoder output held in enc_T
    // optional timing of the cross-attention decode kernel (the dominant HBM stream)
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;   // start/stop pairs
    size_t prof_used = 0;
    long prof_launches = 0;
    long prof_windows = 0;             // windows summed over the launches
    // secondary spans, sampled so that their events do not crowd the stream: 1 self-attention decode launches (one in
    // eight), 2 the encoder pass, 3 the DTW launch, 4 whole decode steps of the greedy loop (one in eight)
    struct Span { std::vector<hipEvent_t> ev; size_t used = 0; long launches = 0; };
    Span span[5];
    // side streams: disjoint window ranges decode concurrently so one range's latency-bound small kernels
    // overlap another range's bandwidth-bound cross-attention stream
    hipStream_t sub_stream[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t sub_event[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t fork_event = nullptr;
    int n_streams = 4;
    int split_mode = -1;           // cross-attention decode: -1 choose by grid size, 0 never split, n > 1 always n parts
    float* split_ws = nullptr;
    int32_t* split_ticket = nullptr;   // [Bm][H] arrival counters of the row-split cross-attention (zero between launches)
    int step_prefetch = -1;        // decode-step products request the following product's weights: -1 by row count, 0 never, 1 always
    int fused_logits = 1;          // up to 16 windows (bf16): LayerNorm + tied projection + arg-max candidates in one launch
    float* cand_val = nullptr;     // [Bm rounded up to 16][uw_logits_rows16_blocks(V)] arg-max candidates of the step's logits
    int32_t* cand_idx = nullptr;
};

namespace {

int dev_alloc(uw_ctx* c, void** p, size_t bytes, bool zero)
{
    bytes = (bytes + 255) & ~(size_t)255;
    UW_HIP_CHECK(hipMalloc(p, bytes));
    c->allocs.push_back(*p);
    c->bytes += (int64_t)bytes;
    if (zero) UW_HIP_CHECK(hipMemset(*p, 0, bytes));
    return UW_OK;
}
#define ALLOC(ptr, bytes, zero)                                         \
    do {                                                                \
        int rc__ = dev_alloc(c, (void**)&(ptr), (size_t)(bytes), zero); \
        if (rc__) return rc__;                                          \
    } while (0)
#define RUN(expr)                 \
    do {                          \
        int rc__ = (expr);        \
        if (rc__) return rc__;    \
    } while (0)

struct Cursor {
    const void* const* w;
    int n, i = 0;
    const float* next() { return (i < n) ? (const float*)w[i++] : (i++, nullptr); }
};

// dst (compute dtype) rows [row0, row0+rows) of a [*, cols] matrix <- src f32 [rows][cols] * scale
int pack_rows(uw_ctx* c, void* dst, long row0, const float* src, long rows, long cols, float scale)
{
    UW_REQUIRE(src, "weight tensor missing");
    return uw_convert(c->dt, src, (char*)dst + (size_t)row0 * cols * c->es, rows * cols, scale, 0);
}
int copy_f32(float* dst, const float* src, long n, float scale)
{
    UW_REQUIRE(src, "weight tensor missing");
    return uw_convert(UW_F32, src, dst, n, scale, 0);
}

int load_attn(uw_ctx* c, Cursor& cur, AttnW& a, int kind /*0 enc self, 1 dec self, 2 dec cross*/)
{
    const long d = c->d;
    const float *wq = cur.next(), *bq = cur.next(), *wk = cur.next(), *wv = cur.next(), *bv = cur.next(), *wo = cur.next(),
                *bo = cur.next(), *lg = cur.next(), *lb = cur.next();
    UW_REQUIRE(wq && bq && wk && wv && bv && wo && bo && lg && lb, "attention weights missing");
    a.src_wq = wq; a.src_wk = wk; a.src_wv = wv; a.src_bq = bq; a.src_bv = bv;
    if (kind == 1) {
        ALLOC(a.wqkv, 3 * d * d * c->es, false);
        ALLOC(a.bqkv, 3 * d * 4, true);
        RUN(pack_rows(c, a.wqkv, 0, wq, d, d, 0.125f));
        RUN(pack_rows(c, a.wqkv, d, wk, d, d, 1.f));
        RUN(pack_rows(c, a.wqkv, 2 * d, wv, d, d, 1.f));
        RUN(copy_f32(a.bqkv, bq, d, 0.125f));
        RUN(copy_f32(a.bqkv + 2 * d, bv, d, 1.f));
        // k|v view for the batched cache-filling product
        a.wkv = (char*)a.wqkv + (size_t)d * d * c->es;
        a.bkv = a.bqkv + d;
        a.wq = a.wqkv;
        a.bq = a.bqkv;
    } else if (kind == 0) {
        ALLOC(a.wq, 2 * d * d * c->es, false);   // q | k
        ALLOC(a.bq, 2 * d * 4, true);
        RUN(pack_rows(c, a.wq, 0, wq, d, d, 0.125f));
        RUN(pack_rows(c, a.wq, d, wk, d, d, 1.f));
        RUN(copy_f32(a.bq, bq, d, 0.125f));
    } else {
        ALLOC(a.wq, d * d * c->es, false);
        ALLOC(a.bq, d * 4, false);
        RUN(pack_rows(c, a.wq, 0, wq, d, d, 0.125f));
        RUN(copy_f32(a.bq, bq, d, 0.125f));
        ALLOC(a.wkv, 2 * d * d * c->es, false);
        ALLOC(a.bkv, 2 * d * 4, true);
        RUN(pack_rows(c, a.wkv, 0, wk, d, d, 1.f));
        RUN(pack_rows(c, a.wkv, d, wv, d, d, 1.f));
        RUN(copy_f32(a.bkv + d, bv, d, 1.f));
    }
    ALLOC(a.wv, d * d * c->es, false);
    ALLOC(a.bv, d * 4, false);
    RUN(pack_rows(c, a.wv, 0, wv, d, d, 1.f));
    RUN(copy_f32(a.bv, bv, d, 1.f));
    ALLOC(a.wo, d * d * c->es, false);
    ALLOC(a.bo, d * 4, false);
    RUN(pack_rows(c, a.wo, 0, wo, d, d, 1.f));
    RUN(copy_f32(a.bo, bo, d, 1.f));
    ALLOC(a.ln_g, d * 4, false);
    ALLOC(a.ln_b, d * 4, false);
    RUN(copy_f32(a.ln_g, lg, d, 1.f));
    RUN(copy_f32(a.ln_b, lb, d, 1.f));
    return UW_OK;
}

int load_mlp(uw_ctx* c, Cursor& cur, LayerW& l)
{
    const long d = c->d, f = 4 * d;
    const float *w1 = cur.next(), *b1 = cur.next(), *w2 = cur.next(), *b2 = cur.next(), *lg = cur.next(), *lb = cur.next();
    UW_REQUIRE(w1 && b1 && w2 && b2 && lg && lb, "mlp weights missing");
    l.src_w1 = w1; l.src_b1 = b1;
    ALLOC(l.w1, f * d * c->es, false);
    ALLOC(l.b1, f * 4, false);
    ALLOC(l.w2, d * f * c->es, false);
    ALLOC(l.b2, d * 4, false);
    ALLOC(l.ln2_g, d * 4, false);
    ALLOC(l.ln2_b, d * 4, false);
    RUN(pack_rows(c, l.w1, 0, w1, f, d, 1.f));
    RUN(copy_f32(l.b1, b1, f, 1.f));
    RUN(pack_rows(c, l.w2, 0, w2, d, f, 1.f));
    RUN(copy_f32(l.b2, b2, d, 1.f));
    RUN(copy_f32(l.ln2_g, lg, d, 1.f));
    RUN(copy_f32(l.ln2_b, lb, d, 1.f));
    return UW_OK;
}

int load_weights(uw_ctx* c, const void* const* weights, int n_weights)
{
    const uw_config& g = c->cfg;
    const long d = c->d;
    const int expect = 5 + g.enc_layers * 15 + 2 + 2 + g.dec_layers * 24 + 2;
    UW_REQUIRE(n_weights == expect, "expected %d weight tensors for this shape, got %d", expect, n_weights);
    Cursor cur{weights, n_weights};
    const float *c1w = cur.next(), *c1b = cur.next(), *c2w = cur.next(), *c2b = cur.next(), *epos = cur.next();
    UW_REQUIRE(c1w && c1b && c2w && c2b && epos, "conv stem weights missing");
    ALLOC(c->conv1_w, d * c->k1pad * c->es, false);
    ALLOC(c->conv2_w, d * 3 * d * c->es, false);
    ALLOC(c->conv1_b, d * 4, false);
    ALLOC(c->conv2_b, d * 4, false);
    ALLOC(c->enc_pos, (long)c->S * d * 4, false);
    RUN(uw_pack_conv_weight(c->dt, c1w, c->conv1_w, (int)d, g.n_mels, c->k1pad, 0));
    RUN(uw_pack_conv_weight(c->dt, c2w, c->conv2_w, (int)d, (int)d, (int)(3 * d), 0));
    RUN(copy_f32(c->conv1_b, c1b, d, 1.f));
    RUN(copy_f32(c->conv2_b, c2b, d, 1.f));
    RUN(copy_f32(c->enc_pos, epos, (long)c->S * d, 1.f));
    c->enc.resize(g.enc_layers);
    for (auto& l : c->enc) {
        RUN(load_attn(c, cur, l.self, 0));
        RUN(load_mlp(c, cur, l));
    }
    const float *elg = cur.next(), *elb = cur.next();
    ALLOC(c->enc_ln_g, d * 4, false);
    ALLOC(c->enc_ln_b, d * 4, false);
    RUN(copy_f32(c->enc_ln_g, elg, d, 1.f));
    RUN(copy_f32(c->enc_ln_b, elb, d, 1.f));
    const float *temb = cur.next(), *dpos = cur.next();
    UW_REQUIRE(temb && dpos, "decoder embeddings missing");
    ALLOC(c->tok_emb32, (long)c->V * d * 4, false);
    ALLOC(c->dec_pos, (long)c->Tm * d * 4, false);
    RUN(copy_f32(c->tok_emb32, temb, (long)c->V * d, 1.f));
    RUN(copy_f32(c->dec_pos, dpos, (long)c->Tm * d, 1.f));
    if (c->dt == UW_F32) {
        c->tok_emb = c->tok_emb32;
    } else {
        ALLOC(c->tok_emb, (long)c->V * d * c->es, false);
        RUN(pack_rows(c, c->tok_emb, 0, temb, c->V, d, 1.f));
        if (c->Bm <= 16 && d % 32 == 0 && d <= 1280) {
            ALLOC(c->tok_emb_p, (long)((c->V + 15) & ~15) * d * c->es, false);
            RUN(uw_pack_frag(c->dt, c->tok_emb, d, c->tok_emb_p, c->V, (int)d, 0));
        }
    }
    c->dec.resize(g.dec_layers);
    for (auto& l : c->dec) {
        RUN(load_attn(c, cur, l.self, 1));
        RUN(load_attn(c, cur, l.cross, 2));
        RUN(load_mlp(c, cur, l));
    }
    // the KV-cached step reads its weights as whole MFMA operand blocks: second, fragment-major copy.  The three
    // products that follow a LayerNorm (q|k|v, cross q, fc1) run with the LayerNorm fused in: its shift goes into
    // the bias, its scale is applied to the operand in the kernel, and the kernel needs gamma-weighted row sums.
    {
        const long f = 4 * d;
        void* tmp = nullptr;   // row-major staging for the folded weights; owned by the context like every other buffer
        ALLOC(tmp, (size_t)f * d * c->es, false);
        for (auto& l : c->dec) {
            ALLOC(l.self.wqkv_p, 3 * d * d * c->es, false);
            ALLOC(l.self.wo_p, d * d * c->es, false);
            ALLOC(l.cross.wq_p, d * d * c->es, false);
            ALLOC(l.cross.wo_p, d * d * c->es, false);
            ALLOC(l.w1_p, f * d * c->es, false);
            ALLOC(l.w2_p, d * f * c->es, false);
            ALLOC(l.self.bln, 3 * d * 4, false);
            ALLOC(l.self.wsum, 3 * d * 4, false);
            ALLOC(l.cross.bln, d * 4, false);
            ALLOC(l.cross.wsum, d * 4, false);
            ALLOC(l.b1ln, f * 4, false);
            ALLOC(l.w1sum, f * 4, false);
            char* t = (char*)tmp;
            const size_t dd = (size_t)d * d * c->es;
            RUN(uw_fold_layernorm(c->dt, l.self.src_wq, l.self.src_bq, l.self.ln_g, l.self.ln_b, 0.125f, t, l.self.bln, l.self.wsum, (int)d, (int)d, 0));
            RUN(uw_fold_layernorm(c->dt, l.self.src_wk, nullptr, l.self.ln_g, l.self.ln_b, 1.f, t + dd, l.self.bln + d, l.self.wsum + d, (int)d, (int)d, 0));
            RUN(uw_fold_layernorm(c->dt, l.self.src_wv, l.self.src_bv, l.self.ln_g, l.self.ln_b, 1.f, t + 2 * dd, l.self.bln + 2 * d, l.self.wsum + 2 * d, (int)d, (int)d, 0));
            RUN(uw_pack_frag(c->dt, t, d, l.self.wqkv_p, (int)(3 * d), (int)d, 0));
            RUN(uw_fold_layernorm(c->dt, l.cross.src_wq, l.cross.src_bq, l.cross.ln_g, l.cross.ln_b, 0.125f, t, l.cross.bln, l.cross.wsum, (int)d, (int)d, 0));
            RUN(uw_pack_frag(c->dt, t, d, l.cross.wq_p, (int)d, (int)d, 0));
            RUN(uw_fold_layernorm(c->dt, l.src_w1, l.src_b1, l.ln2_g, l.ln2_b, 1.f, t, l.b1ln, l.w1sum, (int)f, (int)d, 0));
            RUN(uw_pack_frag(c->dt, t, d, l.w1_p, (int)f, (int)d, 0));
            RUN(uw_pack_frag(c->dt, l.self.wo, d, l.self.wo_p, (int)d, (int)d, 0));
            RUN(uw_pack_frag(c->dt, l.cross.wo, d, l.cross.wo_p, (int)d, (int)d, 0));
            RUN(uw_pack_frag(c->dt, l.w2, f, l.w2_p, (int)d, (int)f, 0));
        }
    }
    // The encoder's and the cross K/V projection's tile GEMMs read K-blocked operands ([K / KT][rows][KT], KT = 64
    // bytes of K: see GemmParams): their weights are re-laid once, here; the activations are produced in that form.
    {
        const long f = 4 * d;
        void* tmp = nullptr;
        ALLOC(tmp, (size_t)f * d * c->es, false);
        auto kblock = [&](void* w, long R, long K) -> int {
            RUN(uw_pack_kblock(c->dt, w, tmp, R, (int)K, 0));
            UW_HIP_CHECK(hipMemcpyAsync(w, tmp, (size_t)R * K * c->es, hipMemcpyDeviceToDevice, 0));
            return UW_OK;
        };
        for (auto& l : c->enc) {
            RUN(kblock(l.self.wq, 2 * d, d));
            RUN(kblock(l.self.wv, d, d));
            RUN(kblock(l.self.wo, d, d));
            RUN(kblock(l.w1, f, d));
            RUN(kblock(l.w2, d, f));
        }
        for (auto& l : c->dec) RUN(kblock(l.cross.wkv, 2 * d, d));
    }
    const float *dlg = cur.next(), *dlb = cur.next();
    ALLOC(c->dec_ln_g, d * 4, false);
    ALLOC(c->dec_ln_b, d * 4, false);
    RUN(copy_f32(c->dec_ln_g, dlg, d, 1.f));
    RUN(copy_f32(c->dec_ln_b, dlb, d, 1.f));
    UW_HIP_CHECK(hipDeviceSynchronize());
    return UW_OK;
}

int create_streams(uw_ctx* c)
{
    const char* env = getenv("UW_DECODE_STREAMS");
    const char* sp = getenv("UW_DECODE_SPLIT");   // development / test knob: force the row-split form on or off
    c->split_mode = sp ? atoi(sp) : -1;
    if (c->split_mode > 16) c->split_mode = 16;
    const char* pfw = getenv("UW_STEP_PREFETCH");   // development / test knob
    c->step_prefetch = pfw ? atoi(pfw) : -1;
    const char* fl = getenv("UW_LOGITS_FUSED");     // development / test knob
    c->fused_logits = fl ? atoi(fl) : 1;
    c->n_streams = env ? atoi(env) : 1;   // concurrent queues did not overlap on MI355X (measured): one stream by default
    if (c->n_streams < 1) c->n_streams = 1;
    if (c->n_streams > 4) c->n_streams = 4;
    for (int k = 0; k < 4; ++k) {
        UW_HIP_CHECK(hipStreamCreateWithFlags(&c->sub_stream[k], hipStreamNonBlocking));
        UW_HIP_CHECK(hipEventCreateWithFlags(&c->sub_event[k], hipEventDisableTiming));
    }
    UW_HIP_CHECK(hipEventCreateWithFlags(&c->fork_event, hipEventDisableTiming));
    return UW_OK;
}

int alloc_workspace(uw_ctx* c)
{
    const long d = c->d, S = c->S, Bm = c->Bm, Tm = c->Tm, es = (long)c->es, Spad = c->Spad;
    const uw_config& g = c->cfg;
    ALLOC(c->mel_max, Bm * 4, true);
    c->melT_stride = (long)(3000 + 2 + 4) * g.n_mels;   // 2 halo rows + slack for the padded conv1 K
    ALLOC(c->melT, Bm * c->melT_stride * es, true);
    ALLOC(c->h1p, Bm * 3002 * d * es, true);
    ALLOC(c->e_x, Bm * S * d * 4, false);
    ALLOC(c->e_ln, Bm * S * d * es, false);
    ALLOC(c->e_qk, Bm * S * 2 * d * es, false);
    ALLOC(c->e_vt, Bm * d * Spad * es, true);
    ALLOC(c->e_attn, Bm * S * d * es, false);
    ALLOC(c->e_mlp, Bm * S * 4 * d * es, false);
    ALLOC(c->enc_T, Bm * S * d * es, false);
    ALLOC(c->enc_out32, Bm * S * d * 4, false);
    const long Ld = g.dec_layers;
    ALLOC(c->cross_kv, Ld * Bm * S * 2 * d * es, false);
    ALLOC(c->cross_vt, Ld * Bm * d * Spad * es, true);
    ALLOC(c->self_kv, Ld * Bm * Tm * 2 * d * es, true);
    ALLOC(c->d_x, Bm * Tm * d * 4, true);
    // LayerNorm output, attention output and MLP hidden state are carried as [hi | lo] bf16 pairs where logits are
    // produced (see GemmParams), hence twice the width
    ALLOC(c->d_ln, Bm * Tm * 2 * d * es, true);
    ALLOC(c->d_qkv, Bm * Tm * 3 * d * es, true);
    ALLOC(c->d_vt, Bm * d * Tm * es, true);
    ALLOC(c->d_attn, Bm * Tm * 2 * d * es, true);
    ALLOC(c->d_mlp, Bm * Tm * 8 * d * es, true);
    ALLOC(c->d_last, ((Bm + 15) & ~15l) * d * 4, true);
    ALLOC(c->split_ws, Bm * (d / 64) * 16 * 66 * 4, true);
    ALLOC(c->split_ticket, Bm * (d / 64) * 4, true);
    ALLOC(c->cand_val, ((Bm + 15) & ~15l) * uw_logits_rows16_blocks(c->V) * 4, true);
    ALLOC(c->cand_idx, ((Bm + 15) & ~15l) * uw_logits_rows16_blocks(c->V) * 4, true);
    c->logits_ld = (c->V + 3) & ~3;
    ALLOC(c->logits, Bm * c->logits_ld * 4, true);
    if (c->n_sel > 0) {
        ALLOC(c->rows, Bm * (long)c->n_sel * Tm * S * 4, false);
        ALLOC(c->cost, Bm * Tm * S * 8, false);
        ALLOC(c->dtw_ws, (size_t)uw_dtw_workspace_bytes((int)Bm, (int)Tm, (int)S), false);
    }
    ALLOC(c->st_tokens, Bm * Tm * 4, true);
    ALLOC(c->st_text, Bm * Tm * 4, true);
    ALLOC(c->st_ids, Bm * Tm * 4, true);
    int32_t** small[] = {&c->st_pos, &c->st_n_new, &c->st_n_text, &c->st_active, &c->st_dirty, &c->st_max_new,
                         &c->st_n_active, &c->st_next, &c->st_qfirst, &c->st_qcount, &c->st_zero, &c->st_failed};
    for (auto pp : small) ALLOC(*pp, (Bm + 8) * 4, true);
    UW_HIP_CHECK(hipHostMalloc((void**)&c->host_pin, (size_t)(8 * Bm + 64) * 4, hipHostMallocDefault));
    return UW_OK;
}

// ---- launch sequences -------------------------------------------------------------------------------------
int gemm(uw_ctx* c, const void* A, long lda, const void* Bw, long ldb, void* C, long ldc, bool out_f32, int M, int N,
         int K, const float* bias, int act, const float* residual, long ldr, hipStream_t s, int batch = 1, long sA = 0,
         long sB = 0, long sC = 0, long sR = 0, int bias_mode = 1, long c_group = 0, int kwrap_a = 0, int kwrap_b = 0,
         int c_split = 0)
{
    GemmParams p;
    p.c_group = c_group;
    p.kwrap_a = kwrap_a; p.kwrap_b = kwrap_b; p.c_split = c_split;
    p.A = A; p.B = Bw; p.C = C; p.bias = bias; p.residual = residual;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr;
    p.sA = sA; p.sB = sB; p.sC = sC; p.sR = sR;
    p.M = M; p.N = N; p.K = K; p.batch = batch;
    p.bias_mode = bias ? bias_mode : 0;
    p.act = act;
    // decode-step shapes: few rows, modest N -> stream the weights once over many small workgroups
    if (batch == 1 && M <= 128 && N <= 8192 && !c_group && !kwrap_a && !kwrap_b && !c_split) return uw_gemm_skinny(c->dt, out_f32 ? 1 : 0, p, s);
    return uw_gemm(c->dt, out_f32 ? 1 : 0, p, s);
}

// The product that follows a decode-step product in the chain: its weights are requested while this one runs
// (GemmParams::pf).  ln: LayerNorm-fused form; otherwise the split-activation form.
struct NextProduct {
    const void* w = nullptr;
    int n = 0, k = 0, ln = 0;
};
void set_prefetch(uw_ctx* c, GemmParams& p, const NextProduct* nx)
{
    // Measured (decode step / pass with one batch in flight / with three): medium 15 windows -9 % / -8 % / -2 %, small
    // 16 windows -9 % / -7 % / -5 %, small 32 -6 % / -5 % / -1 %; from 48 windows on the step gains under 2 % and the
    // extra workgroups and fetches cost other batches in flight about 1 %; at 128 windows the step itself loses 1.6 %.
    // Models whose decoder weights fit the Infinity Cache (tiny, base) find them warm anyway.
    const bool on = c->step_prefetch < 0 ? p.M <= 32 : c->step_prefetch > 0;
    if (!nx || !nx->w || !on) return;
    p.pf = nx->w; p.pf_n = nx->n; p.pf_k = nx->k;
    p.pf_cols = uw_gemm_skinny_block_columns(c->dt, p.M, nx->n, nx->ln, nx->ln ? 0 : 1);
}

// decode-step product: A [M][K] and W [N][K] fragment-major, C row-major (or fragment-major when c_packed)
int gemm_step(uw_ctx* c, const void* A, const void* A_lo, const void* Wp, void* C, long ldc, bool out_f32, int M, int N, int K,
              const float* bias, int act, const float* residual, long ldr, bool c_packed, hipStream_t s, const NextProduct* nx = nullptr)
{
    GemmParams p;
    p.A = A; p.A_lo = A_lo; p.B = Wp; p.C = C; p.bias = bias; p.residual = residual;
    p.ldc = ldc; p.ldr = ldr; p.M = M; p.N = N; p.K = K;
    p.bias_mode = bias ? 1 : 0;
    p.act = act;
    p.ab_packed = 1; p.c_packed = c_packed ? 1 : 0;
    p.c_xlayout = residual ? 1 : 0;   // the step's residual stream lives in the x layout
    set_prefetch(c, p, nx);
    return uw_gemm_skinny(c->dt, out_f32 ? 1 : 0, p, s);
}

// decode-step product with the preceding LayerNorm folded in: A is the residual stream x (x layout)
int gemm_step_ln(uw_ctx* c, const float* x, const void* Wp, const float* gamma, const float* bias_ln, const float* wsum, void* C, long ldc,
                 int M, int N, int K, int act, bool c_packed, hipStream_t s, void* C_lo = nullptr, bool out_f32 = false,
                 const NextProduct* nx = nullptr)
{
    GemmParams p;
    p.A = x; p.B = Wp; p.C = C; p.C_lo = C_lo; p.bias = bias_ln; p.wsum = wsum; p.gamma = gamma;
    p.ldc = ldc; p.M = M; p.N = N; p.K = K;
    p.bias_mode = 1; p.act = act;
    p.ab_packed = 1; p.c_packed = c_packed ? 1 : 0; p.ln_fused = 1;
    set_prefetch(c, p, nx);
    return uw_gemm_skinny(c->dt, out_f32 ? 1 : 0, p, s);
}

// logits[m][:] = LayerNorm(x[m]) . tok_emb^T for M rows of the residual stream held in the step's x layout.  For bf16
// compute the normalised row is written as a [hi | lo] pair and multiplied by uw_gemm_rows_hilo, so the logits see no
// bf16-rounded activations; `ln` is scratch for [M][2 d].
int logits_from_x(uw_ctx* c, const float* x, int M, void* ln, float* logits, hipStream_t s)
{
    const int d = c->d, split = c->dt == UW_BF16 ? 1 : 0;
    RUN(uw_layernorm(c->dt, x, ln, nullptr, c->dec_ln_g, c->dec_ln_b, M, d, s, 0, 1, split));
    if (split) {
        for (int off = 0; off < M; off += 128) {
            GemmParams p;
            p.A = (const char*)ln + (size_t)off * 2 * d * c->es; p.B = c->tok_emb; p.C = logits + (long)off * c->logits_ld;
            p.lda = 2 * d; p.ldb = d; p.ldc = c->logits_ld; p.M = std::min(128, M - off); p.N = c->V; p.K = d;
            RUN(uw_gemm_rows_hilo(p, s));
        }
        return UW_OK;
    }
    return gemm(c, ln, d, c->tok_emb, d, logits, c->logits_ld, true, M, c->V, d, nullptr, 0, nullptr, 0, s);
}

// brackets for the secondary spans: begin returns the event slot (or -1), end closes it
int span_begin(uw_ctx* c, int kind, int every, hipStream_t s)
{
    if (!c->prof_on) return -1;
    uw_ctx::Span& sp = c->span[kind];
    const long n = sp.launches++;
    if (n % every != 0 || sp.used + 2 > sp.ev.size()) return -1;
    if (hipEventRecord(sp.ev[sp.used], s) != hipSuccess) return -1;
    return (int)sp.used;
}
void span_end(uw_ctx* c, int kind, int slot, hipStream_t s)
{
    if (slot < 0) return;
    uw_ctx::Span& sp = c->span[kind];
    if (hipEventRecord(sp.ev[slot + 1], s) == hipSuccess) sp.used = (size_t)slot + 2;
}

int encoder_forward(uw_ctx* c, const float* mel, int B, hipStream_t s)
{
    const int d = c->d, S = c->S, H = c->H, nm = c->cfg.n_mels;
    const long M = (long)B * S;
    RUN(uw_mel_to_time_major(c->dt, mel, c->melT, B, nm, 3000, c->melT_stride, s));
    // conv1 (k3, pad 1) + GELU: row t of the product reads time-major rows t .. t+2 of the padded mel buffer
    RUN(gemm(c, c->melT, nm, c->conv1_w, c->k1pad, (char*)c->h1p + (size_t)d * c->es, d, false, 3000, d, c->k1pad, c->conv1_b, 1,
             nullptr, 0, s, B, c->melT_stride, 0, (long)3002 * d));
    // conv2 (k3, stride 2, pad 1) + GELU + positions: row t reads padded rows 2t .. 2t+2
    RUN(gemm(c, c->h1p, 2 * d, c->conv2_w, 3 * d, c->e_x, d, true, S, d, 3 * d, c->conv2_b, 1, c->enc_pos, d, s, B, (long)3002 * d, 0,
             (long)S * d, 0));
    // Operands of the layer GEMMs are K-blocked (GemmParams::kstep_a): LayerNorm, attention and fc1 write that form,
    // the weights were re-laid at load time.  kt = elements in 64 bytes; a matrix of R rows has K-steps R * kt apart.
    const long kt = 64 / (long)c->es;
    auto product = [&](const void* A, long rowsA, const void* W, long rowsW, void* C, long ldc, bool out_f32, int Mg, int Ng, int Kg,
                       const float* bias, int act, const float* residual, long c_kblock) -> int {
        GemmParams p;
        p.A = A; p.B = W; p.C = C; p.bias = bias; p.residual = residual;
        p.lda = kt; p.ldb = kt; p.ldc = ldc; p.ldr = ldc;
        p.kstep_a = rowsA * kt; p.kstep_b = rowsW * kt; p.c_kblock = c_kblock;
        p.M = Mg; p.N = Ng; p.K = Kg; p.batch = 1;
        p.bias_mode = bias ? 1 : 0; p.act = act;
        return uw_gemm(c->dt, out_f32 ? 1 : 0, p, s);
    };
    // V^T[b] = Wv X[b]^T + bv (rows = model columns): the weights are the A operand, window b's rows of X the B operand
    auto vt_product = [&](const void* Wv, const void* X, void* VT, const float* bv) -> int {
        GemmParams p;
        p.A = Wv; p.B = X; p.C = VT; p.bias = bv;
        p.lda = kt; p.ldb = kt; p.ldc = c->Spad;
        p.kstep_a = (long)d * kt; p.kstep_b = M * kt;
        p.sA = 0; p.sB = (long)S * kt; p.sC = (long)d * c->Spad;
        p.M = d; p.N = S; p.K = d; p.batch = B;
        p.bias_mode = 2;
        return uw_gemm(c->dt, 0, p, s);
    };
    for (auto& l : c->enc) {
        RUN(uw_layernorm(c->dt, c->e_x, c->e_ln, nullptr, l.self.ln_g, l.self.ln_b, M, d, s, 2));
        RUN(product(c->e_ln, M, l.self.wq, 2 * d, c->e_qk, 2 * d, false, (int)M, 2 * d, d, l.self.bq, 0, nullptr, 0));
        RUN(vt_product(l.self.wv, c->e_ln, c->e_vt, l.self.bv));
        AttnParams a;
        a.Q = c->e_qk; a.K = (char*)c->e_qk + (size_t)d * c->es; a.VT = c->e_vt; a.O = c->e_attn;
        a.ldq = a.ldk = 2 * d; a.ldvt = c->Spad; a.ldo = d;
        a.sQ = a.sK = (long)S * 2 * d; a.sVT = (long)d * c->Spad; a.sO = (long)S * d;
        a.B = B; a.H = H; a.Sq = S; a.Sk = S; a.causal = 0;
        a.o_kblock = M;
        RUN(uw_flash_attention(c->dt, a, s));
        RUN(product(c->e_attn, M, l.self.wo, d, c->e_x, d, true, (int)M, d, d, l.self.bo, 0, c->e_x, 0));
        RUN(uw_layernorm(c->dt, c->e_x, c->e_ln, nullptr, l.ln2_g, l.ln2_b, M, d, s, 2));
        RUN(product(c->e_ln, M, l.w1, 4 * d, c->e_mlp, 4 * d, false, (int)M, 4 * d, d, l.b1, 1, nullptr, M));
        RUN(product(c->e_mlp, M, l.w2, d, c->e_x, d, true, (int)M, d, 4 * d, l.b2, 0, c->e_x, 0));
    }
    // the encoder output: f32 row-major for callers, compute dtype K-blocked for the cross K/V projection
    RUN(uw_layernorm(c->dt, c->e_x, c->enc_T, c->enc_out32, c->enc_ln_g, c->enc_ln_b, M, d, s, 2));
    c->enc_rows = M;
    c->enc_valid = true;
    return UW_OK;
}

size_t cross_kv_layer_bytes(const uw_ctx* c) { return (size_t)c->Bm * c->S * 2 * c->d * c->es; }
size_t cross_vt_layer_bytes(const uw_ctx* c) { return (size_t)c->Bm * c->d * c->Spad * c->es; }
size_t self_kv_layer_bytes(const uw_ctx* c) { return (size_t)c->Bm * c->Tm * 2 * c->d * c->es; }

int cross_kv_project(uw_ctx* c, int B, hipStream_t s)
{
    const int d = c->d, S = c->S;
    const long kt = 64 / (long)c->es, M = (long)B * S;
    UW_REQUIRE(c->enc_rows == M, "cross_kv: the encoder output in the context holds %ld rows, this call needs %ld", c->enc_rows, M);
    for (size_t li = 0; li < c->dec.size(); ++li) {
        auto& a = c->dec[li].cross;
        void* kv = (char*)c->cross_kv + li * cross_kv_layer_bytes(c);
        void* vt = (char*)c->cross_vt + li * cross_vt_layer_bytes(c);
        // K|V head-contiguous: window b, group g (K heads, then V heads), frame t at ((b*2H + g)*S + t)*64
        GemmParams p;
        p.A = c->enc_T; p.B = a.wkv; p.C = kv; p.bias = a.bkv;
        p.lda = kt; p.ldb = kt; p.ldc = 64;
        p.kstep_a = M * kt; p.kstep_b = 2l * d * kt;
        p.sA = (long)S * kt; p.sB = 0; p.sC = (long)S * 2 * d;
        p.M = S; p.N = 2 * d; p.K = d; p.batch = B;
        p.bias_mode = 1; p.c_group = (long)S * 64;
        // the V half also leaves transposed ([d][Spad] per window), the form the multi-token passes' attention reads
        p.C2 = vt; p.ldc2 = c->Spad; p.sC2 = (long)d * c->Spad; p.c2_col0 = d;
        RUN(uw_gemm(c->dt, 0, p, s));
    }
    c->cross_batch = B;
    return UW_OK;
}

// row idx[b] of window b -> row b of `out`, in the decode step's x layout (the operand of the logits product)
template <typename TT>
__global__ void gather_last_rows_kernel(const float* __restrict__ x, float* __restrict__ out, const int32_t* __restrict__ idx,
                                        int fixed_idx, int T, int d)
{
    const int b = blockIdx.x;
    int t = idx ? idx[b] : fixed_idx;
    t = min(max(t, 0), T - 1);
    const f32x4* src = (const f32x4*)(x + ((long)b * T + t) * d);
    for (int i = threadIdx.x; i < (d >> 2); i += blockDim.x) *(f32x4*)(out + x_elem_offset<TT>(b, i * 4, d)) = src[i];
}

// Multi-token decoder pass over ids [B][T] (positions 0..T-1); fills the self K/V cache for those positions.
//   rows_first/rows_count (device, per window) select query rows whose alignment-head probabilities go to c->rows
//   all_probs: [Ld][B][H][T][S] or null;  logits_all: [B][T][V] or null;  last_idx: device per-window index of the row
//   whose logits go to c->logits (or fixed_last when null and want_last)
// A pass that produces logits carries the operands of its products as [hi | lo] bf16 pairs (bf16 compute only, see
// GemmParams): LayerNorm outputs, attention outputs and the MLP hidden state then keep ~16 mantissa bits, so the
// logits differ from an fp32 evaluation on the same bf16 weights by the rounding of q, k, v and the probabilities
// alone.  A pass that only emits attention rows (forced alignment, the greedy loop's row fix-up) runs without it.
int decoder_forward(uw_ctx* c, const int32_t* ids, int ld_ids, int B, int T, const int32_t* rows_first,
                    const int32_t* rows_count, float* all_probs, float* logits_all, bool want_last, const int32_t* last_idx,
                    int fixed_last, hipStream_t s)
{
    const int d = c->d, S = c->S, H = c->H, Tm = c->Tm;
    UW_REQUIRE(B > 0 && B <= c->cross_batch, "decoder: batch %d exceeds the %d windows with cross K/V", B, c->cross_batch);
    UW_REQUIRE(T > 0 && T <= Tm, "decoder: T=%d outside 1..%d", T, Tm);
    const long M = (long)B * T;
    const int Tpad = (T + 63) & ~63;
    const bool split = c->dt == UW_BF16 && (logits_all || want_last);
    const int w = split ? 2 : 1;            // operand rows are w * K elements wide
    const int kw = split ? d : 0;           // ... and the weights repeat after K
    const int kw4 = split ? 4 * d : 0;
    RUN(uw_embed_tokens(c->d_x, c->tok_emb32, c->dec_pos, ids, ld_ids, nullptr, B, T, d, c->V, Tm, s));
    for (size_t li = 0; li < c->dec.size(); ++li) {
        auto& l = c->dec[li];
        void* skv = (char*)c->self_kv + li * self_kv_layer_bytes(c);
        void* ckv = (char*)c->cross_kv + li * cross_kv_layer_bytes(c);
        void* cvt = (char*)c->cross_vt + li * cross_vt_layer_bytes(c);
        // --- causal self-attention ---
        RUN(uw_layernorm(c->dt, c->d_x, c->d_ln, nullptr, l.self.ln_g, l.self.ln_b, M, d, s, 0, 0, split));
        RUN(gemm(c, c->d_ln, w * d, l.self.wq, d, c->d_qkv, d, false, (int)M, d, w * d, l.self.bq, 0, nullptr, 0, s, 1, 0, 0, 0, 0, 1, 0, 0, kw));
        RUN(gemm(c, c->d_ln, w * d, l.self.wkv, d, skv, 64, false, T, 2 * d, w * d, l.self.bkv, 0, nullptr, 0, s, B, (long)T * w * d, 0,
                 (long)Tm * 2 * d, 0, 1, (long)Tm * 64, 0, kw));
        RUN(gemm(c, l.self.wv, d, c->d_ln, w * d, c->d_vt, Tpad, false, d, T, w * d, l.self.bv, 0, nullptr, 0, s, B, 0, (long)T * w * d,
                 (long)d * Tpad, 0, 2, 0, kw, 0));
        AttnParams a;
        a.Q = c->d_qkv; a.K = skv; a.VT = c->d_vt; a.O = c->d_attn;
        a.ldq = d; a.ldk = 64; a.sKh = (long)Tm * 64; a.ldvt = Tpad; a.ldo = w * d; a.o_split = split;
        a.sQ = (long)T * d; a.sK = (long)Tm * 2 * d; a.sVT = (long)d * Tpad; a.sO = (long)T * w * d;
        a.B = B; a.H = H; a.Sq = T; a.Sk = T; a.causal = 1;
        RUN(uw_flash_attention(c->dt, a, s));
        RUN(gemm(c, c->d_attn, w * d, l.self.wo, d, c->d_x, d, true, (int)M, d, w * d, l.self.bo, 0, c->d_x, d, s, 1, 0, 0, 0, 0, 1, 0, 0, kw));
        // --- cross-attention over the encoder frames ---
        RUN(uw_layernorm(c->dt, c->d_x, c->d_ln, nullptr, l.cross.ln_g, l.cross.ln_b, M, d, s, 0, 0, split));
        RUN(gemm(c, c->d_ln, w * d, l.cross.wq, d, c->d_qkv, d, false, (int)M, d, w * d, l.cross.bq, 0, nullptr, 0, s, 1, 0, 0, 0, 0, 1, 0, 0, kw));
        ProbParams pp;
        pp.Q = c->d_qkv; pp.K = ckv; pp.ldq = d; pp.ldk = 64; pp.sKh = (long)S * 64; pp.sQ = (long)T * d; pp.sK = (long)S * 2 * d;
        pp.B = B; pp.Sq = T; pp.Sk = S;
        if (rows_count && c->n_sel > 0) {
            pp.n_heads_sel = 0;
            for (int k = 0; k < c->n_sel; ++k)
                if (c->cfg.align_layer[k] == (int)li) { pp.heads[pp.n_heads_sel] = c->cfg.align_head[k]; pp.sel_slot[pp.n_heads_sel++] = k; }
            if (pp.n_heads_sel > 0) {
                pp.P = c->rows; pp.sP_b = (long)c->n_sel * Tm * S; pp.sP_sel = (long)Tm * S;
                pp.q_first = rows_first; pp.q_count = rows_count;
                RUN(uw_attention_probs(c->dt, pp, s));
            }
        }
        if (all_probs) {
            pp.q_first = nullptr; pp.q_count = nullptr;
            pp.sP_b = (long)H * T * S; pp.sP_sel = (long)T * S;
            float* base = all_probs + (long)li * B * H * T * S;
            for (int h0 = 0; h0 < H; h0 += 32) {
                pp.n_heads_sel = std::min(32, H - h0);
                for (int k = 0; k < pp.n_heads_sel; ++k) { pp.heads[k] = h0 + k; pp.sel_slot[k] = h0 + k; }
                pp.P = base;
                RUN(uw_attention_probs(c->dt, pp, s));
            }
        }
        a.Q = c->d_qkv; a.K = ckv; a.VT = cvt; a.O = c->d_attn;
        a.ldq = d; a.ldk = 64; a.sKh = (long)S * 64; a.ldvt = c->Spad; a.ldo = w * d;
        a.sQ = (long)T * d; a.sK = (long)S * 2 * d; a.sVT = (long)d * c->Spad; a.sO = (long)T * w * d;
        a.Sq = T; a.Sk = S; a.causal = 0;
        RUN(uw_flash_attention(c->dt, a, s));
        RUN(gemm(c, c->d_attn, w * d, l.cross.wo, d, c->d_x, d, true, (int)M, d, w * d, l.cross.bo, 0, c->d_x, d, s, 1, 0, 0, 0, 0, 1, 0, 0, kw));
        // --- MLP ---
        RUN(uw_layernorm(c->dt, c->d_x, c->d_ln, nullptr, l.ln2_g, l.ln2_b, M, d, s, 0, 0, split));
        RUN(gemm(c, c->d_ln, w * d, l.w1, d, c->d_mlp, w * 4 * d, false, (int)M, 4 * d, w * d, l.b1, 1, nullptr, 0, s, 1, 0, 0, 0, 0, 1, 0, 0, kw, split));
        RUN(gemm(c, c->d_mlp, w * 4 * d, l.w2, 4 * d, c->d_x, d, true, (int)M, d, w * 4 * d, l.b2, 0, c->d_x, d, s, 1, 0, 0, 0, 0, 1, 0, 0, kw4));
    }
    if (logits_all) {
        RUN(uw_layernorm(c->dt, c->d_x, c->d_ln, nullptr, c->dec_ln_g, c->dec_ln_b, M, d, s, 0, 0, split));
        RUN(gemm(c, c->d_ln, w * d, c->tok_emb, d, logits_all, c->V, true, (int)M, c->V, w * d, nullptr, 0, nullptr, 0, s, 1, 0, 0, 0, 0, 1, 0, 0, kw));
    }
    if (want_last) {
        if (c->dt == UW_F32) hipLaunchKernelGGL(gather_last_rows_kernel<float>, dim3(B), dim3(256), 0, s, c->d_x, c->d_last, last_idx, fixed_last, T, d);
        else hipLaunchKernelGGL(gather_last_rows_kernel<bf16_t>, dim3(B), dim3(256), 0, s, c->d_x, c->d_last, last_idx, fixed_last, T, d);
        UW_LAUNCH_CHECK();
        RUN(logits_from_x(c, c->d_last, B, c->d_ln, c->logits, s));
    }
    return UW_OK;
}

// writes the step's residual stream in the x layout (x_elem_offset): row b of the call = window b0 + b
template <typename T>
__global__ void embed_step_kernel(float* __restrict__ x, const float* __restrict__ tok_emb, const float* __restrict__ pos_emb,
                                  const int32_t* __restrict__ tokens, int ld_tok, const int32_t* __restrict__ pos, int d, int vocab,
                                  int n_ctx)
{
    const int b = blockIdx.x;
    const int p = min(max(pos[b], 0), n_ctx - 1);
    const int id = min(max(tokens[(long)b * ld_tok + p], 0), vocab - 1);
    const f32x4* e = (const f32x4*)(tok_emb + (long)id * d);
    const f32x4* pe = (const f32x4*)(pos_emb + (long)p * d);
    for (int i = threadIdx.x; i < (d >> 2); i += blockDim.x) *(f32x4*)(x + x_elem_offset<T>(b, i * 4, d)) = e[i] + pe[i];
}

int decoder_step_chunk(uw_ctx* c, int b0, int B, bool embed, bool greedy, hipStream_t s);
// a step over at most 16 windows takes the one-launch logits form (uw_logits_rows16), which also leaves arg-max candidates
bool step_uses_rows16(const uw_ctx* c, int B) { return c->fused_logits && c->tok_emb_p && B <= 16; }
// one KV-cached decoder step for the token at st_pos[b] of windows [b0, b0 + B) -> their rows of c->logits.
// The step's products take at most 128 rows per launch: larger batches go through in chunks of 128 windows.
// embed: build the residual-stream rows from st_tokens / st_pos first (the greedy loop's later steps get them from
// the previous step's uw_greedy_next instead).  greedy: the caller is the greedy loop; where the step leaves arg-max
// candidates (step_uses_rows16) it reads those and the logits themselves are not stored
int decoder_step(uw_ctx* c, int b0, int B, bool embed, hipStream_t s, bool greedy = false)
{
    for (int off = 0; off < B; off += 128) RUN(decoder_step_chunk(c, b0 + off, std::min(128, B - off), embed, greedy, s));
    return UW_OK;
}

// Every buffer is indexed by window, so disjoint window ranges (starting on multiples of 16) are independent.
int decoder_step_chunk(uw_ctx* c, int b0, int B, bool embed, bool greedy, hipStream_t s)
{
    const int d = c->d, S = c->S, H = c->H, Tm = c->Tm;
    const size_t es = c->es;
    float* x = c->d_x + (long)b0 * d;
    char* qkv = (char*)c->d_qkv + (size_t)b0 * 3 * d * es;
    char* att = (char*)c->d_attn + (size_t)b0 * d * es;
    char* mlp = (char*)c->d_mlp + (size_t)b0 * 4 * d * es;
    // bf16 compute: attention outputs and the MLP hidden state go to the next product as hi + lo planes (GemmParams)
    const size_t plane = (size_t)((c->Bm + 15) & ~15);
    char* att_lo = c->dt == UW_BF16 ? att + plane * d * es : nullptr;
    char* mlp_lo = c->dt == UW_BF16 ? mlp + plane * 4 * d * es : nullptr;
    float* logits = c->logits + (long)b0 * c->logits_ld;
    const int32_t* pos = c->st_pos + b0;
    const int32_t* active = c->st_active + b0;
    if (embed) {
        if (c->dt == UW_F32)
            hipLaunchKernelGGL(embed_step_kernel<float>, dim3(B), dim3(256), 0, s, x, c->tok_emb32, c->dec_pos, c->st_tokens + (long)b0 * Tm, Tm, pos, d, c->V, Tm);
        else
            hipLaunchKernelGGL(embed_step_kernel<bf16_t>, dim3(B), dim3(256), 0, s, x, c->tok_emb32, c->dec_pos, c->st_tokens + (long)b0 * Tm, Tm, pos, d, c->V, Tm);
        UW_LAUNCH_CHECK();
    }
    for (size_t li = 0; li < c->dec.size(); ++li) {
        auto& l = c->dec[li];
        char* skv = (char*)c->self_kv + li * self_kv_layer_bytes(c) + (size_t)b0 * Tm * 2 * d * es;
        char* ckv = (char*)c->cross_kv + li * cross_kv_layer_bytes(c) + (size_t)b0 * S * 2 * d * es;
        // each product names the one that follows it -- across an attention launch too: at the batch sizes this is
        // used for, the lines survive the attention kernel's non-temporal K/V stream (measured: better than leaving
        // the two out-projections out)
        NextProduct n_wo_s, n_q, n_wo_c, n_fc1, n_fc2, n_qkv;
        n_wo_s.w = l.self.wo_p; n_wo_s.n = d; n_wo_s.k = d;
        n_q.w = l.cross.wq_p; n_q.n = d; n_q.k = d; n_q.ln = 1;
        n_wo_c.w = l.cross.wo_p; n_wo_c.n = d; n_wo_c.k = d;
        n_fc1.w = l.w1_p; n_fc1.n = 4 * d; n_fc1.k = d; n_fc1.ln = 1;
        n_fc2.w = l.w2_p; n_fc2.n = d; n_fc2.k = 4 * d;
        if (li + 1 < c->dec.size()) { n_qkv.w = c->dec[li + 1].self.wqkv_p; n_qkv.n = 3 * d; n_qkv.k = d; n_qkv.ln = 1; }
        RUN(gemm_step_ln(c, x, l.self.wqkv_p, l.self.ln_g, l.self.bln, l.self.wsum, qkv, 3 * d, B, 3 * d, d, 0, false, s, nullptr, false, &n_wo_s));
        SelfDecodeParams sp;
        sp.qkv = qkv; sp.kcache = skv; sp.vcache = skv + (size_t)Tm * d * es; sp.out = att; sp.out_lo = att_lo;
        sp.pos = pos; sp.active = active; sp.ld_cache = 64; sp.cache_sB = (long)Tm * 2 * d; sp.cache_sH = (long)Tm * 64; sp.out_packed = 1; sp.B = B; sp.H = H; sp.d = d; sp.n_ctx = Tm; sp.max_rows = c->step_rows;
        const int self_slot = span_begin(c, 1, 8, s);
        RUN(uw_self_attn_decode(c->dt, sp, s));
        span_end(c, 1, self_slot, s);
        RUN(gemm_step(c, att, att_lo, l.self.wo_p, x, d, true, B, d, d, l.self.bo, 0, x, d, false, s, &n_q));
        RUN(gemm_step_ln(c, x, l.cross.wq_p, l.cross.ln_g, l.cross.bln, l.cross.wsum, qkv, d, B, d, d, 0, false, s, nullptr, false, &n_wo_c));
        CrossDecodeParams cp;
        cp.q = qkv; cp.K = ckv; cp.V = ckv + (size_t)S * d * es; cp.out = att; cp.out_lo = att_lo;
        cp.rows_sB = (long)c->n_sel * Tm * S; cp.rows_sSel = (long)Tm * S;
        cp.rows = c->rows ? c->rows + (long)b0 * cp.rows_sB : nullptr;
        cp.row_index = c->st_n_text + b0; cp.active = active;
        for (int h = 0; h < 32; ++h) cp.sel_slot[h] = -1;
        for (int k = 0; k < c->n_sel; ++k)
            if (c->cfg.align_layer[k] == (int)li) cp.sel_slot[c->cfg.align_head[k]] = k;
        cp.out_packed = 1; cp.B = B; cp.H = H; cp.d = d; cp.S = S; cp.n_rows_max = Tm; cp.ld_kv = 64;
        cp.kv_sB = (long)S * 2 * d; cp.kv_sH = (long)S * 64;
        // a handful of (window, head) pairs (single-window use): cut the frames over several workgroups
        cp.n_split = c->split_mode >= 0 ? std::max(1, c->split_mode) : (H * B >= 64 ? 1 : 8);
        cp.split_ws = c->split_ws + (long)b0 * H * 16 * 66;
        cp.split_ticket = c->split_ticket + (long)b0 * H;
        const bool timed = c->prof_on && c->prof_used + 2 <= c->prof_ev.size();
        if (timed) UW_HIP_CHECK(hipEventRecord(c->prof_ev[c->prof_used], s));
        RUN(uw_cross_attn_decode(c->dt, cp, s));
        if (timed) { UW_HIP_CHECK(hipEventRecord(c->prof_ev[c->prof_used + 1], s)); c->prof_used += 2; }
        if (c->prof_on) { ++c->prof_launches; c->prof_windows += B; }
        RUN(gemm_step(c, att, att_lo, l.cross.wo_p, x, d, true, B, d, d, l.cross.bo, 0, x, d, false, s, &n_fc1));
        RUN(gemm_step_ln(c, x, l.w1_p, l.ln2_g, l.b1ln, l.w1sum, mlp, 4 * d, B, 4 * d, d, 1, true, s, mlp_lo, false, &n_fc2));
        RUN(gemm_step(c, mlp, mlp_lo, l.w2_p, x, d, true, B, d, 4 * d, l.b2, 0, x, d, false, s, &n_qkv));
    }
    if (step_uses_rows16(c, B)) {
        LogitsRows16Params lp;
        lp.x = x; lp.gamma = c->dec_ln_g; lp.beta = c->dec_ln_b; lp.W = c->tok_emb_p;
        lp.logits = greedy ? nullptr : logits; lp.ldl = c->logits_ld;
        const long n_blk = uw_logits_rows16_blocks(c->V);
        lp.cand_val = c->cand_val + (long)b0 * n_blk; lp.cand_idx = c->cand_idx + (long)b0 * n_blk;
        lp.M = B; lp.V = c->V; lp.d = d;
        return uw_logits_rows16(lp, s);
    }
    return logits_from_x(c, x, B, (char*)c->d_ln + (size_t)b0 * 2 * d * es, logits, s);
}

__global__ void greedy_init_kernel(GreedyState st, const int32_t* __restrict__ prompt, int ld_prompt,
                                   const int32_t* __restrict__ plen, int B, int max_tokens, int n_ctx)
{
    const int b = blockIdx.x;
    const int P = plen[b];
    for (int i = threadIdx.x; i < st.ld_tok; i += blockDim.x) {
        st.tokens[(long)b * st.ld_tok + i] = (i < P) ? prompt[(long)b * ld_prompt + i] : 0;
        st.text_ids[(long)b * st.ld_tok + i] = 0;
    }
    if (threadIdx.x == 0) {
        const int budget = max(0, min(max_tokens, n_ctx) - P);
        st.pos[b] = P - 1;
        st.n_new[b] = 0;
        st.n_text[b] = 0;
        st.dirty_from[b] = -1;
        st.failed[b] = 0;
        st.max_new[b] = budget;
        st.active[b] = budget > 0 ? 1 : 0;
    }
}

// ids_forced[b] = prompt + text[:-1]; q_first = P - 1; q_count = n_text (0 for windows that need no forced pass)
__global__ void build_forced_ids_kernel(int32_t* __restrict__ ids, int ld, const int32_t* __restrict__ prompt, int ld_prompt,
                                        const int32_t* __restrict__ plen, const int32_t* __restrict__ text, int ld_text,
                                        const int32_t* __restrict__ n_text, const int32_t* __restrict__ need,
                                        int32_t* __restrict__ q_first, int32_t* __restrict__ q_count)
{
    const int b = blockIdx.x;
    const int P = plen[b], N = n_text[b];
    const bool on = (!need || need[b]) && N > 0;
    for (int i = threadIdx.x; i < ld; i += blockDim.x) {
        int v = 0;
        if (i < P) v = prompt[(long)b * ld_prompt + i];
        else if (i < P + N - 1) v = text[(long)b * ld_text + (i - P)];
        ids[(long)b * ld + i] = v;
    }
    if (threadIdx.x == 0) {
        q_first[b] = P - 1;
        q_count[b] = on ? N : 0;
    }
}

__global__ void copy_tokens_out_kernel(const GreedyState st, const int32_t* __restrict__ plen, int32_t* __restrict__ new_tokens,
                                       int32_t* __restrict__ n_new, int32_t* __restrict__ text_ids, int32_t* __restrict__ n_text,
                                       int ld_out)
{
    const int b = blockIdx.x;
    const int P = plen[b], nn = st.n_new[b], nt = st.n_text[b];
    for (int i = threadIdx.x; i < ld_out; i += blockDim.x) {
        if (new_tokens) new_tokens[(long)b * ld_out + i] = (i < nn) ? st.tokens[(long)b * st.ld_tok + P + i] : 0;
        if (text_ids) text_ids[(long)b * ld_out + i] = (i < nt) ? st.text_ids[(long)b * st.ld_tok + i] : 0;
    }
    if (threadIdx.x == 0) {
        if (n_new) n_new[b] = nn;
        if (n_text) n_text[b] = nt;
    }
}

__global__ void count_active_kernel(const int32_t* __restrict__ active, int n, int32_t* __restrict__ out)
{
    int c = 0;
    for (int i = threadIdx.x; i < n; i += 64) c += active[i] ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (threadIdx.x == 0) *out = c;
}

GreedyState state_of(uw_ctx* c)
{
    GreedyState st;
    st.tokens = c->st_tokens; st.ld_tok = c->Tm; st.pos = c->st_pos; st.n_new = c->st_n_new; st.n_text = c->st_n_text;
    st.active = c->st_active; st.dirty_from = c->st_dirty; st.max_new = c->st_max_new; st.n_active = c->st_n_active;
    st.text_ids = c->st_text; st.failed = c->st_failed;
    return st;
}

int check_batch(uw_ctx* c, int B)
{
    UW_REQUIRE(c, "null context");
    UW_REQUIRE(B > 0 && B <= c->Bm, "batch %d outside 1..max_batch=%d", B, c->Bm);
    // a context lives on the device that was current when it was created; make it current for this call as well
    int dev = -1;
    UW_HIP_CHECK(hipGetDevice(&dev));
    if (dev != c->device) UW_HIP_CHECK(hipSetDevice(c->device));
    return UW_OK;
}

int forced_rows(uw_ctx* c, const int32_t* prompt_ids, int ld_prompt, const int32_t* plen_dev, const int32_t* text_ids,
                int ld_text, const int32_t* n_text_dev, const int32_t* need_dev, int B, int T, hipStream_t s)
{
    hipLaunchKernelGGL(build_forced_ids_kernel, dim3(B), dim3(256), 0, s, c->st_ids, c->Tm, prompt_ids, ld_prompt, plen_dev, text_ids,
                       ld_text, n_text_dev, need_dev, c->st_qfirst, c->st_qcount);
    UW_LAUNCH_CHECK();
    return decoder_forward(c, c->st_ids, c->Tm, B, T, c->st_qfirst, c->st_qcount, nullptr, nullptr, false, nullptr, 0, s);
}

}  // namespace

// ================================================================================================
extern "C" {

const char* uw_last_error(void) { return g_err; }
int uw_abi_version(void) { return 1; }

int uw_ctx_create(const uw_config* cfg, const void* const* weights, int n_weights, const float* mel_filters, uw_ctx** out)
{
    UW_REQUIRE(cfg && weights && mel_filters && out, "uw_ctx_create: null argument");
    UW_REQUIRE(cfg->d_model > 0 && cfg->d_model % 64 == 0 && cfg->n_heads * 64 == cfg->d_model,
               "d_model=%d must be n_heads(%d) * 64", cfg->d_model, cfg->n_heads);
    UW_REQUIRE(cfg->d_model <= 2048 && cfg->n_heads <= 32, "d_model=%d too large", cfg->d_model);
    UW_REQUIRE(cfg->dtype == UW_DTYPE_F32 || cfg->dtype == UW_DTYPE_BF16, "unknown dtype %d", cfg->dtype);
    UW_REQUIRE(cfg->n_audio_ctx == 1500 && cfg->n_text_ctx > 0 && cfg->n_text_ctx <= 448 && cfg->n_text_ctx % 64 == 0,
               "n_audio_ctx must be 1500 and n_text_ctx a multiple of 64 up to 448 (got %d, %d)", cfg->n_audio_ctx, cfg->n_text_ctx);
    UW_REQUIRE(cfg->n_mels == 80 || cfg->n_mels == 128, "n_mels=%d unsupported", cfg->n_mels);
    UW_REQUIRE(cfg->max_batch > 0 && cfg->max_batch <= 4096, "max_batch=%d", cfg->max_batch);
    UW_REQUIRE(cfg->n_align_heads >= 0 && cfg->n_align_heads <= UW_MAX_ALIGN_HEADS, "n_align_heads=%d", cfg->n_align_heads);
    for (int k = 0; k < cfg->n_align_heads; ++k)
        UW_REQUIRE(cfg->align_layer[k] >= 0 && cfg->align_layer[k] < cfg->dec_layers && cfg->align_head[k] >= 0 &&
                       cfg->align_head[k] < cfg->n_heads,
                   "alignment head %d = (%d, %d) out of range", k, cfg->align_layer[k], cfg->align_head[k]);
    uw_ctx* c = new uw_ctx();
    c->cfg = *cfg;
    c->dt = cfg->dtype;
    c->es = uw_elem_size(c->dt);
    c->d = cfg->d_model; c->H = cfg->n_heads; c->S = cfg->n_audio_ctx; c->Spad = (c->S + 63) & ~63; c->Tm = cfg->n_text_ctx;
    c->V = cfg->vocab; c->Bm = cfg->max_batch; c->n_sel = cfg->n_align_heads;
    const int kt = c->dt == UW_F32 ? 32 : 64;
    c->k1pad = ((3 * cfg->n_mels + kt - 1) / kt) * kt;
    int rc = UW_OK;
    if (hipGetDevice(&c->device) != hipSuccess) { uw_set_error("no HIP device"); rc = UW_ERR_HIP; }
    if (!rc) rc = uw_mel_tables_create(cfg->n_mels, mel_filters, &c->mel_tables);
    if (!rc) rc = load_weights(c, weights, n_weights);
    if (!rc) rc = alloc_workspace(c);
    if (!rc) rc = create_streams(c);
    if (!rc && hipDeviceSynchronize() != hipSuccess) { uw_set_error("device error during context creation"); rc = UW_ERR_HIP; }
    if (rc) { uw_ctx_destroy(c); return rc; }
    *out = c;
    return UW_OK;
}

int uw_ctx_destroy(uw_ctx* c)
{
    if (!c) return UW_OK;
    (void)hipDeviceSynchronize();
    for (void* p : c->allocs) (void)hipFree(p);
    for (hipEvent_t e : c->prof_ev) (void)hipEventDestroy(e);
    for (auto& sp : c->span)
        for (hipEvent_t e : sp.ev) (void)hipEventDestroy(e);
    for (int k = 0; k < 4; ++k) {
        if (c->sub_stream[k]) (void)hipStreamDestroy(c->sub_stream[k]);
        if (c->sub_event[k]) (void)hipEventDestroy(c->sub_event[k]);
    }
    if (c->fork_event) (void)hipEventDestroy(c->fork_event);
    if (c->host_pin) (void)hipHostFree(c->host_pin);
    uw_mel_tables_destroy(c->mel_tables);
    delete c;
    return UW_OK;
}

int64_t uw_ctx_bytes(const uw_ctx* c) { return c ? c->bytes : 0; }

int uw_profile_begin(uw_ctx* c, int max_launches)
{
    UW_REQUIRE(c && max_launches > 0 && max_launches <= (1 << 20), "uw_profile_begin: bad arguments");
    while (c->prof_ev.size() < (size_t)2 * max_launches) {
        hipEvent_t e;
        UW_HIP_CHECK(hipEventCreate(&e));
        c->prof_ev.push_back(e);
    }
    for (int k = 1; k < 5; ++k) {
        const size_t want = k == 1 ? (size_t)max_launches / 4 + 2 : (k == 4 ? 512 : 128);
        while (c->span[k].ev.size() < want) {
            hipEvent_t e;
            UW_HIP_CHECK(hipEventCreate(&e));
            c->span[k].ev.push_back(e);
        }
        c->span[k].used = 0;
        c->span[k].launches = 0;
    }
    c->prof_used = 0;
    c->prof_launches = 0;
    c->prof_windows = 0;
    c->prof_on = true;
    return UW_OK;
}

int uw_profile_end(uw_ctx* c, int64_t* launches_total, int64_t* launches_timed, int64_t* windows_total, double* timed_ms)
{
    UW_REQUIRE(c && c->prof_on, "uw_profile_end: profiling was not started");
    c->prof_on = false;
    UW_HIP_CHECK(hipDeviceSynchronize());
    double ms = 0.0;
    for (size_t i = 0; i + 1 < c->prof_used; i += 2) {
        float t = 0.f;
        UW_HIP_CHECK(hipEventElapsedTime(&t, c->prof_ev[i], c->prof_ev[i + 1]));
        ms += t;
    }
    if (launches_total) *launches_total = c->prof_launches;
    if (windows_total) *windows_total = c->prof_windows;
    if (launches_timed) *launches_timed = (int64_t)(c->prof_used / 2);
    if (timed_ms) *timed_ms = ms;
    return UW_OK;
}

int uw_profile_span(uw_ctx* c, int kind, int64_t* launches_total, int64_t* launches_timed, double* timed_ms)
{
    UW_REQUIRE(c && !c->prof_on && kind >= 1 && kind <= 4, "uw_profile_span: call after uw_profile_end with kind 1..4");
    uw_ctx::Span& sp = c->span[kind];
    double ms = 0.0;
    for (size_t i = 0; i + 1 < sp.used; i += 2) {
        float t = 0.f;
        UW_HIP_CHECK(hipEventElapsedTime(&t, sp.ev[i], sp.ev[i + 1]));
        ms += t;
    }
    if (launches_total) *launches_total = sp.launches;
    if (launches_timed) *launches_timed = (int64_t)(sp.used / 2);
    if (timed_ms) *timed_ms = ms;
    return UW_OK;
}

int uw_mel(uw_ctx* c, const float* audio, int64_t stride, const int32_t* n_samples, int B, float* mel_out, void* stream)
{
    RUN(check_batch(c, B));
    return uw_mel_run(c->mel_tables, audio, (long)stride, n_samples, B, mel_out, c->mel_max, (hipStream_t)stream);
}

int uw_encode(uw_ctx* c, const float* mel, int B, float* enc_out, void* stream)
{
    RUN(check_batch(c, B));
    UW_REQUIRE(mel, "uw_encode: null mel");
    hipStream_t s = (hipStream_t)stream;
    RUN(encoder_forward(c, mel, B, s));
    if (enc_out)
        UW_HIP_CHECK(hipMemcpyAsync(enc_out, c->enc_out32, (size_t)B * c->S * c->d * 4, hipMemcpyDeviceToDevice, s));
    return UW_OK;
}

int uw_cross_kv(uw_ctx* c, const float* enc_states, int B, void* stream)
{
    RUN(check_batch(c, B));
    hipStream_t s = (hipStream_t)stream;
    if (enc_states) {
        // row-major in the compute dtype (parked in the LayerNorm scratch), then K-blocked as the projection reads it
        RUN(uw_convert(c->dt, enc_states, c->e_ln, (long)B * c->S * c->d, 1.f, s));
        RUN(uw_pack_kblock(c->dt, c->e_ln, c->enc_T, (long)B * c->S, c->d, s));
        c->enc_rows = (long)B * c->S;
    } else {
        UW_REQUIRE(c->enc_valid, "uw_cross_kv: no encoder output in the context");
    }
    return cross_kv_project(c, B, s);
}

int uw_decoder_forward(uw_ctx* c, const int32_t* ids, int ld_ids, int B, int T, float* logits_all, float* logits_last,
                       float* cross_probs, void* stream)
{
    RUN(check_batch(c, B));
    UW_REQUIRE(ids && ld_ids >= T, "uw_decoder_forward: bad ids");
    hipStream_t s = (hipStream_t)stream;
    RUN(decoder_forward(c, ids, ld_ids, B, T, nullptr, nullptr, cross_probs, logits_all, logits_last != nullptr, nullptr, T - 1, s));
    if (logits_last)
        UW_HIP_CHECK(hipMemcpy2DAsync(logits_last, (size_t)c->V * 4, c->logits, (size_t)c->logits_ld * 4, (size_t)c->V * 4, B,
                                      hipMemcpyDeviceToDevice, s));
    return UW_OK;
}

namespace {
__global__ void step_state_kernel(int32_t* __restrict__ tokens, int ld_tok, int32_t* __restrict__ st_pos, int32_t* __restrict__ st_active,
                                  int32_t* __restrict__ st_n_text, const int32_t* __restrict__ ids, const int32_t* __restrict__ pos,
                                  int B, int n_ctx)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int p = pos[b];
    const bool ok = p >= 0 && p < n_ctx;
    if (ok) tokens[(long)b * ld_tok + p] = ids[b];
    st_pos[b] = ok ? p : 0;
    st_active[b] = ok ? 1 : 0;      // a position outside the context leaves the window's logits untouched
    st_n_text[b] = -1;              // no attention-row capture
}
}  // namespace

int uw_decode_step(uw_ctx* c, const int32_t* ids, const int32_t* pos, int B, float* logits, void* stream)
{
    RUN(check_batch(c, B));
    UW_REQUIRE(ids && pos && logits, "uw_decode_step: null argument");
    UW_REQUIRE(B <= c->cross_batch, "uw_decode_step: call uw_cross_kv for these %d windows first", B);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(step_state_kernel, dim3((B + 63) / 64), dim3(64), 0, s, c->st_tokens, c->Tm, c->st_pos, c->st_active, c->st_n_text,
                       ids, pos, B, c->Tm);
    UW_LAUNCH_CHECK();
    c->step_rows = 0;   // positions are device data here
    RUN(decoder_step(c, 0, B, true, s));
    UW_HIP_CHECK(hipMemcpy2DAsync(logits, (size_t)c->V * 4, c->logits, (size_t)c->logits_ld * 4, (size_t)c->V * 4, B,
                                  hipMemcpyDeviceToDevice, s));
    return UW_OK;
}

int uw_decode_greedy(uw_ctx* c, const int32_t* prompt_ids, int ld_prompt, const int32_t* prompt_len_host, int B, int max_tokens,
                     int eot, int sot, int ts_begin, int ts_end, int32_t* new_tokens, int32_t* n_new, int32_t* text_ids,
                     int32_t* n_text, void* stream)
{
    RUN(check_batch(c, B));
    UW_REQUIRE(prompt_ids && prompt_len_host, "uw_decode_greedy: null prompt");
    UW_REQUIRE(B <= c->cross_batch, "uw_decode_greedy: call uw_cross_kv for these %d windows first", B);
    hipStream_t s = (hipStream_t)stream;
    int Pmax = 0, Pmin = 1 << 30;
    for (int b = 0; b < B; ++b) { Pmax = std::max(Pmax, prompt_len_host[b]); Pmin = std::min(Pmin, prompt_len_host[b]); }
    UW_REQUIRE(Pmin >= 1 && Pmax <= c->Tm && Pmax <= ld_prompt, "prompt lengths must be within 1..%d (got %d..%d)", c->Tm, Pmin, Pmax);
    int32_t* plen_dev = c->st_zero;   // per-window prompt lengths
    memcpy(c->host_pin, prompt_len_host, (size_t)B * 4);
    UW_HIP_CHECK(hipMemcpyAsync(plen_dev, c->host_pin, (size_t)B * 4, hipMemcpyHostToDevice, s));
    GreedyState st = state_of(c);
    hipLaunchKernelGGL(greedy_init_kernel, dim3(B), dim3(256), 0, s, st, prompt_ids, ld_prompt, plen_dev, B, max_tokens, c->Tm);
    UW_LAUNCH_CHECK();
    const int max_steps = std::max(0, std::min(max_tokens, c->Tm) - Pmin);
    if (max_steps > 0) {
        // prompt pass over every prompt token but the last: fills the self K/V cache.  The last prompt token then goes
        // through the KV-cached step like every generated one, so all logits and all captured attention rows come from
        // one set of kernels (positions a shorter prompt does not have are overwritten by that window's own steps).
        if (Pmax > 1) RUN(decoder_forward(c, c->st_tokens, c->Tm, B, Pmax - 1, nullptr, nullptr, nullptr, nullptr, false, nullptr, 0, s));
        // split the windows into up to four ranges, one stream each
        const int n_sub = c->n_streams <= 1 ? 1 : (B >= 64 ? std::min(4, c->n_streams) : (B >= 32 ? 2 : 1));
        int sub_b0[5];
        // ranges start on multiples of 16 windows: the step's fragment-major activations are laid out in 16-row bands
        for (int k = 0; k <= n_sub; ++k) sub_b0[k] = k == n_sub ? B : std::min(B, (int)(((long)B * k / n_sub + 15) & ~15l));
        hipStream_t ss[4];
        if (n_sub > 1) {
            UW_HIP_CHECK(hipEventRecord(c->fork_event, s));
            for (int k = 0; k < n_sub; ++k) { ss[k] = c->sub_stream[k]; UW_HIP_CHECK(hipStreamWaitEvent(ss[k], c->fork_event, 0)); }
        } else {
            ss[0] = s;
        }
        int32_t* h_active = c->host_pin + 4 * c->Bm;       // [n_sub] read back every 8 steps
        bool sub_done[4] = {false, false, false, false};
        for (int k = 0; k < n_sub; ++k) {
            // per-range live counters: recount from the active flags
            hipLaunchKernelGGL(count_active_kernel, dim3(1), dim3(64), 0, ss[k], c->st_active + sub_b0[k], sub_b0[k + 1] - sub_b0[k], c->st_n_active + k);
            UW_LAUNCH_CHECK();
        }
        for (int step = 0; step < max_steps; ++step) {
            bool any = false;
            for (int k = 0; k < n_sub; ++k) {
                if (sub_done[k]) continue;
                any = true;
                const int b0 = sub_b0[k], Bn = sub_b0[k + 1] - sub_b0[k];
                GreedyState sk = st;
                sk.tokens += (long)b0 * st.ld_tok; sk.text_ids += (long)b0 * st.ld_tok;
                sk.pos += b0; sk.n_new += b0; sk.n_text += b0; sk.active += b0; sk.dirty_from += b0; sk.max_new += b0; sk.failed += b0;
                sk.n_active = c->st_n_active + k;
                c->step_rows = std::min(c->Tm, Pmax + step);   // the longest prompt's position + 1 at this step
                const int step_slot = span_begin(c, 4, 8, ss[k]);
                RUN(decoder_step(c, b0, Bn, step == 0, ss[k], true));
                const bool cand = step_uses_rows16(c, Bn);
                const long n_blk = uw_logits_rows16_blocks(c->V);
                RUN(uw_greedy_next(c->dt, sk, c->logits + (long)b0 * c->logits_ld, c->logits_ld, Bn, c->V, eot, sot, ts_begin, ts_end,
                                   c->d_x + (long)b0 * c->d, c->tok_emb32, c->dec_pos, c->d, ss[k], cand ? c->cand_val + (long)b0 * n_blk : nullptr,
                                   cand ? c->cand_idx + (long)b0 * n_blk : nullptr, (int)n_blk));
                span_end(c, 4, step_slot, ss[k]);
                if ((step & 7) == 7 && step + 1 < max_steps) {
                    // non-blocking look at the counter copied 8 steps ago, then queue the next copy
                    if (step >= 15 && hipEventQuery(c->sub_event[k]) == hipSuccess && h_active[k] <= 0) sub_done[k] = true;
                    UW_HIP_CHECK(hipMemcpyAsync(h_active + k, c->st_n_active + k, 4, hipMemcpyDeviceToHost, ss[k]));
                    UW_HIP_CHECK(hipEventRecord(c->sub_event[k], ss[k]));
                }
            }
            if (!any) break;
        }
        if (n_sub > 1) {
            for (int k = 0; k < n_sub; ++k) {
                UW_HIP_CHECK(hipEventRecord(c->sub_event[k], ss[k]));
                UW_HIP_CHECK(hipStreamWaitEvent(s, c->sub_event[k], 0));
            }
        }
    }
    // which windows carry text after a generated non-text token?  Their captured rows differ from the reference's
    // replay (align.py:605-623 feeds prompt + TEXT tokens only), so they get a teacher-forced pass.
    int32_t *h_ntext = c->host_pin, *h_dirty = c->host_pin + c->Bm, *h_need = c->host_pin + 2 * c->Bm;
    UW_HIP_CHECK(hipMemcpyAsync(h_ntext, c->st_n_text, (size_t)B * 4, hipMemcpyDeviceToHost, s));
    UW_HIP_CHECK(hipMemcpyAsync(h_dirty, c->st_dirty, (size_t)B * 4, hipMemcpyDeviceToHost, s));
    int32_t* h_failed = c->host_pin + 6 * c->Bm + 16;   // clear of h_active ([4 Bm, 4 Bm + 4)) for every Bm
    UW_HIP_CHECK(hipMemcpyAsync(h_failed, c->st_failed, (size_t)B * 4, hipMemcpyDeviceToHost, s));
    UW_HIP_CHECK(hipStreamSynchronize(s));
    c->window_status.assign(h_failed, h_failed + B);
    int T_forced = 0;
    for (int b = 0; b < B; ++b) {
        h_need[b] = (h_dirty[b] >= 0 && h_ntext[b] > h_dirty[b]) ? 1 : 0;
        if (h_need[b]) T_forced = std::max(T_forced, prompt_len_host[b] + h_ntext[b] - 1);
    }
    if (T_forced > 0 && c->n_sel > 0) {
        UW_HIP_CHECK(hipMemcpyAsync(c->st_max_new, h_need, (size_t)B * 4, hipMemcpyHostToDevice, s));
        RUN(forced_rows(c, c->st_tokens, c->Tm, plen_dev, c->st_text, c->Tm, c->st_n_text, c->st_max_new, B, std::min(T_forced, c->Tm), s));
    }
    hipLaunchKernelGGL(copy_tokens_out_kernel, dim3(B), dim3(256), 0, s, st, plen_dev, new_tokens, n_new, text_ids, n_text, c->Tm);
    UW_LAUNCH_CHECK();
    UW_HIP_CHECK(hipStreamSynchronize(s));
    return UW_OK;
}

int uw_window_status(uw_ctx* c, int B, int32_t* status_host)
{
    UW_REQUIRE(c && status_host && B > 0, "uw_window_status: bad arguments");
    UW_REQUIRE((size_t)B <= c->window_status.size(), "uw_window_status: the last greedy decode covered %d windows, %d asked for",
               (int)c->window_status.size(), B);
    memcpy(status_host, c->window_status.data(), (size_t)B * 4);
    return UW_OK;
}

int uw_attn_rows_forced(uw_ctx* c, const int32_t* prompt_ids, int ld_prompt, const int32_t* prompt_len_host,
                        const int32_t* text_ids, int ld_text, const int32_t* n_text_host, int B, void* stream)
{
    RUN(check_batch(c, B));
    UW_REQUIRE(prompt_ids && prompt_len_host && text_ids && n_text_host, "uw_attn_rows_forced: null argument");
    UW_REQUIRE(c->n_sel > 0, "uw_attn_rows_forced: the context has no alignment heads");
    hipStream_t s = (hipStream_t)stream;
    int T = 0;
    for (int b = 0; b < B; ++b) {
        UW_REQUIRE(prompt_len_host[b] >= 1 && n_text_host[b] >= 0 && prompt_len_host[b] + n_text_host[b] - 1 <= c->Tm &&
                       prompt_len_host[b] <= ld_prompt && n_text_host[b] <= ld_text,
                   "window %d: prompt %d + text %d does not fit %d positions", b, prompt_len_host[b], n_text_host[b], c->Tm);
        T = std::max(T, prompt_len_host[b] + std::max(n_text_host[b], 1) - 1);
    }
    memcpy(c->host_pin, prompt_len_host, (size_t)B * 4);
    memcpy(c->host_pin + c->Bm, n_text_host, (size_t)B * 4);
    UW_HIP_CHECK(hipMemcpyAsync(c->st_zero, c->host_pin, (size_t)B * 4, hipMemcpyHostToDevice, s));
    UW_HIP_CHECK(hipMemcpyAsync(c->st_n_text, c->host_pin + c->Bm, (size_t)B * 4, hipMemcpyHostToDevice, s));
    RUN(forced_rows(c, prompt_ids, ld_prompt, c->st_zero, text_ids, ld_text, c->st_n_text, nullptr, B, T, s));
    UW_HIP_CHECK(hipStreamSynchronize(s));   // host_pin is reused by the next call
    return UW_OK;
}

int uw_get_attn_rows(uw_ctx* c, int B, int n_rows, float* rows, void* stream)
{
    RUN(check_batch(c, B));
    UW_REQUIRE(rows && n_rows > 0 && n_rows <= c->Tm && c->n_sel > 0, "uw_get_attn_rows: bad arguments");
    UW_HIP_CHECK(hipMemcpy2DAsync(rows, (size_t)n_rows * c->S * 4, c->rows, (size_t)c->Tm * c->S * 4, (size_t)n_rows * c->S * 4,
                                  (size_t)B * c->n_sel, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return UW_OK;
}

int uw_cost_matrix(const float* rows, int n_sel, const int32_t* n_tok, int B, int n_rows_max, int S, double* cost, void* stream)
{
    return uw_cost_matrix_run(rows, (long)n_sel * n_rows_max * S, (long)n_rows_max * S, n_sel, n_tok, B, n_rows_max, S, cost,
                              (hipStream_t)stream);
}

int64_t uw_dtw_workspace_bytes(int B, int n_rows_max, int S)
{
    return (int64_t)B * n_rows_max * (int64_t)uw_dtw_dir_words(S) * 4;
}

int uw_dtw(const double* cost, const int32_t* n_tok, int B, int n_rows_max, int S, void* dir_ws, int32_t* boundaries,
           int32_t* path_i, int32_t* path_j, int32_t* path_len, void* stream)
{
    return uw_dtw_run(cost, n_tok, B, n_rows_max, S, (uint8_t*)dir_ws, boundaries, path_i, path_j, path_len, (hipStream_t)stream);
}

int uw_dtw_static(const double* cost, const int32_t* n_tok, int B, int n_rows_max, int S, int32_t* boundaries, int32_t* path_i,
                  int32_t* path_j, int32_t* path_len, void* stream)
{
    return uw_dtw_static_run(cost, n_tok, B, n_rows_max, S, boundaries, path_i, path_j, path_len, (hipStream_t)stream);
}

int uw_align(uw_ctx* c, const int32_t* n_text, int B, int32_t* boundaries, void* stream)
{
    RUN(check_batch(c, B));
    UW_REQUIRE(n_text && boundaries && c->n_sel > 0, "uw_align: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    RUN(uw_cost_matrix_run(c->rows, (long)c->n_sel * c->Tm * c->S, (long)c->Tm * c->S, c->n_sel, n_text, B, c->Tm, c->S, c->cost, s));
    const int dtw_slot = span_begin(c, 3, 1, s);
    const int rc = uw_dtw_run(c->cost, n_text, B, c->Tm, c->S, c->dtw_ws, boundaries, nullptr, nullptr, nullptr, s);
    span_end(c, 3, dtw_slot, s);
    return rc;
}

int uw_transcribe_windows(uw_ctx* c, const float* audio, int64_t stride, const int32_t* n_samples, const int32_t* prompt_ids,
                          int ld_prompt, const int32_t* prompt_len_host, int B, int max_tokens, int eot, int sot, int ts_begin,
                          int ts_end, int32_t* new_tokens, int32_t* n_new, int32_t* text_ids, int32_t* n_text,
                          int32_t* boundaries, void* stream)
{
    RUN(check_batch(c, B));
    UW_REQUIRE(audio && n_samples && n_text, "uw_transcribe_windows: null argument");
    UW_REQUIRE(boundaries || c->n_sel == 0, "uw_transcribe_windows: boundaries missing");
    hipStream_t s = (hipStream_t)stream;
    // the log-mel tile is parked in the (not yet used) decoder MLP scratch
    float* mel = (float*)c->e_mlp;
    RUN(uw_mel_run(c->mel_tables, audio, (long)stride, n_samples, B, mel, c->mel_max, s));
    const int enc_slot = span_begin(c, 2, 1, s);
    RUN(encoder_forward(c, mel, B, s));
    span_end(c, 2, enc_slot, s);
    RUN(cross_kv_project(c, B, s));
    RUN(uw_decode_greedy(c, prompt_ids, ld_prompt, prompt_len_host, B, max_tokens, eot, sot, ts_begin, ts_end, new_tokens, n_new,
                         text_ids, n_text, stream));
    if (c->n_sel == 0) {
        // a context without alignment heads (unknown model key or a mask of the wrong size, align.py:81-84,595-597)
        // transcribes without timestamps: boundary row b starts with -1, "no alignment for this window"
        if (boundaries) UW_HIP_CHECK(hipMemsetAsync(boundaries, 0xff, (size_t)B * (c->Tm + 1) * 4, s));
        return UW_OK;
    }
    return uw_align(c, n_text, B, boundaries, stream);
}

}  // extern "C"
