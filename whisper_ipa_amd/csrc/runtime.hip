// Model runtime: sequences the kernels of libwipa for a whole encoder pass, the cross-KV
// projection, the KV-cached greedy decode loop (replayed from a hipGraph: every per-step
// quantity -- token position, KV write offset, visible keys -- lives in device memory, so
// one captured step is valid for all steps) and the teacher-forced decoder.
//
// Replaces mlx_whisper.whisper.{AudioEncoder,TextDecoder}.__call__ and
// mlx_whisper.decoding.DecodingTask._main_loop (call sites scripts/train_whisper_ipa.py:223,232,356;
// scripts/transcribe_single.py:54-55; scripts/evaluate_model.py:197-200).
#include <algorithm>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <strings.h>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "wipa_common.h"

// ------------------------------------------------------------------ errors / version
static thread_local char g_err[512] = "";
void wipa_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* wipa_last_error(void) { return g_err; }
extern "C" int wipa_version(void) { return 100; }

extern "C" int wipa_stream_create_cu_limited(int n_cus, wipa_stream_t* out) {
    int dev = 0, total = 0;
    WIPA_CHECK_HIP(hipGetDevice(&dev));
    WIPA_CHECK_HIP(hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, dev));
    WIPA_REQUIRE(out && n_cus >= 8 && n_cus <= total && n_cus % 8 == 0, "wipa_stream_create_cu_limited: n_cus=%d must be a multiple of 8 in [8, %d]",
                 n_cus, total);
    uint32_t mask[32] = {0};
    WIPA_REQUIRE(total <= 32 * 32, "wipa_stream_create_cu_limited: %d CUs", total);
    for (int i = 0; i < n_cus; ++i) mask[i / 32] |= 1u << (i % 32);
    hipStream_t s = nullptr;
    WIPA_CHECK_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)((total + 31) / 32), mask));
    *out = (wipa_stream_t)s;
    return WIPA_OK;
}
extern "C" int wipa_stream_create(wipa_stream_t* out) {
    WIPA_REQUIRE(out, "wipa_stream_create: null out");
    hipStream_t s = nullptr;
    WIPA_CHECK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = (wipa_stream_t)s;
    return WIPA_OK;
}
extern "C" int wipa_stream_destroy(wipa_stream_t s) {
    WIPA_REQUIRE(s, "wipa_stream_destroy: null stream");
    WIPA_CHECK_HIP(hipStreamDestroy((hipStream_t)s));
    return WIPA_OK;
}

namespace {

constexpr float QK_SCALE = 0.35355339059327379f;  // 64 ** -0.25
constexpr int T_ENC_PAD = 1536;                     // V^T row length (>= ceil64(1500))
constexpr int ROWS_IN = WIPA_N_FRAMES + 2;          // 3002 padded frames per clip

inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }
inline int k_multiple(int dtype) { return dtype == WIPA_BF16 ? 64 : 32; }
inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

#define RT_CALL(expr)                 \
    do {                              \
        int _rc = (expr);             \
        if (_rc != WIPA_OK) return _rc; \
    } while (0)

int cfg_check(const wipa_model_cfg* c) {
    WIPA_REQUIRE(c, "null cfg");
    WIPA_REQUIRE(c->dtype == WIPA_F32 || c->dtype == WIPA_BF16, "cfg.dtype %d", c->dtype);
    WIPA_REQUIRE(c->n_audio_state == c->n_audio_head * WIPA_HEAD_DIM && c->n_text_state == c->n_text_head * WIPA_HEAD_DIM,
                 "head_dim must be 64 (state %d heads %d)", c->n_audio_state, c->n_audio_head);
    WIPA_REQUIRE(c->n_audio_ctx == WIPA_N_FRAMES / 2, "n_audio_ctx must be 1500");
    WIPA_REQUIRE(c->n_audio_state % 64 == 0 && c->n_text_state % 64 == 0, "state must be a multiple of 64");
    WIPA_REQUIRE(c->dec_w_dtype == 0 || (c->dec_w_dtype == WIPA_FP8_E4M3 && c->dtype == WIPA_BF16),
                 "cfg.dec_w_dtype %d: fp8 (e4m3) decoder weights need a bf16 model", c->dec_w_dtype);
    WIPA_REQUIRE(c->dec_cross_absorbed == 0 ||
                     (c->dec_cross_absorbed == 1 && c->dtype == WIPA_BF16 && c->dec_w_dtype == 0 && c->n_text_head <= 16 &&
                      (c->n_text_state == 384 || c->n_text_state == 512 || c->n_text_state == 768 || c->n_text_state == 1024)),
                 "cfg.dec_cross_absorbed %d: absorbed cross-attention needs a bf16 model without fp8 decoder tables, <= 16 heads, d in "
                 "{384, 512, 768, 1024}", c->dec_cross_absorbed);
    WIPA_REQUIRE(c->dec_cross_splits >= 0 && c->dec_cross_splits <= 4 && (c->dec_cross_splits == 0 || c->dec_cross_absorbed == 1),
                 "cfg.dec_cross_splits %d: 0 (default) or 1..4 frame splits, with dec_cross_absorbed = 1 only", c->dec_cross_splits);
    WIPA_REQUIRE(c->enc_act_fp8 == 0 || (c->enc_act_fp8 == 1 && c->dtype == WIPA_BF16 && c->n_audio_state % 128 == 0),
                 "cfg.enc_act_fp8 %d: fp8 encoder activations need a bf16 model whose width is a multiple of 128", c->enc_act_fp8);
    return WIPA_OK;
}

// ---- stage profiler (wipa_profile_begin / wipa_profile_end): HIP events around every launch of the encoder forward and
// the cross-K/V projection, summed per kernel class.  Host-thread local; inactive (one pointer test) otherwise.
enum { PROF_GEMM = 0, PROF_ATTN = 1, PROF_NORM = 2, PROF_OTHER = 3, PROF_CLASSES = 4 };
struct Profiler {
    hipStream_t stream;
    std::vector<std::tuple<int, hipEvent_t, hipEvent_t>> spans;
};
thread_local Profiler* t_prof = nullptr;
struct ProfSpan {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int cls;
    explicit ProfSpan(int c) : cls(c) {
        if (!t_prof) return;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { e0 = e1 = nullptr; return; }
        hipEventRecord(e0, t_prof->stream);
    }
    ~ProfSpan() {
        if (!t_prof || !e0) return;
        hipEventRecord(e1, t_prof->stream);
        t_prof->spans.emplace_back(cls, e0, e1);
    }
};
#define PROF(cls, call)        \
    do {                       \
        ProfSpan _span(cls);   \
        RT_CALL(call);         \
    } while (0)

// wipa_model_cfg.f32_split of the runtime call in progress on this host thread (set by the entry points below)
thread_local int t_f32_split = 0;
struct SplitScope {
    int prev;
    explicit SplitScope(const wipa_model_cfg* c) : prev(t_f32_split) { t_f32_split = (c && c->dtype == WIPA_F32 && c->f32_split) ? 1 : 0; }
    ~SplitScope() { t_f32_split = prev; }
};

// fp8 decoder weights (wipa_model_cfg.dec_w_dtype = WIPA_FP8_E4M3): matrix pointer -> per-row scale, built from the weight
// table by the decode entry points for the duration of the call (host-thread local, enqueue-time only).
typedef std::map<const void*, const float*> W8Map;
thread_local const W8Map* t_w8 = nullptr;
struct W8Scope {
    W8Map map;
    const W8Map* prev;
    W8Scope(const wipa_model_cfg* c, const void* const* w) : prev(t_w8) {
        if (c && w && c->dec_w_dtype == WIPA_FP8_E4M3) {
            const int base = WIPA_DEC_GLOBAL + WIPA_DEC_PER_LAYER * c->n_text_layer;
            map[w[0]] = (const float*)w[base];
            static const int slots[WIPA_DEC_FP8_PER_LAYER] = {2, 4, 8, 12, 16, 18};
            for (int l = 0; l < c->n_text_layer; ++l)
                for (int j = 0; j < WIPA_DEC_FP8_PER_LAYER; ++j)
                    map[w[WIPA_DEC_GLOBAL + WIPA_DEC_PER_LAYER * l + slots[j]]] = (const float*)w[base + 1 + WIPA_DEC_FP8_PER_LAYER * l + j];
            t_w8 = &map;
        }
    }
    ~W8Scope() { t_w8 = prev; }
};
inline int emb_dtype(const wipa_model_cfg* c) { return c->dec_w_dtype == WIPA_FP8_E4M3 ? WIPA_FP8_E4M3 : c->dtype; }
inline const float* emb_scale(const wipa_model_cfg* c, const void* const* w) {
    return c->dec_w_dtype == WIPA_FP8_E4M3 ? (const float*)w[WIPA_DEC_GLOBAL + WIPA_DEC_PER_LAYER * c->n_text_layer] : nullptr;
}

int gemm(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int M, int N, int K, int in_dt,
         int out_dt, const float* bias, int act, const void* residual, wipa_stream_t s, wipa_gemm_desc* extra = nullptr) {
    wipa_gemm_desc g;
    if (extra) g = *extra; else memset(&g, 0, sizeof(g));
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.C = C; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.in_dtype = in_dt; g.out_dtype = out_dt;
    g.bias = bias; g.act = act; g.residual = residual;
    g.f32_split = t_f32_split;
    if (t_w8) {  // fp8 decoder weights: matrices are recognised by their table pointer
        auto it = t_w8->find(W);
        if (it != t_w8->end()) {
            g.w_dtype = WIPA_FP8_E4M3;
            g.w_scale = it->second;
        }
    }
    return wipa_gemm(&g, s);
}

// fp8 x fp8 GEMM (cfg.enc_act_fp8): A codes [M, K] + a_scale[M], W codes [N, K] + w_scale[N]
int gemm_f8(const void* A, int64_t lda, const float* a_scale, const void* W, int64_t ldw, const float* w_scale, void* C, int64_t ldc,
            int M, int N, int K, int out_dt, const float* bias, int act, const void* residual, wipa_stream_t s,
            wipa_gemm_desc* extra = nullptr) {
    wipa_gemm_desc g;
    if (extra) g = *extra; else memset(&g, 0, sizeof(g));
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.C = C; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.in_dtype = WIPA_FP8_E4M3; g.out_dtype = out_dt;
    g.bias = bias; g.act = act; g.residual = residual;
    g.a_scale = a_scale; g.w_scale = w_scale;
    return wipa_gemm(&g, s);
}

// ------------------------------------------------------------------ encoder
struct EncWs {
    size_t c1, x, ln, qk, v, ao, h, total;
};
EncWs enc_ws(const wipa_model_cfg* c, int B) {
    const size_t e = wipa_dtype_size(c->dtype), d = c->n_audio_state, M = (size_t)B * c->n_audio_ctx;
    EncWs w;
    size_t o = 0;
    w.c1 = o; o += align256(((size_t)B * ROWS_IN + 4) * d * e);
    w.x = o;  o += align256(M * d * 4);
    w.ln = o; o += align256(M * d * e);
    w.qk = o; o += align256(M * 2 * d * e);
    w.v = o;  o += align256(c->dtype == WIPA_BF16 ? (size_t)B * d * T_ENC_PAD * e + 256 : M * d * e);
    w.ao = o; o += align256(M * d * e);
    w.h = o;  o += align256(M * 4 * d * e);
    w.total = o;
    return w;
}

}  // namespace

extern "C" size_t wipa_encoder_workspace_bytes(const wipa_model_cfg* cfg, int B) {
    if (!cfg || B <= 0) return 0;
    return enc_ws(cfg, B).total;
}

extern "C" int wipa_encoder_forward(const wipa_model_cfg* cfg, const void* const* w, const void* mel_padded, void* out,
                                    void* workspace, size_t workspace_bytes, int B, wipa_stream_t stream) {
    RT_CALL(cfg_check(cfg));
    SplitScope split_scope(cfg);
    WIPA_REQUIRE(w && mel_padded && out && workspace && B > 0, "wipa_encoder_forward: null pointer / bad batch");
    const EncWs L = enc_ws(cfg, B);
    WIPA_REQUIRE(workspace_bytes >= L.total, "wipa_encoder_forward: workspace too small (%zu < %zu)", workspace_bytes, L.total);
    hipStream_t s = (hipStream_t)stream;
    const int dt = cfg->dtype;
    const size_t e = wipa_dtype_size(dt);
    const int d = cfg->n_audio_state, H = cfg->n_audio_head, T = cfg->n_audio_ctx, M = B * T;
    char* ws = (char*)workspace;
    void* c1 = ws + L.c1;
    float* x = (float*)(ws + L.x);
    void* ln = ws + L.ln;
    void* qk = ws + L.qk;
    void* vb = ws + L.v;
    void* ao = ws + L.ao;
    void* hb = ws + L.h;
    const int K1 = round_up(3 * cfg->n_mels, k_multiple(dt));

    // conv1 (k3,p1) + GELU as a GEMM with overlapping rows; output row g lands at c1 row g+1,
    // dead rows (t >= 3000) write the zero halos of conv2.
    WIPA_CHECK_HIP(hipMemsetAsync(c1, 0, (size_t)d * e, s));
    {
        wipa_gemm_desc g;
        memset(&g, 0, sizeof(g));
        g.rg_in = ROWS_IN; g.rg_valid = WIPA_N_FRAMES; g.rg_stride = (int64_t)ROWS_IN * d; g.zero_invalid_rows = 1;
        g.c_offset = d;
        PROF(PROF_GEMM, gemm(mel_padded, cfg->n_mels, w[0], K1, c1, d, B * ROWS_IN, d, K1, dt, dt, (const float*)w[1], 1, nullptr,
                     stream, &g));
    }
    // conv2 (k3,s2,p1) + GELU + positional embedding -> residual stream x (f32)
    {
        wipa_gemm_desc g;
        memset(&g, 0, sizeof(g));
        g.rg_in = T + 1; g.rg_valid = T; g.rg_stride = (int64_t)T * d;
        g.pos = (const float*)w[4]; g.ldpos = d;
        PROF(PROF_GEMM, gemm(c1, 2 * d, w[2], 3 * d, x, d, B * (T + 1), d, 3 * d, dt, WIPA_F32, (const float*)w[3], 1, nullptr,
                     stream, &g));
    }
    if (dt == WIPA_BF16) WIPA_CHECK_HIP(hipMemsetAsync(vb, 0, (size_t)B * d * T_ENC_PAD * e, s));
    // fp8 activations (cfg.enc_act_fp8, BASELINE.json configs[4]): the LayerNorm outputs and the GELU output are quantised per
    // row to e4m3 and q|k, value, mlp1, mlp2 multiply fp8 x fp8 on the block-scaled fp8 MFMA.  Buffers are the bf16 path's:
    // `ln` holds the codes [M, d] followed by their M row scales, the (idle) q|k buffer takes the codes of the GELU output
    // [M, 4d] and the (idle) attention-output buffer their row scales.
    const bool f8 = cfg->enc_act_fp8 != 0;
    float* ln_scale = (float*)((char*)ln + (size_t)M * d);  // M*d bytes of codes, then M floats (M*d*2 bytes available)
    const void* const* w8 = w + WIPA_ENC_GLOBAL + WIPA_ENC_PER_LAYER * cfg->n_audio_layer;
    for (int l = 0; l < cfg->n_audio_layer; ++l) {
        const void* const* lw = w + WIPA_ENC_GLOBAL + WIPA_ENC_PER_LAYER * l;
        const void* const* l8 = w8 + WIPA_ENC_FP8_PER_LAYER * l;
        if (f8) PROF(PROF_NORM, wipa_layernorm_fp8(x, d, ln, d, ln_scale, (const float*)lw[0], (const float*)lw[1], M, d, 1e-5f, stream));
        else PROF(PROF_NORM, wipa_layernorm(x, WIPA_F32, d, ln, dt, d, (const float*)lw[0], (const float*)lw[1], M, d, 1e-5f, stream));
        {
            wipa_gemm_desc g;
            memset(&g, 0, sizeof(g));
            g.col_scale_n = 2 * d; g.col_scale = QK_SCALE;
            if (f8) PROF(PROF_GEMM, gemm_f8(ln, d, ln_scale, l8[0], d, (const float*)l8[1], qk, 2 * d, M, 2 * d, d, dt, (const float*)lw[3], 0,
                                            nullptr, stream, &g));
            else PROF(PROF_GEMM, gemm(ln, d, lw[2], d, qk, 2 * d, M, 2 * d, d, dt, dt, (const float*)lw[3], 0, nullptr, stream, &g));
        }
        if (dt == WIPA_BF16) {
            // V^T per clip: swap the operand roles so the GEMM writes [d][t] directly
            wipa_gemm_desc g;
            memset(&g, 0, sizeof(g));
            g.bias_along_m = 1; g.cg_in = T; g.cg_stride = (int64_t)d * T_ENC_PAD;
            if (f8) PROF(PROF_GEMM, gemm_f8(l8[2], d, (const float*)l8[3], ln, d, ln_scale, vb, T_ENC_PAD, d, M, d, dt, (const float*)lw[5], 0,
                                            nullptr, stream, &g));
            else PROF(PROF_GEMM, gemm(lw[4], d, ln, d, vb, T_ENC_PAD, d, M, d, dt, dt, (const float*)lw[5], 0, nullptr, stream, &g));
            PROF(PROF_ATTN, wipa_flash_attn_enc_bf16(qk, 2 * d, vb, T_ENC_PAD, ao, d, B, H, T, stream));
        } else {
            PROF(PROF_GEMM, gemm(ln, d, lw[4], d, vb, d, M, d, d, dt, dt, (const float*)lw[5], 0, nullptr, stream));
            PROF(PROF_ATTN, wipa_flash_attn_enc_f32((const float*)qk, 2 * d, (const float*)qk + d, 2 * d, (const float*)vb, d, (float*)ao, d,
                                            B, H, T, t_f32_split, stream));
        }
        PROF(PROF_GEMM, gemm(ao, d, lw[6], d, x, d, M, d, d, dt, WIPA_F32, (const float*)lw[7], 0, x, stream));
        if (f8) {
            PROF(PROF_NORM, wipa_layernorm_fp8(x, d, ln, d, ln_scale, (const float*)lw[8], (const float*)lw[9], M, d, 1e-5f, stream));
            PROF(PROF_GEMM, gemm_f8(ln, d, ln_scale, l8[4], d, (const float*)l8[5], hb, 4 * d, M, 4 * d, d, dt, (const float*)lw[11], 1, nullptr,
                                    stream));
            void* h8 = qk;                 // [M, 4d] codes in the q|k buffer (M * 2d * 2 bytes)
            float* h_scale = (float*)ao;   // M floats in the attention-output buffer
            PROF(PROF_NORM, wipa_rowquant_fp8(hb, dt, 4 * d, h8, 4 * d, h_scale, M, 4 * d, stream));
            PROF(PROF_GEMM, gemm_f8(h8, 4 * d, h_scale, l8[6], 4 * d, (const float*)l8[7], x, d, M, d, 4 * d, WIPA_F32, (const float*)lw[13], 0,
                                    x, stream));
        } else {
            PROF(PROF_NORM, wipa_layernorm(x, WIPA_F32, d, ln, dt, d, (const float*)lw[8], (const float*)lw[9], M, d, 1e-5f, stream));
            PROF(PROF_GEMM, gemm(ln, d, lw[10], d, hb, 4 * d, M, 4 * d, d, dt, dt, (const float*)lw[11], 1, nullptr, stream));
            PROF(PROF_GEMM, gemm(hb, 4 * d, lw[12], 4 * d, x, d, M, d, 4 * d, dt, WIPA_F32, (const float*)lw[13], 0, x, stream));
        }
    }
    PROF(PROF_NORM, wipa_layernorm(x, WIPA_F32, d, out, dt, d, (const float*)w[5], (const float*)w[6], M, d, 1e-5f, stream));
    return WIPA_OK;
}

// ------------------------------------------------------------------ decoder state
namespace {

constexpr int MAX_SLABS = 4;
struct DecScratch {
    size_t x, x2, ln, q, ao, h, slabs, posd, absorbed, greedy_part, total;
};
// split-K factor of a decode-step residual GEMM: keep >= 3 fragment steps per wave (4 waves)
inline int k_slices_for(int K, int dtype) {
    const int ksteps = K * (int)wipa_dtype_size(dtype) / 64;
    int s = ksteps / 12;
    return s < 1 ? 1 : (s > MAX_SLABS ? MAX_SLABS : s);
}
int decode_mode(const wipa_model_cfg* cfg, int B);  // 0 unfused, 1 fused blocks, 2 unfused + fused cross block (defined below)
constexpr int MAX_PROMPT = 4;  // prompt positions handled by one prefill pass (wipa_decoder_begin takes 1..4 tokens)
DecScratch dec_scratch(const wipa_model_cfg* c, int B) {
    const size_t e = wipa_dtype_size(c->dtype), d = c->n_text_state;
    const size_t R = (size_t)B * MAX_PROMPT;  // rows: one per clip in a decode step, up to four per clip in the prefill
    DecScratch s;
    size_t o = 0;
    s.x = o;    o += align256(R * d * 4);
    s.x2 = o;   o += align256((size_t)B * d * 4);  // fused step: the residual stream ping-pongs between x and x2
    s.ln = o;   o += align256(R * d * e);
    s.q = o;    o += align256(R * d * e);
    s.ao = o;   o += align256(R * d * e);
    s.h = o;    o += align256(R * 4 * d * e);
    // split-K slabs of the unfused residual GEMMs, or one slab per head from the fused self block
    s.slabs = o; o += align256(std::max((size_t)MAX_SLABS * R, (size_t)c->n_text_head * B) * d * 4);
    s.posd = o; o += 256;
    s.absorbed = o;  // Qp + split partials of the absorbed cross-attention (cfg.dec_cross_absorbed)
    if (c->dec_cross_absorbed) o += align256(wipa_cross_absorbed_scratch_bytes(B, (int)d, c->n_audio_ctx));
    s.greedy_part = o;  // arg-max / log-sum-exp partials of the logits projection (wipa_logits_greedy)
    if (wipa_logits_greedy_supported(B, c->n_vocab, (int)d, c->dtype)) o += align256(wipa_logits_greedy_partials_bytes(B));
    s.total = o;
    return s;
}

wipa_dec_layout dec_layout(const wipa_model_cfg* c, int B) {
    const size_t e = wipa_dtype_size(c->dtype);
    const size_t d = c->n_text_state, H = c->n_text_head;
    wipa_dec_layout L;
    size_t o = 0;
    L.ld_tok = c->n_text_ctx + 8;
    L.tokens = o; o += align256((size_t)B * L.ld_tok * 4);
    L.pos = o; o += 256;
    L.not_done = o; o += 256;
    L.sum_logprobs = o; o += align256((size_t)B * 4);
    L.ld_logits = round_up(c->n_vocab, 8);
    L.logits = o; o += align256((size_t)B * L.ld_logits * 4);
    // cached cross K / V of every layer -- or, with absorbed projections, ONE copy of the encoder output [B, n_audio_ctx, d]
    L.cross_kv = o; o += align256(c->dec_cross_absorbed ? (size_t)B * c->n_audio_ctx * d * e
                                                        : (size_t)c->n_text_layer * B * 2 * H * c->n_audio_ctx * 64 * e);
    L.self_kv = o; o += align256((size_t)c->n_text_layer * 3 * B * c->n_text_ctx * d * e);
    L.scratch = o; o += dec_scratch(c, B).total;
    L.total_bytes = o;
    return L;
}

__global__ void advance_pos_kernel(int32_t* pos, int64_t* posd, int d) {
    const int p = *pos + 1;
    *pos = p;
    *posd = (int64_t)p * d;
}

__global__ void fill_tokens_kernel(int32_t* tokens, int64_t ld_tok, int B, int n_init, int i0, int i1, int i2, int i3,
                                   const int32_t* extra) {
    // rows get the prompt; prompts longer than 4 come through `extra` (device copy)
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int first4[4] = {i0, i1, i2, i3};
    for (int t = 0; t < n_init; ++t) tokens[(int64_t)b * ld_tok + t] = (t < 4 || !extra) ? first4[t & 3] : extra[t];
}

// one decoder step (all layers + logits + greedy update + position advance)
// WIPA_ABS_FUSED_PROLOGUE=0 keeps the separate LayerNorm / query GEMM / absorb launches of the absorbed cross block (A/B runs)
bool absorbed_block_fused() {
    const char* e = getenv("WIPA_ABS_FUSED_PROLOGUE");
    return !(e && atoi(e) == 0);
}

// WIPA_ABS_MERGE_OUT=1 folds the cross-attention out projection of the absorbed block into its merge launch (one slab per head,
// wipa_decode_cross_absorbed_block_out).  Built, parity-tested and NOT the default: measured r04 (whisper-small, 64 clips, MI355X)
// the fused launch takes 17.0 us where merge 10.5 + out-projection GEMM 4.9 take 15.4 -- its 192 workgroups pull the head's
// Wo slice (98 KB) on top of the Wv slice through the same per-CU load path and write 12 slabs instead of 2 -- decode step
// 1.338 vs 1.306 ms, pipelined pass 78.2 vs 75.0 ms.  A launch in a replayed graph costs ~1.6 us; 98 KB more per workgroup costs 2.
bool absorbed_merge_out() {
    const char* e = getenv("WIPA_ABS_MERGE_OUT");
    return e && atoi(e) == 1;
}
// WIPA_DECODE_TAIL=0 keeps the separate greedy_step / advance_pos / embed / first-LayerNorm launches (A/B runs, part of the graph
// key); default: ONE launch, wipa_greedy_step_embed, ends a step and prepares the next one's input rows
bool tail_fused() {
    const char* e = getenv("WIPA_DECODE_TAIL");
    return !(e && atoi(e) == 0);
}
// The logits projection with the greedy partials in its epilogue + a tail that merges them (round 4; WIPA_LOGITS_FUSED=0 restores
// the plain GEMM + the row-scanning tail): bf16 models without fp8 decoder tables, <= 64 rows, tail-fused steps.  Measured on
// whisper-small, 64 clips: the logits' 64-byte stores were 8 of the GEMM's 32 us, the tail's read-back of the 13 MB another ~10.
// A step whose logits nobody reads (every step of a wipa_decoder_run call but the last) does not write them at all: t_lean_logits.
thread_local bool t_lean_logits = false;
bool logits_fused(const wipa_model_cfg* cfg, int B) {
    const char* e = getenv("WIPA_LOGITS_FUSED");  // read per call like the other step variants: part of the graph key
    const bool on = !(e && atoi(e) == 0);
    return on && tail_fused() && cfg->dec_w_dtype == 0 && wipa_logits_greedy_supported(B, cfg->n_vocab, cfg->n_text_state, cfg->dtype);
}
int32_t* done_counter_of(char* st, const wipa_dec_layout& L) { return (int32_t*)(st + L.pos + 64); }  // zeroed with pos by wipa_decoder_begin

// x = embedding of the token at the current position, ln = first block's LayerNorm of it: what a tail-fused step expects to find
int enqueue_step_head(const wipa_model_cfg* cfg, const void* const* w, char* st, const wipa_dec_layout& L, int B, wipa_stream_t stream) {
    const DecScratch S = dec_scratch(cfg, B);
    char* sc = st + L.scratch;
    const void* const* lw0 = w + WIPA_DEC_GLOBAL;
    return wipa_embed_layernorm((const int32_t*)(st + L.tokens), L.ld_tok, B, (const int32_t*)(st + L.pos), w[0], emb_dtype(cfg), emb_scale(cfg, w),
                                (const float*)w[1], cfg->n_text_ctx, (float*)(sc + S.x), (const float*)lw0[0], (const float*)lw0[1], sc + S.ln,
                                cfg->dtype, cfg->n_text_state, 1e-5f, stream);
}

int enqueue_step(const wipa_model_cfg* cfg, const void* const* w, char* st, const wipa_dec_layout& L, int B, int n_init,
                 int eot, const float* mask_first, const float* mask_always, wipa_stream_t stream) {
    const int dt = cfg->dtype;
    const bool tail = tail_fused();
    const size_t e = wipa_dtype_size(dt);
    const int d = cfg->n_text_state, H = cfg->n_text_head, nctx = cfg->n_text_ctx, Ta = cfg->n_audio_ctx;
    const DecScratch S = dec_scratch(cfg, B);
    char* sc = st + L.scratch;
    float* x = (float*)(sc + S.x);
    float* x_other = (float*)(sc + S.x2);
    const bool absorbed = cfg->dec_cross_absorbed != 0;
    const bool cross_fused = !absorbed && decode_mode(cfg, B) == 2;
    void* ln = sc + S.ln;
    void* q = sc + S.q;
    void* ao = sc + S.ao;
    void* hb = sc + S.h;
    int64_t* posd = (int64_t*)(sc + S.posd);
    int32_t* tokens = (int32_t*)(st + L.tokens);
    int32_t* pos = (int32_t*)(st + L.pos);
    // Residual GEMMs (out, cross.out, mlp2) are split along K over workgroups; each slice writes a
    // partial slab and the NEXT LayerNorm kernel adds the slabs to x in a fixed order first
    // (deterministic split-K, no atomics).  `pend` = slabs waiting to be folded into x.
    float* slabs = (float*)(sc + S.slabs);
    const int64_t slab_stride = (int64_t)B * d;
    int pend = 0;
    auto ln_step = [&](const void* lw_w, const void* lw_b) -> int {
        const int rc = wipa_add_slabs_layernorm(x, d, slabs, pend, slab_stride, ln, dt, d, (const float*)lw_w, (const float*)lw_b,
                                                B, d, 1e-5f, stream);
        pend = 0;
        return rc;
    };
    auto residual_gemm = [&](const void* A, int K, const void* W, const void* bias) -> int {
        wipa_gemm_desc g;
        memset(&g, 0, sizeof(g));
        g.k_slices = B <= 256 ? k_slices_for(K, dt) : 1;  // split-K lives in the weight-streaming (M <= 256) kernel
        g.slab_stride = slab_stride;
        pend = g.k_slices;
        return gemm(A, K, W, K, slabs, d, B, d, K, dt, WIPA_F32, (const float*)bias, 0, nullptr, stream, &g);
    };
    // tail-fused steps find x (this position's embedding) and ln (the first block's LayerNorm of it) in place: written by the
    // previous step's last launch, or by enqueue_step_head before the first step of a wipa_decoder_run call
    float* const x_first = x;
    if (!tail) RT_CALL(wipa_embed_tokens(tokens, L.ld_tok, B, 1, 0, pos, w[0], emb_dtype(cfg), emb_scale(cfg, w), (const float*)w[1], x, d, stream));
    for (int l = 0; l < cfg->n_text_layer; ++l) {
        const void* const* lw = w + WIPA_DEC_GLOBAL + WIPA_DEC_PER_LAYER * l;
        char* skv = st + L.self_kv + (size_t)l * 3 * B * nctx * d * e;  // [3][B][nctx][d]
        char* ckv = st + L.cross_kv + (size_t)l * B * 2 * H * Ta * 64 * e;
        bool cross_out_done = false;
        if (!(tail && l == 0)) RT_CALL(ln_step(lw[0], lw[1]));
        {
            // q|k|v of this position -> slot[n / d][b][pos][n % d]
            wipa_gemm_desc g;
            memset(&g, 0, sizeof(g));
            g.col_scale_n = 2 * d; g.col_scale = QK_SCALE;
            g.rg_in = 1; g.rg_valid = 1; g.rg_stride = (int64_t)nctx * d;
            g.cg_in = d; g.cg_stride = (int64_t)B * nctx * d;
            g.c_offset_dev = posd;
            RT_CALL(gemm(ln, d, lw[2], d, skv, d, B, 3 * d, d, dt, dt, (const float*)lw[3], 0, nullptr, stream, &g));
        }
        {
            wipa_attn_desc a;
            memset(&a, 0, sizeof(a));
            const size_t slot = (size_t)B * nctx * d * e;
            a.q = skv; a.k = skv + slot; a.v = skv + 2 * slot; a.out = ao;
            a.q_row_dev = pos; a.tk_dev = pos;
            a.q_bs = (int64_t)nctx * d; a.q_rs = d; a.q_hs = 64;
            a.k_bs = a.q_bs; a.k_rs = d; a.k_hs = 64;
            a.v_bs = a.q_bs; a.v_rs = d; a.v_hs = 64;
            a.o_bs = d; a.o_rs = d; a.o_hs = 64;
            a.B = B; a.H = H; a.Tq = 1; a.Tk = 1; a.causal = 0; a.dtype = dt;
            RT_CALL(wipa_decode_attn(&a, stream));
        }
        RT_CALL(residual_gemm(ao, d, lw[4], lw[5]));
        if (cross_fused) {
            // one launch for [slab sum + residual + cross_attn_ln + cross query + cross-attention]; the residual row moves to
            // the other buffer (H workgroups read the row that one of them writes)
            wipa_cross_block_desc c;
            memset(&c, 0, sizeof(c));
            c.x_in = x; c.x_out = x_other; c.slabs = slabs; c.bias_o = nullptr;  // slab 0 carries the out-projection bias
            c.ln_w = (const float*)lw[6]; c.ln_b = (const float*)lw[7]; c.wq = lw[8]; c.bq = (const float*)lw[9];
            c.kv = ckv; c.out = ao; c.slab_stride = slab_stride;
            c.n_slabs = pend; c.B = B; c.d = d; c.H = H; c.Tk = Ta; c.dtype = dt; c.eps = 1e-5f; c.qk_scale = QK_SCALE;
            RT_CALL(wipa_decode_cross_block(&c, stream));
            pend = 0;
            std::swap(x, x_other);
        } else if (absorbed && absorbed_block_fused() && pend <= 4) {
            // three launches: [slab sum + residual + cross_attn_ln + cross query + absorbed query] -> stream over the encoder output
            // -> [merge + value projection]; the residual row moves to the other buffer like in the fused cross block
            const void* wkT = w[WIPA_DEC_GLOBAL + WIPA_DEC_PER_LAYER * cfg->n_text_layer + WIPA_DEC_ABSORBED_PER_LAYER * l];
            wipa_cross_block_desc c;
            memset(&c, 0, sizeof(c));
            c.x_in = x; c.x_out = x_other; c.slabs = slabs; c.bias_o = nullptr;
            c.ln_w = (const float*)lw[6]; c.ln_b = (const float*)lw[7]; c.wq = lw[8]; c.bq = (const float*)lw[9];
            c.kv = st + L.cross_kv; c.out = ao; c.slab_stride = slab_stride;
            c.n_slabs = pend; c.B = B; c.d = d; c.H = H; c.Tk = Ta; c.dtype = dt; c.eps = 1e-5f; c.qk_scale = QK_SCALE;
            c.cross_splits = cfg->dec_cross_splits;
            if (absorbed_merge_out()) {
                // ... and the cross-attention OUT projection rides in the third launch: one slab per head, summed by the mlp LayerNorm
                RT_CALL(wipa_decode_cross_absorbed_block_out(&c, wkT, (const char*)lw[10] + (size_t)d * d * e, (const float*)lw[11] + d,
                                                             lw[12], (const float*)lw[13], slabs, slab_stride, sc + S.absorbed,
                                                             S.greedy_part - S.absorbed, stream));
                pend = H;
                std::swap(x, x_other);
                cross_out_done = true;
            } else {
                RT_CALL(wipa_decode_cross_absorbed_block(&c, wkT, (const char*)lw[10] + (size_t)d * d * e, (const float*)lw[11] + d,
                                                         sc + S.absorbed, S.greedy_part - S.absorbed, stream));
                pend = 0;
                std::swap(x, x_other);
            }
        } else {
            RT_CALL(ln_step(lw[6], lw[7]));
            {
                wipa_gemm_desc g;
                memset(&g, 0, sizeof(g));
                g.col_scale_n = d; g.col_scale = QK_SCALE;
                RT_CALL(gemm(ln, d, lw[8], d, q, d, B, d, d, dt, dt, (const float*)lw[9], 0, nullptr, stream, &g));
            }
            if (absorbed) {
                // scores and values from ONE pass over the encoder output: Wk absorbed into the query, Wv into the output
                const void* wkT = w[WIPA_DEC_GLOBAL + WIPA_DEC_PER_LAYER * cfg->n_text_layer + WIPA_DEC_ABSORBED_PER_LAYER * l];
                RT_CALL(wipa_cross_absorbed_attention(q, d, wkT, st + L.cross_kv, (const char*)lw[10] + (size_t)d * d * e,
                                                      (const float*)lw[11] + d, ao, d, sc + S.absorbed, S.greedy_part - S.absorbed, B, H, d, Ta,
                                                      QK_SCALE, cfg->dec_cross_splits, stream));
            } else {
                RT_CALL(wipa_decode_cross_attn(q, ckv, ao, B, H, Ta, dt, stream));
            }
        }
        if (!cross_out_done) RT_CALL(residual_gemm(ao, d, lw[12], lw[13]));
        RT_CALL(ln_step(lw[14], lw[15]));
        RT_CALL(gemm(ln, d, lw[16], d, hb, 4 * d, B, 4 * d, d, dt, dt, (const float*)lw[17], 1, nullptr, stream));
        RT_CALL(residual_gemm(hb, 4 * d, lw[18], lw[19]));
    }
    RT_CALL(ln_step(w[2], w[3]));
    float* logits = (float*)(st + L.logits);
    if (logits_fused(cfg, B)) {
        const void* const* lw0 = w + WIPA_DEC_GLOBAL;
        float* part = (float*)(sc + S.greedy_part);
        RT_CALL(wipa_logits_greedy(ln, d, w[0], d, t_lean_logits ? nullptr : logits, L.ld_logits, B, cfg->n_vocab, d, mask_first, mask_always, pos,
                                   n_init, part, wipa_logits_greedy_partials_bytes(B), stream));
        RT_CALL(wipa_greedy_step_embed_partials(part, WIPA_GREEDY_PARTS, B, tokens, L.ld_tok, pos, posd, done_counter_of(st, L), n_init, eot,
                                                (float*)(st + L.sum_logprobs), (int32_t*)(st + L.not_done), w[0], emb_dtype(cfg), emb_scale(cfg, w),
                                                (const float*)w[1], nctx, x_first, (const float*)lw0[0], (const float*)lw0[1], ln, dt, d, 1e-5f,
                                                stream));
        return WIPA_OK;
    }
    RT_CALL(gemm(ln, d, w[0], d, logits, L.ld_logits, B, cfg->n_vocab, d, dt, WIPA_F32, nullptr, 0, nullptr, stream));
    if (tail) {
        // greedy update + embedding of the chosen token + first LayerNorm of the NEXT position + position advance: one launch
        const void* const* lw0 = w + WIPA_DEC_GLOBAL;
        RT_CALL(wipa_greedy_step_embed(logits, L.ld_logits, B, cfg->n_vocab, mask_first, mask_always, tokens, L.ld_tok, pos, posd,
                                       done_counter_of(st, L), n_init, eot, (float*)(st + L.sum_logprobs), (int32_t*)(st + L.not_done), w[0],
                                       emb_dtype(cfg), emb_scale(cfg, w), (const float*)w[1], nctx, x_first, (const float*)lw0[0],
                                       (const float*)lw0[1], ln, dt, d, 1e-5f, stream));
        return WIPA_OK;
    }
    RT_CALL(wipa_greedy_step(logits, L.ld_logits, B, cfg->n_vocab, mask_first, mask_always, tokens, L.ld_tok, pos, n_init,
                             eot, (float*)(st + L.sum_logprobs), (int32_t*)(st + L.not_done), stream));
    hipLaunchKernelGGL(advance_pos_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, pos, posd, d);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

// One decoder step with the fused blocks of decode_fused.hip: embed + 5 launches per layer + final LayerNorm + logits +
// greedy update + position advance (65 launches for whisper-small instead of 137).  Same state transitions as enqueue_step.
int enqueue_step_fused(const wipa_model_cfg* cfg, const void* const* w, char* st, const wipa_dec_layout& L, int B, int n_init,
                       int eot, const float* mask_first, const float* mask_always, wipa_stream_t stream) {
    const int dt = cfg->dtype;
    const size_t e = wipa_dtype_size(dt);
    const int d = cfg->n_text_state, H = cfg->n_text_head, nctx = cfg->n_text_ctx, Ta = cfg->n_audio_ctx;
    const DecScratch S = dec_scratch(cfg, B);
    char* sc = st + L.scratch;
    float* xa = (float*)(sc + S.x);
    float* xb = (float*)(sc + S.x2);
    void* ln = sc + S.ln;
    void* ao = sc + S.ao;
    void* hb = sc + S.h;
    float* slabs = (float*)(sc + S.slabs);
    int64_t* posd = (int64_t*)(sc + S.posd);
    int32_t* tokens = (int32_t*)(st + L.tokens);
    int32_t* pos = (int32_t*)(st + L.pos);
    RT_CALL(wipa_embed_tokens(tokens, L.ld_tok, B, 1, 0, pos, w[0], emb_dtype(cfg), emb_scale(cfg, w), (const float*)w[1], xa, d, stream));
    float* cur = xa;
    float* other = xb;
    for (int l = 0; l < cfg->n_text_layer; ++l) {
        const void* const* lw = w + WIPA_DEC_GLOBAL + WIPA_DEC_PER_LAYER * l;
        char* skv = st + L.self_kv + (size_t)l * 3 * B * nctx * d * e;  // [3][B][nctx][d]: slot 0 unused here, 1 K, 2 V
        char* ckv = st + L.cross_kv + (size_t)l * B * 2 * H * Ta * 64 * e;
        const size_t slot = (size_t)B * nctx * d * e;
        {
            wipa_self_block_desc a;
            memset(&a, 0, sizeof(a));
            a.x = cur; a.ln_w = (const float*)lw[0]; a.ln_b = (const float*)lw[1];
            a.wqkv = lw[2]; a.bqkv = (const float*)lw[3]; a.wo = lw[4];
            a.kcache = skv + slot; a.vcache = skv + 2 * slot; a.pos = pos; a.slabs = slabs;
            a.kv_batch_stride = (int64_t)nctx * d; a.slab_stride = (int64_t)B * d;
            a.B = B; a.d = d; a.H = H; a.dtype = dt; a.eps = 1e-5f; a.qk_scale = QK_SCALE;
            RT_CALL(wipa_decode_self_block(&a, stream));
        }
        {
            wipa_cross_block_desc c;
            memset(&c, 0, sizeof(c));
            c.x_in = cur; c.x_out = other; c.slabs = slabs; c.bias_o = (const float*)lw[5];
            c.ln_w = (const float*)lw[6]; c.ln_b = (const float*)lw[7]; c.wq = lw[8]; c.bq = (const float*)lw[9];
            c.kv = ckv; c.out = ao; c.slab_stride = (int64_t)B * d;
            c.n_slabs = H; c.B = B; c.d = d; c.H = H; c.Tk = Ta; c.dtype = dt; c.eps = 1e-5f; c.qk_scale = QK_SCALE;
            RT_CALL(wipa_decode_cross_block(&c, stream));
        }
        std::swap(cur, other);
        // cross out projection and the MLP update the residual rows in place (no split-K: one launch each)
        RT_CALL(gemm(ao, d, lw[12], d, cur, d, B, d, d, dt, WIPA_F32, (const float*)lw[13], 0, cur, stream));
        {
            wipa_gemm_desc g;
            memset(&g, 0, sizeof(g));
            g.ln_x = cur; g.ln_ldx = d; g.ln_w = (const float*)lw[14]; g.ln_b = (const float*)lw[15]; g.ln_eps = 1e-5f;
            RT_CALL(gemm(nullptr, d, lw[16], d, hb, 4 * d, B, 4 * d, d, dt, dt, (const float*)lw[17], 1, nullptr, stream, &g));
        }
        RT_CALL(gemm(hb, 4 * d, lw[18], 4 * d, cur, d, B, d, 4 * d, dt, WIPA_F32, (const float*)lw[19], 0, cur, stream));
    }
    RT_CALL(wipa_add_slabs_layernorm(cur, d, nullptr, 0, 0, ln, dt, d, (const float*)w[2], (const float*)w[3], B, d, 1e-5f, stream));
    float* logits = (float*)(st + L.logits);
    RT_CALL(gemm(ln, d, w[0], d, logits, L.ld_logits, B, cfg->n_vocab, d, dt, WIPA_F32, nullptr, 0, nullptr, stream));
    RT_CALL(wipa_greedy_step(logits, L.ld_logits, B, cfg->n_vocab, mask_first, mask_always, tokens, L.ld_tok, pos, n_init,
                             eot, (float*)(st + L.sum_logprobs), (int32_t*)(st + L.not_done), stream));
    hipLaunchKernelGGL(advance_pos_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, pos, posd, d);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

// Decode-step variants, WIPA_DECODE_FUSED (read at enqueue / capture time, part of the graph key):
//   2 (default)  the unfused step with ONE fusion: [split-K slab sum + residual + cross_attn_ln + cross query + cross-attention]
//                in a single launch (wipa_decode_cross_block, query weights prefetched under the prologue): 9 launches per
//                layer, 1.311 ms per step against 1.352 ms (whisper-small bf16, 64 rows, MI355X, r02);
//   0            the unfused step, 11 launches per layer;
//   1            every fused block (5 launches per layer).  NOT the default: it measured 2.05 ms per step
//                (profiles/r02_fused_step_by_kernel.txt) -- a fused block is a chain of dependent memory round trips (LayerNorm
//                rows -> weight fragments -> cache rows -> weight fragments) that costs more than the ~4.5 us launch gaps it
//                removes: self block 34.9 us vs 23.4 us for the four launches it replaces, LayerNorm-prologue mlp1 37.2 us vs
//                12 us, and without split-K the mlp2 projection has 48 workgroups to stream 4.7 MB (19.7 us vs 7.4 us).
int decode_mode(const wipa_model_cfg* cfg, int B) {
    const char* e = getenv("WIPA_DECODE_FUSED");
    const int m = e ? atoi(e) : 2;
    // the fused blocks read their weights as T: fp8 tables stay on the unfused step
    if (m == 0 || cfg->dec_w_dtype != 0 || cfg->n_text_state > 1280 || cfg->n_text_head > 20 || B > 65535) return 0;
    return m == 1 ? 1 : 2;
}
bool use_fused_step(const wipa_model_cfg* cfg, int B) { return !cfg->dec_cross_absorbed && decode_mode(cfg, B) == 1; }
// whether the steps of this configuration end with the fused tail (and therefore need enqueue_step_head before the first one)
bool step_needs_head(const wipa_model_cfg* cfg, int B) { return !use_fused_step(cfg, B) && tail_fused(); }

int enqueue_decode_step(const wipa_model_cfg* cfg, const void* const* w, char* st, const wipa_dec_layout& L, int B, int n_init,
                        int eot, const float* mask_first, const float* mask_always, wipa_stream_t stream) {
    if (use_fused_step(cfg, B)) return enqueue_step_fused(cfg, w, st, L, B, n_init, eot, mask_first, mask_always, stream);
    return enqueue_step(cfg, w, st, L, B, n_init, eot, mask_first, mask_always, stream);
}

__global__ void set_pos_kernel(int32_t* pos, int64_t* posd, int value, int d) {
    *pos = value;
    *posd = (int64_t)value * d;
}

// The first n_init decoder steps as ONE batched pass over the prompt (rows (b, t), t < n_init): mlx_whisper runs the
// prompt through the decoder in one forward too.  Self-K/V of positions 0..n_init-1 land in the cache, the cross K/V are
// streamed once for all prompt positions (decode_attn_multi_kernel), logits are computed for the last position only, and
// the greedy update writes token n_init.  Afterwards the state equals that after n_init calls of enqueue_step.
int enqueue_prefill(const wipa_model_cfg* cfg, const void* const* w, char* st, const wipa_dec_layout& L, int B, int n_init,
                    int eot, const float* mask_first, const float* mask_always, wipa_stream_t stream) {
    const int dt = cfg->dtype;
    const size_t e = wipa_dtype_size(dt);
    const int d = cfg->n_text_state, H = cfg->n_text_head, nctx = cfg->n_text_ctx, Ta = cfg->n_audio_ctx;
    const int P = n_init, M = B * P;
    const DecScratch S = dec_scratch(cfg, B);
    char* sc = st + L.scratch;
    float* x = (float*)(sc + S.x);
    char* ln = sc + S.ln;
    void* q = sc + S.q;
    void* ao = sc + S.ao;
    void* hb = sc + S.h;
    int64_t* posd = (int64_t*)(sc + S.posd);
    int32_t* tokens = (int32_t*)(st + L.tokens);
    int32_t* pos = (int32_t*)(st + L.pos);
    float* slabs = (float*)(sc + S.slabs);
    const int64_t slab_stride = (int64_t)M * d;
    int pend = 0;
    auto ln_step = [&](const void* lw_w, const void* lw_b) -> int {
        const int rc = wipa_add_slabs_layernorm(x, d, slabs, pend, slab_stride, ln, dt, d, (const float*)lw_w, (const float*)lw_b,
                                                M, d, 1e-5f, stream);
        pend = 0;
        return rc;
    };
    auto residual_gemm = [&](const void* A, int K, const void* W, const void* bias) -> int {
        wipa_gemm_desc g;
        memset(&g, 0, sizeof(g));
        g.stream_weights = 1;
        g.k_slices = M <= 1024 ? k_slices_for(K, dt) : 1;
        g.slab_stride = slab_stride;
        pend = g.k_slices;
        return gemm(A, K, W, K, slabs, d, M, d, K, dt, WIPA_F32, (const float*)bias, 0, nullptr, stream, &g);
    };
    RT_CALL(wipa_embed_tokens(tokens, L.ld_tok, B, P, 0, nullptr, w[0], emb_dtype(cfg), emb_scale(cfg, w), (const float*)w[1], x, d, stream));
    for (int l = 0; l < cfg->n_text_layer; ++l) {
        const void* const* lw = w + WIPA_DEC_GLOBAL + WIPA_DEC_PER_LAYER * l;
        char* skv = st + L.self_kv + (size_t)l * 3 * B * nctx * d * e;  // [3][B][nctx][d]
        char* ckv = st + L.cross_kv + (size_t)l * B * 2 * H * Ta * 64 * e;
        RT_CALL(ln_step(lw[0], lw[1]));
        {
            // q|k|v of positions 0..P-1 -> slot[n / d][b][t][n % d]
            wipa_gemm_desc g;
            memset(&g, 0, sizeof(g));
        g.stream_weights = 1;
            g.col_scale_n = 2 * d; g.col_scale = QK_SCALE;
            g.rg_in = P; g.rg_valid = P; g.rg_stride = (int64_t)nctx * d;
            g.cg_in = d; g.cg_stride = (int64_t)B * nctx * d;
            RT_CALL(gemm(ln, d, lw[2], d, skv, d, M, 3 * d, d, dt, dt, (const float*)lw[3], 0, nullptr, stream, &g));
        }
        {
            wipa_attn_desc a;
            memset(&a, 0, sizeof(a));
            const size_t slot = (size_t)B * nctx * d * e;
            a.q = skv; a.k = skv + slot; a.v = skv + 2 * slot; a.out = ao;
            a.q_bs = (int64_t)nctx * d; a.q_rs = d; a.q_hs = 64;
            a.k_bs = a.q_bs; a.k_rs = d; a.k_hs = 64;
            a.v_bs = a.q_bs; a.v_rs = d; a.v_hs = 64;
            a.o_bs = (int64_t)P * d; a.o_rs = d; a.o_hs = 64;
            a.B = B; a.H = H; a.Tq = P; a.Tk = P; a.causal = 1; a.dtype = dt;
            RT_CALL(wipa_attention(&a, stream));
        }
        RT_CALL(residual_gemm(ao, d, lw[4], lw[5]));
        RT_CALL(ln_step(lw[6], lw[7]));
        {
            wipa_gemm_desc g;
            memset(&g, 0, sizeof(g));
        g.stream_weights = 1;
            g.col_scale_n = d; g.col_scale = QK_SCALE;
            RT_CALL(gemm(ln, d, lw[8], d, q, d, M, d, d, dt, dt, (const float*)lw[9], 0, nullptr, stream, &g));
        }
        if (cfg->dec_cross_absorbed) {
            // one absorbed pass per prompt position: rows (b, t) of q / ao with row stride P * d
            const void* wkT = w[WIPA_DEC_GLOBAL + WIPA_DEC_PER_LAYER * cfg->n_text_layer + WIPA_DEC_ABSORBED_PER_LAYER * l];
            for (int t = 0; t < P; ++t)
                RT_CALL(wipa_cross_absorbed_attention((const char*)q + (size_t)t * d * e, (int64_t)P * d, wkT, st + L.cross_kv,
                                                      (const char*)lw[10] + (size_t)d * d * e, (const float*)lw[11] + d,
                                                      (char*)ao + (size_t)t * d * e, (int64_t)P * d, sc + S.absorbed, S.greedy_part - S.absorbed, B,
                                                      H, d, Ta, QK_SCALE, cfg->dec_cross_splits, stream));
        } else {
            RT_CALL(wipa_decode_cross_attn_multi(q, ckv, ao, B, H, Ta, P, dt, stream));
        }
        RT_CALL(residual_gemm(ao, d, lw[12], lw[13]));
        RT_CALL(ln_step(lw[14], lw[15]));
        {
            wipa_gemm_desc g;
            memset(&g, 0, sizeof(g));
            g.stream_weights = 1;
            RT_CALL(gemm(ln, d, lw[16], d, hb, 4 * d, M, 4 * d, d, dt, dt, (const float*)lw[17], 1, nullptr, stream, &g));
        }
        RT_CALL(residual_gemm(hb, 4 * d, lw[18], lw[19]));
    }
    RT_CALL(ln_step(w[2], w[3]));
    float* logits = (float*)(st + L.logits);
    // logits of the LAST prompt position only: rows (b, P-1) of ln, i.e. row stride P*d
    RT_CALL(gemm(ln + (size_t)(P - 1) * d * e, (int64_t)P * d, w[0], d, logits, L.ld_logits, B, cfg->n_vocab, d, dt, WIPA_F32, nullptr, 0,
                 nullptr, stream));
    hipLaunchKernelGGL(set_pos_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, pos, posd, P - 1, d);
    RT_CALL(wipa_greedy_step(logits, L.ld_logits, B, cfg->n_vocab, mask_first, mask_always, tokens, L.ld_tok, pos, n_init,
                             eot, (float*)(st + L.sum_logprobs), (int32_t*)(st + L.not_done), stream));
    hipLaunchKernelGGL(advance_pos_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, pos, posd, d);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

// Hardware-counter collection (rocprofv3 --pmc sets ROCPROF_COUNTER_COLLECTION) serialises every dispatch and intercepts the
// queues; a captured step replayed under it hung once in round 1 and no artefact of that run survives.  CANDIDATE cause (the only
// one the code ever offered, not a confirmed one): before round 2 the once-only hipFuncSetAttribute pass ran inside the first
// wipa_gemm call, i.e. INSIDE the relaxed capture whenever a decode step was the first GEMM of the process -- exactly the shape
// of a counter micro-target.  Since round 2 every attribute call sits behind wipa_gemm_init / wipa_decode_fused_init /
// wipa_cross_absorbed_init (each a once-per-process pass), which the decode entry points call BEFORE they begin a capture, so
// nothing but kernel launches is issued between hipStreamBeginCapture and hipStreamEndCapture; with that change the one bounded
// confirmation run (WIPA_DECODE_GRAPH=force under --pmc, tools/pmc_graph_confirm.py, profiles/r03_graph_replay_under_counters.txt)
// captured and replayed a step with counters attached and reproduced the eager ids.  The eager-under-counters DEFAULT stays for a
// different reason: a per-kernel counter run wants the dispatches separate anyway, and graphs_allowed() prints one line when it
// takes that path so that PMC numbers are not mistaken for graph-replay numbers.  WIPA_DECODE_GRAPH=0 asks for the eager path
// explicitly (any other queue-intercepting tool), =force keeps the graphs under counters.  Kernel tracing alone keeps the graphs.
bool counters_attached() {
    const char* e = getenv("ROCPROF_COUNTER_COLLECTION");
    return e && *e && strcmp(e, "0") != 0 && strcasecmp(e, "false") != 0;
}
bool graphs_allowed() {
    const char* g = getenv("WIPA_DECODE_GRAPH");
    if (g && (strcmp(g, "0") == 0 || strcasecmp(g, "off") == 0)) return false;
    if (g && strcasecmp(g, "force") == 0) return true;
    if (counters_attached()) {
        // say so once: numbers collected under --pmc are those of the EAGER step (same kernels, host-enqueued), not of graph replay
        static std::once_flag once;
        std::call_once(once, [] {
            fprintf(stderr, "libwipa: hardware counters attached (ROCPROF_COUNTER_COLLECTION): decode steps are enqueued eagerly, not "
                            "replayed from the captured graph (WIPA_DECODE_GRAPH=force keeps the graphs)\n");
        });
        return false;
    }
    return true;
}
// every kernel attribute the step needs, set outside any capture
int init_before_capture(const wipa_model_cfg* cfg) {
    RT_CALL(wipa_decode_fused_init());
    RT_CALL(wipa_gemm_init());
    if (cfg && cfg->dec_cross_absorbed) RT_CALL(wipa_cross_absorbed_init(cfg->n_text_state));
    return WIPA_OK;
}

// The blob layout depends on the configuration (dtype, absorbed / cached cross-attention, fp8 tables): a blob allocated for one
// cfg and handed in with another would be written far past its end.  Every entry point checks the caller's size.
int state_fits(const char* who, const wipa_dec_layout& L, size_t state_bytes) {
    WIPA_REQUIRE((int64_t)state_bytes >= L.total_bytes,
                 "%s: the state blob holds %zu bytes, this configuration needs %lld (wipa_decoder_layout); re-allocate it after a "
                 "change of dtype / cross-attention form / weight format", who, state_bytes, (long long)L.total_bytes);
    return WIPA_OK;
}

// graph cache: one captured step per (state blob, weights, masks, shape)
// The key carries cfg->weights_generation: the host address of a weight table can be reused by a NEW table after the old
// one was freed, so the address alone does not identify the device pointers baked into a captured graph.
typedef std::tuple<const void*, const void*, const void*, const void*, int, int, int, int, int, int> GraphKey;  // ..., generation, kind: 0 step, 1 prefill, 2 lean step (no logit stores)
std::mutex g_graph_mu;
std::map<GraphKey, hipGraphExec_t> g_graphs;

}  // namespace

extern "C" int wipa_profile_begin(wipa_stream_t stream) {
    WIPA_REQUIRE(!t_prof, "wipa_profile_begin: a profile is already open on this thread");
    t_prof = new Profiler{(hipStream_t)stream, {}};
    return WIPA_OK;
}

extern "C" int wipa_profile_end(float* ms_by_class, int* launches_by_class) {
    WIPA_REQUIRE(t_prof, "wipa_profile_end: no open profile");
    Profiler* p = t_prof;
    t_prof = nullptr;
    const hipError_t se = hipStreamSynchronize(p->stream);
    for (int c = 0; c < PROF_CLASSES; ++c) {
        if (ms_by_class) ms_by_class[c] = 0.f;
        if (launches_by_class) launches_by_class[c] = 0;
    }
    for (auto& sp : p->spans) {
        float ms = 0.f;
        if (se == hipSuccess && hipEventElapsedTime(&ms, std::get<1>(sp), std::get<2>(sp)) == hipSuccess) {
            if (ms_by_class) ms_by_class[std::get<0>(sp)] += ms;
            if (launches_by_class) launches_by_class[std::get<0>(sp)] += 1;
        }
        hipEventDestroy(std::get<1>(sp));
        hipEventDestroy(std::get<2>(sp));
    }
    delete p;
    WIPA_CHECK_HIP(se);
    return WIPA_OK;
}

extern "C" int wipa_decoder_layout(const wipa_model_cfg* cfg, int B, wipa_dec_layout* out) {
    RT_CALL(cfg_check(cfg));
    SplitScope split_scope(cfg);
    WIPA_REQUIRE(out && B > 0, "wipa_decoder_layout: bad arguments");
    *out = dec_layout(cfg, B);
    return WIPA_OK;
}

extern "C" int wipa_decoder_set_audio(const wipa_model_cfg* cfg, const void* const* w, const void* features, void* state,
                                      size_t state_bytes, int B, wipa_stream_t stream) {
    RT_CALL(cfg_check(cfg));
    SplitScope split_scope(cfg);
    WIPA_REQUIRE(w && features && state && B > 0, "wipa_decoder_set_audio: null pointer / bad batch");
    const wipa_dec_layout L = dec_layout(cfg, B);
    RT_CALL(state_fits("wipa_decoder_set_audio", L, state_bytes));
    const int dt = cfg->dtype;
    const size_t e = wipa_dtype_size(dt);
    const int d = cfg->n_text_state, H = cfg->n_text_head, Ta = cfg->n_audio_ctx;
    WIPA_REQUIRE(cfg->n_audio_state == d, "encoder/decoder widths differ");
    if (cfg->dec_cross_absorbed) {
        // no key / value projection and no cache: the decode steps stream the encoder output itself (csrc/cross_absorbed.hip)
        WIPA_CHECK_HIP(hipMemcpyAsync((char*)state + L.cross_kv, features, (size_t)B * Ta * d * e, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return WIPA_OK;
    }
    for (int l = 0; l < cfg->n_text_layer; ++l) {
        const void* const* lw = w + WIPA_DEC_GLOBAL + WIPA_DEC_PER_LAYER * l;
        char* ckv = (char*)state + L.cross_kv + (size_t)l * B * 2 * H * Ta * 64 * e;
        wipa_gemm_desc g;
        memset(&g, 0, sizeof(g));
        g.col_scale_n = d; g.col_scale = QK_SCALE;  // the key half; values stay unscaled
        g.rg_in = Ta; g.rg_valid = Ta; g.rg_stride = (int64_t)2 * H * Ta * 64;
        g.cg_in = 64; g.cg_stride = (int64_t)Ta * 64;
        PROF(PROF_GEMM, gemm(features, d, lw[10], d, ckv, 64, B * Ta, 2 * d, d, dt, dt, (const float*)lw[11], 0, nullptr, stream, &g));
    }
    return WIPA_OK;
}

extern "C" int wipa_decoder_begin(const wipa_model_cfg* cfg, void* state, size_t state_bytes, int B,
                                  const int32_t* initial_tokens_host, int n_init, wipa_stream_t stream) {
    RT_CALL(cfg_check(cfg));
    SplitScope split_scope(cfg);
    WIPA_REQUIRE(state && initial_tokens_host && n_init >= 1 && n_init <= 4 && B > 0,
                 "wipa_decoder_begin: need 1..4 prompt tokens (got %d)", n_init);
    const wipa_dec_layout L = dec_layout(cfg, B);
    RT_CALL(state_fits("wipa_decoder_begin", L, state_bytes));
    hipStream_t s = (hipStream_t)stream;
    char* st = (char*)state;
    const DecScratch S = dec_scratch(cfg, B);
    RT_CALL(wipa_decode_fused_init());  // kernel attributes are set here, outside the stream capture of the step
    RT_CALL(wipa_gemm_init());
    WIPA_CHECK_HIP(hipMemsetAsync(st + L.tokens, 0, (size_t)B * L.ld_tok * 4, s));
    WIPA_CHECK_HIP(hipMemsetAsync(st + L.pos, 0, 512, s));  // pos and not_done
    WIPA_CHECK_HIP(hipMemsetAsync(st + L.sum_logprobs, 0, (size_t)B * 4, s));
    WIPA_CHECK_HIP(hipMemsetAsync(st + L.scratch + S.posd, 0, 8, s));
    int t4[4] = {0, 0, 0, 0};
    for (int i = 0; i < n_init; ++i) t4[i] = initial_tokens_host[i];
    hipLaunchKernelGGL(fill_tokens_kernel, dim3((B + 63) / 64), dim3(64), 0, s, (int32_t*)(st + L.tokens), L.ld_tok, B, n_init,
                       t4[0], t4[1], t4[2], t4[3], (const int32_t*)nullptr);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

extern "C" int wipa_decoder_run(const wipa_model_cfg* cfg, const void* const* w, void* state, size_t state_bytes, int B, int n_init,
                                int eot, const float* mask_first, const float* mask_always, int n_steps, int use_graph,
                                wipa_stream_t stream) {
    RT_CALL(cfg_check(cfg));
    SplitScope split_scope(cfg);
    WIPA_REQUIRE(w && state && mask_first && mask_always && B > 0 && n_steps >= 0, "wipa_decoder_run: bad arguments");
    W8Scope w8_scope(cfg, w);
    const wipa_dec_layout L = dec_layout(cfg, B);
    RT_CALL(state_fits("wipa_decoder_run", L, state_bytes));
    char* st = (char*)state;
    hipStream_t s = (hipStream_t)stream;
    RT_CALL(init_before_capture(cfg));
    if (n_steps == 0) return WIPA_OK;
    // the greedy tail's arrival counter starts every call at zero (4 bytes, outside the step graph)
    WIPA_CHECK_HIP(hipMemsetAsync(done_counter_of(st, L), 0, sizeof(int32_t), s));
    // the first step's input rows (embedding + first LayerNorm of the token at the current position): every later step gets
    // them from the previous step's last launch.  Outside the graph: once per call, whatever wrote the token (the greedy
    // update, the prompt, a forced history)
    if (step_needs_head(cfg, B)) RT_CALL(enqueue_step_head(cfg, w, st, L, B, stream));
    // every step of the call but the LAST is "lean" when the logits projection carries the greedy partials: its logits are not
    // written (state.logits holds those of the last step of a call, which is what callers read)
    const bool lean_ok = !use_fused_step(cfg, B) && logits_fused(cfg, B);
    struct LeanScope {
        explicit LeanScope(bool v) { t_lean_logits = v; }
        ~LeanScope() { t_lean_logits = false; }
    };
    if (!use_graph || !graphs_allowed()) {
        for (int i = 0; i < n_steps; ++i) {
            LeanScope lean(lean_ok && i + 1 < n_steps);
            RT_CALL(enqueue_decode_step(cfg, w, st, L, B, n_init, eot, mask_first, mask_always, stream));
        }
        return WIPA_OK;
    }
    WIPA_REQUIRE(s != nullptr, "wipa_decoder_run: graph capture needs a non-default stream");
    const int variant = cfg->dtype * 2 + t_f32_split + 4 * decode_mode(cfg, B) + 16 * cfg->dec_w_dtype + 64 * cfg->dec_cross_absorbed * (absorbed_block_fused() ? 2 : 1) + 256 * (int)tail_fused() + 512 * (int)absorbed_merge_out() + 1024 * cfg->dec_cross_splits + 8192 * (int)lean_ok;
    auto step_graph = [&](bool lean, hipGraphExec_t* out) -> int {  // kind 0: the full step, 2: the lean one
        hipGraphExec_t exec = nullptr;
        const GraphKey key(state, (const void*)w, (const void*)mask_first, (const void*)mask_always, B, n_init, eot, variant, cfg->weights_generation,
                           lean ? 2 : 0);
        {
            std::lock_guard<std::mutex> lk(g_graph_mu);
            auto it = g_graphs.find(key);
            if (it != g_graphs.end()) exec = it->second;
        }
        if (!exec) {
            hipGraph_t graph = nullptr;
            WIPA_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
            int rc;
            {
                LeanScope scope(lean);
                rc = enqueue_decode_step(cfg, w, st, L, B, n_init, eot, mask_first, mask_always, stream);
            }
            const hipError_t ee = hipStreamEndCapture(s, &graph);
            if (rc != WIPA_OK) {
                if (graph) hipGraphDestroy(graph);
                return rc;
            }
            WIPA_CHECK_HIP(ee);
            WIPA_CHECK_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            WIPA_CHECK_HIP(hipGraphDestroy(graph));
            std::lock_guard<std::mutex> lk(g_graph_mu);
            g_graphs[key] = exec;
        }
        *out = exec;
        return WIPA_OK;
    };
    hipGraphExec_t full = nullptr, lean = nullptr;
    RT_CALL(step_graph(false, &full));
    if (lean_ok && n_steps > 1) RT_CALL(step_graph(true, &lean));
    for (int i = 0; i < n_steps; ++i) WIPA_CHECK_HIP(hipGraphLaunch((lean && i + 1 < n_steps) ? lean : full, s));
    return WIPA_OK;
}

extern "C" int wipa_decoder_prefill(const wipa_model_cfg* cfg, const void* const* w, void* state, size_t state_bytes, int B,
                                    int n_init, int eot, const float* mask_first, const float* mask_always, int use_graph,
                                    wipa_stream_t stream) {
    RT_CALL(cfg_check(cfg));
    SplitScope split_scope(cfg);
    WIPA_REQUIRE(w && state && mask_first && mask_always && B > 0, "wipa_decoder_prefill: bad arguments");
    WIPA_REQUIRE(n_init >= 1 && n_init <= MAX_PROMPT, "wipa_decoder_prefill: 1..%d prompt tokens (got %d)", MAX_PROMPT, n_init);
    W8Scope w8_scope(cfg, w);
    const wipa_dec_layout L = dec_layout(cfg, B);
    RT_CALL(state_fits("wipa_decoder_prefill", L, state_bytes));
    char* st = (char*)state;
    auto enqueue = [&]() -> int {
        if (n_init == 1) {  // a bare [sot] prompt: one ordinary step at position 0 (with its head when the step is tail-fused)
            if (step_needs_head(cfg, B)) RT_CALL(enqueue_step_head(cfg, w, st, L, B, stream));
            return enqueue_decode_step(cfg, w, st, L, B, n_init, eot, mask_first, mask_always, stream);
        }
        return enqueue_prefill(cfg, w, st, L, B, n_init, eot, mask_first, mask_always, stream);
    };
    hipStream_t s = (hipStream_t)stream;
    RT_CALL(init_before_capture(cfg));
    WIPA_CHECK_HIP(hipMemsetAsync(done_counter_of(st, L), 0, sizeof(int32_t), s));  // as wipa_decoder_run: the tail's counter starts at zero
    if (!use_graph || s == nullptr || !graphs_allowed()) return enqueue();
    hipGraphExec_t exec = nullptr;
    const GraphKey key(state, (const void*)w, (const void*)mask_first, (const void*)mask_always, B, n_init, eot, cfg->dtype * 2 + t_f32_split + 4 * decode_mode(cfg, B) + 16 * cfg->dec_w_dtype + 64 * cfg->dec_cross_absorbed * (absorbed_block_fused() ? 2 : 1) + 256 * (int)tail_fused() + 512 * (int)absorbed_merge_out() + 1024 * cfg->dec_cross_splits + 8192 * (int)(!use_fused_step(cfg, B) && logits_fused(cfg, B)), cfg->weights_generation, 1);
    {
        std::lock_guard<std::mutex> lk(g_graph_mu);
        auto it = g_graphs.find(key);
        if (it != g_graphs.end()) exec = it->second;
    }
    if (!exec) {
        hipGraph_t graph = nullptr;
        WIPA_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        const int rc = enqueue();
        const hipError_t ee = hipStreamEndCapture(s, &graph);
        if (rc != WIPA_OK) {
            if (graph) hipGraphDestroy(graph);
            return rc;
        }
        WIPA_CHECK_HIP(ee);
        WIPA_CHECK_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        WIPA_CHECK_HIP(hipGraphDestroy(graph));
        std::lock_guard<std::mutex> lk(g_graph_mu);
        g_graphs[key] = exec;
    }
    WIPA_CHECK_HIP(hipGraphLaunch(exec, s));
    return WIPA_OK;
}

extern "C" int wipa_decoder_release(void* state) {
    std::lock_guard<std::mutex> lk(g_graph_mu);
    for (auto it = g_graphs.begin(); it != g_graphs.end();) {
        if (std::get<0>(it->first) == state) {
            hipGraphExecDestroy(it->second);
            it = g_graphs.erase(it);
        } else {
            ++it;
        }
    }
    return WIPA_OK;
}

// ------------------------------------------------------------------ teacher-forced decoder
namespace {
struct TfWs {
    size_t x, ln, qkv, q, ao, h, ckv, total;
};
TfWs tf_ws(const wipa_model_cfg* c, int B, int T) {
    const size_t e = wipa_dtype_size(c->dtype), d = c->n_text_state, M = (size_t)B * T;
    TfWs w;
    size_t o = 0;
    w.x = o;   o += align256(M * d * 4);
    w.ln = o;  o += align256(M * d * e);
    w.qkv = o; o += align256(M * 3 * d * e);
    w.q = o;   o += align256(M * d * e);
    w.ao = o;  o += align256(M * d * e);
    w.h = o;   o += align256(M * 4 * d * e);
    w.ckv = o; o += align256((size_t)B * 2 * c->n_text_head * c->n_audio_ctx * 64 * e);
    w.total = o;
    return w;
}
}  // namespace

extern "C" size_t wipa_decoder_logits_workspace_bytes(const wipa_model_cfg* cfg, int B, int T) {
    if (!cfg || B <= 0 || T <= 0) return 0;
    return tf_ws(cfg, B, T).total;
}

extern "C" int wipa_decoder_logits(const wipa_model_cfg* cfg, const void* const* w, const int32_t* tokens, const void* features,
                                   float* logits, int64_t ld_logits, void* workspace, size_t workspace_bytes, int B, int T,
                                   wipa_stream_t stream) {
    RT_CALL(cfg_check(cfg));
    SplitScope split_scope(cfg);
    WIPA_REQUIRE(w && tokens && features && logits && workspace && B > 0 && T > 0, "wipa_decoder_logits: bad arguments");
    WIPA_REQUIRE(T <= cfg->n_text_ctx, "wipa_decoder_logits: T=%d exceeds n_text_ctx=%d", T, cfg->n_text_ctx);
    WIPA_REQUIRE(cfg->dec_w_dtype == 0, "wipa_decoder_logits: the teacher-forced decoder runs on a bf16 / f32 weight table (fp8 tables serve the decode step)");
    const TfWs L = tf_ws(cfg, B, T);
    WIPA_REQUIRE(workspace_bytes >= L.total, "wipa_decoder_logits: workspace too small (%zu < %zu)", workspace_bytes, L.total);
    const int dt = cfg->dtype;
    const size_t e = wipa_dtype_size(dt);
    const int d = cfg->n_text_state, H = cfg->n_text_head, Ta = cfg->n_audio_ctx, M = B * T;
    char* ws = (char*)workspace;
    float* x = (float*)(ws + L.x);
    void* ln = ws + L.ln;
    char* qkv = ws + L.qkv;
    void* q = ws + L.q;
    void* ao = ws + L.ao;
    void* hb = ws + L.h;
    char* ckv = ws + L.ckv;
    RT_CALL(wipa_embed_tokens(tokens, T, B, T, 0, nullptr, w[0], dt, nullptr, (const float*)w[1], x, d, stream));
    for (int l = 0; l < cfg->n_text_layer; ++l) {
        const void* const* lw = w + WIPA_DEC_GLOBAL + WIPA_DEC_PER_LAYER * l;
        RT_CALL(wipa_layernorm(x, WIPA_F32, d, ln, dt, d, (const float*)lw[0], (const float*)lw[1], M, d, 1e-5f, stream));
        {
            wipa_gemm_desc g;
            memset(&g, 0, sizeof(g));
            g.col_scale_n = 2 * d; g.col_scale = QK_SCALE;
            RT_CALL(gemm(ln, d, lw[2], d, qkv, 3 * d, M, 3 * d, d, dt, dt, (const float*)lw[3], 0, nullptr, stream, &g));
        }
        {
            wipa_attn_desc a;
            memset(&a, 0, sizeof(a));
            a.q = qkv; a.k = qkv + (size_t)d * e; a.v = qkv + 2 * (size_t)d * e; a.out = ao;
            a.q_bs = (int64_t)T * 3 * d; a.q_rs = 3 * d; a.q_hs = 64;
            a.k_bs = a.q_bs; a.k_rs = 3 * d; a.k_hs = 64;
            a.v_bs = a.q_bs; a.v_rs = 3 * d; a.v_hs = 64;
            a.o_bs = (int64_t)T * d; a.o_rs = d; a.o_hs = 64;
            a.B = B; a.H = H; a.Tq = T; a.Tk = T; a.causal = 1; a.dtype = dt;
            RT_CALL(wipa_attention(&a, stream));
        }
        RT_CALL(gemm(ao, d, lw[4], d, x, d, M, d, d, dt, WIPA_F32, (const float*)lw[5], 0, x, stream));
        RT_CALL(wipa_layernorm(x, WIPA_F32, d, ln, dt, d, (const float*)lw[6], (const float*)lw[7], M, d, 1e-5f, stream));
        {
            wipa_gemm_desc g;
            memset(&g, 0, sizeof(g));
            g.col_scale_n = d; g.col_scale = QK_SCALE;
            RT_CALL(gemm(ln, d, lw[8], d, q, d, M, d, d, dt, dt, (const float*)lw[9], 0, nullptr, stream, &g));
        }
        {
            wipa_gemm_desc g;
            memset(&g, 0, sizeof(g));
            g.col_scale_n = d; g.col_scale = QK_SCALE;
            g.rg_in = Ta; g.rg_valid = Ta; g.rg_stride = (int64_t)2 * H * Ta * 64;
            g.cg_in = 64; g.cg_stride = (int64_t)Ta * 64;
            RT_CALL(gemm(features, d, lw[10], d, ckv, 64, B * Ta, 2 * d, d, dt, dt, (const float*)lw[11], 0, nullptr, stream, &g));
        }
        {
            wipa_attn_desc a;
            memset(&a, 0, sizeof(a));
            a.q = q; a.k = ckv; a.v = ckv + (size_t)H * Ta * 64 * e; a.out = ao;
            a.q_bs = (int64_t)T * d; a.q_rs = d; a.q_hs = 64;
            a.k_bs = (int64_t)2 * H * Ta * 64; a.k_rs = 64; a.k_hs = (int64_t)Ta * 64;
            a.v_bs = a.k_bs; a.v_rs = 64; a.v_hs = a.k_hs;
            a.o_bs = (int64_t)T * d; a.o_rs = d; a.o_hs = 64;
            a.B = B; a.H = H; a.Tq = T; a.Tk = Ta; a.causal = 0; a.dtype = dt;
            RT_CALL(wipa_attention(&a, stream));
        }
        RT_CALL(gemm(ao, d, lw[12], d, x, d, M, d, d, dt, WIPA_F32, (const float*)lw[13], 0, x, stream));
        RT_CALL(wipa_layernorm(x, WIPA_F32, d, ln, dt, d, (const float*)lw[14], (const float*)lw[15], M, d, 1e-5f, stream));
        RT_CALL(gemm(ln, d, lw[16], d, hb, 4 * d, M, 4 * d, d, dt, dt, (const float*)lw[17], 1, nullptr, stream));
        RT_CALL(gemm(hb, 4 * d, lw[18], 4 * d, x, d, M, d, 4 * d, dt, WIPA_F32, (const float*)lw[19], 0, x, stream));
    }
    RT_CALL(wipa_layernorm(x, WIPA_F32, d, ln, dt, d, (const float*)w[2], (const float*)w[3], M, d, 1e-5f, stream));
    RT_CALL(gemm(ln, d, w[0], d, logits, ld_logits, M, cfg->n_vocab, d, dt, WIPA_F32, nullptr, 0, nullptr, stream));
    return WIPA_OK;
}
