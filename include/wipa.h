/* wipa.h -- C ABI of libwipa.so: the MI355X (gfx950) Whisper -> IPA hot path.
 *
 * The reference (barathanaslan/whisper-ipa) has no FFI of its own: its hot path
 * sits behind Python calls into mlx_whisper / mlx.  Each entry
 * point below names the reference call site it replaces.  Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller unless it says "host";
 *     the library never allocates or frees caller-visible memory -- scratch comes in
 *     as (workspace, bytes) with a *_bytes() query;
 *   - row-major, contiguous unless a leading dimension is given (in ELEMENTS);
 *   - asynchronous on the given hipStream_t, no hidden synchronisation;
 *   - returns 0 or a negative error code; wipa_last_error() has the text;
 *   - no torch / C++ types cross this boundary.
 * dtype: "T" below is the matrix/activation type of the call (f32 or bf16);
 * biases, LayerNorm parameters, positional tables, the residual stream and the
 * logits are always f32; all accumulation is f32.
 */
#ifndef WIPA_H
#define WIPA_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* wipa_stream_t; /* hipStream_t */

enum { WIPA_F32 = 0, WIPA_BF16 = 1, WIPA_FP8_E4M3 = 2 /* weights only: OCP e4m3fn codes, one byte per element */ };
enum { WIPA_OK = 0, WIPA_ERR_ARG = -1, WIPA_ERR_HIP = -2, WIPA_ERR_STATE = -3 };

#define WIPA_N_SAMPLES 480000 /* 30 s at 16 kHz (mlx_whisper.audio.N_SAMPLES) */
#define WIPA_N_FRAMES 3000
#define WIPA_N_FFT 400
#define WIPA_HOP 160
#define WIPA_HEAD_DIM 64

int wipa_version(void);
const char* wipa_last_error(void);

/* A HIP stream whose kernels may only run on the first n_cus compute units of the device's CU-mask order (on MI355X mask bit
 * i is CU i/8 of XCD i%8, so every XCD keeps n_cus/8 of its 32 CUs; n_cus a multiple of 8).  Serving keeps several passes in
 * flight; giving the MFMA-bound encoder of one pass fewer than all CUs leaves the rest to the HBM-/latency-bound decode loop of
 * another (bench.py --encoder-cus).  The caller destroys the stream after synchronising it. */
/* A plain non-blocking HIP stream created by the library, in the order the host asks for them.  ROCm hands hardware queues to
 * streams in creation order (GPU_MAX_HW_QUEUES of them, then shared), so a host that keeps several passes in flight
 * (whisper_ipa_amd/pipeline.py; reference: the batch loops of scripts/evaluate_model.py:181-212) creates its pass streams -- and,
 * before them, any idle padding streams -- itself instead of taking them from a framework's pool. */
int wipa_stream_create(wipa_stream_t* out);
int wipa_stream_create_cu_limited(int n_cus, wipa_stream_t* out);
int wipa_stream_destroy(wipa_stream_t s);

/* ------------------------------------------------------------------ K1 log-mel
 * replaces mlx_whisper.audio.log_mel_spectrogram (+ pad_or_trim) at
 * scripts/ipa_data_loader.py:80-82, scripts/transcribe_single.py:44-45,
 * scripts/evaluate_model.py:188-189.
 * tables: DFT(window folded in) + mel filterbank, built once by wipa_logmel_init.
 * audio  [B, 480000] f32 (already pad_or_trim'ed);
 * mel    [B*3002 + 4, n_mels] T ("padded mel"): frame t of clip b at row b*3002 + t + 1,
 *        rows b*3002 and b*3002+3001 (the conv1 halo) and the 4 tail rows written as zero,
 *        values = (max(log10(max(mel,1e-10)), gmax-8)+4)/4. */
#define WIPA_MEL_ROWS(B) ((int64_t)(B) * 3002 + 4)
size_t wipa_logmel_tables_bytes(int n_mels);
int wipa_logmel_init(void* tables, int n_mels, wipa_stream_t s);
size_t wipa_logmel_workspace_bytes(int batch, int n_mels);
int wipa_logmel(const float* audio, int batch, int n_mels, const void* tables, void* mel, int mel_dtype,
                void* workspace, size_t workspace_bytes, wipa_stream_t s);
/* [B,3000,n_mels] f32 (any producer) -> padded mel [B*3002 + 4, n_mels] T for the encoder. */
int wipa_mel_pad_cast(const float* mel, int batch, int n_mels, void* mel_padded, int dtype, wipa_stream_t s);

/* ------------------------------------------------------------------ K4 GEMM
 * C = epilogue(A * W^T): every nn.Linear / Conv1d of mlx_whisper.whisper
 * (AudioEncoder, TextDecoder, MultiHeadAttention, ResidualAttentionBlock).
 * A: row m starts at A + m*lda (rows may OVERLAP: conv-as-GEMM, STFT framing), K contiguous.
 * W: [N,K] row-major (nn.Linear layout [out,in]); K must be a multiple of 128/sizeof(T)
 *    bytes-wise (64 bf16 / 32 f32): pad W with zero columns, A must stay readable.
 * epilogue, in order: + bias, * col_scale (columns < col_scale_n), gelu(erf),
 *    + pos[(m % rg_in) * ldpos + n], + residual, cast, store.
 * output element (m,n) lives at
 *    C + c_offset + (*c_offset_dev) + (m / rg_in) * rg_stride + (m % rg_in) * ldc
 *      + (n / cg_in) * cg_stride + (n % cg_in)
 *    rows with (m % rg_in) >= rg_valid are skipped (or written as 0 if zero_invalid_rows).
 *    Defaults rg_in = M (rg_valid = M), cg_in = N give the plain C[m*ldc + n].
 * residual uses the same addressing and dtype as C and may alias it. */
typedef struct wipa_gemm_desc {
    const void* A;
    const void* W;
    void* C;
    const float* bias;
    const void* residual;
    const float* pos;
    const int64_t* c_offset_dev;
    int64_t lda, ldw, ldc, ldpos;
    int64_t rg_stride, cg_stride, c_offset;
    int64_t slab_stride; /* elements between the k_slices partial outputs */
    int32_t M, N, K;
    int32_t in_dtype, out_dtype;
    int32_t rg_in, rg_valid, cg_in;
    int32_t zero_invalid_rows;
    int32_t bias_along_m;
    int32_t act; /* 0 none, 1 gelu(erf) */
    int32_t col_scale_n;
    float col_scale;
    int32_t k_slices; /* 0/1: whole K.  >1: K is cut into k_slices contiguous slices
                       * computed by different workgroups; slice z writes its PARTIAL sums (bias in
                       * slice 0 only, no act/pos/residual) to C + z*slab_stride -- to be summed in a
                       * fixed order by wipa_add_slabs_layernorm or wipa_sum_slabs (deterministic split-K).  M <= 1024
                       * runs in the weight-streaming kernel, larger M in the 128x128 tile kernel. */
    int32_t f32_split; /* float32 inputs in the tile kernels (M > 256 rows).  0 (default): exact f32 products on the f32 MFMA.
                        * 1: every product a*w is taken as three bf16 MFMA terms on operands split into hi + lo
                        * (a_hi*w_lo + a_lo*w_hi + a_hi*w_hi, f32 accumulation): about twice the rate at ~5e-6 relative error
                        * of a K = 768 dot product (f32 MFMA: ~1.5e-6).  The weight-streaming kernel always multiplies exactly. */
    /* fp8 weights (BASELINE.json configs[4]: whisper-large-v3 fp8-weight inference).  w_dtype = WIPA_FP8_E4M3: W holds OCP
     * e4m3fn codes [N, K] (ldw in bytes = elements), the weight value is code * w_scale[n]; activations must be bf16.
     * Weight-streaming kernel only (M <= 1024): the decode step, where the weight stream is what a projection costs.
     * w_dtype = 0: W has the input dtype. */
    const float* w_scale;
    int32_t w_dtype;
    /* LayerNorm prologue (decode step: mlp_ln folded into mlp1).  ln_x != NULL: A is ignored and the A operand is
     * LayerNorm(ln_x[m, 0:K]; ln_w, ln_b, ln_eps) rounded to the input dtype, computed inside the kernel from the f32 rows
     * ln_x + m*ln_ldx.  Weight-streaming kernel only: M <= 1024, K a multiple of 64 and <= 1280, no k_slices. */
    const float* ln_x;
    const float* ln_w;
    const float* ln_b;
    int64_t ln_ldx;
    float ln_eps;
    int32_t stream_weights; /* 1: the rows are decode rows (one or a few per clip) and W is a weight matrix: use the
                             * weight-streaming kernel up to M = 1024 instead of 256, so that a prompt prefill of
                             * 4 rows per clip rounds exactly like the single-row steps (batch invariance) */
    /* K-major operands (float32 in and out, 128x128 tile kernel, k_slices allowed): a_trans: A is stored [K][M] (lda =
     * elements per k-row, M a multiple of 4); w_trans: W is stored [K][N] (ldw likewise, N a multiple of 4).  The staging
     * pass transposes into LDS, so the backward pass of a linear layer multiplies by W^T (input gradient) and by dy^T, x^T
     * (weight gradient) without a transposed copy: nn.value_and_grad at scripts/train_whisper_ipa.py:284. */
    int32_t a_trans;
    int32_t w_trans;
    /* fp8 ACTIVATIONS x fp8 weights on the block-scaled fp8 matrix instruction (BASELINE.json configs[4]: "CDNA4 fp8 MFMA").
     * in_dtype = WIPA_FP8_E4M3: A [M, K] and W [N, K] both hold OCP e4m3fn codes (lda / ldw in bytes = elements), the values
     * are code * a_scale[m] and code * w_scale[n] (power-of-two scales from wipa_layernorm_fp8 / wipa_rowquant_fp8 and the
     * weight quantiser); K a multiple of 128.  256 x 256 tile kernel, every epilogue option except k_slices. */
    const float* a_scale;
} wipa_gemm_desc;
int wipa_gemm(const wipa_gemm_desc* d, wipa_stream_t s);
/* Dispatch census (measurement / test aid, no reference counterpart): how many wipa_gemm calls of this process went to each
 * kernel family since the last reset.  out (HOST) receives the first n counters; reset != 0 clears all of them afterwards.
 * The parity tests of the fine-tune step (scripts/train_whisper_ipa.py:266-311) use it to prove that the shape-dependent
 * branches tuned for 32 clips x 64 tokens (split-K, 384 x 128 tiles, K-major operands) are the ones under test. */
enum {
    WIPA_GEMM_TILE128 = 0,   /* 128 x 128 register-staged tile kernel */
    WIPA_GEMM_TILE256 = 1,   /* 256 x 256 LDS-DMA tile kernel */
    WIPA_GEMM_TILE384 = 2,   /* 384 x 256 */
    WIPA_GEMM_TILE384N = 3,  /* 384 x 128 (float32, badly quantised wide grids) */
    WIPA_GEMM_TILE256P = 4,  /* phase-interleaved 256 x 256 (WIPA_GEMM_TILE=2568) */
    WIPA_GEMM_SKINNY = 5,    /* weight-streaming kernel (decode rows) */
    WIPA_GEMM_SKINNY_FP8 = 6,
    WIPA_GEMM_SKINNY_LN = 7, /* ... with the LayerNorm prologue */
    WIPA_GEMM_KMAJOR = 8,    /* a_trans / w_trans operands (128 x 128 kernel) */
    WIPA_GEMM_SPLIT_K = 9,   /* calls with k_slices > 1 (counted in addition to their kernel family) */
    WIPA_GEMM_TILE_FP8 = 10, /* fp8 x fp8 on v_mfma_scale_f32_16x16x128_f8f6f4 (in_dtype = WIPA_FP8_E4M3): every such call */
    WIPA_GEMM_TILE_FP8_384 = 11, /* ... those of them that took the 384 x 256 tile (counted in addition) */
    WIPA_GEMM_DISPATCH_CLASSES = 12
};
int wipa_gemm_dispatch_counts(int64_t* out, int n, int reset);

/* ------------------------------------------------------------------ K3 LayerNorm
 * nn.LayerNorm(eps=1e-5) rows of width D (attn_ln, cross_attn_ln, mlp_ln, ln_post, ln). */
int wipa_layernorm(const void* x, int x_dtype, int64_t ldx, void* y, int y_dtype, int64_t ldy, const float* w,
                   const float* b, int rows, int D, float eps, wipa_stream_t s);

/* fp8 activations (cfg.enc_act_fp8): y = e4m3fn codes of LayerNorm(x) [rows, D] bytes (row stride ldy) with ONE power-of-two
 * scale per row, y_scale[r] = the smallest power of two with max|LN(x)[r,:]| / y_scale <= 448; value = code * y_scale[r].
 * x f32 rows; D a multiple of 4. */
int wipa_layernorm_fp8(const float* x, int64_t ldx, void* y, int64_t ldy, float* y_scale, const float* w, const float* b, int rows,
                       int D, float eps, wipa_stream_t s);
/* the same row quantisation of an existing activation matrix x [rows, D] (x_dtype WIPA_BF16 or WIPA_F32; D a multiple of 8). */
int wipa_rowquant_fp8(const void* x, int x_dtype, int64_t ldx, void* y, int64_t ldy, float* y_scale, int rows, int D, wipa_stream_t s);

/* Decode-step fusion: x[r,:] += sum_s slabs[s][r,:] (fixed order s = 0..n_slabs-1; x f32, in place),
 * then y = LayerNorm(x).  slabs are the k_slices partial outputs of the preceding residual GEMM. */
int wipa_add_slabs_layernorm(float* x, int64_t ldx, const float* slabs, int n_slabs, int64_t slab_stride, void* y,
                             int y_dtype, int64_t ldy, const float* w, const float* b, int rows, int D, float eps,
                             wipa_stream_t s);

/* ------------------------------------------------------------------ K8 embedding
 * TextDecoder: token_embedding[tokens] + positional_embedding[offset : offset+T].
 * tokens [B, ld_tok] int32; row (b,t) uses token tokens[b][p] and position p,
 * p = t_start + (pos_dev ? *pos_dev : 0) + t.  x [B*T, D] f32.  emb_dtype WIPA_FP8_E4M3: tok_emb holds e4m3 codes and
 * row v dequantises as bf16(code * emb_scale[v]) (emb_scale is ignored otherwise). */
int wipa_embed_tokens(const int32_t* tokens, int64_t ld_tok, int B, int T, int t_start, const int32_t* pos_dev,
                      const void* tok_emb, int emb_dtype, const float* emb_scale, const float* pos_emb, float* x, int D,
                      wipa_stream_t s);

/* ------------------------------------------------------------------ attention
 * MultiHeadAttention.qkv_attention with head_dim 64; q and k arrive already
 * scaled by 64**-0.25 (GEMM epilogue col_scale), softmax in f32.
 * decoder causal self-attention and teacher-forced cross-attention (mlx_whisper
 * TextDecoder blocks behind scripts/train_whisper_ipa.py:232), any Tq / Tk.  float32 with
 * Tq >= 16 and no device-side offsets runs on the f32 MFMA (exact f32 products, 64 queries
 * per workgroup); the few-row prompt prefill and bf16 use the f32-math VALU kernel.
 * Element (b, t, h, d) of X is at
 *   X + b*x_bs + t*x_rs + h*x_hs + d.
 * Tk = (tk_dev ? *tk_dev : 0) + Tk;  causal: query i sees keys <= i + (Tk - Tq). */
typedef struct wipa_attn_desc {
    const void* q;
    const void* k;
    const void* v;
    void* out;
    const int32_t* tk_dev;
    const int32_t* q_row_dev; /* optional device int: first query row = *q_row_dev (decode step) */
    float* lse;               /* optional out (wipa_attention only): f32 [B,H,Tq] log-sum-exp of each score row */
    int64_t q_bs, q_rs, q_hs;
    int64_t k_bs, k_rs, k_hs;
    int64_t v_bs, v_rs, v_hs;
    int64_t o_bs, o_rs, o_hs;
    int32_t B, H, Tq, Tk;
    int32_t causal, dtype;
} wipa_attn_desc;
int wipa_attention(const wipa_attn_desc* d, wipa_stream_t s);

/* K5 encoder self-attention, bf16 MFMA flash kernel (non-causal, T keys).
 * qk  [B*T, ldqk] bf16: q of head h at column h*64, k at column D + h*64;
 * vt  [B][D][ldvt] bf16: V transposed per clip (row h*64+d, column t), ldvt >= ceil64(T),
 *     columns >= T must hold finite values (zero);
 * out [B*T, ldo] bf16. */
int wipa_flash_attn_enc_bf16(const void* qk, int64_t ldqk, const void* vt, int64_t ldvt, void* out, int64_t ldo,
                             int B, int H, int T, wipa_stream_t s);
/* K5 in f32: the same encoder attention on the f32 MFMA (exact f32 products) for models kept in float32, as the
 * reference's scripts do (transcribe_single.py:13, train_whisper_ipa.py:505).  q, k, v: [B*T, ld*] f32 with head h at
 * column h*64 (q and k pre-scaled by 64^-0.25 each); out [B*T, ldo].  Row strides multiples of 4, pointers 16-byte aligned.
 * f32_split != 0: scores on a three-way and P*V on a two-way bf16 split of the operands (bf16 MFMA, error ~2^-24 of a
 * score) instead of exact f32 products -- the same opt-in as wipa_gemm_desc.f32_split. */
int wipa_flash_attn_enc_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, float* out,
                            int64_t ldo, int B, int H, int T, int f32_split, wipa_stream_t s);

/* K11/K12 decode-step attention (HBM-bound): ONE query row per (b,h) (Tq must be 1) against
 * cached K/V with the strides of wipa_attn_desc; 8 (bf16) / 16 (f32) lanes stream one 64-dim key
 * row with 16-byte loads, online softmax per lane group, merged across 4 waves.  Used for the
 * growing self-attention cache (tk_dev / q_row_dev = position) and, via the wrapper below,
 * for cross-attention. */
int wipa_decode_attn(const wipa_attn_desc* d, wipa_stream_t s);
/* cross-attention wrapper: q [B, H*64] T; kv [B][2H][Tk][64] T (K heads then V heads);
 * out [B, H*64] T. */
int wipa_decode_cross_attn(const void* q, const void* kv, void* out, int B, int H, int Tk, int dtype,
                           wipa_stream_t s);
/* n_q (1..4) query rows per clip against the same cache: q / out [B*n_q, H*64] rows (b, t); every K/V row is read once. */
int wipa_decode_cross_attn_multi(const void* q, const void* kv, void* out, int B, int H, int Tk, int n_q, int dtype,
                                 wipa_stream_t s);

/* ------------------------------------------------------------------ fused decode-step blocks (K11 / K12)
 * One decode step of mlx_whisper's ResidualAttentionBlock with a KV cache (DecodingTask._main_loop,
 * transcribe_single.py:55) in 5 launches per layer instead of 11: see csrc/decode_fused.hip.
 *
 * self block, one workgroup per (head, 16 token rows):
 *   y = LayerNorm(x[b]; ln_w, ln_b) -> (q|k|v)_h = y Wqkv[h]^T + b (q, k scaled by qk_scale, rounded to T) -> k, v appended to
 *   the caches at position *pos -> o_h = softmax(q_h K_h^T) V_h over positions 0..*pos -> slabs[h][b][:] = o_h Wo[:, h*64:(h+1)*64]^T.
 * x [B, d] f32; wqkv [3d, d] T (query | key | value rows), bqkv [3d]; wo [d, d] T; kcache / vcache T [B][n_ctx][d]
 * (kv_batch_stride = n_ctx * d elements); slabs f32 [H][B][d] with slab_stride >= B*d elements.  The residual update
 * x + bias_o + sum_h slabs[h] is left to wipa_decode_cross_block. */
typedef struct wipa_self_block_desc {
    const float* x;
    const float* ln_w;
    const float* ln_b;
    const void* wqkv;
    const float* bqkv;
    const void* wo;
    void* kcache;
    void* vcache;
    const int32_t* pos; /* device */
    float* slabs;
    int64_t kv_batch_stride, slab_stride;
    int32_t B, d, H, dtype;
    float eps, qk_scale;
} wipa_self_block_desc;
int wipa_decode_self_block(const wipa_self_block_desc* d, wipa_stream_t s);
/* cross block, one workgroup per (head, clip):
 *   r = x_in[b] + bias_o + slabs[0][b] + ... + slabs[n_slabs-1][b] (that order; bias_o may be NULL when slab 0 already carries
 *   the bias, as the split-K slabs of wipa_gemm do; the h = 0 workgroup stores r to x_out[b]) ->
 *   y = LayerNorm(r; ln_w, ln_b) -> q_h = (y Wq[h]^T + bq) * qk_scale (rounded to T) -> out[b][h*64:(h+1)*64] =
 *   softmax(q_h K^T) V over the cached cross keys/values kv [B][2H][Tk][64] T (K heads then V heads; every element read
 *   once, non-temporal).  x_out must not alias x_in.  n_slabs <= 20. */
typedef struct wipa_cross_block_desc {
    const float* x_in;
    float* x_out;
    const float* slabs;
    const float* bias_o;
    const float* ln_w;
    const float* ln_b;
    const void* wq;
    const float* bq;
    const void* kv;
    void* out;
    int64_t slab_stride;
    int32_t n_slabs, B, d, H, Tk, dtype;
    float eps, qk_scale;
    int32_t cross_splits; /* wipa_decode_cross_absorbed_block[_out] only: frame splits per clip of the streaming launch, 1..4; 0 = the
                           * default (4).  wipa_decode_cross_block ignores it. */
} wipa_cross_block_desc;
int wipa_decode_cross_block(const wipa_cross_block_desc* d, wipa_stream_t s);

/* ------------------------------------------------------------------ K11 with ABSORBED key / value projections
 * Decode-step cross-attention that streams the encoder output xa itself instead of cached K = xa Wk^T and V = xa Wv^T + bv
 * (mlx_whisper MultiHeadAttention with xa, behind DecodingTask._main_loop, transcribe_single.py:55):
 *     scores_h = (q_h Wk_h) xa^T,   out_h = (softmax(scores_h) xa) Wv_h^T + bv_h
 * -- one pass over xa [Tk, d] per clip, layer and step instead of one over K and one over V (half the bytes), no K/V cache and no
 * cross-K/V projection; the 12x larger contractions run on the matrix cores (csrc/cross_absorbed.hip).  bf16, H <= 16 heads of
 * 64, d in {384, 512, 768, 1024}.
 * q   [B rows, row stride q_row_stride] bf16: the cross query, already multiplied by 64^-0.25 (and with its bias);
 * wkT [d, d] bf16 = Wk^T ([in][out]);  xa [B, Tk, d] bf16;  wv [d, d] bf16 ([out][in]), bv [d] f32;
 * out [B rows, row stride out_row_stride] bf16;  k_scale = 64^-0.25 (the key side's share of the score scale);
 * scratch: wipa_cross_absorbed_scratch_bytes(B, d, Tk) bytes, 16-byte aligned (sized for the largest split count).
 * wipa_cross_absorbed_init(d) raises the kernel's LDS limit (call it once outside any stream capture).
 * Frame splits: the streaming launch has n_splits x B workgroups, each holding a CU for Tk / n_splits frames of one clip, and the
 * merge adds the split partials in split order -- so the count is part of the result's rounding and NEVER depends on B.  4 (the
 * default; want = 0) fills the 256 CUs at 64 clips and gives a lone decode step its shortest launch; 2 leaves half the chip to
 * the other passes in flight (round 4, whisper-small, 64 clips, 4 passes in flight: 72.3 against 75.3 ms per pass, a lone step
 * 1.35 against 1.25 ms; 3: 73.5 ms / 1.26 ms) -- the caller who pipelines passes chooses (wipa_model_cfg.dec_cross_splits).
 * wipa_cross_absorbed_splits(want, Tk) = the count a launch uses: want clamped to 1..4 and to >= two 32-frame tiles per split. */
int wipa_cross_absorbed_splits(int want, int Tk);
size_t wipa_cross_absorbed_scratch_bytes(int B, int d, int Tk);
int wipa_cross_absorbed_init(int d);
int wipa_cross_absorbed_attention(const void* q, int64_t q_row_stride, const void* wkT, const void* xa, const void* wv, const float* bv,
                                  void* out, int64_t out_row_stride, void* scratch, size_t scratch_bytes, int B, int H, int d, int Tk,
                                  float k_scale, int n_splits, wipa_stream_t s);
/* The absorbed cross block of ONE decode step in three launches (the decode loop's form; wipa_cross_absorbed_attention with a
 * given query serves the prompt prefill and the tests): [split-K slab sum + residual + cross_attn_ln + cross query + absorbed query]
 * -> streaming kernel -> [merge + value projection].  The descriptor is wipa_decode_cross_block's with c->kv = the encoder output
 * xa [B][Tk][d] and c->out [B][d] bf16; bias_o must be NULL (slab 0 carries the out-projection bias), n_slabs <= 4;
 * wkT / wv / bv / scratch as above.  c->x_out must not alias c->x_in. */
int wipa_decode_cross_absorbed_block(const wipa_cross_block_desc* c, const void* wkT, const void* wv, const float* bv, void* scratch,
                                     size_t scratch_bytes, wipa_stream_t s);
/* The same block with the cross-attention OUT projection inside its third launch (round 4; what the decode step runs): instead of
 * c->out (may be NULL) the launch writes H split-K slabs slabs_out[h * slab_stride + b * d + n] = v_h[b] . Wo[n][h*64 .. h*64+63]
 * (f32; slab 0 carries bo) -- MultiHeadAttention.out of the cross-attention as a per-head K split -- which the next
 * wipa_add_slabs_layernorm(n_slabs = H <= 16) adds to the residual rows in head order.  wo [d, d] bf16 ([out][in]), bo [d] f32;
 * slabs_out may be the buffer c->slabs points to. */
int wipa_decode_cross_absorbed_block_out(const wipa_cross_block_desc* c, const void* wkT, const void* wv, const float* bv, const void* wo,
                                         const float* bo, float* slabs_out, int64_t slab_stride, void* scratch, size_t scratch_bytes,
                                         wipa_stream_t s);
/* measurement aid: the streaming kernel of wipa_cross_absorbed_attention alone, on a scratch a full call has filled */
int wipa_cross_absorbed_stream(const void* xa, void* scratch, size_t scratch_bytes, int B, int H, int d, int Tk, int n_splits,
                               wipa_stream_t s);

/* ------------------------------------------------------------------ K13 greedy step
 * GreedyDecoder.update + SuppressBlank + SuppressTokens of mlx_whisper.decoding
 * (transcribe_single.py:49-55).  p = *pos_dev is the position whose logits these are.
 * If p + 1 < n_init the next token is already given (prompt) and nothing is written.
 * Otherwise: logits += (p + 1 == n_init ? mask_first : mask_always) (0 / -inf, f32 [V]);
 * next = argmax (lowest index on ties); sum_logprobs += log_softmax[next] unless the
 * previous token was eot; next = eot if previous was eot; tokens[b][p+1] = next.
 * not_done is incremented by the number of rows whose next != eot. */
int wipa_greedy_step(const float* logits, int64_t ldl, int B, int V, const float* mask_first,
                     const float* mask_always, int32_t* tokens, int64_t ld_tok, const int32_t* pos_dev, int n_init,
                     int eot, float* sum_logprobs, int32_t* not_done, wipa_stream_t s);
int wipa_add_i32(int32_t* p, int32_t v, wipa_stream_t s);
/* The decode step's tail in ONE launch (round 4; the step replayed by wipa_decoder_run ends with it): wipa_greedy_step as above
 * (at a prompt position, p + 1 < n_init, the given token is taken), THEN the next step's input row -- x[b] = tok_emb[next] +
 * pos_emb[min(p + 1, n_ctx - 1)] (f32) and y[b] = LayerNorm(x[b]; ln_w, ln_b) in y_dtype (the first block's attn_ln) -- and, by
 * the last workgroup to arrive on *done_counter (int32, zero between launches), *pos_dev = p + 1 and *posd_dev = (p + 1) * D.
 * tok_emb: emb_dtype WIPA_F32 / WIPA_BF16 / WIPA_FP8_E4M3 (codes, with emb_scale [V] f32).  TextDecoder's
 * token_embedding + positional_embedding + blocks[0].attn_ln of the NEXT position (mlx_whisper; transcribe_single.py:55). */
int wipa_greedy_step_embed(const float* logits, int64_t ldl, int B, int V, const float* mask_first, const float* mask_always,
                           int32_t* tokens, int64_t ld_tok, int32_t* pos_dev, int64_t* posd_dev, int32_t* done_counter, int n_init,
                           int eot, float* sum_logprobs, int32_t* not_done, const void* tok_emb, int emb_dtype,
                           const float* emb_scale, const float* pos_emb, int n_ctx, float* x, const float* ln_w, const float* ln_b,
                           void* y, int y_dtype, int D, float eps, wipa_stream_t s);
/* The logits projection of a greedy decode step together with the arg-max / log-sum-exp PARTIALS of its filtered rows (round 4):
 * logits[b][v] = x[b] . W[v] (x [B, d] bf16, row stride ldx; W [V, d] bf16 -- TextDecoder's tied token_embedding, mlx_whisper;
 * transcribe_single.py:55) on the persistent wide kernel of wipa_gemm, whose waves also keep, per row, max / arg-max (lowest column
 * on ties) / sum of exponentials of logit + mask over the columns they compute (mask = mask_first when *pos_dev + 1 == n_init, else
 * mask_always; 0 / -inf, f32 [V]) and write one partial per (row, wave): partials [B][3][WIPA_GREEDY_PARTS] f32 (max | sum exp |
 * arg-max as int32).  logits may be NULL: then the 13 MB of logits are neither written nor read back (the steps of a greedy loop
 * whose logits nobody looks at).  bf16, B <= 64, V >= 8192, d in {384, 512, 768, 1024} (wipa_logits_greedy_supported).
 * wipa_greedy_step_embed_partials is wipa_greedy_step_embed on those partials instead of the logits: the same tokens; the
 * log-probability sum differs from the row scan's in the last bits (another summation order, fixed and batch-independent). */
#define WIPA_GREEDY_PARTS 2048
int wipa_logits_greedy_supported(int B, int V, int d, int dtype);
size_t wipa_logits_greedy_partials_bytes(int B);
int wipa_logits_greedy(const void* x, int64_t ldx, const void* w, int64_t ldw, float* logits, int64_t ldl, int B, int V, int d,
                       const float* mask_first, const float* mask_always, const int32_t* pos_dev, int n_init, float* partials,
                       size_t partials_bytes, wipa_stream_t s);
int wipa_greedy_step_embed_partials(const float* partials, int n_parts, int B, int32_t* tokens, int64_t ld_tok, int32_t* pos_dev,
                                    int64_t* posd_dev, int32_t* done_counter, int n_init, int eot, float* sum_logprobs,
                                    int32_t* not_done, const void* tok_emb, int emb_dtype, const float* emb_scale, const float* pos_emb,
                                    int n_ctx, float* x, const float* ln_w, const float* ln_b, void* y, int y_dtype, int D, float eps,
                                    wipa_stream_t s);
/* The same row routine alone, for the token ALREADY at position p = *pos_dev: x[b] = tok_emb[tokens[b][p]] + pos_emb[p],
 * y[b] = LayerNorm(x[b]).  wipa_decoder_run launches it once before its first step. */
int wipa_embed_layernorm(const int32_t* tokens, int64_t ld_tok, int B, const int32_t* pos_dev, const void* tok_emb, int emb_dtype,
                         const float* emb_scale, const float* pos_emb, int n_ctx, float* x, const float* ln_w, const float* ln_b,
                         void* y, int y_dtype, int D, float eps, wipa_stream_t s);

/* ------------------------------------------------------------------ host-side text plumbing (no GPU work)
 * Byte-level BPE of mlx_whisper.tokenizer (tiktoken's CoreBPE) and the token-batch builder of
 * IPADataset._tokenize_ipa_batch (scripts/ipa_data_loader.py:102-131, 146-152).  HOST pointers throughout.
 * The GPT-2 pre-tokeniser regex stays with the caller: encode one pre-split piece at a time.
 * table: n tokens, their bytes concatenated in `blob` (token i has lens[i] bytes) with rank ranks[i]
 *        (multilingual.tiktoken: 50 257 entries; the 256 single bytes must be present). */
typedef struct wipa_bpe wipa_bpe;
wipa_bpe* wipa_bpe_create(const uint8_t* blob, const int32_t* lens, const int32_t* ranks, int n); /* NULL on error */
void wipa_bpe_free(wipa_bpe* b);
/* lowest-rank-first pair merging; returns the number of ids written (<= max_out) or a negative error */
int wipa_bpe_encode_piece(const wipa_bpe* b, const uint8_t* piece, int len, int32_t* out, int max_out);
/* bytes of the ids, concatenated; returns the byte count, a negative WIPA_ERR_*, or -(i+1)-100 if ids[i] is not in the table */
int wipa_bpe_decode(const wipa_bpe* b, const int32_t* ids, int n, uint8_t* out, int max_out);
/* rows = prefix + ids of row r + eot, eot-padded to the longest row; returns the row width (<= ld_out) */
int wipa_build_token_batch(const int32_t* ids, const int32_t* row_lens, int n_rows, const int32_t* prefix, int n_prefix,
                           int32_t eot, int32_t* out, int64_t ld_out);

/* ------------------------------------------------------------------ model runtime
 * Sequencing of the kernels above for a whole encoder / decoder pass, in C++ so the
 * decode loop runs from a hipGraph with no per-kernel host work. */
typedef struct wipa_model_cfg {
    int32_t n_mels, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer;
    int32_t n_vocab, n_text_ctx, n_text_state, n_text_head, n_text_layer;
    int32_t dtype; /* WIPA_F32 or WIPA_BF16: matrices, activations, KV caches */
    int32_t f32_split; /* dtype == WIPA_F32 only: 1 lets the encoder / teacher-forced GEMMs and the encoder flash attention
                        * use split-bf16 products (see wipa_gemm_desc.f32_split); 0 = exact f32 products (default) */
    int32_t dec_w_dtype; /* 0: the decoder matrices have `dtype`.  WIPA_FP8_E4M3 (bf16 models): the matrices the decode step streams
                          * (token embedding = logits matrix, self q|k|v, self out, cross query, cross out, mlp1, mlp2) are OCP
                          * e4m3fn codes with per-row f32 scales appended to the weight table (see WIPA_DEC_FP8_*); wipa_decoder_run
                          * / _prefill read them through the fp8 weight-streaming GEMM.  cross.kv stays bf16 (its projection is a
                          * tile GEMM over B*1500 rows); wipa_decoder_logits refuses an fp8 table. */
    int32_t weights_generation; /* bumped by the caller whenever any pointer of a weight table changes: part of the key of
                                 * the cached decode-step graphs (a freed table's host address may be reused) */
    int32_t enc_act_fp8; /* bf16 models with an fp8 encoder tail (WIPA_ENC_FP8_PER_LAYER): 1 = the encoder's q|k, value, mlp1 and
                          * mlp2 projections (11/12 of its GEMM FLOPs) run fp8 x fp8 on the block-scaled fp8 MFMA: LayerNorm outputs
                          * and the GELU output are quantised per row (wipa_layernorm_fp8 / wipa_rowquant_fp8), the weights are the
                          * e4m3 codes.  Attention, the out projection, the residual stream and everything downstream stay as
                          * they are.  0 = bf16 activations on the dequantised weights (the default; what the parity tests call
                          * "the bf16 model on quantised weights"). */
    int32_t dec_cross_absorbed; /* bf16 models, <= 16 heads, d in {384, 512, 768, 1024}, dec_w_dtype = 0: 1 = the decode step's
                                 * cross-attention streams the encoder output with absorbed key / value projections
                                 * (wipa_cross_absorbed_attention): the state's cross_kv region then holds a copy of the features
                                 * [B, n_audio_ctx, d], wipa_decoder_set_audio runs no projection, and the decoder table carries one
                                 * more entry per layer after the regular blocks: Wk^T [d, d] (WIPA_DEC_ABSORBED_PER_LAYER).  0 =
                                 * cached K / V (every other configuration). */
    int32_t dec_cross_splits; /* dec_cross_absorbed = 1: frame splits per clip of the decode step's streaming launch
                               * (wipa_cross_block_desc.cross_splits): 0 = default (4: shortest lone step), 2 = half-chip launches
                               * for callers that keep several passes in flight on one GPU (whisper_ipa_amd.pipeline sets it).
                               * Part of the step graph's and the prefill graph's key: wipa_decoder_prefill and wipa_decoder_run
                               * use the SAME count, so a prompt pass followed by steps rounds as the steps alone do
                               * (tests/test_gpu_model.py::test_prompt_prefill_equals_stepwise_prompt_absorbed). */
} wipa_model_cfg;

/* Encoder weight table (const void* [WIPA_ENC_GLOBAL + WIPA_ENC_PER_LAYER * n_layer]):
 *   0 conv1.weight [d, K1] T   (mlx layout [d,3,n_mels] flattened, K1 = 3*n_mels padded to the GEMM K multiple)
 *   1 conv1.bias f32   2 conv2.weight [d, 3d] T   3 conv2.bias   4 positional [n_audio_ctx, d] f32
 *   5 ln_post.weight   6 ln_post.bias
 *   per layer: 0 attn_ln.w 1 attn_ln.b 2 qk.w [2d,d] T (query|key) 3 qk.b [2d] (key half zero)
 *              4 value.w [d,d] T 5 value.b 6 out.w 7 out.b 8 mlp_ln.w 9 mlp_ln.b
 *              10 mlp1.w [4d,d] 11 mlp1.b 12 mlp2.w [d,4d] 13 mlp2.b */
#define WIPA_ENC_GLOBAL 7
#define WIPA_ENC_PER_LAYER 14
/* fp8 encoder tables (cfg.enc_act_fp8 = 1) append, after the n_layer regular blocks, per layer:
 *   0 qk codes [2d,d] u8  1 qk scale [2d] f32  2 value codes [d,d]  3 value scale [d]  4 mlp1 codes [4d,d]  5 mlp1 scale [4d]
 *   6 mlp2 codes [d,4d]  7 mlp2 scale [d]      (value = code * scale[row]) */
#define WIPA_ENC_FP8_PER_LAYER 8
/* Decoder weight table (const void* [WIPA_DEC_GLOBAL + WIPA_DEC_PER_LAYER * n_layer]):
 *   0 token_embedding [V, d] T   1 positional_embedding [n_text_ctx, d] f32   2 ln.w   3 ln.b
 *   per layer: 0 attn_ln.w 1 attn_ln.b 2 qkv.w [3d,d] T 3 qkv.b [3d] (key third zero) 4 out.w 5 out.b
 *              6 cross_attn_ln.w 7 cross_attn_ln.b 8 cross.query.w 9 cross.query.b
 *              10 cross.kv.w [2d,d] T (key|value) 11 cross.kv.b [2d] (key half zero) 12 cross.out.w 13 cross.out.b
 *              14 mlp_ln.w 15 mlp_ln.b 16 mlp1.w 17 mlp1.b 18 mlp2.w 19 mlp2.b */
#define WIPA_DEC_GLOBAL 4
#define WIPA_DEC_PER_LAYER 20
/* fp8 decoder tables (cfg.dec_w_dtype = WIPA_FP8_E4M3) append, after the n_layer regular blocks:
 *   token_embedding scale [V] f32, then per layer: qkv scale [3d], out [d], cross.query [d], cross.out [d], mlp1 [4d], mlp2 [d]
 * (value = code * scale[row]); entries 0 and per-layer 2, 4, 8, 12, 16, 18 then point to e4m3 codes. */
#define WIPA_DEC_FP8_PER_LAYER 6
/* absorbed cross-attention tables (cfg.dec_cross_absorbed = 1, never together with fp8 tables) append per layer:
 *   0 cross.key.w transposed, [d_in, d_out] T */
#define WIPA_DEC_ABSORBED_PER_LAYER 1

/* AudioEncoder.__call__ / Whisper.embed_audio (train_whisper_ipa.py:223,
 * transcribe_single.py:54).  mel_padded [B,3002,n_mels] T -> out [B, n_audio_ctx, d] T. */
size_t wipa_encoder_workspace_bytes(const wipa_model_cfg* cfg, int B);
int wipa_encoder_forward(const wipa_model_cfg* cfg, const void* const* weights, const void* mel_padded, void* out,
                         void* workspace, size_t workspace_bytes, int B, wipa_stream_t s);

/* Stage timing (measurement aid, bench.py's MFMA roofline): between wipa_profile_begin(stream) and wipa_profile_end every
 * launch that wipa_encoder_forward and wipa_decoder_set_audio enqueue from THIS host thread on `stream` is bracketed by a
 * pair of HIP events.  wipa_profile_end synchronises the stream and returns the summed kernel time (ms) and launch count per
 * class: [0] GEMM / conv-as-GEMM (MFMA), [1] encoder flash attention, [2] LayerNorm, [3] other. */
#define WIPA_PROFILE_CLASSES 4
int wipa_profile_begin(wipa_stream_t s);
int wipa_profile_end(float* ms_by_class, int* launches_by_class);

/* KV-cached greedy decoding: mlx_whisper.decoding.decode / DecodingTask.run
 * (transcribe_single.py:55, train_whisper_ipa.py:356, evaluate_model.py:200).
 * All decode state lives in ONE caller-owned blob; wipa_decoder_layout gives byte offsets. */
typedef struct wipa_dec_layout {
    int64_t total_bytes;
    int64_t tokens;       /* int32 [B, ld_tok] */
    int64_t ld_tok;       /* elements */
    int64_t pos;          /* int32 scalar: index of the last filled token */
    int64_t not_done;     /* int32 scalar, accumulated by greedy steps */
    int64_t sum_logprobs; /* f32 [B] */
    int64_t logits;       /* f32 [B, ld_logits]: logits of the last step of the last wipa_decoder_run / wipa_decoder_prefill call
                           * (earlier steps of a call may not write them: wipa_logits_greedy) */
    int64_t ld_logits;
    int64_t cross_kv;     /* T [n_layer][B][2H][n_audio_ctx][64] */
    int64_t self_kv;      /* T [n_layer][3][B][n_text_ctx][d]  (slot 0 q staging, 1 K, 2 V) */
    int64_t scratch;
} wipa_dec_layout;
int wipa_decoder_layout(const wipa_model_cfg* cfg, int B, wipa_dec_layout* out);
/* Every entry point that writes into the blob takes `state_bytes`, the size of the caller's allocation, and refuses
 * (WIPA_ERR_ARG) a blob smaller than wipa_decoder_layout(cfg, B).total_bytes: the layout depends on cfg (dtype,
 * dec_cross_absorbed, dec_w_dtype ...), so a blob sized for one configuration must not be reused for another. */
/* cross K/V projection of the encoder output (MultiHeadAttention with xa, computed once). */
int wipa_decoder_set_audio(const wipa_model_cfg* cfg, const void* const* weights, const void* features, void* state,
                           size_t state_bytes, int B, wipa_stream_t s);
/* write the prompt (host int32 [n_init]) to every row, pos = 0, clear counters. */
int wipa_decoder_begin(const wipa_model_cfg* cfg, void* state, size_t state_bytes, int B, const int32_t* initial_tokens_host,
                       int n_init, wipa_stream_t s);
/* n_steps decoder steps (each: one token position through all layers + greedy update).
 * The prompt is consumed one position per step.  use_graph != 0 captures one step into a
 * hipGraph and replays it.  state.logits holds the logits of the call's LAST step (where the logits projection carries the
 * greedy partials -- wipa_logits_greedy -- the other steps do not write them; a caller that wants every step's logits runs
 * one step per call, as decoding.forced_decode_logits does). */
int wipa_decoder_run(const wipa_model_cfg* cfg, const void* const* weights, void* state, size_t state_bytes, int B, int n_init,
                     int eot, const float* mask_first, const float* mask_always, int n_steps, int use_graph,
                     wipa_stream_t s);
/* The first n_init steps (the prompt positions 0..n_init-1 and the first generated token) as ONE batched pass: same
 * resulting state as wipa_decoder_run(..., n_steps = n_init, ...) right after wipa_decoder_begin, but the cached cross K/V
 * are streamed once for all prompt positions and the small per-step kernels run once (mlx_whisper also feeds the whole
 * prompt through the decoder in one forward).  Continue with wipa_decoder_run for the remaining steps. */
int wipa_decoder_prefill(const wipa_model_cfg* cfg, const void* const* w, void* state, size_t state_bytes, int B, int n_init,
                         int eot, const float* mask_first, const float* mask_always, int use_graph, wipa_stream_t s);
/* drop the cached step graphs that reference this state blob (call before freeing it). */
int wipa_decoder_release(void* state);

/* Teacher-forced TextDecoder.__call__ = Whisper.logits (train_whisper_ipa.py:232).
 * tokens [B,T] int32, features [B, n_audio_ctx, d] T -> logits [B*T, ld_logits] f32. */
size_t wipa_decoder_logits_workspace_bytes(const wipa_model_cfg* cfg, int B, int T);
int wipa_decoder_logits(const wipa_model_cfg* cfg, const void* const* weights, const int32_t* tokens, const void* features,
                        float* logits, int64_t ld_logits, void* workspace, size_t workspace_bytes, int B, int T,
                        wipa_stream_t s);

/* ------------------------------------------------------------------ K9 masked CE
 * compute_loss of train_whisper_ipa.py:207-263 on teacher-forced logits.
 * logits [B*T, ldl] f32 (T = tokens_len - 1), tokens [B, ld_tok] int32: input t, target t+1.
 * mask = (tgt != eot) | (cumsum(tgt == eot) == 1).  row_buf f32 [2*B*T] receives the
 * per-row mask*ce and mask; out2[0] = sum(mask*ce), out2[1] = sum(mask) (fixed-order sum). */
int wipa_masked_ce(const float* logits, int64_t ldl, const int32_t* tokens, int64_t ld_tok, int B, int T, int V, int eot,
                   float* row_buf, float* out2, wipa_stream_t s);

/* ------------------------------------------------------------------ fine-tune step (f32)
 * Backward of the teacher-forced decoder + masked CE w.r.t. the decoder parameters, i.e. what
 * nn.value_and_grad(model, loss_fn) computes at scripts/train_whisper_ipa.py:284 with the encoder
 * frozen (:187), then the per-tensor clip (:287-303) and optim.AdamW (:306,513).  The dense
 * contractions are wipa_gemm calls on operands transposed by wipa_transpose; all kernels are
 * deterministic (no float atomics). */
/* out[c][r] = in[r][c]; out columns rows..rows_pad-1 are written as zero (K padding for wipa_gemm). */
int wipa_transpose(const void* in, int64_t ld_in, void* out, int64_t ld_out, int rows, int cols, int rows_pad, int dtype,
                   wipa_stream_t s);
/* out[c] (+)= sum_r x[r][c]  (bias gradients), summed in a fixed order.  workspace (optional, f32): with at least
 * 2*cols floats the rows are cut into up to 64 chunks reduced by separate workgroups (partials in the workspace). */
int wipa_colsum(const float* x, int64_t ld, int rows, int cols, float* out, int accumulate, float* workspace,
                int64_t workspace_floats, wipa_stream_t s);
/* out[i] (+)= sum over k of slabs[k * slab_stride + i], k ascending: the reduction of a split-K wipa_gemm (k_slices > 1). */
int wipa_sum_slabs(const float* slabs, int n_slabs, int64_t slab_stride, float* out, int64_t n, int accumulate, wipa_stream_t s);
/* out[i] = scale * sum over k of slabs[k * slab_stride + i] (k ascending) + (residual ? residual[i] : 0): finishes a split-K GEMM
 * whose slices cannot apply a column scale or a residual themselves (bias rides on slice 0).  residual may alias out. */
int wipa_sum_slabs_ex(const float* slabs, int n_slabs, int64_t slab_stride, float* out, int64_t n, const float* residual,
                      float scale, wipa_stream_t s);
/* LayerNorm backward: dx (+)= ..., dw, db.  stats: f32 scratch of stats_floats elements, at least 2*rows (mean, rstd per row);
 * with room for 2*rows + 64*D the dw / db column sums are cut into up to 32 row chunks reduced by separate workgroups and
 * added in a fixed order.  x, dy, dx contiguous [rows, D]. */
int wipa_layernorm_bwd(const float* x, const float* dy, const float* w, float* dx, int accumulate_dx, float* dw, float* db,
                       float* stats, int64_t stats_floats, int rows, int D, float eps, wipa_stream_t s);
/* exact-erf GELU forward / backward on n (multiple of 4) contiguous f32 values. */
int wipa_gelu(const float* z, float* u, int64_t n, wipa_stream_t s);
int wipa_gelu_bwd(const float* z, const float* du, float* dz, int64_t n, wipa_stream_t s);
/* In place: logits[r][v] <- mask_r * (softmax_r[v] - [v == target_r]) / max(count[0], 1).
 * row_mask = the second half of wipa_masked_ce's row_buf; count = &out2[1] (or its all-reduced value). */
int wipa_masked_ce_bwd(float* logits, int64_t ldl, const int32_t* tokens, int64_t ld_tok, int B, int T, int V,
                       const float* row_mask, const float* count, wipa_stream_t s);
/* d_tok_emb[tokens[r]] += dx[r] (rows visited in order), d_pos_emb[t] = sum_b dx[b*T+t]. tokens_flat int32 [B*T]. */
int wipa_embed_bwd(const int32_t* tokens_flat, const float* dx, int B, int T, int D, float* d_tok_emb, float* d_pos_emb,
                   wipa_stream_t s);
/* Backward of wipa_attention (f32): d is the forward descriptor (q,k,v pre-scaled); out / d_out share the
 * o_* strides; dq,dk,dv share the q/k/v strides; dvec f32 [B,H,Tq] scratch. dq and dk are multiplied by qk_scale. */
int wipa_attention_bwd(const wipa_attn_desc* d, const float* out, const float* d_out, const float* lse, float* dq, float* dk,
                       float* dv, float* dvec, float qk_scale, wipa_stream_t s);
/* Flat multi-tensor optimiser step: per-tensor g *= min(1, max_norm/(|g|+1e-6)), then mlx-style AdamW
 * (no bias correction): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p = p (1 - lr wd) - lr m / (sqrt(v)+eps).
 * Tensors live in flat f32 buffers; a chunk table (<= 4096 elements per chunk, never crossing tensors)
 * drives the kernels: chunk_off/len/seg [n_chunks], seg_first_chunk [n_seg+1]; partial [n_chunks],
 * coef/norms [n_seg] are scratch/outputs.
 * seg_clip int32 [n_seg] or NULL: which tensors the clip reaches.  The reference's clip_grad_dict
 * (scripts/train_whisper_ipa.py:287-303) recurses through dict values only; a list value (decoder.blocks) is neither a
 * dict nor an array and is passed through (:299-300), so a caller reproducing it passes 0 for every decoder.blocks.*
 * tensor (coefficient exactly 1, norm still reported) and 1 for token_embedding.weight, positional_embedding, ln.*.
 * NULL clips every tensor. */
int wipa_clip_adamw(float* params, float* grads, float* m, float* v, const int64_t* chunk_off, const int32_t* chunk_len,
                    const int32_t* chunk_seg, const int32_t* seg_first_chunk, int n_chunks, int n_seg, float* partial,
                    float* coef, float* norms, const int32_t* seg_clip, double max_norm, double lr, double beta1, double beta2,
                    double eps, double weight_decay, wipa_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* WIPA_H */
