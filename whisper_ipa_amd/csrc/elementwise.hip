// K3 LayerNorm, K8 embedding, K13 greedy step, K9 masked cross-entropy, small utilities.
// All HBM-bound row kernels: one wave (LayerNorm) or one workgroup (vocab reductions) per
// row, 8/16-byte vector loads, wave64 shuffle reductions, f32 math.
#include "wipa_common.h"

namespace {

// ------------------------------------------------------------------ LayerNorm
template <typename TI>
__device__ __forceinline__ f32x4 ld4(const TI* p);
template <>
__device__ __forceinline__ f32x4 ld4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <>
__device__ __forceinline__ f32x4 ld4<__bf16>(const __bf16* p) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <typename TO>
__device__ __forceinline__ void st4(TO* p, f32x4 v);
template <>
__device__ __forceinline__ void st4<float>(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <>
__device__ __forceinline__ void st4<__bf16>(__bf16* p, f32x4 v) {
    *reinterpret_cast<bf16x4*>(p) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
}

constexpr int LN_MAXV = 8;  // D <= 8 * 256 = 2048

// NV = ceil(D / 256) vector iterations per lane: a compile-time bound sized to D keeps the register
// count (and so the waves per SIMD) where an HBM-bound kernel needs it -- with the generic bound of 8
// hipcc hoisted every load and allocated 208 VGPRs = 2 waves/SIMD = 2.1 TB/s on the encoder rows.
template <typename TI, typename TO, int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const TI* __restrict__ x, int64_t ldx, TO* __restrict__ y,
                                                        int64_t ldy, const float* __restrict__ w,
                                                        const float* __restrict__ b, int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const TI* xr = x + (int64_t)row * ldx;
    TO* yr = y + (int64_t)row * ldy;
    f32x4 v[NV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < D) {
            v[i] = ld4<TI>(xr + c);
            sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        } else {
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const float mean = wave_reduce_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < D) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[i][e] - mean;
                sq += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_reduce_sum(sq) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < D) {
            const f32x4 ww = *reinterpret_cast<const f32x4*>(w + c);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(b + c);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * ww[e] + bb[e];
            st4<TO>(yr + c, o);
        }
    }
}

// Decode-step variant: ONE workgroup per row (few rows -> spread them over many CUs), fused with
// the fixed-order sum of the split-K partial slabs of the preceding residual GEMM.
constexpr int LN_MAX_SLABS = 16;  // NS = 4 (the split-K residual GEMMs) or 16 (one slab per head from the fused out projections)
template <typename TO, int NS = 4>
__global__ __launch_bounds__(256) void add_slabs_layernorm_kernel(float* __restrict__ x, int64_t ldx,
                                                                   const float* __restrict__ slabs, int n_slabs,
                                                                   int64_t slab_stride, TO* __restrict__ y, int64_t ldy,
                                                                   const float* __restrict__ w, const float* __restrict__ b,
                                                                   int D, float eps) {
    __shared__ float s_red[4];
    const int row = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* xr = x + (int64_t)row * ldx;
    f32x4 v[2], ww[2], bb[2];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid * 4 + 1024 * i;
        v[i] = ww[i] = bb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < D) {
            // every load of the row (x, up to NS slabs, w, b) is issued before the first use:
            // one memory round trip instead of one per slab
            v[i] = *reinterpret_cast<const f32x4*>(xr + c);
            f32x4 sl[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s)
                sl[s] = (s < n_slabs) ? *reinterpret_cast<const f32x4*>(slabs + (int64_t)s * slab_stride + (int64_t)row * ldx + c)
                                      : f32x4{0.f, 0.f, 0.f, 0.f};
            ww[i] = *reinterpret_cast<const f32x4*>(w + c);
            bb[i] = *reinterpret_cast<const f32x4*>(b + c);
#pragma unroll
            for (int s = 0; s < NS; ++s)
                if (s < n_slabs) v[i] += sl[s];  // fixed order s = 0, 1, ...
            if (n_slabs > 0) *reinterpret_cast<f32x4*>(xr + c) = v[i];
            sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    sum = wave_reduce_sum(sum);
    if (lane == 0) s_red[wave] = sum;
    __syncthreads();
    const float mean = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) / (float)D;
    __syncthreads();
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid * 4 + 1024 * i;
        if (c < D) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[i][e] - mean;
                sq += d * d;
            }
        }
    }
    sq = wave_reduce_sum(sq);
    if (lane == 0) s_red[wave] = sq;
    __syncthreads();
    const float rstd = rsqrtf(((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) / (float)D + eps);
    TO* yr = y + (int64_t)row * ldy;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid * 4 + 1024 * i;
        if (c < D) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * ww[i][e] + bb[i][e];
            st4<TO>(yr + c, o);
        }
    }
}

// ------------------------------------------------------------------ embedding
template <typename TE>
__global__ __launch_bounds__(256) void embed_kernel(const int32_t* __restrict__ tokens, int64_t ld_tok, int T, int t_start,
                                                    const int32_t* __restrict__ pos_dev, const TE* __restrict__ emb,
                                                    const float* __restrict__ pos_emb, float* __restrict__ x, int D) {
    const int row = blockIdx.x;  // b*T + t
    const int b = row / T, t = row - b * T;
    const int p = t_start + (pos_dev ? *pos_dev : 0) + t;
    const int tok = tokens[(int64_t)b * ld_tok + p];
    const TE* e = emb + (int64_t)tok * D;
    const float* pe = pos_emb + (int64_t)p * D;
    float* xr = x + (int64_t)row * D;
    for (int c = threadIdx.x * 4; c < D; c += 1024) {
        const f32x4 a = ld4<TE>(e + c);
        const f32x4 q = *reinterpret_cast<const f32x4*>(pe + c);
        *reinterpret_cast<f32x4*>(xr + c) = a + q;
    }
}

// fp8 (e4m3fn) token embedding: row `tok` of the codes times its per-row scale (the same matrix and scales serve as the
// logits weights, where the scale is per output column)
__global__ __launch_bounds__(256) void embed_fp8_kernel(const int32_t* __restrict__ tokens, int64_t ld_tok, int T, int t_start,
                                                        const int32_t* __restrict__ pos_dev, const unsigned char* __restrict__ emb,
                                                        const float* __restrict__ emb_scale, const float* __restrict__ pos_emb,
                                                        float* __restrict__ x, int D) {
    const int row = blockIdx.x;
    const int b = row / T, t = row - b * T;
    const int p = t_start + (pos_dev ? *pos_dev : 0) + t;
    const int tok = tokens[(int64_t)b * ld_tok + p];
    const unsigned char* e = emb + (int64_t)tok * D;
    const float sc = emb_scale[tok];
    const float* pe = pos_emb + (int64_t)p * D;
    float* xr = x + (int64_t)row * D;
    for (int c = threadIdx.x * 4; c < D; c += 1024) {
        const unsigned int u = *reinterpret_cast<const unsigned int*>(e + c);
        const auto lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)u, false);
        const auto hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)u, true);
        const f32x4 q = *reinterpret_cast<const f32x4*>(pe + c);
        // the dequantised value is rounded to bf16 like every stored weight of the bf16 model (code * scale is not exact in bf16)
        const f32x4 a = {(float)(__bf16)(lo[0] * sc), (float)(__bf16)(lo[1] * sc), (float)(__bf16)(hi[0] * sc), (float)(__bf16)(hi[1] * sc)};
        *reinterpret_cast<f32x4*>(xr + c) = a + q;
    }
}

// ------------------------------------------------------------------ block reductions
struct MaxIdx {
    float v;
    int i;
};
__device__ __forceinline__ MaxIdx better(MaxIdx a, MaxIdx b) {
    // larger value wins; on ties the LOWER index (argmax returns the first maximum)
    if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
    return a;
}
__device__ __forceinline__ MaxIdx wave_argmax(MaxIdx m) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        MaxIdx other;
        other.v = __shfl_xor(m.v, o, 64);
        other.i = __shfl_xor(m.i, o, 64);
        m = better(m, other);
    }
    return m;
}

constexpr int GS_THREADS = 1024;

__global__ __launch_bounds__(GS_THREADS) void greedy_step_kernel(const float* __restrict__ logits, int64_t ldl, int V,
                                                                 const float* __restrict__ mask_first,
                                                                 const float* __restrict__ mask_always,
                                                                 int32_t* __restrict__ tokens, int64_t ld_tok,
                                                                 const int32_t* __restrict__ pos_dev, int n_init, int eot,
                                                                 float* __restrict__ sum_logprobs,
                                                                 int32_t* __restrict__ not_done) {
    __shared__ float s_v[GS_THREADS / 64];
    __shared__ int s_i[GS_THREADS / 64];
    __shared__ float s_sum[GS_THREADS / 64];
    const int b = blockIdx.x;
    const int p = *pos_dev;
    if (p + 1 < n_init) return;  // prompt token already in place
    const float* mask = (p + 1 == n_init) ? mask_first : mask_always;
    const float* row = logits + (int64_t)b * ldl;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    MaxIdx m{-INFINITY, 0x7fffffff};
    for (int i = tid; i < V; i += GS_THREADS) {
        const float v = row[i] + mask[i];
        m = better(m, MaxIdx{v, i});
    }
    m = wave_argmax(m);
    if (lane == 0) {
        s_v[wave] = m.v;
        s_i[wave] = m.i;
    }
    __syncthreads();
    MaxIdx bm{s_v[0], s_i[0]};
#pragma unroll
    for (int w = 1; w < GS_THREADS / 64; ++w) bm = better(bm, MaxIdx{s_v[w], s_i[w]});
    float se = 0.f;
    for (int i = tid; i < V; i += GS_THREADS) se += __expf(row[i] + mask[i] - bm.v);
    se = wave_reduce_sum(se);
    if (lane == 0) s_sum[wave] = se;
    __syncthreads();
    if (tid == 0) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < GS_THREADS / 64; ++w) tot += s_sum[w];
        const int prev = tokens[(int64_t)b * ld_tok + p];
        int next = bm.i;
        if (prev == eot) {
            next = eot;
        } else {
            sum_logprobs[b] += -logf(tot);  // log_softmax at the argmax = -(log sum exp(x - max))
        }
        tokens[(int64_t)b * ld_tok + p + 1] = next;
        if (next != eot) atomicAdd(not_done, 1);
    }
}

// Register-resident variant for V <= 65536 (every Whisper vocabulary): the filtered row is fetched ONCE with 16-byte loads
// that are all in flight together (13 per thread for V = 51 865), the arg-max and the sum of exponentials both run out of
// registers.  The two-pass scalar kernel above took 30 us per step on 64 rows (latency-bound: 2 x 51 dependent loads).
constexpr int GS_MAXQ = 16;
__global__ __launch_bounds__(GS_THREADS) void greedy_step_reg_kernel(const float* __restrict__ logits, int64_t ldl, int V,
                                                                     const float* __restrict__ mask_first,
                                                                     const float* __restrict__ mask_always,
                                                                     int32_t* __restrict__ tokens, int64_t ld_tok,
                                                                     const int32_t* __restrict__ pos_dev, int n_init, int eot,
                                                                     float* __restrict__ sum_logprobs,
                                                                     int32_t* __restrict__ not_done) {
    __shared__ float s_v[GS_THREADS / 64];
    __shared__ int s_i[GS_THREADS / 64];
    __shared__ float s_sum[GS_THREADS / 64];
    const int b = blockIdx.x;
    const int p = *pos_dev;
    if (p + 1 < n_init) return;  // prompt token already in place
    const float* mask = (p + 1 == n_init) ? mask_first : mask_always;
    const float* row = logits + (int64_t)b * ldl;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nq = V >> 2;
    f32x4 vals[GS_MAXQ];
#pragma unroll
    for (int j = 0; j < GS_MAXQ; ++j) {
        const int qi = tid + j * GS_THREADS;
        if (qi < nq) {
            vals[j] = *reinterpret_cast<const f32x4*>(row + 4 * qi) + *reinterpret_cast<const f32x4*>(mask + 4 * qi);
        } else {
            vals[j] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
    }
    const int ti = 4 * nq + tid;  // the (V mod 4) trailing elements
    const bool has_tail = tid < 4 && ti < V;
    const float tailv = has_tail ? row[ti] + mask[ti] : -INFINITY;
    MaxIdx m{tailv, has_tail ? ti : 0x7fffffff};
#pragma unroll
    for (int j = 0; j < GS_MAXQ; ++j) {
        const int qi = tid + j * GS_THREADS;
#pragma unroll
        for (int e = 0; e < 4; ++e) m = better(m, MaxIdx{vals[j][e], qi < nq ? 4 * qi + e : 0x7fffffff});
    }
    m = wave_argmax(m);
    if (lane == 0) {
        s_v[wave] = m.v;
        s_i[wave] = m.i;
    }
    __syncthreads();
    MaxIdx bm{s_v[0], s_i[0]};
#pragma unroll
    for (int w = 1; w < GS_THREADS / 64; ++w) bm = better(bm, MaxIdx{s_v[w], s_i[w]});
    float se = __expf(tailv - bm.v);
#pragma unroll
    for (int j = 0; j < GS_MAXQ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) se += __expf(vals[j][e] - bm.v);
    se = wave_reduce_sum(se);
    if (lane == 0) s_sum[wave] = se;
    __syncthreads();
    if (tid == 0) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < GS_THREADS / 64; ++w) tot += s_sum[w];
        const int prev = tokens[(int64_t)b * ld_tok + p];
        int next = bm.i;
        if (prev == eot) {
            next = eot;
        } else {
            sum_logprobs[b] += -logf(tot);
        }
        tokens[(int64_t)b * ld_tok + p + 1] = next;
        if (next != eot) atomicAdd(not_done, 1);
    }
}

// ------------------------------------------------------------------ decode-step tail / head (round 4)
// A decode step used to end with greedy_step + advance_pos and the next one to begin with embed + the first LayerNorm: four
// launches of 4-16 us whose only content is a few dependent round trips.  greedy_tail_kernel does all of it in the launch that
// already owns the row: arg-max / log-prob of the filtered logits (greedy_step_reg_kernel's arithmetic, unchanged), the EOT
// latch, then x = tok_emb[next] + pos_emb[p + 1] and LayerNorm(x) with the first block's attn_ln for position p + 1, and -- by
// the LAST workgroup to finish (device counter) -- the position advance.  embed_layernorm_kernel is the same row routine alone:
// wipa_decoder_run launches it once before the first step (the prompt walk, forced histories and the first step after a
// prefill start from tokens the tail has not embedded).  The row routine repeats add_slabs_layernorm_kernel's reduction order
// (256 threads, float4 per thread, wave shuffle tree, ((s0+s1)+(s2+s3))), so fused and unfused steps agree bit for bit.
__device__ __forceinline__ f32x4 emb_row_ld4(const void* emb, int emb_dtype, const float* emb_scale, int tok, int D, int c) {
    if (emb_dtype == WIPA_F32) return *reinterpret_cast<const f32x4*>((const float*)emb + (int64_t)tok * D + c);
    if (emb_dtype == WIPA_BF16) return ld4<__bf16>((const __bf16*)emb + (int64_t)tok * D + c);
    // e4m3fn codes x per-row scale, rounded to bf16 like embed_fp8_kernel
    const unsigned int u = *reinterpret_cast<const unsigned int*>((const unsigned char*)emb + (int64_t)tok * D + c);
    const float sc = emb_scale[tok];
    const auto lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)u, false);
    const auto hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)u, true);
    return f32x4{(float)(__bf16)(lo[0] * sc), (float)(__bf16)(lo[1] * sc), (float)(__bf16)(hi[0] * sc), (float)(__bf16)(hi[1] * sc)};
}

// threads 0..255 of the workgroup embed token `tok` at position `pp` into xr[D] and write LayerNorm(xr) to yr[D]; EVERY thread
// of the workgroup must call it (three workgroup barriers); s_red: 4 floats of LDS
template <typename TO>
__device__ __forceinline__ void row_embed_layernorm(int tid, int tok, int pp, const void* emb, int emb_dtype, const float* emb_scale,
                                                    const float* pos_emb, float* xr, const float* w, const float* b, TO* yr, int D,
                                                    float eps, float* s_red) {
    const int lane = tid & 63, wave = tid >> 6;
    const bool active = tid < 256;
    f32x4 v[2], ww[2], bb[2];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid * 4 + 1024 * i;
        v[i] = ww[i] = bb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (active && c < D) {
            const f32x4 a = emb_row_ld4(emb, emb_dtype, emb_scale, tok, D, c);
            const f32x4 q = *reinterpret_cast<const f32x4*>(pos_emb + (int64_t)pp * D + c);
            ww[i] = *reinterpret_cast<const f32x4*>(w + c);
            bb[i] = *reinterpret_cast<const f32x4*>(b + c);
            v[i] = a + q;
            *reinterpret_cast<f32x4*>(xr + c) = v[i];
            sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    sum = wave_reduce_sum(sum);
    if (active && lane == 0) s_red[wave] = sum;
    __syncthreads();
    const float mean = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) / (float)D;
    __syncthreads();
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid * 4 + 1024 * i;
        if (active && c < D) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[i][e] - mean;
                sq += d * d;
            }
        }
    }
    sq = wave_reduce_sum(sq);
    if (active && lane == 0) s_red[wave] = sq;
    __syncthreads();
    const float rstd = rsqrtf(((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid * 4 + 1024 * i;
        if (active && c < D) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * ww[i][e] + bb[i][e];
            st4<TO>(yr + c, o);
        }
    }
}

struct TailParams {
    const float* logits; int64_t ldl; int V;
    const float* mask_first; const float* mask_always;
    int32_t* tokens; int64_t ld_tok;
    int32_t* pos; int64_t* posd; int32_t* done_counter;
    int n_init, eot, n_ctx;
    float* sum_logprobs; int32_t* not_done;
    const void* emb; int emb_dtype; const float* emb_scale; const float* pos_emb;
    float* x; const float* ln_w; const float* ln_b; void* y; int D; float eps;
    const float* part; int n_part;  // wipa_logits_greedy's partials [B][3][n_part] instead of the logits (part != nullptr)
};

template <typename TO>
__global__ __launch_bounds__(256) void embed_layernorm_kernel(TailParams q) {
    __shared__ float s_red[4];
    const int b = blockIdx.x;
    const int p = *q.pos;
    const int tok = q.tokens[(int64_t)b * q.ld_tok + p];
    row_embed_layernorm<TO>(threadIdx.x, tok, min(p, q.n_ctx - 1), q.emb, q.emb_dtype, q.emb_scale, q.pos_emb, q.x + (int64_t)b * q.D, q.ln_w,
                            q.ln_b, (TO*)q.y + (int64_t)b * q.D, q.D, q.eps, s_red);
}

template <typename TO>
__global__ __launch_bounds__(GS_THREADS) void greedy_tail_kernel(TailParams q) {
    __shared__ float s_v[GS_THREADS / 64];
    __shared__ int s_i[GS_THREADS / 64];
    __shared__ float s_sum[GS_THREADS / 64];
    __shared__ float s_red[4];
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // volatile: ONE load per thread at the kernel's start, never re-materialised by the compiler after the barrier below -- the
    // last workgroup to arrive overwrites *q.pos while slower workgroups are still between their barrier and their exit
    const int p = *(volatile const int32_t*)q.pos;
    const int V = q.V;
    // `next` is computed by EVERY thread from values all of them hold (no single-thread section that hands a value to the
    // workgroup through LDS: the pattern whose merge-kernel instance misbehaved); thread 0 alone writes
    int next;
    if (p + 1 < q.n_init) {  // prompt walk: the token is already in place (uniform branch: p is the same for every thread)
        next = q.tokens[(int64_t)b * q.ld_tok + p + 1];
    } else if (q.part) {
        // the logits GEMM left one (max, sum exp, arg-max) per wave of its grid for this row: merge them -- (value, lowest column)
        // for the arg-max, then sum_i s_i exp(m_i - M) in a fixed order (per thread ascending, wave tree, waves in order)
        const float* pm = q.part + (int64_t)b * 3 * q.n_part;
        const float* ps = pm + q.n_part;
        const int* pi = reinterpret_cast<const int*>(pm + 2 * q.n_part);
        MaxIdx m{-INFINITY, 0x7fffffff};
        for (int i = tid; i < q.n_part; i += GS_THREADS) m = better(m, MaxIdx{pm[i], pi[i]});
        m = wave_argmax(m);
        if (lane == 0) {
            s_v[wave] = m.v;
            s_i[wave] = m.i;
        }
        __syncthreads();
        MaxIdx bm{s_v[0], s_i[0]};
#pragma unroll
        for (int w = 1; w < GS_THREADS / 64; ++w) bm = better(bm, MaxIdx{s_v[w], s_i[w]});
        float se = 0.f;
        for (int i = tid; i < q.n_part; i += GS_THREADS) se += ps[i] * __expf(pm[i] - bm.v);
        se = wave_reduce_sum(se);
        if (lane == 0) s_sum[wave] = se;
        __syncthreads();
        const int prev = q.tokens[(int64_t)b * q.ld_tok + p];
        next = (prev == q.eot) ? q.eot : bm.i;
        if (tid == 0) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < GS_THREADS / 64; ++w) tot += s_sum[w];
            if (prev != q.eot) q.sum_logprobs[b] += -logf(tot);
            q.tokens[(int64_t)b * q.ld_tok + p + 1] = next;
            if (next != q.eot) atomicAdd(q.not_done, 1);
        }
    } else {
        const float* mask = (p + 1 == q.n_init) ? q.mask_first : q.mask_always;
        const float* row = q.logits + (int64_t)b * q.ldl;
        const int nq = V >> 2;
        f32x4 vals[GS_MAXQ];
#pragma unroll
        for (int j = 0; j < GS_MAXQ; ++j) {
            const int qi = tid + j * GS_THREADS;
            if (qi < nq) {
                vals[j] = *reinterpret_cast<const f32x4*>(row + 4 * qi) + *reinterpret_cast<const f32x4*>(mask + 4 * qi);
            } else {
                vals[j] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            }
        }
        const int ti = 4 * nq + tid;  // the (V mod 4) trailing elements
        const bool has_tail = tid < 4 && ti < V;
        const float tailv = has_tail ? row[ti] + mask[ti] : -INFINITY;
        MaxIdx m{tailv, has_tail ? ti : 0x7fffffff};
#pragma unroll
        for (int j = 0; j < GS_MAXQ; ++j) {
            const int qi = tid + j * GS_THREADS;
#pragma unroll
            for (int e = 0; e < 4; ++e) m = better(m, MaxIdx{vals[j][e], qi < nq ? 4 * qi + e : 0x7fffffff});
        }
        m = wave_argmax(m);
        if (lane == 0) {
            s_v[wave] = m.v;
            s_i[wave] = m.i;
        }
        __syncthreads();
        MaxIdx bm{s_v[0], s_i[0]};
#pragma unroll
        for (int w = 1; w < GS_THREADS / 64; ++w) bm = better(bm, MaxIdx{s_v[w], s_i[w]});
        float se = __expf(tailv - bm.v);
#pragma unroll
        for (int j = 0; j < GS_MAXQ; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) se += __expf(vals[j][e] - bm.v);
        se = wave_reduce_sum(se);
        if (lane == 0) s_sum[wave] = se;
        __syncthreads();
        const int prev = q.tokens[(int64_t)b * q.ld_tok + p];
        next = (prev == q.eot) ? q.eot : bm.i;
        if (tid == 0) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < GS_THREADS / 64; ++w) tot += s_sum[w];
            if (prev != q.eot) q.sum_logprobs[b] += -logf(tot);
            q.tokens[(int64_t)b * q.ld_tok + p + 1] = next;
            if (next != q.eot) atomicAdd(q.not_done, 1);
        }
    }
    // the next step's input row: embedding of the chosen token at position p + 1, then the first block's LayerNorm
    row_embed_layernorm<TO>(tid, next, min(p + 1, q.n_ctx - 1), q.emb, q.emb_dtype, q.emb_scale, q.pos_emb, q.x + (int64_t)b * q.D, q.ln_w,
                            q.ln_b, (TO*)q.y + (int64_t)b * q.D, q.D, q.eps, s_red);
    // position advance by the LAST workgroup: every thread of every workgroup read *pos at its start and has used it before
    // its workgroup's barrier above, i.e. before its counter increment -- no workgroup can still see the old position late.
    // Needs *done_counter == 0 at launch and one launch in flight per state blob: wipa_decoder_run / _prefill zero the counter
    // at the start of every call (a launch that died mid-grid must not leave later calls without a position advance)
    if (tid == 0) {
        const int arrived = atomicAdd(q.done_counter, 1);  // counts arrivals only: what the step wrote reaches the next launch at the kernel boundary
        if (arrived == (int)gridDim.x - 1) {
            *q.done_counter = 0;
            *q.pos = p + 1;
            *q.posd = (int64_t)(p + 1) * q.D;
        }
    }
}

__global__ void add_i32_kernel(int32_t* p, int32_t v) { *p += v; }

// ------------------------------------------------------------------ masked cross entropy
constexpr int CE_THREADS = 512;
__global__ __launch_bounds__(CE_THREADS) void masked_ce_rows_kernel(const float* __restrict__ logits, int64_t ldl,
                                                                    const int32_t* __restrict__ tokens, int64_t ld_tok,
                                                                    int T, int V, int eot, float* __restrict__ row_buf,
                                                                    int rows) {
    __shared__ float s_red[CE_THREADS / 64];
    const int r = blockIdx.x;
    const int b = r / T, t = r - b * T;
    const int32_t* tk = tokens + (int64_t)b * ld_tok;
    const int tgt = tk[t + 1];
    const float* row = logits + (int64_t)r * ldl;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float mx = -INFINITY;
    for (int i = tid; i < V; i += CE_THREADS) mx = fmaxf(mx, row[i]);
    mx = wave_reduce_max(mx);
    if (lane == 0) s_red[wave] = mx;
    __syncthreads();
    mx = s_red[0];
#pragma unroll
    for (int w = 1; w < CE_THREADS / 64; ++w) mx = fmaxf(mx, s_red[w]);
    __syncthreads();
    float se = 0.f;
    for (int i = tid; i < V; i += CE_THREADS) se += __expf(row[i] - mx);
    se = wave_reduce_sum(se);
    if (lane == 0) s_red[wave] = se;
    __syncthreads();
    if (tid == 0) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < CE_THREADS / 64; ++w) tot += s_red[w];
        const float ce = logf(tot) + mx - row[tgt];
        // mask = (tgt != eot) | (cumsum(tgt == eot) == 1)      (train_whisper_ipa.py:242-247)
        bool keep = true;
        if (tgt == eot) {
            int c = 0;
            for (int u = 0; u <= t; ++u) c += (tk[u + 1] == eot);
            keep = (c == 1);
        }
        row_buf[r] = keep ? ce : 0.f;
        row_buf[rows + r] = keep ? 1.f : 0.f;
    }
}

__global__ __launch_bounds__(256) void sum2_kernel(const float* __restrict__ row_buf, int rows, float* __restrict__ out2) {
    // deterministic: fixed-order tree over a single workgroup
    __shared__ float s_a[256], s_b[256];
    float a = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < rows; i += 256) {
        a += row_buf[i];
        c += row_buf[rows + i];
    }
    s_a[threadIdx.x] = a;
    s_b[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            s_a[threadIdx.x] += s_a[threadIdx.x + o];
            s_b[threadIdx.x] += s_b[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out2[0] = s_a[0];
        out2[1] = s_b[0];
    }
}

// ------------------------------------------------------------------ mel pad + cast
template <typename TO>
__global__ __launch_bounds__(256) void mel_pad_cast_kernel(const float* __restrict__ mel, int n_mels, TO* __restrict__ out,
                                                           int64_t total) {
    // out [B, 3002, n_mels]; row 0 and 3001 zero, row t+1 = mel[b][t]
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int64_t per = (int64_t)(WIPA_N_FRAMES + 2) * n_mels;
    const int64_t b = i / per;
    const int64_t r = i - b * per;
    const int row = (int)(r / n_mels);
    float v = 0.f;
    if (row >= 1 && row <= WIPA_N_FRAMES) v = mel[b * (int64_t)WIPA_N_FRAMES * n_mels + (r - n_mels)];
    out[i] = from_f32<TO>(v);
}

}  // namespace

extern "C" int wipa_layernorm(const void* x, int x_dtype, int64_t ldx, void* y, int y_dtype, int64_t ldy, const float* w,
                              const float* b, int rows, int D, float eps, wipa_stream_t stream) {
    WIPA_REQUIRE(x && y && w && b, "wipa_layernorm: null pointer");
    WIPA_REQUIRE(D % 4 == 0 && D <= LN_MAXV * 256 && D > 0, "wipa_layernorm: D=%d must be a multiple of 4 and <= %d", D,
                 LN_MAXV * 256);
    WIPA_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0, "wipa_layernorm: ldx/ldy must be multiples of 4");
    if (rows <= 0) return WIPA_OK;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((rows + 3) / 4), block(256);
    const int nv = (D + 255) / 256;
#define LN_LAUNCH_NV(TI, TO, NV) \
    hipLaunchKernelGGL((layernorm_kernel<TI, TO, NV>), grid, block, 0, s, (const TI*)x, ldx, (TO*)y, ldy, w, b, rows, D, eps)
#define LN_LAUNCH(TI, TO)                      \
    do {                                       \
        if (nv <= 1) LN_LAUNCH_NV(TI, TO, 1);      \
        else if (nv <= 2) LN_LAUNCH_NV(TI, TO, 2); \
        else if (nv <= 3) LN_LAUNCH_NV(TI, TO, 3); \
        else if (nv <= 4) LN_LAUNCH_NV(TI, TO, 4); \
        else if (nv <= 5) LN_LAUNCH_NV(TI, TO, 5); \
        else LN_LAUNCH_NV(TI, TO, 8);              \
    } while (0)
    if (x_dtype == WIPA_F32 && y_dtype == WIPA_F32) LN_LAUNCH(float, float);
    else if (x_dtype == WIPA_F32 && y_dtype == WIPA_BF16) LN_LAUNCH(float, __bf16);
    else if (x_dtype == WIPA_BF16 && y_dtype == WIPA_BF16) LN_LAUNCH(__bf16, __bf16);
    else if (x_dtype == WIPA_BF16 && y_dtype == WIPA_F32) LN_LAUNCH(__bf16, float);
    else WIPA_REQUIRE(false, "wipa_layernorm: bad dtypes %d %d", x_dtype, y_dtype);
#undef LN_LAUNCH
#undef LN_LAUNCH_NV
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

extern "C" int wipa_add_slabs_layernorm(float* x, int64_t ldx, const float* slabs, int n_slabs, int64_t slab_stride, void* y,
                                        int y_dtype, int64_t ldy, const float* w, const float* b, int rows, int D, float eps,
                                        wipa_stream_t stream) {
    WIPA_REQUIRE(x && y && w && b && (slabs || n_slabs == 0), "wipa_add_slabs_layernorm: null pointer");
    WIPA_REQUIRE(n_slabs >= 0 && n_slabs <= LN_MAX_SLABS, "wipa_add_slabs_layernorm: n_slabs=%d (max %d)", n_slabs, LN_MAX_SLABS);
    WIPA_REQUIRE(D % 4 == 0 && D > 0 && D <= 2048, "wipa_add_slabs_layernorm: D=%d must be a multiple of 4 and <= 2048", D);
    WIPA_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && slab_stride % 4 == 0, "wipa_add_slabs_layernorm: strides must be multiples of 4");
    if (rows <= 0) return WIPA_OK;
    hipStream_t s = (hipStream_t)stream;
    WIPA_REQUIRE(y_dtype == WIPA_F32 || y_dtype == WIPA_BF16, "wipa_add_slabs_layernorm: bad dtype %d", y_dtype);
#define LN_SLABS(TO, NS)                                                                                                          \
    hipLaunchKernelGGL((add_slabs_layernorm_kernel<TO, NS>), dim3(rows), dim3(256), 0, s, x, ldx, slabs, n_slabs, slab_stride, (TO*)y, \
                       ldy, w, b, D, eps)
    if (y_dtype == WIPA_F32) {
        if (n_slabs <= 4) LN_SLABS(float, 4);
        else LN_SLABS(float, 16);
    } else {
        if (n_slabs <= 4) LN_SLABS(__bf16, 4);
        else LN_SLABS(__bf16, 16);
    }
#undef LN_SLABS
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

extern "C" int wipa_embed_tokens(const int32_t* tokens, int64_t ld_tok, int B, int T, int t_start, const int32_t* pos_dev,
                                 const void* tok_emb, int emb_dtype, const float* emb_scale, const float* pos_emb, float* x, int D,
                                 wipa_stream_t stream) {
    WIPA_REQUIRE(tokens && tok_emb && pos_emb && x, "wipa_embed_tokens: null pointer");
    WIPA_REQUIRE(emb_dtype != WIPA_FP8_E4M3 || emb_scale, "wipa_embed_tokens: fp8 embedding needs emb_scale");
    WIPA_REQUIRE(D % 4 == 0, "wipa_embed_tokens: D must be a multiple of 4");
    if (B * T <= 0) return WIPA_OK;
    hipStream_t s = (hipStream_t)stream;
    if (emb_dtype == WIPA_F32)
        hipLaunchKernelGGL((embed_kernel<float>), dim3(B * T), dim3(256), 0, s, tokens, ld_tok, T, t_start, pos_dev,
                           (const float*)tok_emb, pos_emb, x, D);
    else if (emb_dtype == WIPA_BF16)
        hipLaunchKernelGGL((embed_kernel<__bf16>), dim3(B * T), dim3(256), 0, s, tokens, ld_tok, T, t_start, pos_dev,
                           (const __bf16*)tok_emb, pos_emb, x, D);
    else if (emb_dtype == WIPA_FP8_E4M3)
        hipLaunchKernelGGL(embed_fp8_kernel, dim3(B * T), dim3(256), 0, s, tokens, ld_tok, T, t_start, pos_dev,
                           (const unsigned char*)tok_emb, emb_scale, pos_emb, x, D);
    else
        WIPA_REQUIRE(false, "wipa_embed_tokens: bad dtype %d", emb_dtype);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

extern "C" int wipa_greedy_step(const float* logits, int64_t ldl, int B, int V, const float* mask_first,
                                const float* mask_always, int32_t* tokens, int64_t ld_tok, const int32_t* pos_dev,
                                int n_init, int eot, float* sum_logprobs, int32_t* not_done, wipa_stream_t stream) {
    WIPA_REQUIRE(logits && mask_first && mask_always && tokens && pos_dev && sum_logprobs && not_done,
                 "wipa_greedy_step: null pointer");
    const bool reg_ok = V <= 4 * GS_MAXQ * GS_THREADS && ldl % 4 == 0 && ((uintptr_t)logits % 16) == 0 &&
                        ((uintptr_t)mask_first % 16) == 0 && ((uintptr_t)mask_always % 16) == 0;
    if (reg_ok)
        hipLaunchKernelGGL(greedy_step_reg_kernel, dim3(B), dim3(GS_THREADS), 0, (hipStream_t)stream, logits, ldl, V, mask_first,
                           mask_always, tokens, ld_tok, pos_dev, n_init, eot, sum_logprobs, not_done);
    else
        hipLaunchKernelGGL(greedy_step_kernel, dim3(B), dim3(GS_THREADS), 0, (hipStream_t)stream, logits, ldl, V, mask_first,
                           mask_always, tokens, ld_tok, pos_dev, n_init, eot, sum_logprobs, not_done);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

static int tail_params_check(const char* who, const float* logits, int V, int64_t ldl, const float* mask_first, const float* mask_always,
                             int D, int emb_dtype, const float* emb_scale) {
    WIPA_REQUIRE(D % 4 == 0 && D > 0 && D <= 2048, "%s: D=%d must be a multiple of 4 and <= 2048", who, D);
    WIPA_REQUIRE(emb_dtype == WIPA_F32 || emb_dtype == WIPA_BF16 || (emb_dtype == WIPA_FP8_E4M3 && emb_scale), "%s: bad embedding dtype %d", who, emb_dtype);
    if (logits)
        WIPA_REQUIRE(V <= 4 * GS_MAXQ * GS_THREADS && ldl % 4 == 0 && ((uintptr_t)logits % 16) == 0 && ((uintptr_t)mask_first % 16) == 0 &&
                         ((uintptr_t)mask_always % 16) == 0, "%s: vocabulary of %d / unaligned logits or masks", who, V);
    return WIPA_OK;
}

extern "C" int wipa_embed_layernorm(const int32_t* tokens, int64_t ld_tok, int B, const int32_t* pos_dev, const void* tok_emb,
                                    int emb_dtype, const float* emb_scale, const float* pos_emb, int n_ctx, float* x, const float* ln_w,
                                    const float* ln_b, void* y, int y_dtype, int D, float eps, wipa_stream_t stream) {
    WIPA_REQUIRE(tokens && pos_dev && tok_emb && pos_emb && x && ln_w && ln_b && y && B > 0 && n_ctx > 0, "wipa_embed_layernorm: bad arguments");
    const int rc = tail_params_check("wipa_embed_layernorm", nullptr, 0, 0, nullptr, nullptr, D, emb_dtype, emb_scale);
    if (rc != WIPA_OK) return rc;
    TailParams q = {};
    q.tokens = const_cast<int32_t*>(tokens); q.ld_tok = ld_tok; q.pos = const_cast<int32_t*>(pos_dev); q.n_ctx = n_ctx;
    q.emb = tok_emb; q.emb_dtype = emb_dtype; q.emb_scale = emb_scale; q.pos_emb = pos_emb;
    q.x = x; q.ln_w = ln_w; q.ln_b = ln_b; q.y = y; q.D = D; q.eps = eps;
    if (y_dtype == WIPA_F32) hipLaunchKernelGGL((embed_layernorm_kernel<float>), dim3(B), dim3(256), 0, (hipStream_t)stream, q);
    else if (y_dtype == WIPA_BF16) hipLaunchKernelGGL((embed_layernorm_kernel<__bf16>), dim3(B), dim3(256), 0, (hipStream_t)stream, q);
    else WIPA_REQUIRE(false, "wipa_embed_layernorm: bad dtype %d", y_dtype);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

extern "C" int wipa_greedy_step_embed(const float* logits, int64_t ldl, int B, int V, const float* mask_first, const float* mask_always,
                                      int32_t* tokens, int64_t ld_tok, int32_t* pos_dev, int64_t* posd_dev, int32_t* done_counter,
                                      int n_init, int eot, float* sum_logprobs, int32_t* not_done, const void* tok_emb, int emb_dtype,
                                      const float* emb_scale, const float* pos_emb, int n_ctx, float* x, const float* ln_w,
                                      const float* ln_b, void* y, int y_dtype, int D, float eps, wipa_stream_t stream) {
    WIPA_REQUIRE(logits && mask_first && mask_always && tokens && pos_dev && posd_dev && done_counter && sum_logprobs && not_done &&
                     tok_emb && pos_emb && x && ln_w && ln_b && y && B > 0 && n_ctx > 0, "wipa_greedy_step_embed: bad arguments");
    const int rc = tail_params_check("wipa_greedy_step_embed", logits, V, ldl, mask_first, mask_always, D, emb_dtype, emb_scale);
    if (rc != WIPA_OK) return rc;
    TailParams q = {};
    q.logits = logits; q.ldl = ldl; q.V = V; q.mask_first = mask_first; q.mask_always = mask_always;
    q.tokens = tokens; q.ld_tok = ld_tok; q.pos = pos_dev; q.posd = posd_dev; q.done_counter = done_counter;
    q.n_init = n_init; q.eot = eot; q.n_ctx = n_ctx; q.sum_logprobs = sum_logprobs; q.not_done = not_done;
    q.emb = tok_emb; q.emb_dtype = emb_dtype; q.emb_scale = emb_scale; q.pos_emb = pos_emb;
    q.x = x; q.ln_w = ln_w; q.ln_b = ln_b; q.y = y; q.D = D; q.eps = eps;
    if (y_dtype == WIPA_F32) hipLaunchKernelGGL((greedy_tail_kernel<float>), dim3(B), dim3(GS_THREADS), 0, (hipStream_t)stream, q);
    else if (y_dtype == WIPA_BF16) hipLaunchKernelGGL((greedy_tail_kernel<__bf16>), dim3(B), dim3(GS_THREADS), 0, (hipStream_t)stream, q);
    else WIPA_REQUIRE(false, "wipa_greedy_step_embed: bad dtype %d", y_dtype);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

extern "C" int wipa_greedy_step_embed_partials(const float* partials, int n_parts, int B, int32_t* tokens, int64_t ld_tok, int32_t* pos_dev,
                                               int64_t* posd_dev, int32_t* done_counter, int n_init, int eot, float* sum_logprobs,
                                               int32_t* not_done, const void* tok_emb, int emb_dtype, const float* emb_scale,
                                               const float* pos_emb, int n_ctx, float* x, const float* ln_w, const float* ln_b, void* y,
                                               int y_dtype, int D, float eps, wipa_stream_t stream) {
    WIPA_REQUIRE(partials && n_parts > 0 && tokens && pos_dev && posd_dev && done_counter && sum_logprobs && not_done && tok_emb && pos_emb && x &&
                     ln_w && ln_b && y && B > 0 && n_ctx > 0, "wipa_greedy_step_embed_partials: bad arguments");
    const int rc = tail_params_check("wipa_greedy_step_embed_partials", nullptr, 0, 0, nullptr, nullptr, D, emb_dtype, emb_scale);
    if (rc != WIPA_OK) return rc;
    TailParams q = {};
    q.part = partials; q.n_part = n_parts;
    q.tokens = tokens; q.ld_tok = ld_tok; q.pos = pos_dev; q.posd = posd_dev; q.done_counter = done_counter;
    q.n_init = n_init; q.eot = eot; q.n_ctx = n_ctx; q.sum_logprobs = sum_logprobs; q.not_done = not_done;
    q.emb = tok_emb; q.emb_dtype = emb_dtype; q.emb_scale = emb_scale; q.pos_emb = pos_emb;
    q.x = x; q.ln_w = ln_w; q.ln_b = ln_b; q.y = y; q.D = D; q.eps = eps;
    if (y_dtype == WIPA_F32) hipLaunchKernelGGL((greedy_tail_kernel<float>), dim3(B), dim3(GS_THREADS), 0, (hipStream_t)stream, q);
    else if (y_dtype == WIPA_BF16) hipLaunchKernelGGL((greedy_tail_kernel<__bf16>), dim3(B), dim3(GS_THREADS), 0, (hipStream_t)stream, q);
    else WIPA_REQUIRE(false, "wipa_greedy_step_embed_partials: bad dtype %d", y_dtype);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

extern "C" int wipa_add_i32(int32_t* p, int32_t v, wipa_stream_t stream) {
    WIPA_REQUIRE(p, "wipa_add_i32: null pointer");
    hipLaunchKernelGGL(add_i32_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, p, v);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

extern "C" int wipa_masked_ce(const float* logits, int64_t ldl, const int32_t* tokens, int64_t ld_tok, int B, int T, int V,
                              int eot, float* row_buf, float* out2, wipa_stream_t stream) {
    WIPA_REQUIRE(logits && tokens && row_buf && out2, "wipa_masked_ce: null pointer");
    const int rows = B * T;
    if (rows <= 0) return WIPA_OK;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(masked_ce_rows_kernel, dim3(rows), dim3(CE_THREADS), 0, s, logits, ldl, tokens, ld_tok, T, V, eot,
                       row_buf, rows);
    hipLaunchKernelGGL(sum2_kernel, dim3(1), dim3(256), 0, s, row_buf, rows, out2);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

extern "C" int wipa_mel_pad_cast(const float* mel, int batch, int n_mels, void* mel_padded, int dtype,
                                 wipa_stream_t stream) {
    WIPA_REQUIRE(mel && mel_padded, "wipa_mel_pad_cast: null pointer");
    const int64_t total = (int64_t)batch * (WIPA_N_FRAMES + 2) * n_mels;
    if (total <= 0) return WIPA_OK;
    const dim3 grid((unsigned)((total + 255) / 256));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == WIPA_F32)
        hipLaunchKernelGGL((mel_pad_cast_kernel<float>), grid, dim3(256), 0, s, mel, n_mels, (float*)mel_padded, total);
    else if (dtype == WIPA_BF16)
        hipLaunchKernelGGL((mel_pad_cast_kernel<__bf16>), grid, dim3(256), 0, s, mel, n_mels, (__bf16*)mel_padded, total);
    else
        WIPA_REQUIRE(false, "wipa_mel_pad_cast: bad dtype %d", dtype);
    WIPA_LAUNCH_CHECK();
    const size_t esz = wipa_dtype_size(dtype);
    WIPA_CHECK_HIP(hipMemsetAsync((char*)mel_padded + (size_t)total * esz, 0, 4 * (size_t)n_mels * esz, s));
    return WIPA_OK;
}
