// Fused kernels of the KV-cached decode step (K11-K13 of SURVEY.md section 8a; mlx_whisper.whisper.ResidualAttentionBlock
// with kv_cache, called through DecodingTask._main_loop at scripts/transcribe_single.py:55, train_whisper_ipa.py:356).
//
// A decode step touches 64 token rows: every kernel of it finishes in a few microseconds, so the step time is the NUMBER of
// dependent launches (~4.5 us each inside a hipGraph), not their work.  The unfused step has 11 launches per decoder layer;
// here a layer is 5:
//
//   wipa_decode_self_block   LayerNorm(attn_ln) -> q|k|v projection of ONE head -> K/V cache append -> self-attention over
//                            the cache -> that head's slice of the out projection, written as a partial ("slab") of the
//                            residual update.  One workgroup per (head, 16 token rows).
//   wipa_decode_cross_block  residual + out-bias + the H head slabs (fixed order) -> LayerNorm(cross_attn_ln) -> cross query
//                            of one head -> streaming cross-attention over the cached K/V (the HBM-bound part of the step:
//                            every cached key and value exactly once, 16-byte non-temporal loads).  One workgroup per
//                            (head, clip); the h = 0 workgroup of a clip also writes the updated residual row.
//   wipa_gemm (skinny)       cross out projection, residual updated in place
//   wipa_gemm (ln prologue)  LayerNorm(mlp_ln) folded into the A operand of mlp1 (+ GELU)          [gemm.hip]
//   wipa_gemm (skinny)       mlp2, residual updated in place
//
// All reductions run in a fixed order (no float atomics): results do not depend on the batch a row rides in.
#include <cstdlib>
#include <mutex>

#include "wipa_common.h"

namespace {

constexpr float NEG_BIG = -1.0e30f;
constexpr float NEG_TEST = -1.0e29f;
constexpr int RG = 16;  // token rows per self-block workgroup (one MFMA row tile)

struct SelfBlockParams {
    const float* x;       // [B, d] f32 residual stream (read only)
    const float* ln_w;
    const float* ln_b;
    const char* wqkv;     // [3d, d] T: query | key | value rows
    const float* bqkv;    // [3d] f32 (key third zero)
    const char* wo;       // [d, d] T
    char* kcache;         // T [B][n_ctx][d]
    char* vcache;
    const int32_t* pos;   // device: position being decoded (= number of cached positions)
    float* slabs;         // f32 [H][B][d]: slab h = attention(head h) @ Wo[:, head h]^T
    int64_t kv_bs;        // elements between clips in the caches (n_ctx * d)
    int64_t slab_stride;  // elements between slabs (B * d)
    int B, d, H;
    float eps, qk_scale;
};

// LayerNorm of one row held by a wave (lane owns columns lane*4 + 256*i), same arithmetic as layernorm_kernel.
constexpr int LN_NV = 5;  // d <= 1280
__device__ __forceinline__ void wave_layernorm_row(const float* __restrict__ xr, const float* __restrict__ w,
                                                   const float* __restrict__ b, int d, float eps, int lane, f32x4 (&o)[LN_NV]) {
    f32x4 v[LN_NV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < LN_NV; ++i) {
        const int c = lane * 4 + 256 * i;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < d) {
            v[i] = *reinterpret_cast<const f32x4*>(xr + c);
            sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    const float mean = wave_reduce_sum(sum) / (float)d;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < LN_NV; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < d) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dv = v[i][e] - mean;
                sq += dv * dv;
            }
        }
    }
    const float rstd = rsqrtf(wave_reduce_sum(sq) / (float)d + eps);
#pragma unroll
    for (int i = 0; i < LN_NV; ++i) {
        const int c = lane * 4 + 256 * i;
        o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < d) {
            const f32x4 ww = *reinterpret_cast<const f32x4*>(w + c);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(b + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[i][e] = (v[i][e] - mean) * rstd * ww[e] + bb[e];
        }
    }
}

template <typename T>
__device__ __forceinline__ void lds_store4(char* p, const f32x4& v);
template <>
__device__ __forceinline__ void lds_store4<float>(char* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
template <>
__device__ __forceinline__ void lds_store4<__bf16>(char* p, const f32x4& v) {
    *reinterpret_cast<bf16x4*>(p) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
}

template <typename T>
__device__ __forceinline__ float round_through(float v) { return to_f32<T>(from_f32<T>(v)); }

// ------------------------------------------------------------------------------------------------------------------
// self block
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void decode_self_block_kernel(SelfBlockParams p) {
    typedef typename Mma<T>::Frag Frag;
    constexpr int EPL = Vec16<T>::EPL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    const int h = blockIdx.x, r0 = blockIdx.y * RG;
    const int d = p.d;
    const int pitch_a = d * (int)sizeof(T) + 16;     // LN rows, padded: conflict-light 16-byte fragment reads
    constexpr int PITCH_O = 64 * (int)sizeof(T) + 16;  // attention output rows
    char* a_s = smem;
    f32x4* red = reinterpret_cast<f32x4*>(smem + RG * pitch_a);                      // [4 waves][12 tiles][64 lanes]
    float* q_s = reinterpret_cast<float*>(reinterpret_cast<char*>(red) + 4 * 12 * 64 * 16);  // [RG][64]
    float* k_s = q_s + RG * 64;
    float* v_s = k_s + RG * 64;
    char* o_s = reinterpret_cast<char*>(v_s + RG * 64);                               // [RG][PITCH_O]
    const int pos = *p.pos;

    // ---- 1. LayerNorm of the 16 rows -> a_s (T)
    for (int r = wave; r < RG; r += 4) {
        const int b = min(r0 + r, p.B - 1);
        f32x4 o[LN_NV];
        wave_layernorm_row(p.x + (int64_t)b * d, p.ln_w, p.ln_b, d, p.eps, lane, o);
#pragma unroll
        for (int i = 0; i < LN_NV; ++i) {
            const int c = lane * 4 + 256 * i;
            if (c < d) lds_store4<T>(a_s + r * pitch_a + c * (int)sizeof(T), o[i]);
        }
    }
    __syncthreads();

    // ---- 2. q|k|v of head h: [16 rows] x [3 x 64 columns], K = d split over the 4 waves (contiguous k ranges)
    {
        const int ksteps_all = d * (int)sizeof(T) / 64;
        const int per = ksteps_all / 4, rem = ksteps_all % 4;
        const int kb = wave * per + min(wave, rem);
        const int ke = kb + per + (wave < rem ? 1 : 0);
        const char* wp[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int n = (i >> 2) * d + h * 64 + (i & 3) * 16 + frow;
            wp[i] = p.wqkv + (int64_t)n * d * sizeof(T) + fq * 16;
        }
        const char* ap = a_s + frow * pitch_a + fq * 16;
        f32x4 acc[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        int ks = kb;
        for (; ks + 2 <= ke; ks += 2) {
            Frag fw[2][12];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 12; ++i) fw[u][i] = *reinterpret_cast<const Frag*>(wp[i] + (int64_t)(ks + u) * 64);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const Frag fx = *reinterpret_cast<const Frag*>(ap + (ks + u) * 64);
#pragma unroll
                for (int i = 0; i < 12; ++i) Mma<T>::run(fw[u][i], fx, acc[i]);
            }
        }
        for (; ks < ke; ++ks) {
            Frag fw[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) fw[i] = *reinterpret_cast<const Frag*>(wp[i] + (int64_t)ks * 64);
            const Frag fx = *reinterpret_cast<const Frag*>(ap + ks * 64);
#pragma unroll
            for (int i = 0; i < 12; ++i) Mma<T>::run(fw[i], fx, acc[i]);
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) red[(wave * 12 + i) * 64 + lane] = acc[i];
    }
    __syncthreads();
    {
        // wave w finishes the 16-column slice w of q, k and v: + bias, q and k scaled, rounded to T; the new K/V row goes to the
        // cache AND stays in LDS for this step's attention
        const int b = r0 + frow;
#pragma unroll
        for (int part = 0; part < 3; ++part) {
            const int i = part * 4 + wave;
            f32x4 s = red[i * 64 + lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) s += red[(w * 12 + i) * 64 + lane];
            const int nl = wave * 16 + 4 * fq;  // column within the head
            const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bqkv + part * d + h * 64 + nl);
            const float sc = part < 2 ? p.qk_scale : 1.0f;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = round_through<T>((s[e] + bias[e]) * sc);
            float* dst = (part == 0 ? q_s : (part == 1 ? k_s : v_s)) + frow * 64 + nl;
            *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
            if (part > 0 && b < p.B) {
                char* cache = part == 1 ? p.kcache : p.vcache;
                T* cp = reinterpret_cast<T*>(cache) + (int64_t)b * p.kv_bs + (int64_t)pos * d + h * 64 + nl;
                lds_store4<T>(reinterpret_cast<char*>(cp), f32x4{v[0], v[1], v[2], v[3]});  // (plain 8/16-byte global store)
            }
        }
    }
    __syncthreads();

    // ---- 3. self-attention: 16 lanes per row; keys 0..pos-1 from the cache, key `pos` from LDS
    {
        constexpr int LPK = 64 / EPL;   // lanes per 64-dim key row (8 bf16 / 16 f32)
        constexpr int G = 16 / LPK;     // keys per row per load instruction (2 / 1)
        constexpr int U = 8;
        const int r = tid >> 4, sub = tid & 15;
        const int g = sub / LPK, c = sub % LPK;
        const int b = min(r0 + r, p.B - 1);
        const int Tk = pos + 1;
        float qf[EPL], kc[EPL], vc[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            qf[e] = q_s[r * 64 + c * EPL + e];
            kc[e] = k_s[r * 64 + c * EPL + e];
            vc[e] = v_s[r * 64 + c * EPL + e];
        }
        const T* Kb = reinterpret_cast<const T*>(p.kcache) + (int64_t)b * p.kv_bs + h * 64 + c * EPL;
        const T* Vb = reinterpret_cast<const T*>(p.vcache) + (int64_t)b * p.kv_bs + h * 64 + c * EPL;
        const int last_cached = max(pos - 1, 0);
        float m = NEG_BIG, l = 0.f, acc[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
        for (int t0 = 0; t0 < Tk; t0 += G * U) {
            Vec16<T> ka[U], va[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = min(t0 + u * G + g, last_cached);
                ka[u] = *reinterpret_cast<const Vec16<T>*>(Kb + (int64_t)t * d);
                va[u] = *reinterpret_cast<const Vec16<T>*>(Vb + (int64_t)t * d);
            }
            float s[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t0 + u * G + g;
                float a = 0.f;
#pragma unroll
                for (int e = 0; e < EPL; ++e) a = fmaf(qf[e], t == pos ? kc[e] : ka[u].get(e), a);
#pragma unroll
                for (int o = 1; o < LPK; o <<= 1) a += __shfl_xor(a, o, 64);
                s[u] = t < Tk ? a : NEG_BIG;
            }
            float m_new = m;
#pragma unroll
            for (int u = 0; u < U; ++u) m_new = fmaxf(m_new, s[u]);
            const float alpha = __expf(m - m_new);
            l *= alpha;
#pragma unroll
            for (int e = 0; e < EPL; ++e) acc[e] *= alpha;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t0 + u * G + g;
                const float pr = (s[u] <= NEG_TEST) ? 0.f : __expf(s[u] - m_new);
                l += pr;
                // keys beyond Tk were loaded from a clamped (at pos = 0: not yet written) slot: their values must not reach the
                // accumulator even multiplied by zero (0 * NaN)
                const bool live = t < Tk;
#pragma unroll
                for (int e = 0; e < EPL; ++e) acc[e] = fmaf(pr, live ? (t == pos ? vc[e] : va[u].get(e)) : 0.f, acc[e]);
            }
            m = m_new;
        }
#pragma unroll
        for (int o = LPK; o < 16; o <<= 1) {  // merge the key groups of this row
            const float m_o = __shfl_xor(m, o, 64);
            const float l_o = __shfl_xor(l, o, 64);
            const float m_n = fmaxf(m, m_o);
            const float a = __expf(m - m_n), bsc = __expf(m_o - m_n);
            l = l * a + l_o * bsc;
#pragma unroll
            for (int e = 0; e < EPL; ++e) acc[e] = acc[e] * a + __shfl_xor(acc[e], o, 64) * bsc;
            m = m_n;
        }
        if (g == 0) {
            const float inv = 1.f / l;
            T* op = reinterpret_cast<T*>(o_s + r * PITCH_O) + c * EPL;
#pragma unroll
            for (int e = 0; e < EPL; ++e) op[e] = from_f32<T>(acc[e] * inv);
        }
    }
    __syncthreads();

    // ---- 4. this head's slice of the out projection: slab[h][b][n] = sum_{j<64} o[b][j] * Wo[n][h*64 + j]
    {
        constexpr int KS = 64 * (int)sizeof(T) / 64;  // fragment steps over the 64 head dims (2 bf16 / 4 f32)
        constexpr int TB = 4;                          // column tiles whose weight fragments are in flight together
        Frag fx[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) fx[ks] = *reinterpret_cast<const Frag*>(o_s + frow * PITCH_O + ks * 64 + fq * 16);
        const int ntiles = d / 16;
        const int b = r0 + frow;
        float* out = p.slabs + (int64_t)h * p.slab_stride + (int64_t)b * d;
        for (int i0 = wave * TB; i0 < ntiles; i0 += 4 * TB) {
            Frag fw[TB][KS];
#pragma unroll
            for (int t = 0; t < TB; ++t) {
                const int n = min((i0 + t) * 16 + frow, d - 1);
                const char* wp = p.wo + ((int64_t)n * d + h * 64) * sizeof(T) + fq * 16;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) fw[t][ks] = *reinterpret_cast<const Frag*>(wp + ks * 64);
            }
#pragma unroll
            for (int t = 0; t < TB; ++t) {
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) Mma<T>::run(fw[t][ks], fx[ks], acc);
                if (i0 + t < ntiles && b < p.B) *reinterpret_cast<f32x4*>(out + (i0 + t) * 16 + 4 * fq) = acc;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// cross block
// ------------------------------------------------------------------------------------------------------------------
struct CrossBlockParams {
    const float* x_in;     // [B, d] f32 residual before the self-attention update
    float* x_out;          // [B, d] f32: x_in + bias_o + sum of slabs (written by the h = 0 workgroup of each clip)
    const float* slabs;    // f32 [n_slabs][B][d]
    const float* bias_o;   // [d] f32 bias of the self-attention out projection
    const float* ln_w;
    const float* ln_b;
    const char* wq;        // [d, d] T cross query
    const float* bq;       // [d] f32
    const char* kv;        // T [B][2H][Tk][64]: K heads then V heads
    char* out;             // T [B, d]
    int64_t slab_stride;
    int n_slabs, B, d, H, Tk;
    float eps, qk_scale;
};

constexpr int MAX_SLABS_X = 20;  // heads of whisper-large

template <typename T>
__global__ __launch_bounds__(256, 6) void decode_cross_block_kernel(CrossBlockParams p) {
    constexpr int EPL = Vec16<T>::EPL;
    constexpr int LPK = 64 / EPL;
    constexpr int G = 64 / LPK;
    constexpr int U = 4;
    __shared__ __attribute__((aligned(16))) float xn[1280];
    __shared__ float s_red[8];
    __shared__ float q_s[64];
    __shared__ float s_m[4], s_l[4];
    __shared__ float s_acc[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, b = blockIdx.y;
    const int d = p.d;

    // The cross-query weights do not depend on the residual row: the first batch of this thread's 16-byte chunks of row
    // (h*64 + j) is requested before anything else, so its round trip overlaps the prologue's.
    constexpr int UQ = 3;
    const int qj = tid >> 2, qpart = tid & 3;
    const T* wr = reinterpret_cast<const T*>(p.wq) + (int64_t)(h * 64 + qj) * d;
    const int nch = d / EPL;  // 16-byte chunks per weight row
    Vec16<T> wcur[UQ];
#pragma unroll
    for (int u = 0; u < UQ; ++u)
        if (qpart + 4 * u < nch) wcur[u] = *reinterpret_cast<const Vec16<T>*>(wr + (qpart + 4 * u) * EPL);
    // streaming roles (phase 3), fixed by the lane alone: the K rows of this wave's FIRST key group are requested right behind
    // the prologue's own loads -- every workgroup of the launch is resident at once, so without this HBM idles for the whole
    // prologue (slab sum, LayerNorm, query GEMV: ~5 us of a 55 us kernel)
    const int g = lane / LPK, c = lane % LPK;
    const int Tk = p.Tk;
    const int64_t head = (int64_t)Tk * 64;
    const T* Kb = reinterpret_cast<const T*>(p.kv) + ((int64_t)b * 2 * p.H + h) * head + c * EPL;
    const T* Vb = Kb + (int64_t)p.H * head;
    typedef decltype(wcur[0].v) VT;

    constexpr int PRE = 3;  // key rows of the first group requested ahead (all U would spill at 80 registers)
    Vec16<T> kpre[PRE];
    // ---- 1. residual row: x + out-bias + slabs in order 0..n-1; LayerNorm -> xn (rounded through T)
    {
        f32x4 v[2];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid * 4 + 1024 * i;
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < d) {
                v[i] = *reinterpret_cast<const f32x4*>(p.x_in + (int64_t)b * d + c);
                if (p.bias_o) v[i] += *reinterpret_cast<const f32x4*>(p.bias_o + c);
                // slabs in groups of SG loads in flight (all 20 at once cost 80 registers and a third of the occupancy the
                // streaming phase wants); the additions stay in slab order
                constexpr int SG = 6;
                const float* sp = p.slabs + (int64_t)b * d + c;
                for (int s0 = 0; s0 < p.n_slabs; s0 += SG) {
                    f32x4 sl[SG];
#pragma unroll
                    for (int s = 0; s < SG; ++s)
                        if (s0 + s < p.n_slabs) sl[s] = *reinterpret_cast<const f32x4*>(sp + (int64_t)(s0 + s) * p.slab_stride);
#pragma unroll
                    for (int s = 0; s < SG; ++s)
                        if (s0 + s < p.n_slabs) v[i] += sl[s];
                }
                if (h == 0) *reinterpret_cast<f32x4*>(p.x_out + (int64_t)b * d + c) = v[i];
                sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
            }
        }
        sum = wave_reduce_sum(sum);
        if (lane == 0) s_red[wave] = sum;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PRE; ++u)
            kpre[u].v = __builtin_nontemporal_load(reinterpret_cast<const VT*>(Kb + (int64_t)min(wave * G * U + u * G + g, Tk - 1) * 64));
        const float mean = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) / (float)d;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid * 4 + 1024 * i;
            if (c < d) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dv = v[i][e] - mean;
                    sq += dv * dv;
                }
            }
        }
        sq = wave_reduce_sum(sq);
        if (lane == 0) s_red[4 + wave] = sq;
        __syncthreads();
        const float rstd = rsqrtf(((s_red[4] + s_red[5]) + (s_red[6] + s_red[7])) / (float)d + p.eps);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid * 4 + 1024 * i;
            if (c < d) {
                const f32x4 ww = *reinterpret_cast<const f32x4*>(p.ln_w + c);
                const f32x4 bb = *reinterpret_cast<const f32x4*>(p.ln_b + c);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = round_through<T>((v[i][e] - mean) * rstd * ww[e] + bb[e]);
                *reinterpret_cast<f32x4*>(xn + c) = o;
            }
        }
    }
    __syncthreads();

    // ---- 2. cross query of head h: q[j] = (xn . Wq[h*64 + j, :] + bq) * scale, 4 threads per output
    {
        float a = 0.f;
        for (int ch = qpart; ch < nch; ch += 4 * UQ) {  // two register sets: the next batch is in flight while this one is spent
            Vec16<T> wnext[UQ];
            const int chn = ch + 4 * UQ;
#pragma unroll
            for (int u = 0; u < UQ; ++u)
                if (chn + 4 * u < nch) wnext[u] = *reinterpret_cast<const Vec16<T>*>(wr + (chn + 4 * u) * EPL);
#pragma unroll
            for (int u = 0; u < UQ; ++u) {
                if (ch + 4 * u < nch) {
                    const float* xp = xn + (ch + 4 * u) * EPL;
#pragma unroll
                    for (int e = 0; e < EPL; ++e) a = fmaf(xp[e], wcur[u].get(e), a);
                }
            }
#pragma unroll
            for (int u = 0; u < UQ; ++u) wcur[u] = wnext[u];
        }
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        if (qpart == 0) q_s[qj] = round_through<T>((a + p.bq[h * 64 + qj]) * p.qk_scale);
    }
    __syncthreads();

    // ---- 3. streaming cross-attention (the arithmetic of decode_attn_kernel<T, 4>: 4 waves split the keys)
    float qf[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) qf[e] = q_s[c * EPL + e];
    float m = NEG_BIG, l = 0.f, acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
    constexpr int STEP = 4 * G * U;
    auto key_group = [&](Vec16<T> (&ka)[U], int t0) {
        Vec16<T> va[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            va[u].v = __builtin_nontemporal_load(reinterpret_cast<const VT*>(Vb + (int64_t)min(t0 + u * G + g, Tk - 1) * 64));
        float s[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) a = fmaf(qf[e], ka[u].get(e), a);
#pragma unroll
            for (int o = 1; o < LPK; o <<= 1) a += __shfl_xor(a, o, 64);
            s[u] = (t0 + u * G + g < Tk) ? a : NEG_BIG;
        }
        float m_new = m;
#pragma unroll
        for (int u = 0; u < U; ++u) m_new = fmaxf(m_new, s[u]);
        const float alpha = __expf(m - m_new);
        l *= alpha;
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] *= alpha;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float pr = (s[u] <= NEG_TEST) ? 0.f : __expf(s[u] - m_new);
            l += pr;
#pragma unroll
            for (int e = 0; e < EPL; ++e) acc[e] = fmaf(pr, va[u].get(e), acc[e]);
        }
        m = m_new;
    };
    if (wave * G * U < Tk) {
        const int t0 = wave * G * U;
        Vec16<T> ka[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (u < PRE) ka[u] = kpre[u];
            else ka[u].v = __builtin_nontemporal_load(reinterpret_cast<const VT*>(Kb + (int64_t)min(t0 + u * G + g, Tk - 1) * 64));
        }
        key_group(ka, t0);
    }
    for (int t0 = wave * G * U + STEP; t0 < Tk; t0 += STEP) {
        Vec16<T> ka[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            ka[u].v = __builtin_nontemporal_load(reinterpret_cast<const VT*>(Kb + (int64_t)min(t0 + u * G + g, Tk - 1) * 64));
        key_group(ka, t0);
    }
#pragma unroll
    for (int o = LPK; o < 64; o <<= 1) {
        const float m_o = __shfl_xor(m, o, 64);
        const float l_o = __shfl_xor(l, o, 64);
        const float m_n = fmaxf(m, m_o);
        const float a = __expf(m - m_n), bsc = __expf(m_o - m_n);
        l = l * a + l_o * bsc;
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] = acc[e] * a + __shfl_xor(acc[e], o, 64) * bsc;
        m = m_n;
    }
    if (lane < LPK) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) s_acc[wave][c * EPL + e] = acc[e];
        if (lane == 0) {
            s_m[wave] = m;
            s_l[wave] = l;
        }
    }
    __syncthreads();
    if (tid < 64) {
        const float mm = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float sc = __expf(s_m[w] - mm);
            num += s_acc[w][tid] * sc;
            den += s_l[w] * sc;
        }
        reinterpret_cast<T*>(p.out)[(int64_t)b * d + h * 64 + tid] = from_f32<T>(num / den);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The cross block with the head of the K/V stream staged in LDS UNDER the prologue (round 3).
//
// At the benchmark shape (12 heads x 64 clips = 768 workgroups) every workgroup of a launch is resident at once, all of them
// run the prologue (slab sum, LayerNorm, 64 x d query GEMV) at the same time, and HBM idles meanwhile: 7.7 us of a 55 us
// kernel whose streaming loop alone runs at the achievable HBM rate (profiles/r02_pmc_cross_block.json).  Registers cannot
// carry the stream through the prologue (80 VGPRs per thread at six workgroups per CU), LDS can: three workgroups per CU leave
// 53 KB each.  So, in program order:
//   1. every load the prologue needs is issued FIRST -- residual row, slabs, LayerNorm parameters and ALL 24 16-byte chunks of
//      this thread's query-weight row (96 VGPRs, affordable at three workgroups per CU): nothing the prologue waits for is
//      younger than the stream (vector-memory loads return in order within a wave);
//   2. the wave's first U x 8 key rows and value rows go to LDS by LDS-DMA (buffer_load ... lds, 1 KiB per instruction, no
//      registers, non-temporal): 8U KiB per wave, 32U KiB per workgroup -- 37.7 MB of a launch's 295 MB with U = 6;
//   3. the prologue computes while those bytes arrive;
//   4. the streaming loop takes its first step from LDS and every later step from registers loaded ONE STEP AHEAD (K and V
//      rows of step s + 1 are requested before step s is spent), so HBM never waits for arithmetic.
// bf16, d <= 768 (24 weight chunks per thread); other shapes and larger grids keep decode_cross_block_kernel.
typedef __attribute__((address_space(3))) void* lds_ptr_x;

template <int U>
__global__ __launch_bounds__(256, 3) void decode_cross_block_pre_kernel(CrossBlockParams p) {
    typedef __bf16 T;
    constexpr int EPL = 8, LPK = 8, G = 8;
    constexpr int NQ = 24;            // 16-byte chunks of a query-weight row per thread (d <= 768)
    constexpr int STEP = 4 * G * U;   // key rows per step of the whole workgroup
    __shared__ __attribute__((aligned(1024))) char pre[4][2][U][1024];  // [wave][K | V][row group][8 rows x 128 B]
    __shared__ __attribute__((aligned(16))) float xn[768];
    __shared__ float s_red[8];
    __shared__ float q_s[64];
    __shared__ float s_m[4], s_l[4];
    __shared__ float s_acc[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, b = blockIdx.y;
    const int d = p.d;
    typedef bf16x8 VT;

    // ---- 1. everything the prologue reads, requested before the stream.  Unconditional loads from clamped (always valid)
    // addresses: a conditional load makes the compiler wait for it on the spot, which would put a vmcnt(0) in front of the
    // stream; what is out of range is simply not used below.
    const int qj = tid >> 2, qpart = tid & 3;
    const T* wr = reinterpret_cast<const T*>(p.wq) + (int64_t)(h * 64 + qj) * d;
    const int nch = d / EPL;
    Vec16<T> wq[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) wq[u] = *reinterpret_cast<const Vec16<T>*>(wr + min(qpart + 4 * u, nch - 1) * EPL);
    const int cc = tid * 4;                 // this thread's four columns of the row (d <= 768 < 1024: one chunk per thread)
    const int ccl = min(cc, d - 4);         // clamped copy for the loads
    constexpr int SG = 4;                   // 1 <= n_slabs <= SG and no separate out-bias (checked by the host): the split-K
                                            // slabs of the default decode step (slab 0 carries the bias)
    f32x4 v = *reinterpret_cast<const f32x4*>(p.x_in + (int64_t)b * d + ccl);
    const f32x4 lnw = *reinterpret_cast<const f32x4*>(p.ln_w + ccl);
    const f32x4 lnb = *reinterpret_cast<const f32x4*>(p.ln_b + ccl);
    f32x4 sl[SG];
#pragma unroll
    for (int s = 0; s < SG; ++s)
        sl[s] = *reinterpret_cast<const f32x4*>(p.slabs + (int64_t)b * d + ccl + (int64_t)min(s, p.n_slabs - 1) * p.slab_stride);
    const float bqv = p.bq[h * 64 + qj];
    // the scheduler must not sink any of the loads above below the stream (or hoist stream pieces above them): a load issued
    // after an LDS-DMA transfer returns after it
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // ---- 2. the head of this wave's K / V stream -> LDS (rows wave*8U + u*8 + g, the rows of its first streaming step)
    const int g = lane / LPK, c = lane % LPK;
    const int Tk = p.Tk;
    const int64_t head = (int64_t)Tk * 64;
    const T* Kh = reinterpret_cast<const T*>(p.kv) + ((int64_t)b * 2 * p.H + h) * head;
    const T* Vh = Kh + (int64_t)p.H * head;
    {
        const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc((void*)Kh, 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc((void*)Vh, 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int off = min(wave * G * U + u * G + g, Tk - 1) * 128 + c * 16;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rK, (lds_ptr_x)&pre[wave][0][u][0], 16, off, 0, 0, 2);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int off = min(wave * G * U + u * G + g, Tk - 1) * 128 + c * 16;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rV, (lds_ptr_x)&pre[wave][1][u][0], 16, off, 0, 0, 2);
        }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // ---- 3. prologue: residual row (x + out-bias + slabs in order), LayerNorm -> xn (rounded through T), query GEMV.
    // Workgroup barriers are RAW s_barriers behind an LDS-only wait: __syncthreads() carries a workgroup fence, and a fence
    // makes the compiler drain vmcnt(0) -- i.e. wait for the whole LDS-DMA stream -- at the first barrier of the prologue.
#define XBAR()                                                \
    do {                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
        __builtin_amdgcn_s_barrier();                         \
        asm volatile("" ::: "memory");                        \
    } while (0)
    {
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < SG; ++s)
            if (s < p.n_slabs) v += sl[s];
        if (cc < d) {
            if (h == 0) *reinterpret_cast<f32x4*>(p.x_out + (int64_t)b * d + cc) = v;
            sum = (v[0] + v[1]) + (v[2] + v[3]);
        }
        sum = wave_reduce_sum(sum);
        if (lane == 0) s_red[wave] = sum;
        XBAR();
        const float mean = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) / (float)d;
        float sq = 0.f;
        if (cc < d) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dv = v[e] - mean;
                sq += dv * dv;
            }
        }
        sq = wave_reduce_sum(sq);
        if (lane == 0) s_red[4 + wave] = sq;
        XBAR();
        const float rstd = rsqrtf(((s_red[4] + s_red[5]) + (s_red[6] + s_red[7])) / (float)d + p.eps);
        if (cc < d) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = round_through<T>((v[e] - mean) * rstd * lnw[e] + lnb[e]);
            *reinterpret_cast<f32x4*>(xn + cc) = o;
        }
    }
    XBAR();
    {
        float a = 0.f;
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            if (qpart + 4 * u < nch) {
                const float* xp = xn + (qpart + 4 * u) * EPL;
#pragma unroll
                for (int e = 0; e < EPL; ++e) a = fmaf(xp[e], wq[u].get(e), a);
            }
        }
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        if (qpart == 0) q_s[qj] = round_through<T>((a + bqv) * p.qk_scale);
    }
    XBAR();

    // ---- 4. streaming cross-attention: step 0 from LDS, later steps from registers loaded one step ahead
    const T* Kb = Kh + c * EPL;
    const T* Vb = Vh + c * EPL;
    float qf[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) qf[e] = q_s[c * EPL + e];
    float m = NEG_BIG, l = 0.f, acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
    auto key_group = [&](const Vec16<T> (&ka)[U], const Vec16<T> (&va)[U], int t0) {
        float s[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) a = fmaf(qf[e], ka[u].get(e), a);
#pragma unroll
            for (int o = 1; o < LPK; o <<= 1) a += __shfl_xor(a, o, 64);
            s[u] = (t0 + u * G + g < Tk) ? a : NEG_BIG;
        }
        float m_new = m;
#pragma unroll
        for (int u = 0; u < U; ++u) m_new = fmaxf(m_new, s[u]);
        const float alpha = __expf(m - m_new);
        l *= alpha;
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] *= alpha;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float pr = (s[u] <= NEG_TEST) ? 0.f : __expf(s[u] - m_new);
            l += pr;
#pragma unroll
            for (int e = 0; e < EPL; ++e) acc[e] = fmaf(pr, va[u].get(e), acc[e]);
        }
        m = m_new;
    };
    auto load_step = [&](Vec16<T> (&ka)[U], Vec16<T> (&va)[U], int t0) {
#pragma unroll
        for (int u = 0; u < U; ++u)
            ka[u].v = __builtin_nontemporal_load(reinterpret_cast<const VT*>(Kb + (int64_t)min(t0 + u * G + g, Tk - 1) * 64));
#pragma unroll
        for (int u = 0; u < U; ++u)
            va[u].v = __builtin_nontemporal_load(reinterpret_cast<const VT*>(Vb + (int64_t)min(t0 + u * G + g, Tk - 1) * 64));
    };
    const int t_first = wave * G * U;
    Vec16<T> kn[U], vn[U];
    const bool more = t_first + STEP < Tk;  // wave-uniform
    if (more) load_step(kn, vn, t_first + STEP);  // younger than the LDS-DMA: they return after it
    // the 2U LDS-DMA transfers of this wave have landed once at most the 2U register loads above are outstanding
    if (more) {
        if constexpr (U == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if constexpr (U == 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (t_first < Tk) {
        Vec16<T> ka[U], va[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            ka[u].v = *reinterpret_cast<const VT*>(&pre[wave][0][u][lane * 16]);
            va[u].v = *reinterpret_cast<const VT*>(&pre[wave][1][u][lane * 16]);
        }
        key_group(ka, va, t_first);
    }
    for (int t0 = t_first + STEP; t0 < Tk; t0 += STEP) {
        Vec16<T> ka[U], va[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            ka[u] = kn[u];
            va[u] = vn[u];
        }
        if (t0 + STEP < Tk) load_step(kn, vn, t0 + STEP);
        key_group(ka, va, t0);
    }
#pragma unroll
    for (int o = LPK; o < 64; o <<= 1) {
        const float m_o = __shfl_xor(m, o, 64);
        const float l_o = __shfl_xor(l, o, 64);
        const float m_n = fmaxf(m, m_o);
        const float a = __expf(m - m_n), bsc = __expf(m_o - m_n);
        l = l * a + l_o * bsc;
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] = acc[e] * a + __shfl_xor(acc[e], o, 64) * bsc;
        m = m_n;
    }
    if (lane < LPK) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) s_acc[wave][c * EPL + e] = acc[e];
        if (lane == 0) {
            s_m[wave] = m;
            s_l[wave] = l;
        }
    }
    __syncthreads();
    if (tid < 64) {
        const float mm = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float sc = __expf(s_m[w] - mm);
            num += s_acc[w][tid] * sc;
            den += s_l[w] * sc;
        }
        reinterpret_cast<T*>(p.out)[(int64_t)b * d + h * 64 + tid] = from_f32<T>(num / den);
    }
}

#undef XBAR

template <typename T>
size_t self_block_lds(int d) {
    return (size_t)RG * (d * sizeof(T) + 16) + 4 * 12 * 64 * 16 + 3 * RG * 64 * 4 + RG * (64 * sizeof(T) + 16);
}

constexpr int LDS_LIMIT = 160 * 1024;

}  // namespace

// Raise the dynamic-LDS limit of the self-block kernels once.  Called from wipa_decoder_begin (always eager), so the first
// launch inside a stream capture finds the attribute already set.
int wipa_decode_fused_init() {
    static std::once_flag once;
    static hipError_t err = hipSuccess;
    std::call_once(once, [] {
        const void* fns[2] = {reinterpret_cast<const void*>(&decode_self_block_kernel<__bf16>),
                              reinterpret_cast<const void*>(&decode_self_block_kernel<float>)};
        for (const void* f : fns) {
            const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT);
            if (e != hipSuccess) err = e;
        }
    });
    WIPA_CHECK_HIP(err);
    return WIPA_OK;
}

extern "C" int wipa_decode_self_block(const wipa_self_block_desc* d, wipa_stream_t stream) {
    WIPA_REQUIRE(d && d->x && d->ln_w && d->ln_b && d->wqkv && d->bqkv && d->wo && d->kcache && d->vcache && d->pos && d->slabs,
                 "wipa_decode_self_block: null pointer");
    WIPA_REQUIRE(d->dtype == WIPA_F32 || d->dtype == WIPA_BF16, "wipa_decode_self_block: dtype %d", d->dtype);
    WIPA_REQUIRE(d->B > 0 && d->H > 0 && d->d == d->H * 64 && d->d <= 1280, "wipa_decode_self_block: d=%d must be 64*H and <= 1280", d->d);
    WIPA_REQUIRE(d->slab_stride >= (int64_t)d->B * d->d && d->slab_stride % 4 == 0, "wipa_decode_self_block: slab_stride too small");
    WIPA_REQUIRE(((uintptr_t)d->x % 16) == 0 && ((uintptr_t)d->wqkv % 16) == 0 && ((uintptr_t)d->wo % 16) == 0 &&
                     ((uintptr_t)d->kcache % 16) == 0 && ((uintptr_t)d->vcache % 16) == 0 && ((uintptr_t)d->slabs % 16) == 0,
                 "wipa_decode_self_block: operands must be 16-byte aligned");
    SelfBlockParams p;
    p.x = d->x; p.ln_w = d->ln_w; p.ln_b = d->ln_b;
    p.wqkv = (const char*)d->wqkv; p.bqkv = d->bqkv; p.wo = (const char*)d->wo;
    p.kcache = (char*)d->kcache; p.vcache = (char*)d->vcache; p.pos = d->pos; p.slabs = d->slabs;
    p.kv_bs = d->kv_batch_stride; p.slab_stride = d->slab_stride;
    p.B = d->B; p.d = d->d; p.H = d->H; p.eps = d->eps; p.qk_scale = d->qk_scale;
    const dim3 grid(d->H, (d->B + RG - 1) / RG);
    if (wipa_decode_fused_init() != WIPA_OK) return WIPA_ERR_HIP;
    if (d->dtype == WIPA_BF16) {
        const size_t lds = self_block_lds<__bf16>(d->d);
        hipLaunchKernelGGL(decode_self_block_kernel<__bf16>, grid, dim3(256), lds, (hipStream_t)stream, p);
    } else {
        const size_t lds = self_block_lds<float>(d->d);
        hipLaunchKernelGGL(decode_self_block_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, p);
    }
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

extern "C" int wipa_decode_cross_block(const wipa_cross_block_desc* d, wipa_stream_t stream) {
    WIPA_REQUIRE(d && d->x_in && d->x_out && d->ln_w && d->ln_b && d->wq && d->bq && d->kv && d->out &&
                     (d->slabs || d->n_slabs == 0), "wipa_decode_cross_block: null pointer");
    WIPA_REQUIRE(d->dtype == WIPA_F32 || d->dtype == WIPA_BF16, "wipa_decode_cross_block: dtype %d", d->dtype);
    WIPA_REQUIRE(d->B > 0 && d->B <= 65535 && d->H > 0 && d->d == d->H * 64 && d->d <= 1280 && d->Tk > 0,
                 "wipa_decode_cross_block: bad shape (d=%d must be 64*H and <= 1280)", d->d);
    WIPA_REQUIRE(d->n_slabs >= 0 && d->n_slabs <= MAX_SLABS_X, "wipa_decode_cross_block: n_slabs=%d (max %d)", d->n_slabs, MAX_SLABS_X);
    WIPA_REQUIRE(d->x_in != d->x_out, "wipa_decode_cross_block: x_out must not alias x_in (H workgroups read the row one writes)");
    WIPA_REQUIRE(((uintptr_t)d->x_in % 16) == 0 && ((uintptr_t)d->x_out % 16) == 0 && ((uintptr_t)d->slabs % 16) == 0 &&
                     ((uintptr_t)d->wq % 16) == 0 && ((uintptr_t)d->kv % 16) == 0 && d->slab_stride % 4 == 0,
                 "wipa_decode_cross_block: operands must be 16-byte aligned");
    CrossBlockParams p;
    p.x_in = d->x_in; p.x_out = d->x_out; p.slabs = d->slabs; p.bias_o = d->bias_o; p.ln_w = d->ln_w; p.ln_b = d->ln_b;
    p.wq = (const char*)d->wq; p.bq = d->bq; p.kv = (const char*)d->kv; p.out = (char*)d->out;
    p.slab_stride = d->slab_stride; p.n_slabs = d->n_slabs; p.B = d->B; p.d = d->d; p.H = d->H; p.Tk = d->Tk;
    p.eps = d->eps; p.qk_scale = d->qk_scale;
    const dim3 grid(d->H, d->B);
    // WIPA_CROSS_PRE: 0 = the plain kernel, 4 / 6 = the LDS-staged kernel with U x 8 key rows per wave and step (A/B runs).
    // Default: staged with U = 4 (32 KiB of LDS, bit-identical to the plain kernel: same row groups, same order) when the
    // whole grid is resident at once, i.e. when nothing else would keep HBM busy under the prologue; larger grids overlap one
    // workgroup's prologue with its neighbours' streams by themselves and are better off with six workgroups per CU.
    // Measured r03, whisper-small, 12 x 64 workgroups: plain 54.5 us, U = 4 51.3 us (0.72 of the HBM peak), U = 6 52.5 us.
    const char* pre_env = getenv("WIPA_CROSS_PRE");  // read per call (enqueue / capture time): tests flip it within a process
    int pre = pre_env ? atoi(pre_env) : -1;
    if (pre < 0) pre = (d->H * d->B <= 3 * 256) ? 4 : 0;
    if (d->dtype == WIPA_BF16 && d->d <= 768 && d->d % 8 == 0 && d->Tk >= 8 && d->n_slabs >= 1 && d->n_slabs <= 4 && !d->bias_o && (pre == 4 || pre == 6)) {
        if (pre == 4) hipLaunchKernelGGL(decode_cross_block_pre_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL(decode_cross_block_pre_kernel<6>, grid, dim3(256), 0, (hipStream_t)stream, p);
    } else if (d->dtype == WIPA_BF16)
        hipLaunchKernelGGL(decode_cross_block_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(decode_cross_block_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, p);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}
