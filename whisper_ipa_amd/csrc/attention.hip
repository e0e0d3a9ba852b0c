// Attention kernels (head_dim = 64; q and k arrive pre-scaled by 64^-0.25 each).
//   * attn_generic_kernel  : f32-math flash-style kernel for any (Tq, Tk), causal or not,
//                            T in {f32, bf16} storage.  Parity path + decoder self-attn.
//   * flash_enc_bf16_kernel: K5 encoder self-attention on bf16 MFMA 32x32x16 (swapped
//                            QK^T so a P row is lane-local; P feeds PV from registers).
//   * decode_cross_attn_kernel: K11, one query per (b,h) against the cached cross K/V,
//                            HBM-bound streaming with 16-byte coalesced loads.
// Replaces MultiHeadAttention.qkv_attention of mlx_whisper.whisper
// (call sites scripts/train_whisper_ipa.py:223,232; scripts/transcribe_single.py:54-55).
#include <cstring>

#include "wipa_common.h"

namespace {

constexpr float NEG_BIG = -1.0e30f;
constexpr float NEG_TEST = -1.0e29f;

// =============================================================================
// generic attention, f32 math
// =============================================================================
struct AttnParams {
    const char* q;
    const char* k;
    const char* v;
    char* out;
    const int32_t* tk_dev;
    const int32_t* q_row_dev;
    float* lse;  // optional [B, H, Tq]: log-sum-exp of every score row (saved for the backward pass)
    int64_t q_bs, q_rs, q_hs, k_bs, k_rs, k_hs, v_bs, v_rs, v_hs, o_bs, o_rs, o_hs;
    int Tq, Tk, causal, H;
};

constexpr int GA_LD = 68;  // padded f32 row (64 + 4): conflict-free ds_read_b128 across 16 keys

template <typename T>
__global__ __launch_bounds__(256) void attn_generic_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) float Ks[64 * GA_LD];
    __shared__ __attribute__((aligned(16))) float Vs[64 * GA_LD];
    constexpr int EPL = Vec16<T>::EPL;
    const int tid = threadIdx.x;
    const int ql = tid >> 4, kl = tid & 15;
    const int h = blockIdx.y, b = blockIdx.z;
    const int Tk = p.Tk + (p.tk_dev ? *p.tk_dev : 0);
    const int qi = blockIdx.x * 16 + ql;
    const int qc = min(qi, p.Tq - 1);
    const int q_row0 = p.q_row_dev ? *p.q_row_dev : 0;
    const T* qp = reinterpret_cast<const T*>(p.q) + b * p.q_bs + (int64_t)(q_row0 + qc) * p.q_rs + h * p.q_hs;
    float q[64];
#pragma unroll
    for (int c = 0; c < 64 / EPL; ++c) {
        Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(qp + c * EPL);
#pragma unroll
        for (int e = 0; e < EPL; ++e) q[c * EPL + e] = v.get(e);
    }
    float acc[64];
#pragma unroll
    for (int d = 0; d < 64; ++d) acc[d] = 0.f;
    float m = NEG_BIG, l = 0.f;
    const int kmax = p.causal ? (qc + (Tk - p.Tq)) : (Tk - 1);  // last visible key of this query
    // staging role: key row srow, 16-element segment sseg
    const int srow = tid >> 2, sseg = (tid & 3) * 16;
    const T* kb = reinterpret_cast<const T*>(p.k) + b * p.k_bs + h * p.k_hs;
    const T* vb = reinterpret_cast<const T*>(p.v) + b * p.v_bs + h * p.v_hs;
    // causal: no query of this block sees keys beyond the block's last query
    const int q_last = min(blockIdx.x * 16 + 15, p.Tq - 1);
    const int k_end = p.causal ? min(Tk, q_last + (Tk - p.Tq) + 1) : Tk;
    for (int k0 = 0; k0 < k_end; k0 += 64) {
        __syncthreads();
        {
            const int key = k0 + srow;
            float kv[16], vv[16];
            if (key < Tk) {
                const T* kp = kb + (int64_t)key * p.k_rs + sseg;
                const T* vp = vb + (int64_t)key * p.v_rs + sseg;
#pragma unroll
                for (int c = 0; c < 16 / EPL; ++c) {
                    Vec16<T> a = *reinterpret_cast<const Vec16<T>*>(kp + c * EPL);
                    Vec16<T> bb = *reinterpret_cast<const Vec16<T>*>(vp + c * EPL);
#pragma unroll
                    for (int e = 0; e < EPL; ++e) {
                        kv[c * EPL + e] = a.get(e);
                        vv[c * EPL + e] = bb.get(e);
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) kv[e] = vv[e] = 0.f;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                *reinterpret_cast<f32x4*>(&Ks[srow * GA_LD + sseg + 4 * c]) =
                    f32x4{kv[4 * c], kv[4 * c + 1], kv[4 * c + 2], kv[4 * c + 3]};
                *reinterpret_cast<f32x4*>(&Vs[srow * GA_LD + sseg + 4 * c]) =
                    f32x4{vv[4 * c], vv[4 * c + 1], vv[4 * c + 2], vv[4 * c + 3]};
            }
        }
        __syncthreads();
        float s[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kk = kl + 16 * i;
            const float* kr = &Ks[kk * GA_LD];
            float a = 0.f;
#pragma unroll
            for (int d4 = 0; d4 < 16; ++d4) {
                const f32x4 k4 = *reinterpret_cast<const f32x4*>(kr + 4 * d4);
                a = fmaf(q[4 * d4], k4[0], a);
                a = fmaf(q[4 * d4 + 1], k4[1], a);
                a = fmaf(q[4 * d4 + 2], k4[2], a);
                a = fmaf(q[4 * d4 + 3], k4[3], a);
            }
            s[i] = (k0 + kk <= kmax) ? a : NEG_BIG;
        }
        const float m_new = fmaxf(fmaxf(fmaxf(m, s[0]), fmaxf(s[1], s[2])), s[3]);
        const float alpha = __expf(m - m_new);
        l *= alpha;
#pragma unroll
        for (int d = 0; d < 64; ++d) acc[d] *= alpha;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float pr = (s[i] <= NEG_TEST) ? 0.f : __expf(s[i] - m_new);
            l += pr;
            const float* vr = &Vs[(kl + 16 * i) * GA_LD];
#pragma unroll
            for (int d4 = 0; d4 < 16; ++d4) {
                const f32x4 v4 = *reinterpret_cast<const f32x4*>(vr + 4 * d4);
                acc[4 * d4] = fmaf(pr, v4[0], acc[4 * d4]);
                acc[4 * d4 + 1] = fmaf(pr, v4[1], acc[4 * d4 + 1]);
                acc[4 * d4 + 2] = fmaf(pr, v4[2], acc[4 * d4 + 2]);
                acc[4 * d4 + 3] = fmaf(pr, v4[3], acc[4 * d4 + 3]);
            }
        }
        m = m_new;
    }
    // merge the 16 key lanes of this query (consecutive lanes of one wave)
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        const float m_o = __shfl_xor(m, o, 64);
        const float l_o = __shfl_xor(l, o, 64);
        const float m_n = fmaxf(m, m_o);
        const float a = __expf(m - m_n), bsc = __expf(m_o - m_n);
        l = l * a + l_o * bsc;
#pragma unroll
        for (int d = 0; d < 64; ++d) acc[d] = acc[d] * a + __shfl_xor(acc[d], o, 64) * bsc;
        m = m_n;
    }
    if (qi < p.Tq) {
        const float inv = 1.f / l;
        if (p.lse && kl == 0) p.lse[((int64_t)b * gridDim.y + h) * p.Tq + qi] = m + __logf(l);
        T* op = reinterpret_cast<T*>(p.out) + b * p.o_bs + (int64_t)qi * p.o_rs + h * p.o_hs;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (kl == j) {
#pragma unroll
                for (int e = 0; e < 4; ++e) op[4 * j + e] = from_f32<T>(acc[4 * j + e] * inv);
            }
        }
    }
}

// ---- the same attention on the f32 MFMA, for float32 with at least one 16-query fragment (the teacher-forced decoder of a
// fine-tune step: 64 target positions against 64 causal keys, and against the 1500 encoder positions -- 0.5 ms per layer on
// the VALU kernel above).  64 queries per workgroup, wave w owns queries 16w .. 16w+15 with the query row as MFMA fragment
// in registers and walks the keys in tiles of 64 staged as K [key][d] and V^T [d][key].  The keys sit on the ROW side of the
// score product, S[key][q], so a lane ends up holding 4 consecutive keys of one query: after the softmax these ARE the
// fragment of O[q][d] += sum_key P[q][key] V[key][d] (no LDS round trip for P; attention_bwd.hip uses the same orientation).
// Online softmax per query: its 64 scores of a tile are spread over the 4 lanes (frow = query, fq = 0..3).
__global__ __launch_bounds__(256) void attn_fwd_f32_mfma_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) float Ks[64 * GA_LD];
    __shared__ __attribute__((aligned(16))) float VsT[64 * GA_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * 64;
    const int qi = q0 + 16 * wave + frow;
    const int qc = min(qi, p.Tq - 1);
    f32x4 qx[4];
    {
        const float* qp = reinterpret_cast<const float*>(p.q) + b * p.q_bs + (int64_t)qc * p.q_rs + h * p.q_hs + 4 * fq;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qx[ks] = *reinterpret_cast<const f32x4*>(qp + 16 * ks);
    }
    f32x4 O[4];  // O[j][r] <-> query q0 + 16 wave + 4 fq + r, d = 16 j + frow
#pragma unroll
    for (int j = 0; j < 4; ++j) O[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = NEG_BIG, l = 0.f;  // of query frow; l is this lane's share (its 16 keys of every tile)
    const int off = p.Tk - p.Tq;
    const int kmax = p.causal ? qc + off : p.Tk - 1;  // last visible key of the lane's query
    const int q_last = min(q0 + 63, p.Tq - 1);
    const int k_end = p.causal ? min(p.Tk, q_last + off + 1) : p.Tk;
    const int srow = tid >> 2, sseg = (tid & 3) * 16;
    const float* kb = reinterpret_cast<const float*>(p.k) + b * p.k_bs + h * p.k_hs;
    const float* vb = reinterpret_cast<const float*>(p.v) + b * p.v_bs + h * p.v_hs;
    for (int k0 = 0; k0 < k_end; k0 += 64) {
        __syncthreads();
        {
            const int key = k0 + srow;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f}, bb = {0.f, 0.f, 0.f, 0.f};
                if (key < p.Tk) {
                    a = *reinterpret_cast<const f32x4*>(kb + (int64_t)key * p.k_rs + sseg + 4 * c);
                    bb = *reinterpret_cast<const f32x4*>(vb + (int64_t)key * p.v_rs + sseg + 4 * c);
                }
                *reinterpret_cast<f32x4*>(&Ks[srow * GA_LD + sseg + 4 * c]) = a;
#pragma unroll
                for (int e = 0; e < 4; ++e) VsT[(sseg + 4 * c + e) * GA_LD + srow] = bb[e];
            }
        }
        __syncthreads();
        f32x4 S[4];  // S[i][e] <-> key k0 + 16 i + 4 fq + e, query = the lane's
        float mx = NEG_BIG;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            S[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(&Ks[(16 * i + frow) * GA_LD + 16 * ks + 4 * fq]);
                Mma<float>::run(kf, qx[ks], S[i]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (k0 + 16 * i + 4 * fq + e > kmax) S[i][e] = NEG_BIG;
                mx = fmaxf(mx, S[i][e]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m, mx);
        if (__any(m_new > m)) {
            const float alpha = __expf(m - m_new);
            l *= alpha;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ar = __shfl(alpha, 4 * fq + r, 64);  // lane 4fq + r holds query 4fq + r of this wave
#pragma unroll
                for (int j = 0; j < 4; ++j) O[j][r] *= ar;
            }
            m = m_new;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pr = (S[i][e] <= NEG_TEST) ? 0.f : __expf(S[i][e] - m);
                S[i][e] = pr;
                l += pr;
            }
        // O[q][d] += sum_key P[q][key] V[key][d]   (k-step i = keys 16i .. 16i+15 of the tile)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 vt = *reinterpret_cast<const f32x4*>(&VsT[(16 * j + frow) * GA_LD + 16 * i + 4 * fq]);
                Mma<float>::run(S[i], vt, O[j]);
            }
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (p.lse && fq == 0 && qi < p.Tq) p.lse[((int64_t)b * gridDim.y + h) * p.Tq + qi] = m + __logf(l);
    const float inv = 1.f / l;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float ir = __shfl(inv, 4 * fq + r, 64);
        const int qq = q0 + 16 * wave + 4 * fq + r;
        if (qq < p.Tq) {
            float* op = reinterpret_cast<float*>(p.out) + b * p.o_bs + (int64_t)qq * p.o_rs + h * p.o_hs + frow;
#pragma unroll
            for (int j = 0; j < 4; ++j) op[16 * j] = O[j][r] * ir;
        }
    }
}

// =============================================================================
// K11 decode-step cross-attention
// =============================================================================
// WPH = waves per (b,h): 4 (the workgroup's waves split the keys; long caches, e.g. 1500 cross keys) or
// 1 (every wave owns one head and walks all its keys; short self-attention caches: no LDS merge, 4x fewer workgroups).
// NT: non-temporal loads (a cache read exactly once per step: the cached cross K / V); off for the growing self-attention cache,
// which the next step reads again.  Round 4: the self-attention of a decode step takes WPH = 4 with NT off once the cache can
// exceed 32 keys -- a wave of the one-wave form walks 32 keys per dependent round trip (7 of them at 224 keys: 6.6 us at 32
// keys, 9.9 us at 64, growing), four waves take 128 per round trip.
template <typename T, int WPH, bool NT = (WPH == 4)>
__global__ __launch_bounds__(256) void decode_attn_kernel(AttnParams p) {
    constexpr int EPL = Vec16<T>::EPL;  // elements per 16-byte load
    constexpr int LPK = 64 / EPL;       // lanes per key row (64 dims)
    constexpr int G = 64 / LPK;         // keys per wave instruction
    constexpr int U = 4;                // independent key groups in flight per iteration
    __shared__ float s_m[4], s_l[4];
    __shared__ float s_acc[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = WPH == 4 ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave, b = blockIdx.y;
    if (WPH == 1 && h >= p.H) return;  // no workgroup barrier on this path
    const int kw = WPH == 4 ? wave : 0;
    const int g = lane / LPK, c = lane % LPK;
    const int Tk = p.Tk + (p.tk_dev ? *p.tk_dev : 0);
    const int q_row0 = p.q_row_dev ? *p.q_row_dev : 0;
    float qf[EPL];
    {
        const T* qp = reinterpret_cast<const T*>(p.q) + b * p.q_bs + (int64_t)q_row0 * p.q_rs + h * p.q_hs + c * EPL;
        Vec16<T> qv = *reinterpret_cast<const Vec16<T>*>(qp);
#pragma unroll
        for (int e = 0; e < EPL; ++e) qf[e] = qv.get(e);
    }
    const T* Kb = reinterpret_cast<const T*>(p.k) + b * p.k_bs + h * p.k_hs + c * EPL;
    const T* Vb = reinterpret_cast<const T*>(p.v) + b * p.v_bs + h * p.v_hs + c * EPL;
    const int64_t k_rs = p.k_rs, v_rs = p.v_rs;
    float m = NEG_BIG, l = 0.f;
    float acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
    // software pipeline: the loads of key group i+1 are in flight while group i is reduced
    auto load_group = [&](int t0, Vec16<T>(&kv_)[U], Vec16<T>(&vv_)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = min(t0 + u * G + g, Tk - 1);
            if constexpr (NT) {  // cross-attention: 3.5 GB per step read exactly once -> non-temporal (nt) loads
                typedef decltype(kv_[u].v) VT;
                kv_[u].v = __builtin_nontemporal_load(reinterpret_cast<const VT*>(Kb + (int64_t)t * k_rs));
                vv_[u].v = __builtin_nontemporal_load(reinterpret_cast<const VT*>(Vb + (int64_t)t * v_rs));
            } else {
                kv_[u] = *reinterpret_cast<const Vec16<T>*>(Kb + (int64_t)t * k_rs);
                vv_[u] = *reinterpret_cast<const Vec16<T>*>(Vb + (int64_t)t * v_rs);
            }
        }
    };
    auto consume_group = [&](int t0, const Vec16<T>(&kvec)[U], const Vec16<T>(&vvec)[U]) {
        float s[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) a = fmaf(qf[e], kvec[u].get(e), a);
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
            for (int e = 0; e < EPL; ++e) acc[e] = fmaf(pr, vvec[u].get(e), acc[e]);
        }
        m = m_new;
    };
    // One register set (U K-rows + U V-rows in flight per lane), latency hidden by the other waves of the CU.
    // A register double-buffer measured the same 5.4 TB/s stand-alone but needs 135 VGPRs; the lean form keeps
    // the kernel small enough to share a SIMD with the 2 x 212-VGPR waves of the encoder GEMM of another
    // in-flight pass (see bench.py --pipeline), where it can use the HBM bandwidth the GEMM leaves idle.
    constexpr int STEP = WPH * G * U;
    for (int t0 = kw * G * U; t0 < Tk; t0 += STEP) {
        Vec16<T> ka[U], va[U];
        load_group(t0, ka, va);
        consume_group(t0, ka, va);
    }
    // merge the G key groups of this wave (lanes with equal c)
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
    if (WPH == 1) {
        if (lane < LPK) {
            const float inv = 1.f / l;
            T* op = reinterpret_cast<T*>(p.out) + b * p.o_bs + h * p.o_hs + c * EPL;
#pragma unroll
            for (int e = 0; e < EPL; ++e) op[e] = from_f32<T>(acc[e] * inv);
        }
        return;
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
        reinterpret_cast<T*>(p.out)[b * p.o_bs + h * p.o_hs + tid] = from_f32<T>(num / den);
    }
}

// Prompt prefill: the NQ prompt positions of a clip attend to the SAME cached cross K/V, so one pass over the cache serves
// all of them (the per-position decode step would stream the 3.5 GB cache NQ times).  Same decomposition and the same
// per-query arithmetic order as decode_attn_kernel<T, 4>: 4 waves split the keys, LDS merge; q / out rows are (b, t).
template <typename T, int NQ>
__global__ __launch_bounds__(256) void decode_attn_multi_kernel(AttnParams p) {
    constexpr int EPL = Vec16<T>::EPL;
    constexpr int LPK = 64 / EPL;
    constexpr int G = 64 / LPK;
    constexpr int U = 4;
    __shared__ float s_m[NQ][4], s_l[NQ][4];
    __shared__ float s_acc[NQ][4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, b = blockIdx.y;
    const int g = lane / LPK, c = lane % LPK;
    const int Tk = p.Tk;
    float qf[NQ][EPL];
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        const T* qp = reinterpret_cast<const T*>(p.q) + b * p.q_bs + (int64_t)t * p.q_rs + h * p.q_hs + c * EPL;
        Vec16<T> qv = *reinterpret_cast<const Vec16<T>*>(qp);
#pragma unroll
        for (int e = 0; e < EPL; ++e) qf[t][e] = qv.get(e);
    }
    const T* Kb = reinterpret_cast<const T*>(p.k) + b * p.k_bs + h * p.k_hs + c * EPL;
    const T* Vb = reinterpret_cast<const T*>(p.v) + b * p.v_bs + h * p.v_hs + c * EPL;
    const int64_t k_rs = p.k_rs, v_rs = p.v_rs;
    float m[NQ], l[NQ], acc[NQ][EPL];
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        m[t] = NEG_BIG;
        l[t] = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[t][e] = 0.f;
    }
    constexpr int STEP = 4 * G * U;
    for (int t0 = wave * G * U; t0 < Tk; t0 += STEP) {
        Vec16<T> ka[U], va[U];
        typedef decltype(ka[0].v) VT;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int key = min(t0 + u * G + g, Tk - 1);
            ka[u].v = __builtin_nontemporal_load(reinterpret_cast<const VT*>(Kb + (int64_t)key * k_rs));
            va[u].v = __builtin_nontemporal_load(reinterpret_cast<const VT*>(Vb + (int64_t)key * v_rs));
        }
#pragma unroll
        for (int t = 0; t < NQ; ++t) {
            float s[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float a = 0.f;
#pragma unroll
                for (int e = 0; e < EPL; ++e) a = fmaf(qf[t][e], ka[u].get(e), a);
#pragma unroll
                for (int o = 1; o < LPK; o <<= 1) a += __shfl_xor(a, o, 64);
                s[u] = (t0 + u * G + g < Tk) ? a : NEG_BIG;
            }
            float m_new = m[t];
#pragma unroll
            for (int u = 0; u < U; ++u) m_new = fmaxf(m_new, s[u]);
            const float alpha = __expf(m[t] - m_new);
            l[t] *= alpha;
#pragma unroll
            for (int e = 0; e < EPL; ++e) acc[t][e] *= alpha;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float pr = (s[u] <= NEG_TEST) ? 0.f : __expf(s[u] - m_new);
                l[t] += pr;
#pragma unroll
                for (int e = 0; e < EPL; ++e) acc[t][e] = fmaf(pr, va[u].get(e), acc[t][e]);
            }
            m[t] = m_new;
        }
    }
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
#pragma unroll
        for (int o = LPK; o < 64; o <<= 1) {
            const float m_o = __shfl_xor(m[t], o, 64);
            const float l_o = __shfl_xor(l[t], o, 64);
            const float m_n = fmaxf(m[t], m_o);
            const float a = __expf(m[t] - m_n), bsc = __expf(m_o - m_n);
            l[t] = l[t] * a + l_o * bsc;
#pragma unroll
            for (int e = 0; e < EPL; ++e) acc[t][e] = acc[t][e] * a + __shfl_xor(acc[t][e], o, 64) * bsc;
            m[t] = m_n;
        }
        if (lane < LPK) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) s_acc[t][wave][c * EPL + e] = acc[t][e];
            if (lane == 0) {
                s_m[t][wave] = m[t];
                s_l[t][wave] = l[t];
            }
        }
    }
    __syncthreads();
    if (tid < 64 * NQ) {
        const int t = tid >> 6, dd = tid & 63;
        const float mm = fmaxf(fmaxf(s_m[t][0], s_m[t][1]), fmaxf(s_m[t][2], s_m[t][3]));
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float sc = __expf(s_m[t][w] - mm);
            num += s_acc[t][w][dd] * sc;
            den += s_l[t][w] * sc;
        }
        reinterpret_cast<T*>(p.out)[b * p.o_bs + (int64_t)t * p.o_rs + h * p.o_hs + dd] = from_f32<T>(num / den);
    }
}

// =============================================================================
// K5 encoder flash attention, bf16 MFMA 32x32x16
// =============================================================================
constexpr int FA_ROWB = 128;                // bytes per LDS row (64 bf16)
constexpr int FA_TILE = 64 * FA_ROWB;       // 8 KiB
constexpr float LOG2E = 1.4426950408889634f;

// blockDim.x / 64 = 4 or 8 waves: 128 or 256 queries share every K / V^T tile (8 waves halve the L2->LDS fill per FLOP).
__global__ __launch_bounds__(512) void flash_enc_bf16_kernel(const __bf16* __restrict__ qk, int64_t ldqk,
                                                             const __bf16* __restrict__ vt, int64_t ldvt,
                                                             __bf16* __restrict__ out, int64_t ldo, int H, int T) {
    __shared__ __attribute__((aligned(16))) char smem[4 * FA_TILE];  // [buf][K tile | V^T tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int D = H * 64;
    const int nw = blockDim.x >> 6;
    const int q0 = blockIdx.x * (nw * 32) + wave * 32;
    const int nkt = (T + 63) / 64;

    // Q fragments: B operand of S^T = K * Q^T.  lane (r,hh) holds Q[q0+r][16s + 8hh + 0..7]
    bf16x8 qf[4];
    {
        const int qrow = min(q0 + r, T - 1);
        const __bf16* qp = qk + ((int64_t)b * T + qrow) * ldqk + h * 64 + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
    }
    // Staging by LDS-DMA (buffer_load_dwordx4 ... lds): no VGPR round trip, no ds_write.  A 1-KiB piece is 8 tile rows of
    // 128 B; lane l lands at byte 16 l of the piece, i.e. (row l>>3, physical chunk l&7), and fetches the SOURCE chunk
    // (l&7) ^ ((row>>1)&7) -- the same XOR the fragment reads apply.  Wave w moves pieces 2w, 2w+1 of the K tile and of
    // the V^T tile.  Keys beyond T are out of the K resource's range and read as zero; V^T is zero there already.
    typedef __attribute__((address_space(3))) void* lds_t;
    const __bf16* kbase = qk + (int64_t)b * T * ldqk + D + h * 64;
    const __bf16* vbase = vt + ((int64_t)b * D + h * 64) * ldvt;
    const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, (int)(((int64_t)(T - 1) * ldqk + 64) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, (int)(((int64_t)63 * ldvt + ldvt) * 2), 0x00020000);
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int ppw = 8 / nw;  // 1-KiB pieces of each tile per wave: 2 (four waves) or 1 (eight)
    int offK[2], offV[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 8 * (ppw * wv + i) + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        offK[i] = row * (int)ldqk * 2 + c * 16;   // + k0 * ldqk * 2 (scalar)
        offV[i] = row * (int)ldvt * 2 + c * 16;   // + k0 * 2 (scalar)
    }
    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * 2 * FA_TILE + (ppw * wv) * 1024;
        const int k0 = kt * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i < ppw) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rK, (lds_t)(base + i * 1024), 16, offK[i], k0 * (int)ldqk * 2, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rV, (lds_t)(base + FA_TILE + i * 1024), 16, offV[i], k0 * 2, 0, 0);
            }
        }
    };

    f32x16 O[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) O[0][i] = O[1][i] = 0.f;
    float m = NEG_BIG, l = 0.f;  // m in the log2 domain (scores * log2e)
    const int rsw = (r >> 1) & 7;  // swizzle term of rows 32x + r

    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) stage(kt + 1, (kt + 1) & 1);
        const char* kbuf = smem + (kt & 1) * 2 * FA_TILE;
        const char* vbuf = kbuf + FA_TILE;
        f32x16 S[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) S[u][i] = 0.f;
            const char* krow = kbuf + (32 * u + r) * FA_ROWB;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(krow + (((2 * s + hh) ^ rsw) << 4));
                S[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], S[u], 0, 0, 0);
            }
        }
        // row max on the raw scores (ragged last tile masked), then p = exp2(S*log2e - m) as one FMA + v_exp
        const bool ragged = (kt * 64 + 64 > T);
        if (ragged) {  // last tile only.  The empty asm keeps this a scalar BRANCH: if-converted, the 32 compares and
                       // selects ran on every tile and cost as much VALU time as the softmax itself.
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kt * 64 + 32 * u + (i & 3) + 8 * (i >> 2) + 4 * hh;
                    if (key >= T) S[u][i] = NEG_BIG;
                }
        }
        float mx = NEG_BIG;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, S[u][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m, mx * LOG2E);  // every tile holds at least one real key, so m_new is finite
        // two scores per VALU slot (v_pk_fma_f32 / v_pk_add_f32); only the exponentials stay scalar
        f32x2 psum2 = {0.f, 0.f};
        const f32x2 l2e = {LOG2E, LOG2E}, mneg = {-m_new, -m_new};
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const f32x2 x = __builtin_elementwise_fma(f32x2{S[u][i], S[u][i + 1]}, l2e, mneg);  // masked: exp2(-1e30) = 0
                const f32x2 pv = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
                S[u][i] = pv.x;
                S[u][i + 1] = pv.y;
                psum2 += pv;
            }
        const float psum = psum2.x + psum2.y;
        if (__any(m_new > m)) {  // wave-uniform: the running max rarely moves after the first tiles
            const float alpha = __builtin_amdgcn_exp2f(m - m_new);
            l *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                O[0][i] *= alpha;
                O[1][i] *= alpha;
            }
            m = m_new;
        }
        l += psum;
        // O^T += V^T * P^T : P^T comes straight from the S accumulators (k order of the
        // 32x32 C layout: element j of half hh is key 16s' + 8(j>>2) + 4hh + (j&3))
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                bf16x8 pb;
#pragma unroll
                for (int j = 0; j < 8; ++j) pb[j] = (__bf16)S[u][8 * sp + j];
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const char* vrow = vbuf + (32 * dt + r) * FA_ROWB + 8 * hh;
                    const int c0 = 4 * u + 2 * sp;
                    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(vrow + ((c0 ^ rsw) << 4));
                    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(vrow + (((c0 + 1) ^ rsw) << 4));
                    const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb, O[dt], 0, 0, 0);
                    // keep the dt = 0 / dt = 1 reads apart: merged into ds_read2st64_b64 their halves land in the wrong
                    // register pairs and cost 24 v_mov per tile on the (busier) VALU
                    asm volatile("" ::: "memory");
                }
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    l += __shfl_xor(l, 32, 64);
    const int qrow = q0 + r;
    if (qrow < T) {
        const float inv = 1.f / l;
        __bf16* op = out + ((int64_t)b * T + qrow) * ldo + h * 64 + 4 * hh;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                bf16x4 o = {(__bf16)(O[dt][4 * g4] * inv), (__bf16)(O[dt][4 * g4 + 1] * inv),
                            (__bf16)(O[dt][4 * g4 + 2] * inv), (__bf16)(O[dt][4 * g4 + 3] * inv)};
                *reinterpret_cast<bf16x4*>(op + 32 * dt + 8 * g4) = o;
            }
    }
}

// =============================================================================
// K5 in f32 (the reference's own dtype: set_dtype(float32)): encoder flash attention on the f32 MFMA
// (v_mfma_f32_32x32x2_f32, exact f32 products, f32 accumulation).  Same structure as the bf16 kernel:
// S^T = K Q^T so that a lane owns one query column, P stays in the S accumulators and feeds O^T += V^T P^T.
// The contraction index of an f32 MFMA step is 2 wide (lane half hh picks the element), so
//   * Q is held as qf[s] = Q[q][2s + hh] and the K tile is stored with even and odd k separated
//     ([key][parity][32]): a lane reads its 32 operands of a key row with 8 ds_read_b128;
//   * the keys of P^T are contracted in the order the 32x32 accumulator layout holds them
//     (step s of half hh = key 8(s>>2) + 4hh + (s&3)), which are 4 consecutive keys per ds_read_b128 of V^T.
// V arrives [key][d]; the staging transposes it into the V^T tile.
constexpr int FF_LD = 68;                  // floats per LDS row: 64 + 4 (conflict-free ds_read_b128 down a column)
constexpr int FF_TILE = 64 * FF_LD;        // floats per operand tile

__global__ __launch_bounds__(256, 2) void flash_enc_f32_kernel(const float* __restrict__ q, int64_t ldq,
                                                               const float* __restrict__ k, int64_t ldk,
                                                               const float* __restrict__ v, int64_t ldv,
                                                               float* __restrict__ out, int64_t ldo, int T) {
    __shared__ __attribute__((aligned(16))) float smem[4 * FF_TILE];  // [buf][K tile | V^T tile], 69 632 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int nkt = (T + 63) / 64;

    float qf[32];
    {
        const float* qp = q + ((int64_t)b * T + min(q0 + r, T - 1)) * ldq + h * 64 + hh;
#pragma unroll
        for (int s = 0; s < 32; ++s) qf[s] = qp[2 * s];
    }
    // staging roles.  K: key row tid>>2, 16 consecutive k (segment tid&3).  V: key tid&63, 16 consecutive d (tid>>6).
    const int krow = tid >> 2, kseg = tid & 3;
    const int vkey = tid & 63, vseg = tid >> 6;
    const float* kg = k + (int64_t)b * T * ldk + h * 64 + 16 * kseg;
    const float* vg = v + (int64_t)b * T * ldv + h * 64 + 16 * vseg;
    f32x4 rk[4], rv[4];
    auto gload = [&](int kt) {
        const float* kp = kg + (int64_t)min(kt * 64 + krow, T - 1) * ldk;
        const float* vp = vg + (int64_t)min(kt * 64 + vkey, T - 1) * ldv;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            rk[c] = *reinterpret_cast<const f32x4*>(kp + 4 * c);
            rv[c] = *reinterpret_cast<const f32x4*>(vp + 4 * c);
        }
    };
    auto swrite = [&](int buf) {
        float* kt = smem + buf * 2 * FF_TILE + krow * FF_LD + 8 * kseg;
        // even k -> [0, 32), odd k -> [32, 64) of the key row
        *reinterpret_cast<f32x4*>(kt) = f32x4{rk[0][0], rk[0][2], rk[1][0], rk[1][2]};
        *reinterpret_cast<f32x4*>(kt + 4) = f32x4{rk[2][0], rk[2][2], rk[3][0], rk[3][2]};
        *reinterpret_cast<f32x4*>(kt + 32) = f32x4{rk[0][1], rk[0][3], rk[1][1], rk[1][3]};
        *reinterpret_cast<f32x4*>(kt + 36) = f32x4{rk[2][1], rk[2][3], rk[3][1], rk[3][3]};
        float* vt = smem + buf * 2 * FF_TILE + FF_TILE + (16 * vseg) * FF_LD + vkey;  // V^T[d][key]
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) vt[(4 * c + e) * FF_LD] = rv[c][e];
    };

    f32x16 O[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) O[0][i] = O[1][i] = 0.f;
    float m = NEG_BIG, l = 0.f;  // m in the log2 domain

    gload(0);
    swrite(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) gload(kt + 1);
        const float* kbuf = smem + (kt & 1) * 2 * FF_TILE;
        const float* vbuf = kbuf + FF_TILE;
        f32x16 S[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) S[u][i] = 0.f;
            const float* kr = kbuf + (32 * u + r) * FF_LD + 32 * hh;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(kr + 4 * c);
#pragma unroll
                for (int e = 0; e < 4; ++e) S[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[4 * c + e], S[u], 0, 0, 0);
            }
        }
        if (kt * 64 + 64 > T) {  // ragged last tile: a scalar branch (see the bf16 kernel)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kt * 64 + 32 * u + (i & 3) + 8 * (i >> 2) + 4 * hh;
                    if (key >= T) S[u][i] = NEG_BIG;
                }
        }
        float mx = NEG_BIG;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, S[u][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m, mx * LOG2E);
        f32x2 psum2 = {0.f, 0.f};
        const f32x2 l2e = {LOG2E, LOG2E}, mneg = {-m_new, -m_new};
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const f32x2 x = __builtin_elementwise_fma(f32x2{S[u][i], S[u][i + 1]}, l2e, mneg);
                const f32x2 pv = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
                S[u][i] = pv.x;
                S[u][i + 1] = pv.y;
                psum2 += pv;
            }
        if (__any(m_new > m)) {
            const float alpha = __builtin_amdgcn_exp2f(m - m_new);
            l *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                O[0][i] *= alpha;
                O[1][i] *= alpha;
            }
            m = m_new;
        }
        l += psum2.x + psum2.y;
        // O^T[d][q] += V^T[d][key] P^T[key][q]; step s of accumulator element s: key 32u + 8(s>>2) + 4hh + (s&3)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const float* vr = vbuf + (32 * dt + r) * FF_LD + 32 * u + 4 * hh;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 vf = *reinterpret_cast<const f32x4*>(vr + 8 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        O[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[e], S[u][4 * g + e], O[dt], 0, 0, 0);
                }
            }
        if (kt + 1 < nkt) swrite((kt + 1) & 1);
        __syncthreads();
    }
    l += __shfl_xor(l, 32, 64);
    const int qrow = q0 + r;
    if (qrow < T) {
        const float inv = 1.f / l;
        float* op = out + ((int64_t)b * T + qrow) * ldo + h * 64 + 4 * hh;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                *reinterpret_cast<f32x4*>(op + 32 * dt + 8 * g4) =
                    f32x4{O[dt][4 * g4] * inv, O[dt][4 * g4 + 1] * inv, O[dt][4 * g4 + 2] * inv, O[dt][4 * g4 + 3] * inv};
    }
}

// K5 in f32, fast form: every product on the bf16 MFMA (32x32x16) with the f32 operands split in registers.  The scores
// S^T = K Q^T use THREE terms per operand and the six largest cross products (error ~2^-24: an error eps in a score is a
// relative error eps*|s| in its probability, and scores can be large); O^T += V^T P^T -- probabilities in [0, 1] against
// values -- uses two terms per operand and three products, as the f32 tile GEMMs do.  9 bf16 MFMAs replace 16 f32-MFMA
// equivalents that run 16x slower; the conversions (VALU) set the pace.  K tile [key][64 d], V^T tile [d][64 keys], f32.
__global__ __launch_bounds__(256, 2) void flash_enc_f32s_kernel(const float* __restrict__ q, int64_t ldq,
                                                                const float* __restrict__ k, int64_t ldk,
                                                                const float* __restrict__ v, int64_t ldv,
                                                                float* __restrict__ out, int64_t ldo, int T) {
    __shared__ __attribute__((aligned(16))) float smem[4 * FF_TILE];  // [buf][K tile | V^T tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int nkt = (T + 63) / 64;

    // Q fragments (B operand): lane (r, hh) holds Q[q0+r][16s + 8hh + 0..7], split once
    bf16x8 qh[4], qm[4], ql[4];
    {
        const float* qp = q + ((int64_t)b * T + min(q0 + r, T - 1)) * ldq + h * 64 + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            split_bf16x3(*reinterpret_cast<const f32x4*>(qp + 16 * s), *reinterpret_cast<const f32x4*>(qp + 16 * s + 4), qh[s], qm[s], ql[s]);
    }
    const int krow = tid >> 2, kseg = tid & 3;
    const int vkey = tid & 63, vseg = tid >> 6;
    const float* kg = k + (int64_t)b * T * ldk + h * 64 + 16 * kseg;
    const float* vg = v + (int64_t)b * T * ldv + h * 64 + 16 * vseg;
    f32x4 rk[4], rv[4];
    auto gload = [&](int kt) {
        const float* kp = kg + (int64_t)min(kt * 64 + krow, T - 1) * ldk;
        const float* vp = vg + (int64_t)min(kt * 64 + vkey, T - 1) * ldv;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            rk[c] = *reinterpret_cast<const f32x4*>(kp + 4 * c);
            rv[c] = *reinterpret_cast<const f32x4*>(vp + 4 * c);
        }
    };
    auto swrite = [&](int buf) {
        float* kt = smem + buf * 2 * FF_TILE + krow * FF_LD + 16 * kseg;
#pragma unroll
        for (int c = 0; c < 4; ++c) *reinterpret_cast<f32x4*>(kt + 4 * c) = rk[c];
        float* vt = smem + buf * 2 * FF_TILE + FF_TILE + (16 * vseg) * FF_LD + vkey;  // V^T[d][key]
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) vt[(4 * c + e) * FF_LD] = rv[c][e];
    };

    f32x16 O[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) O[0][i] = O[1][i] = 0.f;
    float m = NEG_BIG, l = 0.f;  // m in the log2 domain

    gload(0);
    swrite(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) gload(kt + 1);
        const float* kbuf = smem + (kt & 1) * 2 * FF_TILE;
        const float* vbuf = kbuf + FF_TILE;
        f32x16 S[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) S[u][i] = 0.f;
            const float* kr = kbuf + (32 * u + r) * FF_LD + 8 * hh;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bf16x8 kh, km, kl;
                split_bf16x3(*reinterpret_cast<const f32x4*>(kr + 16 * s), *reinterpret_cast<const f32x4*>(kr + 16 * s + 4), kh, km, kl);
                S[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[s], S[u], 0, 0, 0);  // smallest terms first
                S[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(km, qm[s], S[u], 0, 0, 0);
                S[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[s], S[u], 0, 0, 0);
                S[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(km, qh[s], S[u], 0, 0, 0);
                S[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qm[s], S[u], 0, 0, 0);
                S[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[s], S[u], 0, 0, 0);
            }
        }
        if (kt * 64 + 64 > T) {  // ragged last tile: a scalar branch
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kt * 64 + 32 * u + (i & 3) + 8 * (i >> 2) + 4 * hh;
                    if (key >= T) S[u][i] = NEG_BIG;
                }
        }
        float mx = NEG_BIG;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, S[u][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m, mx * LOG2E);
        f32x2 psum2 = {0.f, 0.f};
        const f32x2 l2e = {LOG2E, LOG2E}, mneg = {-m_new, -m_new};
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const f32x2 x = __builtin_elementwise_fma(f32x2{S[u][i], S[u][i + 1]}, l2e, mneg);
                const f32x2 pv = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
                S[u][i] = pv.x;
                S[u][i + 1] = pv.y;
                psum2 += pv;
            }
        if (__any(m_new > m)) {
            const float alpha = __builtin_amdgcn_exp2f(m - m_new);
            l *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                O[0][i] *= alpha;
                O[1][i] *= alpha;
            }
            m = m_new;
        }
        l += psum2.x + psum2.y;
        // O^T += V^T P^T; MFMA k slot j of half hh is key 16 sp + 8 (j>>2) + 4 hh + (j&3) of the 32-key block u
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                bf16x8 ph, pl;
                split_bf16x2(f32x4{S[u][8 * sp], S[u][8 * sp + 1], S[u][8 * sp + 2], S[u][8 * sp + 3]},
                             f32x4{S[u][8 * sp + 4], S[u][8 * sp + 5], S[u][8 * sp + 6], S[u][8 * sp + 7]}, ph, pl);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const float* vr = vbuf + (32 * dt + r) * FF_LD + 32 * u + 16 * sp + 4 * hh;
                    bf16x8 vh, vl;
                    split_bf16x2(*reinterpret_cast<const f32x4*>(vr), *reinterpret_cast<const f32x4*>(vr + 8), vh, vl);
                    O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl, O[dt], 0, 0, 0);
                    O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph, O[dt], 0, 0, 0);
                    O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, O[dt], 0, 0, 0);
                }
            }
        if (kt + 1 < nkt) swrite((kt + 1) & 1);
        __syncthreads();
    }
    l += __shfl_xor(l, 32, 64);
    const int qrow = q0 + r;
    if (qrow < T) {
        const float inv = 1.f / l;
        float* op = out + ((int64_t)b * T + qrow) * ldo + h * 64 + 4 * hh;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                *reinterpret_cast<f32x4*>(op + 32 * dt + 8 * g4) =
                    f32x4{O[dt][4 * g4] * inv, O[dt][4 * g4 + 1] * inv, O[dt][4 * g4 + 2] * inv, O[dt][4 * g4 + 3] * inv};
    }
}

}  // namespace

extern "C" int wipa_attention(const wipa_attn_desc* d, wipa_stream_t stream) {
    WIPA_REQUIRE(d && d->q && d->k && d->v && d->out, "wipa_attention: null pointer");
    WIPA_REQUIRE(d->B > 0 && d->H > 0 && d->Tq > 0, "wipa_attention: bad shape");
    WIPA_REQUIRE(d->dtype == WIPA_F32 || d->dtype == WIPA_BF16, "wipa_attention: bad dtype %d", d->dtype);
    const int64_t al = d->dtype == WIPA_BF16 ? 8 : 4;
    WIPA_REQUIRE(d->q_rs % al == 0 && d->k_rs % al == 0 && d->v_rs % al == 0 && d->q_hs % al == 0 && d->k_hs % al == 0 &&
                     d->v_hs % al == 0 && d->q_bs % al == 0 && d->k_bs % al == 0 && d->v_bs % al == 0,
                 "wipa_attention: strides must keep 16-byte alignment");
    AttnParams p;
    p.q = (const char*)d->q;
    p.k = (const char*)d->k;
    p.v = (const char*)d->v;
    p.out = (char*)d->out;
    p.tk_dev = d->tk_dev;
    p.q_row_dev = d->q_row_dev;
    p.lse = d->lse;
    p.q_bs = d->q_bs; p.q_rs = d->q_rs; p.q_hs = d->q_hs;
    p.k_bs = d->k_bs; p.k_rs = d->k_rs; p.k_hs = d->k_hs;
    p.v_bs = d->v_bs; p.v_rs = d->v_rs; p.v_hs = d->v_hs;
    p.o_bs = d->o_bs; p.o_rs = d->o_rs; p.o_hs = d->o_hs;
    p.Tq = d->Tq;
    p.Tk = d->Tk;
    p.causal = d->causal;
    dim3 grid((d->Tq + 15) / 16, d->H, d->B);
    // float32 with whole 16-query fragments and no device-side offsets (teacher-forced decoder, fine-tune step): f32 MFMA kernel;
    // the few-row prefill and everything bf16 stay on the VALU kernel.  WIPA_ATTN_FWD=valu keeps it for A/B runs.
    static const bool valu = [] { const char* e = getenv("WIPA_ATTN_FWD"); return e && !strcmp(e, "valu"); }();
    if (d->dtype == WIPA_F32 && !valu && d->Tq >= 16 && !d->tk_dev && !d->q_row_dev && d->o_rs % 4 == 0 && d->o_hs % 4 == 0 &&
        d->o_bs % 4 == 0) {
        hipLaunchKernelGGL(attn_fwd_f32_mfma_kernel, dim3((d->Tq + 63) / 64, d->H, d->B), dim3(256), 0, (hipStream_t)stream, p);
        WIPA_LAUNCH_CHECK();
        return WIPA_OK;
    }
    if (d->dtype == WIPA_F32)
        hipLaunchKernelGGL((attn_generic_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((attn_generic_kernel<__bf16>), grid, dim3(256), 0, (hipStream_t)stream, p);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

extern "C" int wipa_flash_attn_enc_bf16(const void* qk, int64_t ldqk, const void* vt, int64_t ldvt, void* out, int64_t ldo,
                                        int B, int H, int T, wipa_stream_t stream) {
    WIPA_REQUIRE(qk && vt && out, "wipa_flash_attn_enc_bf16: null pointer");
    WIPA_REQUIRE(ldqk % 8 == 0 && ldvt % 8 == 0 && ldo % 4 == 0, "wipa_flash_attn_enc_bf16: ld alignment");
    WIPA_REQUIRE(ldvt >= ((T + 63) / 64) * 64, "wipa_flash_attn_enc_bf16: ldvt=%lld must cover ceil64(T)", (long long)ldvt);
    WIPA_REQUIRE(B > 0 && H > 0 && T > 0, "wipa_flash_attn_enc_bf16: bad shape");
    static const int q128 = [] { const char* e = getenv("WIPA_FLASH_Q128"); return e ? atoi(e) : 0; }();
    const int nw = (T > 256 && !q128) ? 8 : 4;
    dim3 grid((T + nw * 32 - 1) / (nw * 32), H, B);
    hipLaunchKernelGGL(flash_enc_bf16_kernel, grid, dim3(nw * 64), 0, (hipStream_t)stream, (const __bf16*)qk, ldqk,
                       (const __bf16*)vt, ldvt, (__bf16*)out, ldo, H, T);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

extern "C" int wipa_flash_attn_enc_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                       float* out, int64_t ldo, int B, int H, int T, int f32_split, wipa_stream_t stream) {
    WIPA_REQUIRE(q && k && v && out, "wipa_flash_attn_enc_f32: null pointer");
    WIPA_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0, "wipa_flash_attn_enc_f32: row strides must be multiples of 4");
    WIPA_REQUIRE(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0 && ((uintptr_t)out % 16) == 0,
                 "wipa_flash_attn_enc_f32: operands must be 16-byte aligned");
    WIPA_REQUIRE(B > 0 && H > 0 && T > 0, "wipa_flash_attn_enc_f32: bad shape");
    dim3 grid((T + 127) / 128, H, B);
    // exact f32 products unless the caller opted into the split-bf16 kernel
    if (!f32_split)
        hipLaunchKernelGGL(flash_enc_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, q, ldq, k, ldk, v, ldv, out, ldo, T);
    else
        hipLaunchKernelGGL(flash_enc_f32s_kernel, grid, dim3(256), 0, (hipStream_t)stream, q, ldq, k, ldk, v, ldv, out, ldo, T);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

static int fill_attn_params(const wipa_attn_desc* d, AttnParams& p) {
    p.q = (const char*)d->q;
    p.k = (const char*)d->k;
    p.v = (const char*)d->v;
    p.out = (char*)d->out;
    p.tk_dev = d->tk_dev;
    p.q_row_dev = d->q_row_dev;
    p.lse = d->lse;
    p.q_bs = d->q_bs; p.q_rs = d->q_rs; p.q_hs = d->q_hs;
    p.k_bs = d->k_bs; p.k_rs = d->k_rs; p.k_hs = d->k_hs;
    p.v_bs = d->v_bs; p.v_rs = d->v_rs; p.v_hs = d->v_hs;
    p.o_bs = d->o_bs; p.o_rs = d->o_rs; p.o_hs = d->o_hs;
    p.Tq = d->Tq;
    p.Tk = d->Tk;
    p.causal = d->causal;
    p.H = d->H;
    return WIPA_OK;
}

extern "C" int wipa_decode_attn(const wipa_attn_desc* d, wipa_stream_t stream) {
    WIPA_REQUIRE(d && d->q && d->k && d->v && d->out, "wipa_decode_attn: null pointer");
    WIPA_REQUIRE(d->B > 0 && d->H > 0 && d->Tq == 1, "wipa_decode_attn: one query row per (b,h) (Tq=%d)", d->Tq);
    WIPA_REQUIRE(d->dtype == WIPA_F32 || d->dtype == WIPA_BF16, "wipa_decode_attn: bad dtype %d", d->dtype);
    const int64_t al = d->dtype == WIPA_BF16 ? 8 : 4;
    WIPA_REQUIRE(d->q_rs % al == 0 && d->k_rs % al == 0 && d->v_rs % al == 0 && d->q_hs % al == 0 && d->k_hs % al == 0 &&
                     d->v_hs % al == 0 && d->q_bs % al == 0 && d->k_bs % al == 0 && d->v_bs % al == 0,
                 "wipa_decode_attn: strides must keep 16-byte alignment");
    AttnParams p;
    fill_attn_params(d, p);
    // short caches (the growing self-attention cache, <= n_text_ctx keys): default-policy loads, and four waves per head
    // (WIPA_SELF_ATTN_WAVES=1: the one-wave-per-head form of rounds 1-3); long ones (cached cross K / V): 4 waves, nt loads
    const bool short_cache = d->Tk + (d->tk_dev ? 448 : 0) <= 512;
    static const int self_waves = [] { const char* e = getenv("WIPA_SELF_ATTN_WAVES"); return e ? atoi(e) : 4; }();
    if (short_cache && self_waves == 4) {
        dim3 grid(d->H, d->B);
        if (d->dtype == WIPA_F32)
            hipLaunchKernelGGL((decode_attn_kernel<float, 4, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else
            hipLaunchKernelGGL((decode_attn_kernel<__bf16, 4, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
    } else if (short_cache) {
        dim3 grid((d->H + 3) / 4, d->B);
        if (d->dtype == WIPA_F32)
            hipLaunchKernelGGL((decode_attn_kernel<float, 1>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else
            hipLaunchKernelGGL((decode_attn_kernel<__bf16, 1>), grid, dim3(256), 0, (hipStream_t)stream, p);
    } else {
        dim3 grid(d->H, d->B);
        if (d->dtype == WIPA_F32)
            hipLaunchKernelGGL((decode_attn_kernel<float, 4>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else
            hipLaunchKernelGGL((decode_attn_kernel<__bf16, 4>), grid, dim3(256), 0, (hipStream_t)stream, p);
    }
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

extern "C" int wipa_decode_cross_attn(const void* q, const void* kv, void* out, int B, int H, int Tk, int dtype,
                                      wipa_stream_t stream) {
    WIPA_REQUIRE(q && kv && out, "wipa_decode_cross_attn: null pointer");
    WIPA_REQUIRE(B > 0 && H > 0 && Tk > 0, "wipa_decode_cross_attn: bad shape");
    WIPA_REQUIRE(dtype == WIPA_F32 || dtype == WIPA_BF16, "wipa_decode_cross_attn: bad dtype %d", dtype);
    wipa_attn_desc d;
    memset(&d, 0, sizeof(d));
    const size_t e = wipa_dtype_size(dtype);
    d.q = q; d.k = kv; d.v = (const char*)kv + (size_t)H * Tk * 64 * e; d.out = out;
    d.q_bs = (int64_t)H * 64; d.q_rs = (int64_t)H * 64; d.q_hs = 64;
    d.k_bs = (int64_t)2 * H * Tk * 64; d.k_rs = 64; d.k_hs = (int64_t)Tk * 64;
    d.v_bs = d.k_bs; d.v_rs = 64; d.v_hs = d.k_hs;
    d.o_bs = (int64_t)H * 64; d.o_rs = (int64_t)H * 64; d.o_hs = 64;
    d.B = B; d.H = H; d.Tq = 1; d.Tk = Tk; d.causal = 0; d.dtype = dtype;
    return wipa_decode_attn(&d, stream);
}

extern "C" int wipa_decode_cross_attn_multi(const void* q, const void* kv, void* out, int B, int H, int Tk, int n_q, int dtype,
                                            wipa_stream_t stream) {
    WIPA_REQUIRE(q && kv && out, "wipa_decode_cross_attn_multi: null pointer");
    WIPA_REQUIRE(B > 0 && H > 0 && Tk > 0 && n_q >= 1 && n_q <= 4, "wipa_decode_cross_attn_multi: bad shape (n_q=%d)", n_q);
    WIPA_REQUIRE(dtype == WIPA_F32 || dtype == WIPA_BF16, "wipa_decode_cross_attn_multi: bad dtype %d", dtype);
    if (n_q == 1) return wipa_decode_cross_attn(q, kv, out, B, H, Tk, dtype, stream);
    wipa_attn_desc d;
    memset(&d, 0, sizeof(d));
    const size_t e = wipa_dtype_size(dtype);
    d.q = q; d.k = kv; d.v = (const char*)kv + (size_t)H * Tk * 64 * e; d.out = out;
    d.q_bs = (int64_t)n_q * H * 64; d.q_rs = (int64_t)H * 64; d.q_hs = 64;
    d.k_bs = (int64_t)2 * H * Tk * 64; d.k_rs = 64; d.k_hs = (int64_t)Tk * 64;
    d.v_bs = d.k_bs; d.v_rs = 64; d.v_hs = d.k_hs;
    d.o_bs = d.q_bs; d.o_rs = d.q_rs; d.o_hs = 64;
    d.B = B; d.H = H; d.Tq = n_q; d.Tk = Tk; d.causal = 0; d.dtype = dtype;
    AttnParams p;
    fill_attn_params(&d, p);
    const dim3 grid(H, B);
    hipStream_t s = (hipStream_t)stream;
#define WIPA_MULTI(T, NQ) hipLaunchKernelGGL((decode_attn_multi_kernel<T, NQ>), grid, dim3(256), 0, s, p)
    if (dtype == WIPA_F32) {
        if (n_q == 2) WIPA_MULTI(float, 2); else if (n_q == 3) WIPA_MULTI(float, 3); else WIPA_MULTI(float, 4);
    } else {
        if (n_q == 2) WIPA_MULTI(__bf16, 2); else if (n_q == 3) WIPA_MULTI(__bf16, 3); else WIPA_MULTI(__bf16, 4);
    }
#undef WIPA_MULTI
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

