// Backward of MultiHeadAttention.qkv_attention (head_dim 64, f32) for the decoder fine-tune step:
// causal self-attention and cross-attention of the teacher-forced decoder
// (scripts/train_whisper_ipa.py:232,284).  Flash-style: the probabilities are recomputed from the
// saved log-sum-exp, nothing of size Tq x Tk is stored.  Deterministic (no atomics):
//   kernel 1 (query-major): D_i = dO_i . O_i ;  dQ_i = sum_j dS_ij K_j
//   kernel 2 (key-major)  : dV_j = sum_i P_ij dO_i ;  dK_j = sum_i dS_ij Q_i
// with P_ij = exp(q_i.k_j - lse_i), dS_ij = P_ij (dO_i.v_j - D_i).  q and k are the PRE-SCALED
// values the forward used (64^-0.25 each); dq/dk are multiplied by that scale on the way out so
// they are gradients of the unscaled projections.
#include <cstdlib>
#include <cstring>

#include "wipa_common.h"

namespace {

struct AttnBwdParams {
    const float* q;
    const float* k;
    const float* v;
    const float* o;
    const float* d_o;
    const float* lse;
    float* dq;
    float* dk;
    float* dv;
    float* dvec;  // [B,H,Tq] scratch: D_i
    int64_t q_bs, q_rs, q_hs, k_bs, k_rs, k_hs, v_bs, v_rs, v_hs, o_bs, o_rs, o_hs;
    int H, Tq, Tk, causal;
    float qk_scale;
};

constexpr int LD = 68;

// ---- kernel 1: 16 queries x 16 key lanes per workgroup (mirror of the forward kernel)
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnBwdParams p) {
    __shared__ __attribute__((aligned(16))) float Ks[64 * LD];
    __shared__ __attribute__((aligned(16))) float Vs[64 * LD];
    const int tid = threadIdx.x, ql = tid >> 4, kl = tid & 15;
    const int h = blockIdx.y, b = blockIdx.z;
    const int qi = blockIdx.x * 16 + ql;
    const int qc = min(qi, p.Tq - 1);
    const float* qp = p.q + b * p.q_bs + (int64_t)qc * p.q_rs + h * p.q_hs;
    const float* op = p.o + b * p.o_bs + (int64_t)qc * p.o_rs + h * p.o_hs;
    const float* gp = p.d_o + b * p.o_bs + (int64_t)qc * p.o_rs + h * p.o_hs;
    float q[64], g[64], dq[64];
    float dsum = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(qp + 4 * c);
        const f32x4 gg = *reinterpret_cast<const f32x4*>(gp + 4 * c);
        const f32x4 oo = *reinterpret_cast<const f32x4*>(op + 4 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            q[4 * c + e] = a[e];
            g[4 * c + e] = gg[e];
            dq[4 * c + e] = 0.f;
            dsum = fmaf(gg[e], oo[e], dsum);
        }
    }
    const float lse = p.lse[((int64_t)b * p.H + h) * p.Tq + qc];
    if (kl == 0 && qi < p.Tq) p.dvec[((int64_t)b * p.H + h) * p.Tq + qi] = dsum;
    const int kmax = p.causal ? (qc + (p.Tk - p.Tq)) : (p.Tk - 1);
    const int q_last = min(blockIdx.x * 16 + 15, p.Tq - 1);
    const int k_end = p.causal ? min(p.Tk, q_last + (p.Tk - p.Tq) + 1) : p.Tk;
    const int srow = tid >> 2, sseg = (tid & 3) * 16;
    const float* kb = p.k + b * p.k_bs + h * p.k_hs;
    const float* vb = p.v + b * p.v_bs + h * p.v_hs;
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
                *reinterpret_cast<f32x4*>(&Ks[srow * LD + sseg + 4 * c]) = a;
                *reinterpret_cast<f32x4*>(&Vs[srow * LD + sseg + 4 * c]) = bb;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kk = kl + 16 * i;
            if (k0 + kk > kmax) continue;
            const float* kr = &Ks[kk * LD];
            const float* vr = &Vs[kk * LD];
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d4 = 0; d4 < 16; ++d4) {
                const f32x4 k4 = *reinterpret_cast<const f32x4*>(kr + 4 * d4);
                const f32x4 v4 = *reinterpret_cast<const f32x4*>(vr + 4 * d4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s = fmaf(q[4 * d4 + e], k4[e], s);
                    dp = fmaf(g[4 * d4 + e], v4[e], dp);
                }
            }
            const float ds = __expf(s - lse) * (dp - dsum);
#pragma unroll
            for (int d4 = 0; d4 < 16; ++d4) {
                const f32x4 k4 = *reinterpret_cast<const f32x4*>(kr + 4 * d4);
#pragma unroll
                for (int e = 0; e < 4; ++e) dq[4 * d4 + e] = fmaf(ds, k4[e], dq[4 * d4 + e]);
            }
        }
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1)
#pragma unroll
        for (int d = 0; d < 64; ++d) dq[d] += __shfl_xor(dq[d], o, 64);
    if (qi < p.Tq) {
        float* dp_ = p.dq + b * p.q_bs + (int64_t)qi * p.q_rs + h * p.q_hs;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (kl == j)
#pragma unroll
                for (int e = 0; e < 4; ++e) dp_[4 * j + e] = dq[4 * j + e] * p.qk_scale;
    }
}

// ---- kernel 2: 16 keys per workgroup; the 16 lanes of a key own 4 dims each and walk all queries
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnBwdParams p) {
    __shared__ __attribute__((aligned(16))) float Qs[64 * LD];
    __shared__ __attribute__((aligned(16))) float Gs[64 * LD];
    __shared__ float s_lse[64], s_d[64];
    const int tid = threadIdx.x, kl = tid >> 4, dl = tid & 15;  // key in tile, dim group
    const int h = blockIdx.y, b = blockIdx.z;
    const int kj = blockIdx.x * 16 + kl;
    const int kc = min(kj, p.Tk - 1);
    const f32x4 k4 = *reinterpret_cast<const f32x4*>(p.k + b * p.k_bs + (int64_t)kc * p.k_rs + h * p.k_hs + 4 * dl);
    const f32x4 v4 = *reinterpret_cast<const f32x4*>(p.v + b * p.v_bs + (int64_t)kc * p.v_rs + h * p.v_hs + 4 * dl);
    f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
    // causal: key j is seen by queries i >= j - (Tk - Tq)
    const int k_first = blockIdx.x * 16;
    const int q_begin = p.causal ? max(0, k_first - (p.Tk - p.Tq)) : 0;
    const int srow = tid >> 2, sseg = (tid & 3) * 16;
    const float* qb = p.q + b * p.q_bs + h * p.q_hs;
    const float* gb = p.d_o + b * p.o_bs + h * p.o_hs;
    for (int q0 = (q_begin / 64) * 64; q0 < p.Tq; q0 += 64) {
        __syncthreads();
        {
            const int qi = q0 + srow;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f}, g = {0.f, 0.f, 0.f, 0.f};
                if (qi < p.Tq) {
                    a = *reinterpret_cast<const f32x4*>(qb + (int64_t)qi * p.q_rs + sseg + 4 * c);
                    g = *reinterpret_cast<const f32x4*>(gb + (int64_t)qi * p.o_rs + sseg + 4 * c);
                }
                *reinterpret_cast<f32x4*>(&Qs[srow * LD + sseg + 4 * c]) = a;
                *reinterpret_cast<f32x4*>(&Gs[srow * LD + sseg + 4 * c]) = g;
            }
            if (tid < 64) {
                const int qq = q0 + tid;
                s_lse[tid] = qq < p.Tq ? p.lse[((int64_t)b * p.H + h) * p.Tq + qq] : 0.f;
                s_d[tid] = qq < p.Tq ? p.dvec[((int64_t)b * p.H + h) * p.Tq + qq] : 0.f;
            }
        }
        __syncthreads();
        const int n = min(64, p.Tq - q0);
        for (int i = 0; i < n; ++i) {
            const f32x4 q4 = *reinterpret_cast<const f32x4*>(&Qs[i * LD + 4 * dl]);
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(&Gs[i * LD + 4 * dl]);
            float s = (q4[0] * k4[0] + q4[1] * k4[1]) + (q4[2] * k4[2] + q4[3] * k4[3]);
            float dp = (g4[0] * v4[0] + g4[1] * v4[1]) + (g4[2] * v4[2] + g4[3] * v4[3]);
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                s += __shfl_xor(s, o, 64);
                dp += __shfl_xor(dp, o, 64);
            }
            const bool visible = !p.causal || (kj <= q0 + i + (p.Tk - p.Tq));
            const float pr = visible ? __expf(s - s_lse[i]) : 0.f;
            const float ds = pr * (dp - s_d[i]);
            dv += pr * g4;
            dk += ds * q4;
        }
    }
    if (kj < p.Tk) {
        *reinterpret_cast<f32x4*>(p.dk + b * p.k_bs + (int64_t)kj * p.k_rs + h * p.k_hs + 4 * dl) = dk * p.qk_scale;
        *reinterpret_cast<f32x4*>(p.dv + b * p.v_bs + (int64_t)kj * p.v_rs + h * p.v_hs + 4 * dl) = dv;
    }
}

// ---- kernel 2 on the f32 MFMA (16x16x4, exact f32 products): 64 keys per workgroup, wave w owns keys 16w .. 16w+15 and walks
// the queries in tiles of 64.  The trick that keeps P out of LDS: the scores are computed with the QUERIES on the MFMA's row
// side, S[q][key] = sum_d Q[q][d] K[key][d], so a lane ends up with 4 consecutive q of ONE key -- exactly the fragment the
// next contractions over q need (dV[key][d] += sum_q P[q][key] dO[q][d], dK[key][d] += sum_q dS[q][key] Q[q][d]); the
// probabilities go from the accumulators straight back into the matrix pipe.  K / V rows of the wave live in registers,
// the Q / dO tile in LDS in both orientations ([q][d] for the scores, [d][q] for the two accumulations).
constexpr int BQ = 68;  // padded row (floats): 16-byte aligned, conflict-light
__global__ __launch_bounds__(256) void attn_bwd_dkv_mfma_kernel(AttnBwdParams p) {
    __shared__ __attribute__((aligned(16))) float Qs[64 * BQ];
    __shared__ __attribute__((aligned(16))) float Gs[64 * BQ];
    __shared__ __attribute__((aligned(16))) float QsT[64 * BQ];
    __shared__ __attribute__((aligned(16))) float GsT[64 * BQ];
    __shared__ float s_lse[64], s_d[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int k0 = blockIdx.x * 64;
    const int key = k0 + 16 * wave + frow;  // the key whose K / V row this lane feeds to the matrix pipe
    const int kc = min(key, p.Tk - 1);
    f32x4 kx[4], vx[4];
    {
        const float* kp = p.k + b * p.k_bs + (int64_t)kc * p.k_rs + h * p.k_hs + 4 * fq;
        const float* vp = p.v + b * p.v_bs + (int64_t)kc * p.v_rs + h * p.v_hs + 4 * fq;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kx[ks] = *reinterpret_cast<const f32x4*>(kp + 16 * ks);
            vx[ks] = *reinterpret_cast<const f32x4*>(vp + 16 * ks);
        }
    }
    f32x4 dV[4], dK[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) dV[j] = dK[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int off = p.Tk - p.Tq;  // causal: key j is seen by queries i >= j - off
    const int q_begin = p.causal ? max(0, k0 - off) : 0;
    const int srow = tid >> 2, sseg = (tid & 3) * 16;
    const float* qb = p.q + b * p.q_bs + h * p.q_hs;
    const float* gb = p.d_o + b * p.o_bs + h * p.o_hs;
    for (int q0 = (q_begin / 64) * 64; q0 < p.Tq; q0 += 64) {
        __syncthreads();
        {
            const int qi = q0 + srow;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f}, g = {0.f, 0.f, 0.f, 0.f};
                if (qi < p.Tq) {
                    a = *reinterpret_cast<const f32x4*>(qb + (int64_t)qi * p.q_rs + sseg + 4 * c);
                    g = *reinterpret_cast<const f32x4*>(gb + (int64_t)qi * p.o_rs + sseg + 4 * c);
                }
                *reinterpret_cast<f32x4*>(&Qs[srow * BQ + sseg + 4 * c]) = a;
                *reinterpret_cast<f32x4*>(&Gs[srow * BQ + sseg + 4 * c]) = g;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    QsT[(sseg + 4 * c + e) * BQ + srow] = a[e];
                    GsT[(sseg + 4 * c + e) * BQ + srow] = g[e];
                }
            }
            if (tid < 64) {
                const int qq = q0 + tid;
                s_lse[tid] = qq < p.Tq ? p.lse[((int64_t)b * p.H + h) * p.Tq + qq] : 0.f;
                s_d[tid] = qq < p.Tq ? p.dvec[((int64_t)b * p.H + h) * p.Tq + qq] : 0.f;
            }
        }
        __syncthreads();
        // S[q][key] and dP[q][key] for the wave's 16 keys and the tile's 64 queries: acc[i][e] <-> q = 16i + 4fq + e
        f32x4 S[4], dP[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            S[i] = dP[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f32x4 qf = *reinterpret_cast<const f32x4*>(&Qs[(16 * i + frow) * BQ + 16 * ks + 4 * fq]);
                const f32x4 gf = *reinterpret_cast<const f32x4*>(&Gs[(16 * i + frow) * BQ + 16 * ks + 4 * fq]);
                Mma<float>::run(qf, kx[ks], S[i]);
                Mma<float>::run(gf, vx[ks], dP[i]);
            }
        }
        // P = exp(S - lse), dS = P (dP - D); masked where the query or the key does not exist / is not visible
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ql = 16 * i + 4 * fq + e, qi = q0 + ql;
                const bool vis = qi < p.Tq && key < p.Tk && (!p.causal || key <= qi + off);
                const float pr = vis ? __expf(S[i][e] - s_lse[ql]) : 0.f;
                S[i][e] = pr;
                dP[i][e] = pr * (dP[i][e] - s_d[ql]);
            }
        // dV[key][d] += sum_q P[q][key] dO[q][d];  dK[key][d] += sum_q dS[q][key] Q[q][d]   (k-step i = queries 16i .. 16i+15)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 gt = *reinterpret_cast<const f32x4*>(&GsT[(16 * j + frow) * BQ + 16 * i + 4 * fq]);
                const f32x4 qt = *reinterpret_cast<const f32x4*>(&QsT[(16 * j + frow) * BQ + 16 * i + 4 * fq]);
                Mma<float>::run(S[i], gt, dV[j]);
                Mma<float>::run(dP[i], qt, dK[j]);
            }
    }
    // acc[j][r] <-> key = k0 + 16 wave + 4 fq + r, d = 16 j + frow
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int kj = k0 + 16 * wave + 4 * fq + r;
        if (kj < p.Tk) {
            float* dkp = p.dk + b * p.k_bs + (int64_t)kj * p.k_rs + h * p.k_hs + frow;
            float* dvp = p.dv + b * p.v_bs + (int64_t)kj * p.v_rs + h * p.v_hs + frow;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dkp[16 * j] = dK[j][r] * p.qk_scale;
                dvp[16 * j] = dV[j][r];
            }
        }
    }
}

// ---- kernel 1 on the f32 MFMA: 64 queries per workgroup, wave w owns queries 16w .. 16w+15 (their q / dO rows live in
// registers as MFMA fragments) and walks the keys in tiles of 64 staged in LDS as [key][d] (scores, dP) and [d][key] (dQ).
// Mirror image of the key-major kernel: here the KEYS sit on the row side of the score product, S[key][q], so a lane holds
// 4 consecutive keys of one query -- the fragment of dQ[q][d] += sum_key dS[q][key] K[key][d].
__global__ __launch_bounds__(256) void attn_bwd_dq_mfma_kernel(AttnBwdParams p) {
    __shared__ __attribute__((aligned(16))) float Ks[64 * BQ];
    __shared__ __attribute__((aligned(16))) float Vs[64 * BQ];
    __shared__ __attribute__((aligned(16))) float KsT[64 * BQ];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * 64;
    const int qi = q0 + 16 * wave + frow;
    const int qc = min(qi, p.Tq - 1);
    f32x4 qx[4], gx[4];
    float dsum = 0.f;
    {
        const float* qp = p.q + b * p.q_bs + (int64_t)qc * p.q_rs + h * p.q_hs + 4 * fq;
        const float* gp = p.d_o + b * p.o_bs + (int64_t)qc * p.o_rs + h * p.o_hs + 4 * fq;
        const float* op = p.o + b * p.o_bs + (int64_t)qc * p.o_rs + h * p.o_hs + 4 * fq;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qx[ks] = *reinterpret_cast<const f32x4*>(qp + 16 * ks);
            gx[ks] = *reinterpret_cast<const f32x4*>(gp + 16 * ks);
            const f32x4 oo = *reinterpret_cast<const f32x4*>(op + 16 * ks);
#pragma unroll
            for (int e = 0; e < 4; ++e) dsum = fmaf(gx[ks][e], oo[e], dsum);
        }
        dsum += __shfl_xor(dsum, 16, 64);  // the four fq lanes of a query hold 16 of its 64 dims each
        dsum += __shfl_xor(dsum, 32, 64);
    }
    const float lse = p.lse[((int64_t)b * p.H + h) * p.Tq + qc];
    if (fq == 0 && qi < p.Tq) p.dvec[((int64_t)b * p.H + h) * p.Tq + qi] = dsum;
    f32x4 dQ[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) dQ[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int off = p.Tk - p.Tq;
    const int q_last = min(q0 + 63, p.Tq - 1);
    const int k_end = p.causal ? min(p.Tk, q_last + off + 1) : p.Tk;
    const int srow = tid >> 2, sseg = (tid & 3) * 16;
    const float* kb = p.k + b * p.k_bs + h * p.k_hs;
    const float* vb = p.v + b * p.v_bs + h * p.v_hs;
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
                *reinterpret_cast<f32x4*>(&Ks[srow * BQ + sseg + 4 * c]) = a;
                *reinterpret_cast<f32x4*>(&Vs[srow * BQ + sseg + 4 * c]) = bb;
#pragma unroll
                for (int e = 0; e < 4; ++e) KsT[(sseg + 4 * c + e) * BQ + srow] = a[e];
            }
        }
        __syncthreads();
        f32x4 S[4], dP[4];  // acc[i][e] <-> key = k0 + 16 i + 4 fq + e, query = the lane's
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            S[i] = dP[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(&Ks[(16 * i + frow) * BQ + 16 * ks + 4 * fq]);
                const f32x4 vf = *reinterpret_cast<const f32x4*>(&Vs[(16 * i + frow) * BQ + 16 * ks + 4 * fq]);
                Mma<float>::run(kf, qx[ks], S[i]);
                Mma<float>::run(vf, gx[ks], dP[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = k0 + 16 * i + 4 * fq + e;
                const bool vis = qi < p.Tq && key < p.Tk && (!p.causal || key <= qi + off);
                dP[i][e] = vis ? __expf(S[i][e] - lse) * (dP[i][e] - dsum) : 0.f;
            }
        // dQ[q][d] += sum_key dS[q][key] K[key][d]   (k-step i = keys 16i .. 16i+15 of the tile)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 kt = *reinterpret_cast<const f32x4*>(&KsT[(16 * j + frow) * BQ + 16 * i + 4 * fq]);
                Mma<float>::run(dP[i], kt, dQ[j]);
            }
    }
    // acc[j][r] <-> query = q0 + 16 wave + 4 fq + r, d = 16 j + frow
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qq = q0 + 16 * wave + 4 * fq + r;
        if (qq < p.Tq) {
            float* dqp = p.dq + b * p.q_bs + (int64_t)qq * p.q_rs + h * p.q_hs + frow;
#pragma unroll
            for (int j = 0; j < 4; ++j) dqp[16 * j] = dQ[j][r] * p.qk_scale;
        }
    }
}

}  // namespace

extern "C" int wipa_attention_bwd(const wipa_attn_desc* d, const float* out, const float* d_out, const float* lse, float* dq,
                                  float* dk, float* dv, float* dvec, float qk_scale, wipa_stream_t stream) {
    WIPA_REQUIRE(d && d->q && d->k && d->v && out && d_out && lse && dq && dk && dv && dvec, "wipa_attention_bwd: null pointer");
    WIPA_REQUIRE(d->dtype == WIPA_F32, "wipa_attention_bwd: f32 only");
    WIPA_REQUIRE(d->B > 0 && d->H > 0 && d->Tq > 0 && d->Tk > 0, "wipa_attention_bwd: bad shape");
    WIPA_REQUIRE(d->q_rs % 4 == 0 && d->k_rs % 4 == 0 && d->v_rs % 4 == 0 && d->o_rs % 4 == 0 && d->q_hs % 4 == 0 &&
                     d->k_hs % 4 == 0 && d->v_hs % 4 == 0 && d->o_hs % 4 == 0,
                 "wipa_attention_bwd: strides must keep 16-byte alignment");
    AttnBwdParams p;
    p.q = (const float*)d->q; p.k = (const float*)d->k; p.v = (const float*)d->v;
    p.o = out; p.d_o = d_out; p.lse = lse; p.dq = dq; p.dk = dk; p.dv = dv; p.dvec = dvec;
    p.q_bs = d->q_bs; p.q_rs = d->q_rs; p.q_hs = d->q_hs;
    p.k_bs = d->k_bs; p.k_rs = d->k_rs; p.k_hs = d->k_hs;
    p.v_bs = d->v_bs; p.v_rs = d->v_rs; p.v_hs = d->v_hs;
    p.o_bs = d->o_bs; p.o_rs = d->o_rs; p.o_hs = d->o_hs;
    p.H = d->H; p.Tq = d->Tq; p.Tk = d->Tk; p.causal = d->causal;
    p.qk_scale = qk_scale;
    hipStream_t s = (hipStream_t)stream;
    // both halves on the f32 MFMA (64 queries / 64 keys per workgroup); WIPA_ATTN_BWD=valu keeps the VALU kernels for A/B runs
    static const bool valu = [] { const char* e = getenv("WIPA_ATTN_BWD"); return e && !strcmp(e, "valu"); }();
    if (valu) {
        hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((d->Tq + 15) / 16, d->H, d->B), dim3(256), 0, s, p);
        hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3((d->Tk + 15) / 16, d->H, d->B), dim3(256), 0, s, p);
    } else {
        hipLaunchKernelGGL(attn_bwd_dq_mfma_kernel, dim3((d->Tq + 63) / 64, d->H, d->B), dim3(256), 0, s, p);
        hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel, dim3((d->Tk + 63) / 64, d->H, d->B), dim3(256), 0, s, p);
    }
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}
