// Backward / optimiser kernels of the decoder fine-tune step (f32, parity-first versions).
// Replaces what mlx's nn.value_and_grad / optim.AdamW do for scripts/train_whisper_ipa.py:266-311:
// gradients of the teacher-forced decoder + masked CE w.r.t. the decoder parameters, the
// PER-TENSOR clip g *= min(1, 1/(|g|+1e-6)) (:287-303) and mlx-style AdamW without bias
// correction (:513).  Dense contractions reuse wipa_gemm (A * W^T) on transposed operands
// produced by transpose_kernel; everything here is deterministic (no float atomics).
#include "wipa_common.h"

namespace {

// ------------------------------------------------------------------ transpose (+ zero pad)
// out[c][r] = in[r][c] for r < rows, c < cols; out columns rows..rows_pad-1 are written as zero
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ in, int64_t ld_in, T* __restrict__ out,
                                                        int64_t ld_out, int rows, int cols, int rows_pad) {
    __shared__ T tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? in[(int64_t)r * ld_in + c] : from_f32<T>(0.f);
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows_pad) out[(int64_t)c * ld_out + r] = tile[tx][i];
    }
}

// ------------------------------------------------------------------ column sums (bias grads)
// out[c] (+)= sum_r x[r][c] in a FIXED order (deterministic): grid (cols / 64, R); workgroup (bx, k) sums row chunk k of
// 64 columns (4 row lanes x 8 independent accumulators per thread, so 32 loads per thread are in flight) into
// partial[k][c]; a second launch adds the R partials in order.  R = 1 writes straight to out.  The old form (one
// workgroup per 64 columns walking ALL rows serially) took 690 us on the 48 000-row cross-attention value bias.
constexpr int CS_UNROLL = 8;
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int64_t ld, int rows, int cols,
                                                     float* __restrict__ dst, int accumulate, int rows_per_chunk) {
    __shared__ float part[4][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    float acc[CS_UNROLL];
#pragma unroll
    for (int u = 0; u < CS_UNROLL; ++u) acc[u] = 0.f;
    if (c < cols) {
        const float* xp = x + c;
        int r = r0 + g;
        for (; r + 4 * (CS_UNROLL - 1) < r1; r += 4 * CS_UNROLL) {
#pragma unroll
            for (int u = 0; u < CS_UNROLL; ++u) acc[u] += xp[(int64_t)(r + 4 * u) * ld];
        }
        for (; r < r1; r += 4) acc[0] += xp[(int64_t)r * ld];
    }
    part[g][cl] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (g == 0 && c < cols) {
        const float t = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
        float* o = dst + (int64_t)blockIdx.y * cols + c;
        *o = (accumulate && gridDim.y == 1) ? *o + t : t;
    }
}
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ partial, int R, int cols,
                                                            float* __restrict__ out, int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float t = 0.f;
    for (int k = 0; k < R; ++k) t += partial[(int64_t)k * cols + c];
    out[c] = accumulate ? out[c] + t : t;
}

// ------------------------------------------------------------------ slab sum (split-K weight gradients)
// out[i] (+)= sum_k slabs[k * stride + i], k ascending (deterministic)
__global__ __launch_bounds__(256) void sum_slabs_kernel(const float* __restrict__ slabs, int n_slabs, int64_t stride,
                                                        float* __restrict__ out, int64_t n, int accumulate) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        f32x4 t = *reinterpret_cast<const f32x4*>(slabs + i);
        for (int k = 1; k < n_slabs; ++k) t += *reinterpret_cast<const f32x4*>(slabs + k * stride + i);
        if (accumulate) t += *reinterpret_cast<const f32x4*>(out + i);
        *reinterpret_cast<f32x4*>(out + i) = t;
    } else {
        for (int64_t j = i; j < n; ++j) {
            float t = slabs[j];
            for (int k = 1; k < n_slabs; ++k) t += slabs[k * stride + j];
            out[j] = accumulate ? out[j] + t : t;
        }
    }
}

// out[i] = scale * sum_k slabs[k * stride + i] (+ residual[i]): the epilogue of a split-K GEMM whose slices could not apply it.
// residual MAY alias out (in-place accumulation of an input gradient): neither pointer is __restrict__, and every thread
// reads residual[i..i+3] before it stores out[i..i+3].
__global__ __launch_bounds__(256) void sum_slabs_ex_kernel(const float* __restrict__ slabs, int n_slabs, int64_t stride,
                                                           float* out, int64_t n, const float* residual, float scale) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        f32x4 t = *reinterpret_cast<const f32x4*>(slabs + i);
        for (int k = 1; k < n_slabs; ++k) t += *reinterpret_cast<const f32x4*>(slabs + k * stride + i);
        t *= scale;
        if (residual) t += *reinterpret_cast<const f32x4*>(residual + i);
        *reinterpret_cast<f32x4*>(out + i) = t;
    } else {
        for (int64_t j = i; j < n; ++j) {
            float t = slabs[j];
            for (int k = 1; k < n_slabs; ++k) t += slabs[k * stride + j];
            t *= scale;
            out[j] = residual ? residual[j] + t : t;
        }
    }
}

// ------------------------------------------------------------------ LayerNorm backward
// dx (+)= rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * w ; stats[r] = (mean, rstd)
constexpr int LNB_MAXV = 8;
__global__ __launch_bounds__(256) void ln_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        const float* __restrict__ w, float* __restrict__ dx,
                                                        float* __restrict__ stats, int rows, int D, float eps, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (int64_t)row * D;
    const float* gr = dy + (int64_t)row * D;
    float* dr = dx + (int64_t)row * D;
    f32x4 v[LNB_MAXV], g[LNB_MAXV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < LNB_MAXV; ++i) {
        const int c = lane * 4 + 256 * i;
        v[i] = g[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < D) {
            v[i] = *reinterpret_cast<const f32x4*>(xr + c);
            g[i] = *reinterpret_cast<const f32x4*>(gr + c) * *reinterpret_cast<const f32x4*>(w + c);
            sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    const float mean = wave_reduce_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < LNB_MAXV; ++i)
        if (lane * 4 + 256 * i < D)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[i][e] - mean;
                sq += d * d;
            }
    const float rstd = rsqrtf(wave_reduce_sum(sq) / (float)D + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < LNB_MAXV; ++i)
        if (lane * 4 + 256 * i < D)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (v[i][e] - mean) * rstd;
                sg += g[i][e];
                sgx += g[i][e] * xh;
            }
    sg = wave_reduce_sum(sg) / (float)D;
    sgx = wave_reduce_sum(sgx) / (float)D;
#pragma unroll
    for (int i = 0; i < LNB_MAXV; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < D) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (v[i][e] - mean) * rstd;
                o[e] = rstd * (g[i][e] - sg - xh * sgx);
            }
            if (accumulate) o += *reinterpret_cast<const f32x4*>(dr + c);
            *reinterpret_cast<f32x4*>(dr + c) = o;
        }
    }
    if (lane == 0) {
        stats[2 * row] = mean;
        stats[2 * row + 1] = rstd;
    }
}

// dw[c] = sum_r dy*xhat, db[c] = sum_r dy  (one workgroup per 64 columns, fixed order)
// Column sums over the rows, in two deterministic stages: workgroup (column tile, row chunk) reduces its rows [r0, r1) and
// either writes dw / db directly (one chunk) or a partial pair into `part` [n_chunks][2][D]; ln_bwd_params_finish_kernel adds
// the chunks in order.  (One workgroup per column tile walking ALL rows took 147 us for 2048 x 768: 12 workgroups on a
// 256-CU chip, 5.4 ms of a fine-tune step.)
__global__ __launch_bounds__(256) void ln_bwd_params_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ stats, int rows, int D, int rows_per_chunk,
                                                            float* __restrict__ dw, float* __restrict__ db, float* __restrict__ part) {
    __shared__ float pw[4][64], pb[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    float sw = 0.f, sb = 0.f;
    if (c < D)
        for (int r = r0 + g; r < r1; r += 4) {
            const float d = dy[(int64_t)r * D + c];
            sw += d * (x[(int64_t)r * D + c] - stats[2 * r]) * stats[2 * r + 1];
            sb += d;
        }
    pw[g][threadIdx.x & 63] = sw;
    pb[g][threadIdx.x & 63] = sb;
    __syncthreads();
    if (g == 0 && c < D) {
        const float tw = (pw[0][threadIdx.x] + pw[1][threadIdx.x]) + (pw[2][threadIdx.x] + pw[3][threadIdx.x]);
        const float tb = (pb[0][threadIdx.x] + pb[1][threadIdx.x]) + (pb[2][threadIdx.x] + pb[3][threadIdx.x]);
        if (part) {
            part[((int64_t)blockIdx.y * 2) * D + c] = tw;
            part[((int64_t)blockIdx.y * 2 + 1) * D + c] = tb;
        } else {
            dw[c] = tw;
            db[c] = tb;
        }
    }
}
__global__ __launch_bounds__(256) void ln_bwd_params_finish_kernel(const float* __restrict__ part, int n_chunks, int D,
                                                                   float* __restrict__ dw, float* __restrict__ db) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= D) return;
    float tw = 0.f, tb = 0.f;
    for (int k = 0; k < n_chunks; ++k) {  // fixed order
        tw += part[((int64_t)k * 2) * D + c];
        tb += part[((int64_t)k * 2 + 1) * D + c];
    }
    dw[c] = tw;
    db[c] = tb;
}

// ------------------------------------------------------------------ GELU forward / backward (exact erf form)
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* __restrict__ z, float* __restrict__ u, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const f32x4 a = *reinterpret_cast<const f32x4*>(z + i);
    *reinterpret_cast<f32x4*>(u + i) = f32x4{gelu_erf(a[0]), gelu_erf(a[1]), gelu_erf(a[2]), gelu_erf(a[3])};
}
__device__ __forceinline__ float gelu_grad(float z) {
    const float cdf = 0.5f * (1.0f + erf_fast(z * 0.70710678118654752440f));
    const float pdf = 0.3989422804014327f * __builtin_amdgcn_exp2f(-0.72134752044448170f * z * z);  // exp(-z^2/2)/sqrt(2 pi)
    return cdf + z * pdf;
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ z, const float* __restrict__ du,
                                                       float* __restrict__ dz, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const f32x4 a = *reinterpret_cast<const f32x4*>(z + i);
    const f32x4 d = *reinterpret_cast<const f32x4*>(du + i);
    *reinterpret_cast<f32x4*>(dz + i) =
        f32x4{d[0] * gelu_grad(a[0]), d[1] * gelu_grad(a[1]), d[2] * gelu_grad(a[2]), d[3] * gelu_grad(a[3])};
}

// ------------------------------------------------------------------ masked CE backward (in place over the logits)
// logits[r][v] <- mask_r * (softmax_r[v] - [v == tgt_r]) / max(count, 1)
constexpr int CEB_THREADS = 512;
__global__ __launch_bounds__(CEB_THREADS) void ce_bwd_kernel(float* __restrict__ logits, int64_t ldl,
                                                             const int32_t* __restrict__ tokens, int64_t ld_tok, int T, int V,
                                                             const float* __restrict__ row_mask,
                                                             const float* __restrict__ count) {
    __shared__ float s_red[CEB_THREADS / 64];
    const int r = blockIdx.x;
    const int b = r / T, t = r - b * T;
    const int tgt = tokens[(int64_t)b * ld_tok + t + 1];
    float* row = logits + (int64_t)r * ldl;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float keep = row_mask[r];
    if (keep == 0.f) {
        for (int i = tid; i < V; i += CEB_THREADS) row[i] = 0.f;
        return;
    }
    float mx = -INFINITY;
    for (int i = tid; i < V; i += CEB_THREADS) mx = fmaxf(mx, row[i]);
    mx = wave_reduce_max(mx);
    if (lane == 0) s_red[wave] = mx;
    __syncthreads();
    mx = s_red[0];
#pragma unroll
    for (int w = 1; w < CEB_THREADS / 64; ++w) mx = fmaxf(mx, s_red[w]);
    __syncthreads();
    float se = 0.f;
    for (int i = tid; i < V; i += CEB_THREADS) se += __expf(row[i] - mx);
    se = wave_reduce_sum(se);
    if (lane == 0) s_red[wave] = se;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < CEB_THREADS / 64; ++w) tot += s_red[w];
    const float scale = 1.0f / fmaxf(count[0], 1.0f);
    const float inv = scale / tot;
    for (int i = tid; i < V; i += CEB_THREADS) row[i] = __expf(row[i] - mx) * inv - (i == tgt ? scale : 0.f);
}

// ------------------------------------------------------------------ embedding backward
// dE[tok[r]][c] += dx[r][c] : each thread owns a column and walks the rows in order (deterministic)
__global__ __launch_bounds__(64) void embed_bwd_tok_kernel(const int32_t* __restrict__ tokens, int rows,
                                                           const float* __restrict__ dx, float* __restrict__ dE, int D) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= D) return;
    for (int r = 0; r < rows; ++r) dE[(int64_t)tokens[r] * D + c] += dx[(int64_t)r * D + c];
}
// dpos[t][c] = sum_b dx[b*T + t][c]
__global__ __launch_bounds__(256) void embed_bwd_pos_kernel(const float* __restrict__ dx, int B, int T, int D,
                                                            float* __restrict__ dpos) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)T * D) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dx[(int64_t)b * T * D + i];
    dpos[i] = s;
}

// ------------------------------------------------------------------ per-tensor clip + AdamW (mlx defaults)
// chunk table: every 4096-element chunk belongs to one tensor (segment)
constexpr int OPT_CHUNK = 4096;
__global__ __launch_bounds__(256) void sumsq_chunks_kernel(const float* __restrict__ g, const int64_t* __restrict__ chunk_off,
                                                           const int32_t* __restrict__ chunk_len, float* __restrict__ partial) {
    __shared__ float s_red[4];
    const int ch = blockIdx.x;
    const float* p = g + chunk_off[ch];
    const int n = chunk_len[ch];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += p[i] * p[i];
    s = wave_reduce_sum(s);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[ch] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}
// one workgroup per segment (tensor): thread i adds chunks i, i + 256, ... in order, then a fixed tree over the 256 partial
// sums -> clip coefficient.  Deterministic; the token embedding alone has 9 724 chunks, which one thread per segment walked
// in 0.7 ms.
// seg_clip (may be null = every segment): 0 marks a tensor the reference's clip_grad_dict never reaches (it walks dicts only,
// decoder.blocks is a list: train_whisper_ipa.py:290-300) -- its norm is still reported, its coefficient is exactly 1.
__global__ __launch_bounds__(256) void clip_coef_kernel(const float* __restrict__ partial, const int32_t* __restrict__ seg_first_chunk,
                                                        int n_seg, float max_norm, const int32_t* __restrict__ seg_clip,
                                                        float* __restrict__ coef, float* __restrict__ norms) {
    __shared__ float s_red[4];
    const int s = blockIdx.x;
    float t = 0.f;
    for (int c = seg_first_chunk[s] + threadIdx.x; c < seg_first_chunk[s + 1]; c += 256) t += partial[c];
    t = wave_reduce_sum(t);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float nrm = sqrtf((s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
        norms[s] = nrm;
        const bool clip = !seg_clip || seg_clip[s] != 0;
        coef[s] = clip ? fminf(max_norm / (nrm + 1e-6f), 1.0f) : 1.0f;  // train_whisper_ipa.py:295-297 / :299-300
    }
}
__global__ __launch_bounds__(256) void adamw_chunks_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, const int64_t* __restrict__ chunk_off,
                                                           const int32_t* __restrict__ chunk_len,
                                                           const int32_t* __restrict__ chunk_seg, const float* __restrict__ coef,
                                                           float lr, float b1, float omb1, float b2, float omb2, float eps,
                                                           float decay) {
    const int ch = blockIdx.x;
    const int64_t off = chunk_off[ch];
    const int n = chunk_len[ch];
    const float c = coef[chunk_seg[ch]];
    for (int i = threadIdx.x; i < n; i += 256) {
        const float gi = g[off + i] * c;
        const float mi = b1 * m[off + i] + omb1 * gi;
        const float vi = b2 * v[off + i] + omb2 * gi * gi;
        g[off + i] = gi;  // the clipped gradient (train_step returns it)
        m[off + i] = mi;
        v[off + i] = vi;
        p[off + i] = p[off + i] * decay - lr * mi / (sqrtf(vi) + eps);  // mlx AdamW, no bias correction
    }
}

}  // namespace

extern "C" int wipa_transpose(const void* in, int64_t ld_in, void* out, int64_t ld_out, int rows, int cols, int rows_pad,
                              int dtype, wipa_stream_t stream) {
    WIPA_REQUIRE(in && out && rows > 0 && cols > 0 && rows_pad >= rows, "wipa_transpose: bad arguments");
    dim3 grid((cols + 63) / 64, (rows_pad + 63) / 64);
    if (dtype == WIPA_F32)
        hipLaunchKernelGGL((transpose_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)in, ld_in,
                           (float*)out, ld_out, rows, cols, rows_pad);
    else if (dtype == WIPA_BF16)
        hipLaunchKernelGGL((transpose_kernel<__bf16>), grid, dim3(256), 0, (hipStream_t)stream, (const __bf16*)in, ld_in,
                           (__bf16*)out, ld_out, rows, cols, rows_pad);
    else
        WIPA_REQUIRE(false, "wipa_transpose: bad dtype %d", dtype);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

extern "C" int wipa_colsum(const float* x, int64_t ld, int rows, int cols, float* out, int accumulate, float* workspace,
                           int64_t workspace_floats, wipa_stream_t stream) {
    WIPA_REQUIRE(x && out && rows > 0 && cols > 0, "wipa_colsum: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    // row chunks of >= 256 rows, as many as the workspace holds (at most 64)
    int R = 1;
    if (workspace && rows >= 512) {
        R = (rows + 255) / 256;
        if (R > 64) R = 64;
        if ((int64_t)R * cols > workspace_floats) R = (int)(workspace_floats / cols);
        if (R < 2) R = 1;
    }
    const int per = (rows + R - 1) / R;
    const dim3 grid((cols + 63) / 64, R);
    if (R == 1) {
        hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, s, x, ld, rows, cols, out, accumulate, per);
    } else {
        hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, s, x, ld, rows, cols, workspace, 0, per);
        hipLaunchKernelGGL(colsum_finish_kernel, dim3((cols + 255) / 256), dim3(256), 0, s, workspace, R, cols, out, accumulate);
    }
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

extern "C" int wipa_sum_slabs(const float* slabs, int n_slabs, int64_t slab_stride, float* out, int64_t n, int accumulate,
                              wipa_stream_t stream) {
    WIPA_REQUIRE(slabs && out && n_slabs >= 1 && n > 0, "wipa_sum_slabs: bad arguments");
    WIPA_REQUIRE(slab_stride % 4 == 0 && ((uintptr_t)slabs % 16) == 0 && ((uintptr_t)out % 16) == 0,
                 "wipa_sum_slabs: slabs / out must be 16-byte aligned, slab_stride a multiple of 4");
    hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, slabs, n_slabs,
                       slab_stride, out, n, accumulate);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

extern "C" int wipa_sum_slabs_ex(const float* slabs, int n_slabs, int64_t slab_stride, float* out, int64_t n, const float* residual,
                                 float scale, wipa_stream_t stream) {
    WIPA_REQUIRE(slabs && out && n_slabs >= 1 && n > 0, "wipa_sum_slabs_ex: bad arguments");
    WIPA_REQUIRE(slab_stride % 4 == 0 && ((uintptr_t)slabs % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)residual % 16) == 0,
                 "wipa_sum_slabs_ex: slabs / out / residual must be 16-byte aligned, slab_stride a multiple of 4");
    hipLaunchKernelGGL(sum_slabs_ex_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, slabs, n_slabs,
                       slab_stride, out, n, residual, scale);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

constexpr int LNB_CHUNKS = 32;
extern "C" int wipa_layernorm_bwd(const float* x, const float* dy, const float* w, float* dx, int accumulate_dx, float* dw,
                                  float* db, float* stats, int64_t stats_floats, int rows, int D, float eps, wipa_stream_t stream) {
    WIPA_REQUIRE(x && dy && w && dx && dw && db && stats, "wipa_layernorm_bwd: null pointer");
    WIPA_REQUIRE(stats_floats >= (int64_t)2 * rows, "wipa_layernorm_bwd: stats needs at least 2*rows floats");
    WIPA_REQUIRE(D % 4 == 0 && D <= LNB_MAXV * 256 && rows > 0, "wipa_layernorm_bwd: bad shape");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ln_bwd_dx_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, dy, w, dx, stats, rows, D, eps, accumulate_dx);
    // dw / db: rows cut into up to LNB_CHUNKS chunks when the scratch has room for the partials (stats_floats >= 2*rows + 2*chunks*D)
    int chunks = (rows + 63) / 64;
    if (chunks > LNB_CHUNKS) chunks = LNB_CHUNKS;
    if (stats_floats < (int64_t)2 * rows + (int64_t)2 * chunks * D) chunks = 1;
    const int per = (rows + chunks - 1) / chunks;
    float* part = chunks > 1 ? stats + (int64_t)2 * rows : nullptr;
    hipLaunchKernelGGL(ln_bwd_params_kernel, dim3((D + 63) / 64, chunks), dim3(256), 0, s, x, dy, stats, rows, D, per, dw, db, part);
    if (chunks > 1)
        hipLaunchKernelGGL(ln_bwd_params_finish_kernel, dim3((D + 255) / 256), dim3(256), 0, s, part, chunks, D, dw, db);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

extern "C" int wipa_gelu(const float* z, float* u, int64_t n, wipa_stream_t stream) {
    WIPA_REQUIRE(z && u && n > 0 && n % 4 == 0, "wipa_gelu: bad arguments");
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, u, n);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

extern "C" int wipa_gelu_bwd(const float* z, const float* du, float* dz, int64_t n, wipa_stream_t stream) {
    WIPA_REQUIRE(z && du && dz && n > 0 && n % 4 == 0, "wipa_gelu_bwd: bad arguments");
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, du, dz, n);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

extern "C" int wipa_masked_ce_bwd(float* logits, int64_t ldl, const int32_t* tokens, int64_t ld_tok, int B, int T, int V,
                                  const float* row_mask, const float* count, wipa_stream_t stream) {
    WIPA_REQUIRE(logits && tokens && row_mask && count && B * T > 0, "wipa_masked_ce_bwd: bad arguments");
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(B * T), dim3(CEB_THREADS), 0, (hipStream_t)stream, logits, ldl, tokens, ld_tok, T, V,
                       row_mask, count);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

extern "C" int wipa_embed_bwd(const int32_t* tokens_flat, const float* dx, int B, int T, int D, float* d_tok_emb, float* d_pos_emb,
                              wipa_stream_t stream) {
    WIPA_REQUIRE(tokens_flat && dx && d_tok_emb && d_pos_emb && B * T > 0, "wipa_embed_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(embed_bwd_tok_kernel, dim3((D + 63) / 64), dim3(64), 0, s, tokens_flat, B * T, dx, d_tok_emb, D);
    hipLaunchKernelGGL(embed_bwd_pos_kernel, dim3((unsigned)(((int64_t)T * D + 255) / 256)), dim3(256), 0, s, dx, B, T, D, d_pos_emb);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

extern "C" int wipa_clip_adamw(float* params, float* grads, float* m, float* v, const int64_t* chunk_off,
                               const int32_t* chunk_len, const int32_t* chunk_seg, const int32_t* seg_first_chunk, int n_chunks,
                               int n_seg, float* partial, float* coef, float* norms, const int32_t* seg_clip, double max_norm,
                               double lr, double beta1, double beta2, double eps, double weight_decay, wipa_stream_t stream) {
    WIPA_REQUIRE(params && grads && m && v && chunk_off && chunk_len && chunk_seg && seg_first_chunk && partial && coef && norms,
                 "wipa_clip_adamw: null pointer");
    WIPA_REQUIRE(n_chunks > 0 && n_seg > 0, "wipa_clip_adamw: empty");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sumsq_chunks_kernel, dim3(n_chunks), dim3(256), 0, s, grads, chunk_off, chunk_len, partial);
    hipLaunchKernelGGL(clip_coef_kernel, dim3(n_seg), dim3(256), 0, s, partial, seg_first_chunk, n_seg, (float)max_norm,
                       seg_clip, coef, norms);
    // the scalar coefficients are formed in double like the Python reference forms them, then rounded once
    hipLaunchKernelGGL(adamw_chunks_kernel, dim3(n_chunks), dim3(256), 0, s, params, grads, m, v, chunk_off, chunk_len, chunk_seg,
                       coef, (float)lr, (float)beta1, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                       (float)(1.0 - lr * weight_decay));
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}
