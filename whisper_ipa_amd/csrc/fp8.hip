// fp8 ACTIVATIONS for the encoder of BASELINE.json configs[4] ("whisper-large-v3 fp8-weight inference (CDNA4 fp8 MFMA)"):
// row quantisers that feed the fp8 x fp8 tile GEMM (gemm.hip gemm_fp8_256_kernel, v_mfma_scale_f32_16x16x128_f8f6f4).
//
//   wipa_layernorm_fp8   nn.LayerNorm rows (attn_ln / mlp_ln of mlx_whisper's ResidualAttentionBlock, behind
//                        scripts/transcribe_single.py:54) -> OCP e4m3fn codes + ONE power-of-two scale per row
//   wipa_rowquant_fp8    the same quantisation of an existing activation matrix (the GELU output that feeds mlp2)
//
// scale[r] = the smallest power of two with max|row| / scale <= 448 (the largest finite e4m3fn value), exactly the rule of
// Whisper.quantize_weights for weight rows: a product code_a * code_w * scale_a * scale_w is exact in f32, so the GEMM's only
// rounding is the e4m3 rounding of its operands (3 mantissa bits, round to nearest even by v_cvt_pk_fp8_f32).
#include "wipa_common.h"

namespace {

constexpr float FP8_MAX = 448.0f;

template <typename TI>
__device__ __forceinline__ f32x4 ld4(const TI* p);
template <>
__device__ __forceinline__ f32x4 ld4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <>
__device__ __forceinline__ f32x4 ld4<__bf16>(const __bf16* p) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}

// smallest power of two s with amax / s <= 448 (2^-100 for an all-zero row)
__device__ __forceinline__ float pow2_scale(float amax) {
    if (!(amax > 0.f)) return 7.888609052210118e-31f;  // 2^-100
    int k;
    const float m = frexpf(amax * (1.0f / FP8_MAX), &k);  // amax / 448 = m * 2^k, m in [0.5, 1)
    return ldexpf(1.0f, m == 0.5f ? k - 1 : k);
}

__device__ __forceinline__ uint32_t pack4_fp8(const f32x4& v, float inv) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, w, true);
    return (uint32_t)w;
}

// one wave per row, lane owns columns lane*4 + 256*i (the arithmetic of layernorm_kernel up to the rounding)
template <int NV>
__global__ __launch_bounds__(256) void layernorm_fp8_kernel(const float* __restrict__ x, int64_t ldx, uint8_t* __restrict__ y, int64_t ldy,
                                                            float* __restrict__ y_scale, const float* __restrict__ w,
                                                            const float* __restrict__ b, int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (int64_t)row * ldx;
    f32x4 v[NV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < D) {
            v[i] = *reinterpret_cast<const f32x4*>(xr + c);
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
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < D) {
            const f32x4 ww = *reinterpret_cast<const f32x4*>(w + c);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(b + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[i][e] = (v[i][e] - mean) * rstd * ww[e] + bb[e];
                amax = fmaxf(amax, fabsf(v[i][e]));
            }
        }
    }
    const float scale = pow2_scale(wave_reduce_max(amax));
    const float inv = 1.0f / scale;  // a power of two: exact
    uint8_t* yr = y + (int64_t)row * ldy;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < D) *reinterpret_cast<uint32_t*>(yr + c) = pack4_fp8(v[i], inv);
    }
    if (lane == 0) y_scale[row] = scale;
}

// one wave per row; a lane owns 8 consecutive columns per pass of 512.  NP > 0: the row (D <= 512 NP) is held in registers
// between the maximum and the conversion -- one read of the matrix instead of two (the GELU output of whisper-large-v3 is
// 1.97 GB per layer at 128 clips); NP = 0: any D, two reads.
template <typename T, int NP>
__global__ __launch_bounds__(256) void rowquant_fp8_kernel(const T* __restrict__ x, int64_t ldx, uint8_t* __restrict__ y, int64_t ldy,
                                                           float* __restrict__ y_scale, int rows, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* xr = x + (int64_t)row * ldx;
    uint8_t* yr = y + (int64_t)row * ldy;
    float amax = 0.f;
    if constexpr (NP > 0) {
        f32x4 a[NP], bq[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int c = lane * 8 + 512 * i;
            a[i] = bq[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < D) {
                a[i] = ld4<T>(xr + c);
                bq[i] = ld4<T>(xr + c + 4);
            }
        }
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(a[i][e]), fabsf(bq[i][e])));
        const float scale = pow2_scale(wave_reduce_max(amax));
        const float inv = 1.0f / scale;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int c = lane * 8 + 512 * i;
            if (c < D) {
                uint2 o;
                o.x = pack4_fp8(a[i], inv);
                o.y = pack4_fp8(bq[i], inv);
                *reinterpret_cast<uint2*>(yr + c) = o;
            }
        }
        if (lane == 0) y_scale[row] = scale;
    } else {
        for (int c = lane * 8; c < D; c += 512) {
            const f32x4 a = ld4<T>(xr + c), bq = ld4<T>(xr + c + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(a[e]), fabsf(bq[e])));
        }
        const float scale = pow2_scale(wave_reduce_max(amax));
        const float inv = 1.0f / scale;
        for (int c = lane * 8; c < D; c += 512) {
            const f32x4 a = ld4<T>(xr + c), bq = ld4<T>(xr + c + 4);
            uint2 o;
            o.x = pack4_fp8(a, inv);
            o.y = pack4_fp8(bq, inv);
            *reinterpret_cast<uint2*>(yr + c) = o;
        }
        if (lane == 0) y_scale[row] = scale;
    }
}

}  // namespace

extern "C" int wipa_layernorm_fp8(const float* x, int64_t ldx, void* y, int64_t ldy, float* y_scale, const float* w, const float* b,
                                  int rows, int D, float eps, wipa_stream_t stream) {
    WIPA_REQUIRE(x && y && y_scale && w && b, "wipa_layernorm_fp8: null pointer");
    WIPA_REQUIRE(D > 0 && D % 4 == 0 && D <= 2048 && ldx % 4 == 0 && ldy % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 4) == 0,
                 "wipa_layernorm_fp8: D=%d must be a multiple of 4 and <= 2048, rows 16-byte aligned", D);
    if (rows <= 0) return WIPA_OK;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((rows + 3) / 4), block(256);
    const int nv = (D + 255) / 256;
#define LNF8(NV) hipLaunchKernelGGL((layernorm_fp8_kernel<NV>), grid, block, 0, s, x, ldx, (uint8_t*)y, ldy, y_scale, w, b, rows, D, eps)
    if (nv <= 1) LNF8(1);
    else if (nv <= 2) LNF8(2);
    else if (nv <= 3) LNF8(3);
    else if (nv <= 4) LNF8(4);
    else if (nv <= 5) LNF8(5);
    else LNF8(8);
#undef LNF8
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

extern "C" int wipa_rowquant_fp8(const void* x, int x_dtype, int64_t ldx, void* y, int64_t ldy, float* y_scale, int rows, int D,
                                 wipa_stream_t stream) {
    WIPA_REQUIRE(x && y && y_scale, "wipa_rowquant_fp8: null pointer");
    WIPA_REQUIRE(x_dtype == WIPA_BF16 || x_dtype == WIPA_F32, "wipa_rowquant_fp8: x_dtype %d", x_dtype);
    WIPA_REQUIRE(D > 0 && D % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 8) == 0,
                 "wipa_rowquant_fp8: D=%d must be a multiple of 8, rows 16-byte aligned", D);
    if (rows <= 0) return WIPA_OK;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((rows + 3) / 4), block(256);
    const int np = (D + 511) / 512;
#define RQ(T, NP) hipLaunchKernelGGL((rowquant_fp8_kernel<T, NP>), grid, block, 0, s, (const T*)x, ldx, (uint8_t*)y, ldy, y_scale, rows, D)
    if (x_dtype == WIPA_BF16) {
        if (np <= 2) RQ(__bf16, 2);
        else if (np <= 4) RQ(__bf16, 4);
        else if (np <= 6) RQ(__bf16, 6);
        else if (np <= 8) RQ(__bf16, 8);
        else if (np <= 10) RQ(__bf16, 10);
        else RQ(__bf16, 0);
    } else {
        if (np <= 4) RQ(float, 4);
        else if (np <= 10) RQ(float, 10);
        else RQ(float, 0);
    }
#undef RQ
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}
