// Shared device/host helpers for libwipa (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "wipa.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define WIPA_WAVE 64

// ---- error plumbing (never throw across the C ABI) -------------------------
void wipa_set_error(const char* fmt, ...);
// one-time kernel attribute setup (dynamic LDS limits); must first run OUTSIDE a stream capture
int wipa_decode_fused_init();
int wipa_gemm_init();

#define WIPA_CHECK_HIP(expr)                                                        \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) {                                                     \
            wipa_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return WIPA_ERR_HIP;                                                    \
        }                                                                           \
    } while (0)

#define WIPA_REQUIRE(cond, ...)                \
    do {                                       \
        if (!(cond)) {                         \
            wipa_set_error(__VA_ARGS__);       \
            return WIPA_ERR_ARG;               \
        }                                      \
    } while (0)

#define WIPA_LAUNCH_CHECK()                                                          \
    do {                                                                             \
        hipError_t _e = hipGetLastError();                                           \
        if (_e != hipSuccess) {                                                      \
            wipa_set_error("%s:%d launch -> %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
            return WIPA_ERR_HIP;                                                     \
        }                                                                            \
    } while (0)

static inline size_t wipa_dtype_size(int dt) { return dt == WIPA_BF16 ? 2 : 4; }

// ---- device helpers --------------------------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ float wipa_bf16_to_f32(unsigned short h) {
    return __uint_as_float(((unsigned int)h) << 16);
}

template <typename T>
__device__ __forceinline__ float to_f32(T v);
template <>
__device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ float to_f32<__bf16>(__bf16 v) { return (float)v; }

template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ __bf16 from_f32<__bf16>(float v) { return (__bf16)v; }

// 16-byte vector of T, unpacked to floats.  EPL = elements per 16-byte lane load.
template <typename T>
struct Vec16;
template <>
struct Vec16<float> {
    static constexpr int EPL = 4;
    f32x4 v;
    __device__ __forceinline__ float get(int i) const { return v[i]; }
};
template <>
struct Vec16<__bf16> {
    static constexpr int EPL = 8;
    bf16x8 v;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
};

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// MFMA 16x16 wrappers shared by the GEMM and the fused decode kernels.  Operand roles are swapped w.r.t. the textbook (W
// feeds the "A" side): acc[e] of lane (frow = lane & 15, fq = lane >> 4) is C[m = frow][n = 4*fq + e].  A fragment is the
// 16 bytes at (row frow, 16-byte chunk fq) of a 64-byte K step, for both dtypes.
template <typename T>
struct Mma;
template <>
struct Mma<__bf16> {
    typedef bf16x8 Frag;
    static __device__ __forceinline__ void run(const Frag& w, const Frag& x, f32x4& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, acc, 0, 0, 0);
    }
};
template <>
struct Mma<float> {
    typedef f32x4 Frag;
    static __device__ __forceinline__ void run(const Frag& w, const Frag& x, f32x4& acc) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e], x[e], acc, 0, 0, 0);
    }
};

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. f32 round-off level): one v_rcp, one
// v_exp and a degree-5 Horner instead of libm's branchy erff -- the GELU epilogue of the MLP
// GEMM is VALU-bound otherwise (K = 768 gives the MFMA only ~1.5 SIMD-cycles per output).
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    const float y = fmaf(-p * t, e, 1.0f);
    return copysignf(y, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }

// Two GELUs at once for the GEMM epilogues, written so that the arithmetic maps onto the packed-f32 VALU
// (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth per issue slot).  Same Abramowitz-Stegun 7.1.26 series as
// erf_fast, folded into the normal CDF:  Phi(x) = 1 - h (x >= 0), h (x < 0),  h = 0.5 P(t) exp(-x^2/2),
// t = 1 / (1 + 0.3275911 |x| / sqrt(2)),  and  gelu(x) = x Phi(x) = max(x, 0) - |x| h.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
    const f32x2 ax = __builtin_elementwise_abs(x);
    const f32x2 d = __builtin_elementwise_fma(f32x2{0.2316418882f, 0.2316418882f}, ax, f32x2{1.f, 1.f});
    const f32x2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    f32x2 p = __builtin_elementwise_fma(f32x2{0.5307027145f, 0.5307027145f}, t, f32x2{-0.7265760135f, -0.7265760135f});
    p = __builtin_elementwise_fma(p, t, f32x2{0.7107068705f, 0.7107068705f});
    p = __builtin_elementwise_fma(p, t, f32x2{-0.142248368f, -0.142248368f});
    p = __builtin_elementwise_fma(p, t, f32x2{0.127414796f, 0.127414796f});
    p = p * t;
    const f32x2 s = x * 0.8493218003f;  // sqrt(log2(e) / 2): exp(-x^2/2) = exp2(-s^2)
    const f32x2 s2 = s * s;
    const f32x2 e = {__builtin_amdgcn_exp2f(-s2.x), __builtin_amdgcn_exp2f(-s2.y)};
    return __builtin_elementwise_fma(-ax, p * e, __builtin_elementwise_max(x, f32x2{0.f, 0.f}));
}
// f32 operands as two bf16 terms: x = hi + lo + O(2^-17 |x|).  Eight consecutive-in-k floats of a lane (two 16-byte LDS
// chunks) become the hi and lo fragments of one bf16 MFMA K-step; a product a*w is then taken as
// a_hi*w_lo + a_lo*w_hi + a_hi*w_hi (the dropped a_lo*w_lo and the representation residuals are ~2^-16 relative), each
// term accumulated in f32 by the matrix pipe at 16x the rate of the f32 MFMA.
__device__ __forceinline__ void split_bf16x2(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h = (__bf16)x[e];
        hi[e] = h;
        lo[e] = (__bf16)(x[e] - (float)h);
    }
}


// Three bf16 terms: x = hi + mid + lo + O(2^-25 |x|) -- f32-exact for practical purposes (used where a product feeds an
// exponential: attention scores).
__device__ __forceinline__ void split_bf16x3(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h = (__bf16)x[e];
        const float r1 = x[e] - (float)h;
        const __bf16 m = (__bf16)r1;
        hi[e] = h;
        mid[e] = m;
        lo[e] = (__bf16)(r1 - (float)m);
    }
}

#endif
