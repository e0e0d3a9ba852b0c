// K1: 80/128-bin log-mel front-end (replaces mlx_whisper.audio.log_mel_spectrogram,
// call sites scripts/ipa_data_loader.py:82, scripts/transcribe_single.py:45).
//
//   reflect-pad 200 | periodic Hann(400) | 400-point real DFT, hop 160 | |.|^2 of frames
//   0..2999 | mel filterbank (Slaney) | log10(max(.,1e-10)) | max(., clip_max - 8) | (.+4)/4
//
// MI355X mapping (round 3): ONE kernel from the audio samples to log10(mel) -- the spectrum never reaches HBM.
//   * The DFT of a real frame is folded in half: with the window inside the matrix, re[f] = sum_{k=1..200} C[f][k] e[k] and
//     im[f] = sum_{k=1..199} S[f][k] o[k], e[k] = x[k] + x[400-k], o[k] = x[k] - x[400-k] (e[200] = x[200]; w[0] = 0 drops
//     k = 0).  Half the multiply-adds of the 402 x 400 matrix, still an exact f32 fma chain on v_mfma_f32_16x16x4_f32.
//   * A workgroup owns 128 frames of one clip: its 20 720-sample span is reflect-padded straight into LDS, every wave builds
//     the e / o values of its 32 frames ONCE as MFMA B fragments (200 registers) and keeps them; the 13 frequency tiles'
//     matrix chunks (16 bins x 200 x cos|sin = 25.6 KB) stream through a double-buffered LDS image shared by the 4 waves.
//   * Transposed product S^T[f][t]: its accumulator layout (lane = frame, 4 registers = 4 bins) IS the B-fragment layout of
//     the next MFMA, so power = re^2 + im^2 feeds the mel projection (again f32 MFMA, mel^T[j][t] += W[j][f] P^T[f][t],
//     only the mel tiles a frequency tile overlaps) without leaving registers.
//   * log10, (x + 4) / 4 and the per-clip max (atomic) in the epilogue, written straight into the conv1 halo layout
//     [B,3002,n_mels] in the output dtype; the clamp at clip_max - 8 commutes with that map and with the rounding, so a small
//     in-place pass applies it afterwards (bit-identical to clamping first) and no f32 log-mel ever goes through HBM.
// The previous form (f32 STFT GEMM with overlapping rows -> 320 MB spectrum -> VALU mel kernel: 1.4 ms per 64 clips) stays
// selectable with WIPA_LOGMEL=gemm for A/B runs and serves n_mels > 128.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "wipa_common.h"

namespace {

constexpr int N_BINS = WIPA_N_FFT / 2 + 1;          // 201
constexpr int DFT_N = 2 * N_BINS;                   // 402 outputs: re | im
constexpr int DFT_K = 416;                          // 400 padded to a multiple of 32
constexpr int SPEC_LD = 416;                        // spectrum row stride (f32)
constexpr int ROWS_PER_CLIP = 3003;                 // 3001 frames + 2 dead rows -> clip stride 3003*160
constexpr int PAD_CLIP = ROWS_PER_CLIP * WIPA_HOP;  // 480480 samples per padded clip
constexpr int PAD_SLACK = 1024;                     // readable tail for the K padding of the last rows

// fused kernel geometry
constexpr int FT_TILES = 13;                        // frequency tiles of 16 bins (208 >= 201)
constexpr int FK = 200;                             // folded reduction length
constexpr int CHUNK_F = 2 * FK * 16;                // floats per frequency tile: [cos | sin][200 k][16 bins]
constexpr int WG_FRAMES = 128;                      // frames per workgroup (4 waves x 2 frame tiles)
constexpr int SPAN = (WG_FRAMES - 1) * WIPA_HOP + WIPA_N_FFT + 1;  // 20 721 padded samples a workgroup touches
constexpr int SPAN_LDS = SPAN + SPAN / 32 + 2;      // one skew word per 32 samples: frame starts (160 t) spread over the banks

struct TableLayout {
    size_t dft, melw, lo, hi, chunks, melt, jrange, total;
};
int mel_tiles(int n_mels) { return (n_mels + 15) / 16; }
TableLayout table_layout(int n_mels) {
    TableLayout t;
    t.dft = 0;
    t.melw = t.dft + sizeof(float) * DFT_N * DFT_K;
    t.lo = t.melw + sizeof(float) * (size_t)n_mels * N_BINS;
    t.hi = t.lo + sizeof(int) * (size_t)n_mels;
    t.chunks = ((t.hi + sizeof(int) * (size_t)n_mels + 255) / 256) * 256;  // [13]{[cos|sin][200][16], [JT][4 r][64 lanes]} f32
    t.melt = t.chunks;                                                      // (mel fragments live inside the chunks)
    t.jrange = t.chunks + sizeof(float) * (size_t)FT_TILES * (CHUNK_F + mel_tiles(n_mels) * 256);  // [13][2] int: mel tile range
    t.total = ((t.jrange + sizeof(int) * 2 * FT_TILES + 255) / 256) * 256;
    return t;
}

struct WsLayout {
    size_t padded, spec, logmel, gmax, total;
};
bool use_fused(int n_mels);
WsLayout ws_layout(int B, int n_mels) {
    WsLayout w;
    const bool fused = use_fused(n_mels);  // the fused kernel needs neither the padded copy of the audio nor the spectrum
    w.padded = 0;
    w.spec = w.padded + (fused ? 0 : (((size_t)B * PAD_CLIP + PAD_SLACK) * sizeof(float) + 255) / 256 * 256);
    w.logmel = w.spec + (fused ? 0 : ((size_t)B * ROWS_PER_CLIP * SPEC_LD * sizeof(float) + 255) / 256 * 256);
    w.gmax = w.logmel + (fused ? 0 : ((size_t)B * WIPA_N_FRAMES * n_mels * sizeof(float) + 255) / 256 * 256);
    w.total = w.gmax + (((size_t)B * sizeof(unsigned)) + 255) / 256 * 256;
    return w;
}

double hz_to_mel(double f) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
double mel_to_hz(double m) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

__global__ __launch_bounds__(256) void reflect_pad_kernel(const float* __restrict__ audio, float* __restrict__ padded,
                                                          int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int64_t b = i / PAD_CLIP;
    const int j = (int)(i - b * PAD_CLIP);
    float v = 0.f;
    if (j < WIPA_N_SAMPLES + WIPA_N_FFT) {
        int src = j - WIPA_N_FFT / 2;
        if (src < 0) src = -src;                                          // reflect, edge not repeated
        if (src >= WIPA_N_SAMPLES) src = 2 * (WIPA_N_SAMPLES - 1) - src;
        v = audio[b * (int64_t)WIPA_N_SAMPLES + src];
    }
    padded[i] = v;
}

__device__ __forceinline__ unsigned f32_key(float x) {
    const unsigned u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_f32(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// one thread per (frame, mel); a workgroup stays inside one clip -> one atomicMax per workgroup
__global__ __launch_bounds__(256) void mel_log_kernel(const float* __restrict__ spec, const float* __restrict__ melw,
                                                      const int* __restrict__ lo, const int* __restrict__ hi, int n_mels,
                                                      float* __restrict__ logmel, unsigned* __restrict__ gmax) {
    __shared__ float s_red[4];
    const int b = blockIdx.y;
    const int per_clip = WIPA_N_FRAMES * n_mels;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    float val = -INFINITY;
    if (idx < per_clip) {
        const int t = idx / n_mels, j = idx - t * n_mels;
        const float* row = spec + ((int64_t)b * ROWS_PER_CLIP + t) * SPEC_LD;
        const float* w = melw + (int64_t)j * N_BINS;
        float acc = 0.f;
        for (int f = lo[j]; f < hi[j]; ++f) {
            const float re = row[f], im = row[N_BINS + f];
            acc = fmaf(w[f], fmaf(re, re, im * im), acc);
        }
        val = log10f(fmaxf(acc, 1e-10f));
        logmel[(int64_t)b * per_clip + idx] = val;
    }
    val = wave_reduce_max(val);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = val;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
        atomicMax(gmax + b, f32_key(mx));
    }
}

template <typename TO>
__global__ __launch_bounds__(256) void mel_norm_kernel(const float* __restrict__ logmel, const unsigned* __restrict__ gmax,
                                                       int n_mels, TO* __restrict__ out) {
    // out [B, 3002, n_mels]: halo rows 0 and 3001 are zero
    const int b = blockIdx.y;
    const int per_out = (WIPA_N_FRAMES + 2) * n_mels;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= per_out) return;
    const int row = idx / n_mels;
    float v = 0.f;
    if (row >= 1 && row <= WIPA_N_FRAMES) {
        const float x = logmel[(int64_t)b * WIPA_N_FRAMES * n_mels + (idx - n_mels)];
        const float floor_v = key_f32(gmax[b]) - 8.0f;
        v = (fmaxf(x, floor_v) + 4.0f) / 4.0f;
    }
    out[(int64_t)b * per_out + idx] = from_f32<TO>(v);
}

typedef float f32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int span_addr(int s) { return s + (s >> 5); }

typedef __attribute__((address_space(3))) void* lm_lds_ptr;

__device__ __forceinline__ void store4(float* p, const f32x4_t& v) { *reinterpret_cast<f32x4_t*>(p) = v; }
__device__ __forceinline__ void store4(__bf16* p, const f32x4_t& v) {
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    bf16x4_t o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (__bf16)v[r];
    *reinterpret_cast<bf16x4_t*>(p) = o;
}

// audio [B][480000] -> out [B][3002][n_mels] rows 1..3000 = (log10(mel) + 4) / 4, NOT yet clamped, already in the output dtype and
// the conv1 halo layout, + the per-clip maximum of log10(mel).  The clamp max(x, clip_max - 8) commutes with the monotone map
// x -> (x + 4) / 4 and with the rounding to the output dtype, so mel_clamp_kernel can apply it in place afterwards and the result
// is bit-identical to clamping first: no f32 copy of the log-mel goes through HBM.
template <int JT, typename TO>
__global__ __launch_bounds__(256, 1) void logmel_fused_kernel(const float* __restrict__ audio, const float* __restrict__ chunks,
                                                              const int* __restrict__ jrange, int n_mels, TO* __restrict__ out,
                                                              unsigned* __restrict__ gmax) {
    constexpr int CHUNK = CHUNK_F + JT * 256;   // floats per frequency tile: cos | sin | mel fragments
    constexpr int NDMA = CHUNK * 4 / 1024;      // 1-KiB LDS-DMA transfers per chunk (25 + JT)
    extern __shared__ __attribute__((aligned(1024))) float lds[];
    float* cbuf = lds;                          // [2][CHUNK]
    float* span = lds + 2 * CHUNK;              // [SPAN_LDS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int b = blockIdx.y, t0 = blockIdx.x * WG_FRAMES;
    const float* clip = audio + (int64_t)b * WIPA_N_SAMPLES;
    // matrix chunks go global -> LDS by LDS-DMA (lane-linear 1-KiB pieces: the image is the table as it lies in memory), the
    // pieces dealt round-robin to the four waves; chunk 0 is requested before the span is built
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void*)chunks, 0, 0x7fffffff, 0x00020000);
    auto stage = [&](int p) {
        float* dst = cbuf + (p & 1) * CHUNK;
        for (int i = wave; i < NDMA; i += 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rC, (lm_lds_ptr)(dst + i * 256), 16, p * (CHUNK * 4) + i * 1024 + lane * 16, 0, 0, 0);
    };
    stage(0);
    // reflect-padded samples of frames t0 .. t0 + 127: padded index j = 160 t0 + s, source j - 200 mirrored at both ends
    // (edge not repeated), zero past the padded clip.  Eight loads in flight per thread.
    for (int s0 = 0; s0 < SPAN; s0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int sidx = s0 + 256 * u + tid;
            const int j = t0 * WIPA_HOP + sidx;
            int src = j - WIPA_N_FFT / 2;
            if (src < 0) src = -src;
            if (src >= WIPA_N_SAMPLES) src = 2 * (WIPA_N_SAMPLES - 1) - src;
            const bool ok = sidx < SPAN && j < WIPA_N_SAMPLES + WIPA_N_FFT;
            v[u] = clip[ok ? src : 0];
            if (!ok) v[u] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int sidx = s0 + 256 * u + tid;
            if (sidx < SPAN) span[span_addr(sidx)] = v[u];
        }
    }
    __syncthreads();
    // B fragments of this wave's two frame tiles: k-step ks, lane (frame l15, k = 4 ks + g + 1)
    float ef[2][FK / 4], of[2][FK / 4];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int base = (32 * wave + 16 * tt + l15) * WIPA_HOP;
#pragma unroll
        for (int ks = 0; ks < FK / 4; ++ks) {
            const int k = 4 * ks + g + 1;
            const float a = span[span_addr(base + k)];
            const float c = span[span_addr(base + WIPA_N_FFT - k)];
            ef[tt][ks] = k < 200 ? a + c : a;   // k = 200 pairs with itself
            of[tt][ks] = k < 200 ? a - c : 0.f;
        }
    }
    f32x4_t macc[JT][2];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) macc[jt][0] = macc[jt][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < FT_TILES; ++p) {
        // chunk p has landed for every wave; the other buffer (read in iteration p - 1) is free for chunk p + 1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (p + 1 < FT_TILES) stage(p + 1);
        const float* cc = cbuf + (p & 1) * CHUNK;  // cos rows [k][16], sin rows, mel fragments
        const int j_lo = jrange[2 * p], j_hi = jrange[2 * p + 1];
        f32x4_t re[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
        f32x4_t im[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < FK / 4; ++ks) {
            const float ac = cc[(4 * ks + g) * 16 + l15];
            const float as = cc[FK * 16 + (4 * ks + g) * 16 + l15];
            re[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac, ef[0][ks], re[0], 0, 0, 0);
            im[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(as, of[0][ks], im[0], 0, 0, 0);
            re[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac, ef[1][ks], re[1], 0, 0, 0);
            im[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(as, of[1][ks], im[1], 0, 0, 0);
        }
        // power of bins 16 p + 4 g + r for frame l15: the B fragments of the mel projection (k-step r <-> bin 4 g + r)
        f32x4_t pw[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) pw[tt][r] = fmaf(re[tt][r], re[tt][r], im[tt][r] * im[tt][r]);
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            if (jt >= j_lo && jt < j_hi) {  // uniform
                const float* wt = cc + CHUNK_F + jt * 256 + lane;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float wf = wt[r * 64];
                    macc[jt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf, pw[0][r], macc[jt][0], 0, 0, 0);
                    macc[jt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf, pw[1][r], macc[jt][1], 0, 0, 0);
                }
            }
        }
    }
    // macc[jt][tt][r] = mel energy of mel 16 jt + 4 g + r, frame t0 + 32 wave + 16 tt + l15
    float mx = -INFINITY;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int t = t0 + 32 * wave + 16 * tt + l15;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            const int j = 16 * jt + 4 * g;
            f32x4_t v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = log10f(fmaxf(macc[jt][tt][r], 1e-10f));
                if (t < WIPA_N_FRAMES && j + r < n_mels) mx = fmaxf(mx, x);
                v[r] = (x + 4.0f) / 4.0f;
            }
            TO* row = out + ((int64_t)b * (WIPA_N_FRAMES + 2) + t + 1) * n_mels;  // frame t is row t + 1 of the halo layout
            if (t < WIPA_N_FRAMES && j + 3 < n_mels)
                store4(row + j, v);
            else if (t < WIPA_N_FRAMES) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (j + r < n_mels) row[j + r] = from_f32<TO>(v[r]);
            }
        }
    }
    mx = wave_reduce_max(mx);
    if (lane == 0) atomicMax(gmax + b, f32_key(mx));
}

constexpr size_t fused_lds(int JT) { return ((size_t)((SPAN_LDS + 3) & ~3) + 2 * (size_t)(CHUNK_F + JT * 256)) * sizeof(float); }

// in place on out [B][3002][n_mels]: halo rows 0 and 3001 zero, every other value raised to the clip's floor ((max - 8) + 4) / 4;
// only values below the floor (and the halo) are written
template <typename TO>
__global__ __launch_bounds__(256) void mel_clamp_kernel(const unsigned* __restrict__ gmax, int n_mels, TO* __restrict__ out) {
    const int b = blockIdx.y;
    const int per_out = (WIPA_N_FRAMES + 2) * n_mels;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= per_out) return;
    const int row = idx / n_mels;
    TO* p = out + (int64_t)b * per_out + idx;
    if (row < 1 || row > WIPA_N_FRAMES) {
        *p = from_f32<TO>(0.f);
        return;
    }
    const float floor_y = ((key_f32(gmax[b]) - 8.0f) + 4.0f) / 4.0f;
    if ((float)*p < floor_y) *p = from_f32<TO>(floor_y);
}

// the dynamic-LDS limit of the fused kernel is raised by wipa_logmel_init (once per n_mels, outside any stream capture)
template <int JT>
int fused_attrs() {
    WIPA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&logmel_fused_kernel<JT, float>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused_lds(JT)));
    WIPA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&logmel_fused_kernel<JT, __bf16>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused_lds(JT)));
    return WIPA_OK;
}
int fused_attrs_for(int n_mels) {
    switch (mel_tiles(n_mels)) {
        case 1: return fused_attrs<1>();
        case 2: return fused_attrs<2>();
        case 3: return fused_attrs<3>();
        case 4: return fused_attrs<4>();
        case 5: return fused_attrs<5>();
        case 6: return fused_attrs<6>();
        case 7: return fused_attrs<7>();
        default: return fused_attrs<8>();
    }
}

template <int JT, typename TO>
int launch_fused(const float* audio, int batch, int n_mels, const char* tb, const TableLayout& L, TO* out, unsigned* gmax, hipStream_t s) {
    hipLaunchKernelGGL((logmel_fused_kernel<JT, TO>), dim3((WIPA_N_FRAMES + WG_FRAMES - 1) / WG_FRAMES, batch), dim3(256), fused_lds(JT), s,
                       audio, (const float*)(tb + L.chunks), (const int*)(tb + L.jrange), n_mels, out, gmax);
    const int per_out = (WIPA_N_FRAMES + 2) * n_mels;
    hipLaunchKernelGGL((mel_clamp_kernel<TO>), dim3((per_out + 255) / 256, batch), dim3(256), 0, s, gmax, n_mels, out);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

template <typename TO>
int launch_fused_jt(const float* audio, int batch, int n_mels, const char* tb, const TableLayout& L, TO* out, unsigned* gmax, hipStream_t s) {
    switch (mel_tiles(n_mels)) {
        case 1: return launch_fused<1, TO>(audio, batch, n_mels, tb, L, out, gmax, s);
        case 2: return launch_fused<2, TO>(audio, batch, n_mels, tb, L, out, gmax, s);
        case 3: return launch_fused<3, TO>(audio, batch, n_mels, tb, L, out, gmax, s);
        case 4: return launch_fused<4, TO>(audio, batch, n_mels, tb, L, out, gmax, s);
        case 5: return launch_fused<5, TO>(audio, batch, n_mels, tb, L, out, gmax, s);
        case 6: return launch_fused<6, TO>(audio, batch, n_mels, tb, L, out, gmax, s);
        case 7: return launch_fused<7, TO>(audio, batch, n_mels, tb, L, out, gmax, s);
        default: return launch_fused<8, TO>(audio, batch, n_mels, tb, L, out, gmax, s);
    }
}

bool use_fused(int n_mels) {
    static const bool gemm_form = [] { const char* e = getenv("WIPA_LOGMEL"); return e && strcmp(e, "gemm") == 0; }();
    return !gemm_form && n_mels <= 128 && n_mels % 4 == 0;
}

}  // namespace

extern "C" size_t wipa_logmel_tables_bytes(int n_mels) { return table_layout(n_mels).total; }

extern "C" int wipa_logmel_init(void* tables, int n_mels, wipa_stream_t stream) {
    WIPA_REQUIRE(tables && n_mels > 0 && n_mels <= 256, "wipa_logmel_init: bad arguments");
    if (use_fused(n_mels)) {
        const int rc = fused_attrs_for(n_mels);
        if (rc != WIPA_OK) return rc;
    }
    const TableLayout L = table_layout(n_mels);
    std::vector<char> host(L.total, 0);
    float* dft = reinterpret_cast<float*>(host.data() + L.dft);
    const double two_pi = 6.283185307179586476925286766559;
    for (int n = 0; n < N_BINS; ++n)
        for (int k = 0; k < WIPA_N_FFT; ++k) {
            const double win = 0.5 - 0.5 * std::cos(two_pi * k / WIPA_N_FFT);  // periodic Hann
            const int ph = (int)(((int64_t)n * k) % WIPA_N_FFT);               // exact angle reduction
            const double ang = two_pi * ph / WIPA_N_FFT;
            dft[(size_t)n * DFT_K + k] = (float)(win * std::cos(ang));
            dft[(size_t)(N_BINS + n) * DFT_K + k] = (float)(-win * std::sin(ang));
        }
    // Slaney mel filterbank == librosa.filters.mel(sr=16000, n_fft=400, n_mels, norm="slaney")
    float* melw = reinterpret_cast<float*>(host.data() + L.melw);
    int* lo = reinterpret_cast<int*>(host.data() + L.lo);
    int* hi = reinterpret_cast<int*>(host.data() + L.hi);
    std::vector<double> hz(n_mels + 2);
    const double m_lo = hz_to_mel(0.0), m_hi = hz_to_mel(8000.0);
    for (int i = 0; i < n_mels + 2; ++i) hz[i] = mel_to_hz(m_lo + (m_hi - m_lo) * i / (n_mels + 1));
    for (int j = 0; j < n_mels; ++j) {
        const double enorm = 2.0 / (hz[j + 2] - hz[j]);
        int first = N_BINS, last = 0;
        for (int f = 0; f < N_BINS; ++f) {
            const double freq = 8000.0 * f / (N_BINS - 1);
            const double lower = (freq - hz[j]) / (hz[j + 1] - hz[j]);
            const double upper = (hz[j + 2] - freq) / (hz[j + 2] - hz[j + 1]);
            const double w = std::fmax(0.0, std::fmin(lower, upper)) * enorm;
            melw[(size_t)j * N_BINS + f] = (float)w;
            if (w > 0.0) {
                if (f < first) first = f;
                last = f + 1;
            }
        }
        lo[j] = first < last ? first : 0;
        hi[j] = first < last ? last : 0;
    }
    // fused kernel: folded DFT chunks [13][cos|sin][kk = k - 1][bin in tile] with the window inside, bins >= 201 zero
    const int JT = mel_tiles(n_mels);
    const size_t chunk = CHUNK_F + (size_t)JT * 256;
    float* ch = reinterpret_cast<float*>(host.data() + L.chunks);
    for (int p = 0; p < FT_TILES; ++p)
        for (int kk = 0; kk < FK; ++kk)
            for (int fl = 0; fl < 16; ++fl) {
                const int f = 16 * p + fl, k = kk + 1;
                double c = 0.0, sn = 0.0;
                if (f < N_BINS) {
                    const double win = 0.5 - 0.5 * std::cos(two_pi * k / WIPA_N_FFT);
                    const int ph = (int)(((int64_t)f * k) % WIPA_N_FFT);
                    const double ang = two_pi * ph / WIPA_N_FFT;
                    c = win * std::cos(ang);
                    sn = k < 200 ? -win * std::sin(ang) : 0.0;
                }
                ch[(size_t)p * chunk + (size_t)kk * 16 + fl] = (float)c;
                ch[(size_t)p * chunk + (size_t)FK * 16 + (size_t)kk * 16 + fl] = (float)sn;
            }
    // mel weights as A fragments: [p][jt][r][lane (g, l15)] = W[16 jt + l15][16 p + 4 g + r]; mel tiles a frequency tile meets
    int* jr = reinterpret_cast<int*>(host.data() + L.jrange);
    for (int p = 0; p < FT_TILES; ++p) {
        int first = JT, last = 0;
        for (int jt = 0; jt < JT; ++jt) {
            bool any = false;
            for (int r = 0; r < 4; ++r)
                for (int ln = 0; ln < 64; ++ln) {
                    const int j = 16 * jt + (ln & 15), f = 16 * p + 4 * (ln >> 4) + r;
                    const float wgt = (j < n_mels && f < N_BINS) ? melw[(size_t)j * N_BINS + f] : 0.f;
                    ch[(size_t)p * chunk + CHUNK_F + ((size_t)jt * 4 + r) * 64 + ln] = wgt;
                    any = any || wgt != 0.f;
                }
            if (any) {
                if (jt < first) first = jt;
                last = jt + 1;
            }
        }
        jr[2 * p] = first < last ? first : 0;
        jr[2 * p + 1] = first < last ? last : 0;
    }
    WIPA_CHECK_HIP(hipMemcpyAsync(tables, host.data(), L.total, hipMemcpyHostToDevice, (hipStream_t)stream));
    WIPA_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));  // host staging buffer dies here (init-time only)
    return WIPA_OK;
}

extern "C" size_t wipa_logmel_workspace_bytes(int batch, int n_mels) { return ws_layout(batch, n_mels).total; }

extern "C" int wipa_logmel(const float* audio, int batch, int n_mels, const void* tables, void* mel, int mel_dtype,
                           void* workspace, size_t workspace_bytes, wipa_stream_t stream) {
    WIPA_REQUIRE(audio && tables && mel && workspace, "wipa_logmel: null pointer");
    WIPA_REQUIRE(batch > 0, "wipa_logmel: batch must be positive");
    const WsLayout W = ws_layout(batch, n_mels);
    WIPA_REQUIRE(workspace_bytes >= W.total, "wipa_logmel: workspace too small (%zu < %zu)", workspace_bytes, W.total);
    const TableLayout L = table_layout(n_mels);
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    float* padded = (float*)(ws + W.padded);
    float* spec = (float*)(ws + W.spec);
    float* logmel = (float*)(ws + W.logmel);
    unsigned* gmax = (unsigned*)(ws + W.gmax);
    const char* tb = (const char*)tables;

    const int per_clip = WIPA_N_FRAMES * n_mels;
    const int per_out = (WIPA_N_FRAMES + 2) * n_mels;
    if (use_fused(n_mels)) {
        WIPA_REQUIRE(mel_dtype == WIPA_F32 || mel_dtype == WIPA_BF16, "wipa_logmel: bad mel dtype %d", mel_dtype);
        WIPA_CHECK_HIP(hipMemsetAsync(gmax, 0, batch * sizeof(unsigned), s));
        const int rc = mel_dtype == WIPA_F32 ? launch_fused_jt<float>(audio, batch, n_mels, tb, L, (float*)mel, gmax, s)
                                             : launch_fused_jt<__bf16>(audio, batch, n_mels, tb, L, (__bf16*)mel, gmax, s);
        if (rc != WIPA_OK) return rc;
        const size_t esz = wipa_dtype_size(mel_dtype);
        WIPA_CHECK_HIP(hipMemsetAsync((char*)mel + (size_t)batch * per_out * esz, 0, 4 * (size_t)n_mels * esz, s));
        return WIPA_OK;
    } else {
    const int64_t total = (int64_t)batch * PAD_CLIP + PAD_SLACK;
    hipLaunchKernelGGL(reflect_pad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, audio, padded,
                       (int64_t)batch * PAD_CLIP);
    WIPA_CHECK_HIP(hipMemsetAsync(padded + (int64_t)batch * PAD_CLIP, 0, PAD_SLACK * sizeof(float), s));
    WIPA_CHECK_HIP(hipMemsetAsync(gmax, 0, batch * sizeof(unsigned), s));

    wipa_gemm_desc g = {};
    g.A = padded;
    g.lda = WIPA_HOP;
    g.W = tb + L.dft;
    g.ldw = DFT_K;
    g.C = spec;
    g.ldc = SPEC_LD;
    g.M = batch * ROWS_PER_CLIP;
    g.N = DFT_N;
    g.K = DFT_K;
    g.in_dtype = WIPA_F32;
    g.out_dtype = WIPA_F32;
    int rc = wipa_gemm(&g, stream);
    if (rc != WIPA_OK) return rc;

    hipLaunchKernelGGL(mel_log_kernel, dim3((per_clip + 255) / 256, batch), dim3(256), 0, s, spec,
                       (const float*)(tb + L.melw), (const int*)(tb + L.lo), (const int*)(tb + L.hi), n_mels, logmel, gmax);
    }
    if (mel_dtype == WIPA_F32)
        hipLaunchKernelGGL((mel_norm_kernel<float>), dim3((per_out + 255) / 256, batch), dim3(256), 0, s, logmel, gmax,
                           n_mels, (float*)mel);
    else if (mel_dtype == WIPA_BF16)
        hipLaunchKernelGGL((mel_norm_kernel<__bf16>), dim3((per_out + 255) / 256, batch), dim3(256), 0, s, logmel, gmax,
                           n_mels, (__bf16*)mel);
    else
        WIPA_REQUIRE(false, "wipa_logmel: bad mel dtype %d", mel_dtype);
    WIPA_LAUNCH_CHECK();
    // 4 zero tail rows: the conv1 GEMM's halo / K-padding over-read of the last clip
    const size_t esz = wipa_dtype_size(mel_dtype);
    WIPA_CHECK_HIP(hipMemsetAsync((char*)mel + (size_t)batch * per_out * esz, 0, 4 * (size_t)n_mels * esz, s));
    return WIPA_OK;
}
