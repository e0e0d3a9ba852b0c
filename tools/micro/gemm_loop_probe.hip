// The encoder GEMM's loop as a probe: what bounds the 384 x 256 tile's K loop, and does a second,
// independent workgroup per CU hide the epilogue?  Not a GEMM library: the staging pattern (LDS-DMA of swizzled 16-byte chunks,
// MFMA fragments read back with ds_read_b128), the tile -> XCD mapping and the epilogue's memory volume are those of
// csrc/gemm.hip's gemm_nt384_body; tile shape, K-slice width, stage count, prefetch distance and waves per workgroup are
// template parameters, and the parts of the loop can be switched off one by one:
//   MODE bit 0: fill the stages (buffer_load ... lds)      bit 1: fragment reads + MFMAs      bit 2: f32 residual
//   read-modify-write epilogue (C += acc, the out / mlp2 projections' epilogue volume: 8 bytes per output)
//   bit 3: GELU-type epilogue instead (mlp1): bias + an exp / rcp / polynomial per element on the VALU, bf16 store, no read
//   bit 4: the residual epilogue PACED (drained after every 16-row slice, so that it never queues more than 8 loads per wave)
//
//   hipcc --offload-arch=gfx950 -O3 -o gemm_loop_probe tools/micro/gemm_loop_probe.hip && ./gemm_loop_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct Params {
    const char* A;  // [M][K] bf16
    const char* W;  // [N][K] bf16
    float* C;       // [M][N] f32 (epilogue) or scratch
    int M, N, K, tiles_m, tiles_n;
    int persistent;   // 1: gridDim.x workgroups walk the tiles with stride gridDim.x (tile = blockIdx.x + i * gridDim.x)
    int delay_ticks;  // persistent: workgroups of the grid's second half (the second one on each CU) start this many 10 ns ticks late,
                      // so that their main loops fall into the first half's epilogues
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// WPS: waves per SIMD the register allocation must leave room for (2 = one 8-wave workgroup or two 4-wave ones per CU: <= 256
// VGPRs + AGPRs per wave; 3 = three 4-wave workgroups: <= 168)
// (the body is a __device__ function and the kernel a thin wrapper, as in csrc/gemm.hip: the host pass cannot digest the LDS-DMA
// builtin inside a __global__ function's own body)
template <int BM, int BN, int WM, int WN, int KB, int STAGES, int DIST, int MODE, int PERSIST>
__device__ __forceinline__ void loop_body(Params p) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN, MT = TM / 16, NT = TN / 16;
    static_assert(MT * NT * 4 <= 192, "accumulators");
    constexpr int ROWS = BN + BM;                 // W rows first, then A rows
    constexpr int STAGE = ROWS * KB;              // bytes
    constexpr int RPP = 1024 / KB;                // rows per DMA piece (one wave instruction = 1 KiB)
    constexpr int PIECES = ROWS / RPP;
    static_assert(PIECES % NW == 0 && BN % RPP == 0, "pieces per wave");
    constexpr int PPW = PIECES / NW;
    constexpr int CPR = KB / 16;                  // 16-byte chunks per row: 8 | 4
    static_assert(STAGES >= DIST + 1, "a stage is refilled only after its last reader has passed the barrier");
    static_assert((DIST - 1) * PPW < 64, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int nblocks = p.tiles_m * p.tiles_n;
    if (PERSIST && p.delay_ticks > 0 && (int)blockIdx.x >= (int)gridDim.x / 2) {
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (unsigned long long)p.delay_ticks) __builtin_amdgcn_s_sleep(32);
    }
    const int64_t ldb = (int64_t)p.K * 2;
    const int frow = lane & 15, fq = lane >> 4;
    const int nk = p.K * 2 / KB;
    auto swz = [](int row) { return CPR == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3); };
    // PERSIST = 0: one tile per workgroup, compiled as the plain kernel (the loop and its live values vanish)
    for (int vb = blockIdx.x; vb < (PERSIST ? nblocks : (int)blockIdx.x + 1); vb += PERSIST ? (int)gridDim.x : 1) {
    // tile of this step: gemm.hip's mapping (consecutive ids on one XCD; groups of 8 row tiles walk the columns)
    int id;
    {
        const int bid = vb, q = nblocks >> 3, r = nblocks & 7, xcd = bid & 7, idx = bid >> 3;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int group_size = 8 * p.tiles_n, group = id / group_size, first_m = group * 8;
    const int gm = min(p.tiles_m - first_m, 8), in_group = id - group * group_size;
    const int m0 = (first_m + in_group % gm) * BM, n0 = (in_group / gm) * BN;
    if (PERSIST && vb != (int)blockIdx.x) __syncthreads();  // the previous tile's last fragment reads are done before its stages are refilled
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)n0 * ldb), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)m0 * ldb), 0, 0x7fffffff, 0x00020000);
    int off[PPW];  // per-lane byte offset of the piece's source chunk inside its matrix (K offset added per slice)
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int piece = i * NW + wave;
        const int row = piece * RPP + lane / CPR;  // LDS row (W rows, then A rows)
        const int c = (lane % CPR) ^ swz(row);
        off[i] = row < BN ? (min(n0 + row, p.N - 1) - n0) * (int)ldb + c * 16 : (min(m0 + row - BN, p.M - 1) - m0) * (int)ldb + c * 16;
    }
    auto stage = [&](int kt, int buf) {
        if constexpr (MODE & 1) {
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                const int piece = i * NW + wave;  // wave-uniform
                char* dst = smem + buf * STAGE + piece * 1024;
                if (piece * RPP < BN) __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_ptr_t)dst, 16, off[i], kt * KB, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)dst, 16, off[i], kt * KB, 0, 0);
            }
        }
    };
    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int buf) {
        if constexpr (MODE & 2) {
            const char* wb = smem + buf * STAGE + (wn * TN + frow) * KB;
            const char* ab = smem + buf * STAGE + BN * KB + (wm * TM + frow) * KB;
#pragma unroll
            for (int kk = 0; kk < KB / 64; ++kk) {
                bf16x8 fw[NT];
#pragma unroll
                for (int i = 0; i < NT; ++i) fw[i] = *reinterpret_cast<const bf16x8*>(wb + i * 16 * KB + (((fq + 4 * kk) ^ swz(wn * TN + 16 * i + frow)) << 4));
                constexpr int JG = MT > 6 ? 4 : MT;
#pragma unroll
                for (int jg = 0; jg < MT / JG; ++jg) {
                    bf16x8 fx[JG];
#pragma unroll
                    for (int j = 0; j < JG; ++j)
                        fx[j] = *reinterpret_cast<const bf16x8*>(ab + (JG * jg + j) * 16 * KB + (((fq + 4 * kk) ^ swz(wm * TM + 16 * (JG * jg + j) + frow)) << 4));
#pragma unroll
                    for (int i = 0; i < NT; ++i)
#pragma unroll
                        for (int j = 0; j < JG; ++j)
                            acc[i][JG * jg + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fx[j], acc[i][JG * jg + j], 0, 0, 0);
                }
            }
        }
    };
    // prologue: DIST slices in flight
#pragma unroll
    for (int d = 0; d < DIST; ++d)
        if (d < nk) stage(d, d % STAGES);
    for (int kt = 0; kt < nk; ++kt) {
        // slice kt has landed when at most the DIST - 1 younger slices of this wave are still in flight
        if (kt + DIST - 1 < nk) wait_vmcnt<(DIST - 1) * PPW>();
        else wait_vmcnt<0>();
        __syncthreads();  // every wave's pieces of slice kt are in LDS; everybody has finished reading slice kt - 1
        if (kt + DIST < nk) stage(kt + DIST, (kt + DIST) % STAGES);
        compute(kt % STAGES);
    }
    if constexpr (MODE & 8) {
        // mlp1's epilogue class: per element bias + exp + rcp + a short polynomial (the erf-GELU's instruction mix), 2 bytes stored
        __bf16* Cb = reinterpret_cast<__bf16*>(p.C);
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int m = m0 + wm * TM + 16 * j + frow;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int n = n0 + wn * TN + 16 * i + 4 * fq;
                typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = acc[i][j][e] + 0.01f * (float)(n + e);
                    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * fabsf(x) * 0.70710678f);
                    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
                    const float erfv = 1.0f - poly * __expf(-0.5f * x * x);
                    o[e] = (__bf16)(0.5f * x * (1.0f + (x < 0.f ? -erfv : erfv)));
                }
                if (m < p.M && n + 3 < p.N) *reinterpret_cast<bf16x4*>(Cb + (int64_t)m * p.N + n) = o;
            }
        }
    } else if constexpr (MODE & 4) {
        // the out / mlp2 projections' epilogue volume: C (the f32 residual stream) += acc, on the MFMA layout
        // (a lane holds 4 consecutive columns of one row: 16 rows x 64 bytes per wave instruction)
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int m = m0 + wm * TM + 16 * j + frow;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int n = n0 + wn * TN + 16 * i + 4 * fq;
                if (m < p.M && n + 3 < p.N) {
                    f32x4* c = reinterpret_cast<f32x4*>(p.C + (int64_t)m * p.N + n);
                    *c = *c + acc[i][j];
                }
            }
            if constexpr (MODE & 16) {
                wait_vmcnt<0>();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        p.C[(int64_t)vb * blockDim.x + tid] = s;  // keeps the MFMAs alive: 4 bytes per thread
    }
    }  // tiles
}

template <int BM, int BN, int WM, int WN, int KB, int STAGES, int DIST, int MODE, int WPS = 2, int PERSIST = 0>
__global__ __launch_bounds__(64 * WM * WN, WPS) void loop_kernel(Params p) {
    loop_body<BM, BN, WM, WN, KB, STAGES, DIST, MODE, PERSIST>(p);
}

__global__ void fill_kernel(uint16_t* x, size_t n, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + seed;
        h ^= h >> 15;
        h *= 2246822519u;
        h ^= h >> 13;
        // a bf16 in roughly [-1, 1): sign + exponent 0x3d..0x3f + 7 mantissa bits (random data: the clock a real GEMM sees)
        x[i] = (uint16_t)(((h & 1u) << 15) | ((0x3d + (h >> 1) % 3u) << 7) | ((h >> 8) & 0x7f));
    }
}

template <int BM, int BN, int WM, int WN, int KB, int STAGES, int DIST, int MODE, int WPS = 2, int PERSIST = 0>
static void run(const char* name, const char* A, const char* W, float* C, int M, int N, int K, double delay_us = 0.0) {
    const int persistent = PERSIST;
    Params p{A, W, C, M, N, K, (M + BM - 1) / BM, (N + BN - 1) / BN, persistent, (int)(delay_us * 100.0)};
    const int smem = STAGES * (BN + BM) * KB, threads = 64 * WM * WN, tiles = p.tiles_m * p.tiles_n;
    int grid = tiles;
    auto kern = loop_kernel<BM, BN, WM, WN, KB, STAGES, DIST, MODE, WPS, PERSIST>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    int per_cu = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, smem));
    if (persistent) grid = 256 * per_cu;  // every workgroup resident; the second half of the grid = each CU's second workgroup
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), smem, 0, p);
    CK(hipEventRecord(e0, 0));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), smem, 0, p);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
    const double rounds = (double)tiles / (256.0 * per_cu);
    printf("  %-34s %s%s%s  %3dx%-3d %d waves KB %3d stages %d dist %d | LDS %3d KB, %d wg/CU, %5d tiles = %5.2f rounds | %7.1f us  %6.0f TF/s"
           "  | %5.2f us per 64-element K step and round\n",
           name, (MODE & 1) ? "F" : "-", (MODE & 2) ? "M" : "-", (MODE & 8) ? "G" : (MODE & 16) ? "e" : (MODE & 4) ? "E" : "-", BM, BN, WM * WN, KB, STAGES, DIST, smem >> 10, per_cu, tiles, rounds, us,
           (MODE & 2) ? tf : 0.0, us / (K / 64.0) / (rounds < 1 ? 1 : rounds));
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
}

int main() {
    const int M = 96000;  // 64 clips x 1500 frames
    char *A, *W;
    float* C;
    const int Kmax = 3072, Nmax = 3072;
    CK(hipMalloc(&A, (size_t)M * Kmax * 2));
    CK(hipMalloc(&W, (size_t)Nmax * Kmax * 2));
    CK(hipMalloc(&C, (size_t)M * 768 * 4 > (size_t)M * Nmax * 2 ? (size_t)M * 768 * 4 : (size_t)M * Nmax * 2));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (uint16_t*)A, (size_t)M * Kmax, 1u);
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, (uint16_t*)W, (size_t)Nmax * Kmax, 2u);
    CK(hipMemset(C, 0, (size_t)M * 768 * 4));
    CK(hipDeviceSynchronize());
    for (int K : {768, 3072}) {
        const int N = 768;
        printf("M = %d, N = %d, K = %d (%s projection's shape); F = fills, M = MFMAs, E = f32 residual read-modify-write epilogue\n", M, N, K,
               K == 768 ? "out" : "mlp2");
        // the shipped design: one 8-wave workgroup per CU, 2 stages of 80 KB, one slice ahead, full drain per slice
        run<384, 256, 2, 4, 128, 2, 1, 7>("shipped loop", A, W, C, M, N, K);
        run<384, 256, 2, 4, 128, 2, 1, 3>("  without the epilogue", A, W, C, M, N, K);
        run<384, 256, 2, 4, 128, 2, 1, 1>("  fills only", A, W, C, M, N, K);
        run<384, 256, 2, 4, 128, 2, 1, 2>("  MFMAs only", A, W, C, M, N, K);
        run<384, 256, 2, 4, 128, 2, 1, 4>("  epilogue only", A, W, C, M, N, K);
        // the same tile, thinner slices, more of them in flight (counted vmcnt, no drain)
        run<384, 256, 2, 4, 64, 4, 2, 3>("4 x 40 KB, two slices ahead", A, W, C, M, N, K);
        run<384, 256, 2, 4, 64, 4, 3, 3>("4 x 40 KB, three slices ahead", A, W, C, M, N, K);
        run<384, 256, 2, 4, 64, 4, 3, 7>("  ... with the epilogue", A, W, C, M, N, K);
        // two independent 4-wave workgroups per CU: 192 x 256 tiles (192 accumulators per wave as before)
        run<192, 256, 2, 2, 64, 2, 1, 7>("2 wg/CU 192x256, 2 x 28 KB", A, W, C, M, N, K);
        run<192, 256, 2, 2, 64, 2, 1, 3>("  without the epilogue", A, W, C, M, N, K);
        run<192, 256, 2, 2, 64, 2, 1, 1>("  fills only", A, W, C, M, N, K);
        run<192, 256, 2, 2, 64, 2, 1, 2>("  MFMAs only", A, W, C, M, N, K);
        run<192, 256, 2, 2, 64, 2, 1, 4>("  epilogue only", A, W, C, M, N, K);
        // the same two workgroups per CU as a PERSISTENT tile loop, the second one of each CU starting late by about one main loop
        run<192, 256, 2, 2, 64, 2, 1, 7, 2, 1>("  persistent, no offset", A, W, C, M, N, K, 0.0);
        for (double d : {8.0, 16.0, 24.0, 32.0, 48.0})
            run<192, 256, 2, 2, 64, 2, 1, 7, 2, 1>(d == 8.0 ? "  persistent, offset 8/16/24/32/48 us" : "", A, W, C, M, N, K, d * (K / 768.0));
        // ... and with 128-byte slices in ONE 56 KB stage plus a half: not expressible; 3 workgroups of 192 x 128 instead
        run<192, 128, 2, 2, 64, 2, 1, 7, 3>("3 wg/CU 192x128 (96 acc), 2 x 20 KB", A, W, C, M, N, K);
        run<192, 128, 2, 2, 128, 2, 1, 7>("2 wg/CU 192x128, 2 x 40 KB slices", A, W, C, M, N, K);
    }
    {
        const int N = 3072, K = 768;
        printf("M = %d, N = %d, K = %d (mlp1 projection's shape); G = GELU-type epilogue (VALU + bf16 store, no read)\n", M, N, K);
        run<384, 256, 2, 4, 128, 2, 1, 3 | 8>("shipped loop + GELU-type epilogue", A, W, C, M, N, K);
        run<384, 256, 2, 4, 128, 2, 1, 3>("  without the epilogue", A, W, C, M, N, K);
        run<384, 256, 2, 4, 128, 2, 1, 8>("  epilogue only", A, W, C, M, N, K);
        run<192, 256, 2, 2, 64, 2, 1, 3 | 8>("2 wg/CU 192x256 + GELU-type", A, W, C, M, N, K);
        run<192, 256, 2, 2, 64, 2, 1, 3 | 8, 2, 1>("  persistent, no offset", A, W, C, M, N, K, 0.0);
        for (double d : {8.0, 16.0, 24.0, 32.0})
            run<192, 256, 2, 2, 64, 2, 1, 3 | 8, 2, 1>(d == 8.0 ? "  persistent, offset 8/16/24/32 us" : "", A, W, C, M, N, K, d);
    }
    {
        const int N = 768, K = 768;
        printf("M = %d, N = %d, K = %d (out projection) with the residual epilogue PACED (e): drained after every 16-row slice\n", M, N, K);
        run<384, 256, 2, 4, 128, 2, 1, 7 | 16>("shipped loop, paced epilogue", A, W, C, M, N, K);
        run<192, 256, 2, 2, 64, 2, 1, 7 | 16>("2 wg/CU 192x256, paced", A, W, C, M, N, K);
        run<192, 256, 2, 2, 64, 2, 1, 7 | 16, 2, 1>("  persistent, no offset", A, W, C, M, N, K, 0.0);
        for (double d : {16.0, 32.0})
            run<192, 256, 2, 2, 64, 2, 1, 7 | 16, 2, 1>(d == 16.0 ? "  persistent, offset 16/32 us" : "", A, W, C, M, N, K, d);
    }
    return 0;
}
