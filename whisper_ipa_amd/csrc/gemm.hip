// K4: C = epilogue(A * W^T) on the MFMA units of gfx950.
//
// Replaces every nn.Linear / Conv1d of mlx_whisper.whisper on the path
// (call sites: scripts/train_whisper_ipa.py:223,232; scripts/transcribe_single.py:54-55).
//
// Tiling (v1): 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per
// wave = 4x4 MFMA 16x16 tiles), K-step = 128 BYTES per row (64 bf16 / 32 f32) so both
// dtypes share one LDS image: [rows][8 x 16-byte chunks], chunk index XOR ((row>>1)&7)
// -> every ds_read_b128 lane group hits 16 distinct 16-byte slots (conflict free).
// Operand roles are swapped w.r.t. the textbook (W feeds the MFMA "A" side) so each lane
// ends up with 4 CONSECUTIVE output columns of one row -> 8/16-byte epilogue stores.
// f32 inputs use v_mfma_f32_16x16x4_f32 (exact f32 fma chain); a 16-byte fragment feeds
// four K=4 steps with a consistent K permutation on both operands.
// Global->register->LDS staging, double buffered, one barrier per K-step; loads for
// step k+1 are issued before the MFMAs of step k (async-STAGE split).
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include <type_traits>

#include "wipa_common.h"

namespace {

struct GemmParams {
    int f32_split = 0;  // f32 inputs: every product as three bf16 MFMA terms (split_bf16x2) instead of the f32 MFMA
    const char* A;
    const char* W;
    char* C;
    const float* bias;
    const char* residual;
    const float* pos;
    const int64_t* c_offset_dev;
    int64_t lda_b, ldw_b;  // bytes
    int64_t ldc, ldpos;    // elements
    int64_t rg_stride, cg_stride, c_offset;
    int M, N, K;
    int rg_in, rg_valid, cg_in;
    int zero_invalid, bias_along_m, act, col_scale_n;
    float col_scale;
    int tiles_m, tiles_n;
    int vec_ok;
    int stage_ok;  // epilogue_staged may be used (16-byte row-major stores are legal for this output mapping)
    int k_slices;
    int64_t slab_stride;
    int a_trans = 0, w_trans = 0;    // operand given K-major ([K][rows], lda_b / ldw_b = bytes per k-row): 128x128 f32 kernel only
    const float* w_scale = nullptr;  // fp8 (e4m3) weights: per-output-column dequantisation scale
    const float* a_scale = nullptr;  // fp8 (e4m3) A operand: per-row dequantisation scale
    // LayerNorm prologue (weight-streaming kernel only): A = LayerNorm(ln_x) computed in the kernel
    const float* ln_x = nullptr;
    const float* ln_w = nullptr;
    const float* ln_b = nullptr;
    int64_t ln_ldx = 0;
    float ln_eps = 0.f;
    // greedy partials (gemm_wide_persistent_kernel<.., GREEDY = true>: wipa_logits_greedy)
    const float* g_mask_first = nullptr;
    const float* g_mask_always = nullptr;
    const int32_t* g_pos = nullptr;
    int g_n_init = 0;
    float* g_part = nullptr;  // [M rows][3: max | sum exp | arg-max][gridDim.x * 8 waves]
    int g_store = 1;          // 0: the logits themselves are not written
};

template <typename OutT>
__device__ __forceinline__ void store4(char* C, int64_t off, const float (&v)[4], int nvalid, bool vec) {
    OutT* p = reinterpret_cast<OutT*>(C) + off;
    if (vec && nvalid == 4) {
        if constexpr (sizeof(OutT) == 4) {
            f32x4 o = {v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(p) = o;
        } else {
            bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
            *reinterpret_cast<bf16x4*>(p) = o;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < nvalid) p[r] = from_f32<OutT>(v[r]);
    }
}

template <typename OutT>
__device__ __forceinline__ void load4(const char* R, int64_t off, float (&v)[4], int nvalid, bool vec) {
    const OutT* p = reinterpret_cast<const OutT*>(R) + off;
    if (vec && nvalid == 4) {
        if constexpr (sizeof(OutT) == 4) {
            f32x4 o = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = o[r];
        } else {
            bf16x4 o = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (float)o[r];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (r < nvalid) ? to_f32<OutT>(p[r]) : 0.f;
    }
}

// Shared epilogue (see wipa_gemm in wipa.h).  Row- and column-dependent address parts, the
// remap divisions and the bias loads are hoisted: once per output row / per 4-column group
// of a lane, not once per accumulator tile.
struct EpiRow {
    int64_t roff;
    int gr;
    float bm;
    bool store, valid;
};
struct EpiCol {
    int64_t coff;
    int n, nvalid;
    float b[4], sc[4];
    bool ok;
};
__device__ __forceinline__ EpiRow epi_row(const GemmParams& p, int m, int64_t coff_dev) {
    EpiRow r;
    r.store = m < p.M;
    int gi = 0, gr = m;
    if (p.rg_in < p.M) {  // uniform: remapped rows (conv halo layouts, KV caches)
        gi = m / p.rg_in;
        gr = m - gi * p.rg_in;
    }
    r.gr = gr;
    r.valid = gr < p.rg_valid;
    if (!r.valid && !p.zero_invalid) r.store = false;
    r.roff = coff_dev + (int64_t)gi * p.rg_stride + (int64_t)gr * p.ldc;
    r.bm = (p.bias && p.bias_along_m && m < p.M) ? p.bias[m] : 0.f;
    return r;
}
__device__ __forceinline__ EpiCol epi_col(const GemmParams& p, int n) {
    EpiCol c;
    c.n = n;
    c.ok = n < p.N;
    c.nvalid = min(4, p.N - n);
    int cgi = 0, cgr = n;
    if (p.cg_in < p.N) {
        cgi = n / p.cg_in;
        cgr = n - cgi * p.cg_in;
    }
    c.coff = (int64_t)cgi * p.cg_stride + cgr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        c.b[r] = (p.bias && !p.bias_along_m && c.ok && r < c.nvalid) ? p.bias[n + r] : 0.f;
        c.sc[r] = (n + r < p.col_scale_n) ? p.col_scale : 1.0f;
    }
    return c;
}
template <typename OutT>
__device__ __forceinline__ void epilogue4(const GemmParams& p, const f32x4& a, const EpiRow& R, const EpiCol& Cc, bool vec) {
    if (!R.store || !Cc.ok) return;
    const int64_t off = R.roff + Cc.coff;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (a[r] + Cc.b[r] + R.bm) * Cc.sc[r];
    if (p.act == 1) {
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
            const f32x2 g = gelu_erf2(f32x2{v[r], v[r + 1]});
            v[r] = g.x;
            v[r + 1] = g.y;
        }
    }
    if (p.pos) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < Cc.nvalid) v[r] += p.pos[(int64_t)R.gr * p.ldpos + Cc.n + r];
    }
    if (p.residual) {
        float rr[4];
        load4<OutT>(p.residual, off, rr, Cc.nvalid, vec);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += rr[r];
    }
    if (!R.valid) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = 0.f;
    }
    store4<OutT>(p.C, off, v, Cc.nvalid, vec);
}

// ---------------------------------------------------------------------------------------
// Staged epilogue of one wave's [16*MTILES rows x 64 columns] accumulator block.  The MFMA C layout
// gives a lane 4 consecutive columns of ONE row, i.e. 8/16-byte stores that touch 16 different rows per
// instruction: 32 store instructions per lane for a 128x64 block, issue-bound (the store tail cost as
// much as half the K=768 main loop).  Here every 16-row slice goes through a wave-private 4 KiB LDS
// scratch (f32, 256-byte rows, 16-byte chunk index XOR row) and comes back ROW-major: a lane then owns
// 16 output bytes of one row and 8 (bf16) / 16 (f32) consecutive lanes cover a whole 128/256-byte row
// segment -> half / equal the store instructions, all full-line, and bias / residual / positional
// operands are read with the same coalesced shape.
// row slices J0 .. J0+NJ-1 of the MTILES the wave holds; ACT: GELU path compiled in; RES_EARLY: residual fetched ahead
template <typename OutT, int MTILES, int J0 = 0, int NJ = MTILES, bool ACT = true, bool RES_EARLY = true>
__device__ __forceinline__ void epilogue_staged(const GemmParams& p, const f32x4 (&acc)[4][MTILES], char* scratch, int m_base,
                                                int n_base, int64_t coff_dev, int lane) {
    constexpr int EPC = 16 / (int)sizeof(OutT);  // output elements per lane per store (8 bf16 / 4 f32)
    constexpr int LPR = 64 / EPC;                // lanes per 64-column row segment
    constexpr int RPP = 64 / LPR;                // rows per pass
    constexpr int NCH = EPC / 4;                 // 16-byte f32 chunks a lane reads back
    const int frow = lane & 15, fq = lane >> 4;
    const int rrow = lane / LPR, cch = lane % LPR;
    const int n = n_base + cch * EPC;
    // column context of this lane (constant over the rows)
    const bool col_ok = n < p.N;
    const int nvalid = min(EPC, p.N - n);
    int cgi = 0, cgr = n;
    if (p.cg_in < p.N) {
        cgi = n / p.cg_in;
        cgr = n - cgi * p.cg_in;
    }
    const int64_t coff = (int64_t)cgi * p.cg_stride + cgr;
    float bias[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) bias[e] = (p.bias && !p.bias_along_m && col_ok && e < nvalid) ? p.bias[n + e] : 0.f;
    const float sc = n < p.col_scale_n ? p.col_scale : 1.0f;  // stage_ok guarantees col_scale_n % EPC == 0: one decision per lane
    // The residual of slice j+1 is fetched while slice j goes through the LDS transpose: without this each pass issued its
    // own dependent 16-byte load and the f32-residual epilogue ran at ~1.7x its HBM time.
    constexpr int NPASS = 16 / RPP;
    // with 12 row slices (192 accumulators) a slice-ahead prefetch would spill: there the residual of a slice is issued at the
    // START of that slice instead, so it at least overlaps the slice's LDS round trip
    constexpr bool PREFETCH = RES_EARLY;  // (the GELU instantiation of the 12-slice tile has no registers to spare)
    constexpr bool AHEAD = MTILES <= 8;
    const bool res_vec = PREFETCH && p.residual != nullptr && col_ok && nvalid == EPC;
    Vec16<OutT> rnext[NPASS];
    auto fetch_residual = [&](int j, Vec16<OutT> (&dst)[NPASS]) {
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const EpiRow R = epi_row(p, m_base + 16 * j + pass * RPP + rrow, coff_dev);
            if (R.store) {
                typedef decltype(dst[pass].v) VT;
                dst[pass].v = __builtin_nontemporal_load(reinterpret_cast<const VT*>(reinterpret_cast<const OutT*>(p.residual) + R.roff + coff));
            }
        }
    };
    if (AHEAD && res_vec) fetch_residual(J0, rnext);
#pragma unroll
    for (int j = J0; j < J0 + NJ; ++j) {
        Vec16<OutT> rcur[NPASS];
        if (res_vec) {
            if constexpr (AHEAD) {
#pragma unroll
                for (int pass = 0; pass < NPASS; ++pass) rcur[pass] = rnext[pass];
                if (j + 1 < J0 + NJ) fetch_residual(j + 1, rnext);
            } else {
                fetch_residual(j, rcur);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<f32x4*>(scratch + frow * 256 + (((4 * i + fq) ^ frow) << 4)) = acc[i][j];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private scratch: in-order DS, no barrier needed
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int row = pass * RPP + rrow;
            float v[EPC];
#pragma unroll
            for (int t = 0; t < NCH; ++t) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(scratch + row * 256 + (((cch * NCH + t) ^ row) << 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * t + e] = a[e];
            }
            const EpiRow R = epi_row(p, m_base + 16 * j + row, coff_dev);
            if (!R.store || !col_ok) continue;
            const int64_t off = R.roff + coff;
#pragma unroll
            for (int e = 0; e < EPC; ++e) v[e] = (v[e] + bias[e] + R.bm) * sc;
            if constexpr (ACT) {
                if (p.act == 1) {
#pragma unroll
                    for (int e = 0; e < EPC; e += 2) {
                        const f32x2 g = gelu_erf2(f32x2{v[e], v[e + 1]});
                        v[e] = g.x;
                        v[e + 1] = g.y;
                    }
                }
            }
            if (p.pos) {
#pragma unroll
                for (int e = 0; e < EPC; ++e)
                    if (e < nvalid) v[e] += p.pos[(int64_t)R.gr * p.ldpos + n + e];
            }
            const bool full = nvalid == EPC;
            if (p.residual) {
                const OutT* rp = reinterpret_cast<const OutT*>(p.residual) + off;
                if (full) {
                    if constexpr (PREFETCH) {
#pragma unroll
                        for (int e = 0; e < EPC; ++e) v[e] += rcur[pass].get(e);
                    } else {
                        typedef decltype(rcur[pass].v) VT;
                        Vec16<OutT> rr;
                        rr.v = __builtin_nontemporal_load(reinterpret_cast<const VT*>(rp));
#pragma unroll
                        for (int e = 0; e < EPC; ++e) v[e] += rr.get(e);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < EPC; ++e)
                        if (e < nvalid) v[e] += to_f32<OutT>(rp[e]);
                }
            }
            if (!R.valid) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) v[e] = 0.f;
            }
            OutT* cp = reinterpret_cast<OutT*>(p.C) + off;
            if (full) {
                if constexpr (sizeof(OutT) == 4) {
                    __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4*>(cp));
                } else {
                    __builtin_nontemporal_store(bf16x8{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3],
                                                            (__bf16)v[4], (__bf16)v[5], (__bf16)v[6], (__bf16)v[7]}, reinterpret_cast<bf16x8*>(cp));
                }
            } else {
#pragma unroll
                for (int e = 0; e < EPC; ++e)
                    if (e < nvalid) cp[e] = from_f32<OutT>(v[e]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the next slice overwrites the scratch
    }
}

constexpr int BM = 128, BN = 128, ROWB = 128;  // ROWB: bytes of K per LDS row
constexpr int TILE_BYTES = BM * ROWB;          // 16 KiB per operand tile
constexpr int GROUP_M = 8;

// TA / TW (float32 only): the operand arrives K-major -- A as [K][M], W as [K][N] -- and is transposed by the staging pass
// (16-byte loads along the rows of one k, four 4-byte LDS writes into the usual [row][k] image), so a backward pass can
// multiply by x^T, dy^T or W^T without a transpose kernel and a second copy in HBM.
template <typename T, typename OutT, bool TA = false, bool TW = false>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmParams p) {
    static_assert(!(TA || TW) || sizeof(T) == 4, "K-major operands: float32 only");
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][W tile | A tile]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wn = wave >> 1, wm = wave & 1;

    // ---- workgroup -> tile: XCD-contiguous chunks (bijective), then GROUP_M super-rows
    const int nblocks = p.tiles_m * p.tiles_n;
    int id;
    {
        const int bid = blockIdx.x;
        const int q = nblocks >> 3, r = nblocks & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int group_size = GROUP_M * p.tiles_n;
    const int group = id / group_size;
    const int first_m = group * GROUP_M;
    const int gm = min(p.tiles_m - first_m, GROUP_M);
    const int in_group = id - group * group_size;
    const int tile_m = first_m + in_group % gm;
    const int tile_n = in_group / gm;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- staging assignment: 4 rows of each operand per thread, one 16-byte chunk each
    const int srow = tid >> 3, schunk = tid & 7;
    const char* gA[4];
    const char* gW[4];
    int lds_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = srow + 32 * i;
        const int m = min(m0 + row, p.M - 1);
        const int n = min(n0 + row, p.N - 1);
        gA[i] = p.A + (int64_t)m * p.lda_b + schunk * 16;
        gW[i] = p.W + (int64_t)n * p.ldw_b + schunk * 16;
        lds_off[i] = row * ROWB + ((schunk ^ ((row >> 1) & 7)) << 4);
    }
    // K-major operand: thread (k = tid >> 3 of the step's 32, vector v = (tid & 7) + 8 i) loads rows 4v .. 4v+3 of that k
    const int tk = tid >> 3;
    const char* tA[4];
    const char* tW[4];
    int t_off[4];  // LDS offset of (row 4v, k): row e adds e*ROWB and its own chunk swizzle
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 4 * ((tid & 7) + 8 * i);
        if constexpr (TA) tA[i] = p.A + (int64_t)tk * p.lda_b + (int64_t)min(m0 + r, p.M - 4) * 4;
        if constexpr (TW) tW[i] = p.W + (int64_t)tk * p.ldw_b + (int64_t)min(n0 + r, p.N - 4) * 4;
        t_off[i] = r * ROWB + (tk & 3) * 4;
    }
    f32x4 ra[4], rw[4];
    auto gload = [&](int kt) {
        const int64_t kb = (int64_t)kt * ROWB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (TW) rw[i] = *reinterpret_cast<const f32x4*>(tW[i] + (int64_t)kt * 32 * p.ldw_b);
            else rw[i] = *reinterpret_cast<const f32x4*>(gW[i] + kb);
            if constexpr (TA) ra[i] = *reinterpret_cast<const f32x4*>(tA[i] + (int64_t)kt * 32 * p.lda_b);
            else ra[i] = *reinterpret_cast<const f32x4*>(gA[i] + kb);
        }
    };
    auto swrite_t = [&](char* tile, const f32x4 (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = 4 * ((tid & 7) + 8 * i) + e;
                *reinterpret_cast<float*>(tile + t_off[i] + e * ROWB + (((tk >> 2) ^ ((row >> 1) & 7)) << 4)) = v[i][e];
            }
    };
    auto swrite = [&](int buf) {
        char* base = smem + buf * (2 * TILE_BYTES);
        if constexpr (TW) swrite_t(base, rw);
        if constexpr (TA) swrite_t(base + TILE_BYTES, ra);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (!TW) *reinterpret_cast<f32x4*>(base + lds_off[i]) = rw[i];
            if constexpr (!TA) *reinterpret_cast<f32x4*>(base + TILE_BYTES + lds_off[i]) = ra[i];
        }
    };

    f32x4 acc[4][4];  // [n tile i][m tile j]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;
    const int fsw = (lane >> 1) & 7;  // == ((row >> 1) & 7) for row = 16*x + (lane & 15)
    const int fq = lane >> 4;
    // split-K (weight gradients: small M x N, very long K): slice blockIdx.y of k_slices owns a contiguous range of
    // K-steps and writes its partial tile to slab blockIdx.y; the caller adds the slabs in order (wipa_sum_slabs)
    const int nk_all = p.K * (int)sizeof(T) / ROWB;
    const int kz = blockIdx.y;
    const int k_per = nk_all / p.k_slices, k_rem = nk_all % p.k_slices;
    const int kt0 = kz * k_per + min(kz, k_rem);
    const int nk = kt0 + k_per + (kz < k_rem ? 1 : 0);

    if (kt0 < nk) {
        gload(kt0);
        swrite(kt0 & 1);
    }
    __syncthreads();
    for (int kt = kt0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload(kt + 1);
        const char* wb = smem + (kt & 1) * (2 * TILE_BYTES) + (wn * 64 + frow) * ROWB;
        const char* ab = smem + (kt & 1) * (2 * TILE_BYTES) + TILE_BYTES + (wm * 64 + frow) * ROWB;
        bool done = false;
        if constexpr (sizeof(T) == 4) {
            if (p.f32_split) {  // same element order as the 256 / 384 kernels: results do not depend on the tile chosen
                const int c0 = (fq ^ fsw) << 4, c1 = ((fq + 4) ^ fsw) << 4;
                bf16x8 wh[4], wl[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    split_bf16x2(*reinterpret_cast<const f32x4*>(wb + i * 16 * ROWB + c0),
                                 *reinterpret_cast<const f32x4*>(wb + i * 16 * ROWB + c1), wh[i], wl[i]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bf16x8 xh, xl;
                    split_bf16x2(*reinterpret_cast<const f32x4*>(ab + j * 16 * ROWB + c0),
                                 *reinterpret_cast<const f32x4*>(ab + j * 16 * ROWB + c1), xh, xl);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[i], xl, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[i], xh, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[i], xh, acc[i][j], 0, 0, 0);
                    }
                }
                done = true;
            }
        }
        if (!done) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int coff = ((fq + 4 * kk) ^ fsw) << 4;
                typename Mma<T>::Frag fw[4], fx[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    fw[i] = *reinterpret_cast<const typename Mma<T>::Frag*>(wb + i * 16 * ROWB + coff);
                    fx[i] = *reinterpret_cast<const typename Mma<T>::Frag*>(ab + i * 16 * ROWB + coff);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) Mma<T>::run(fw[i], fx[j], acc[i][j]);
            }
        }
        if (kt + 1 < nk) swrite((kt + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue
    const bool vec = p.vec_ok != 0;
    int64_t coff_dev = p.c_offset + (int64_t)kz * p.slab_stride;
    if (p.c_offset_dev) coff_dev += *p.c_offset_dev;
    if (kz > 0) p.bias = nullptr;  // partial slabs: the bias rides on slice 0 only
    EpiCol cols[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cols[i] = epi_col(p, n0 + wn * 64 + 16 * i + 4 * fq);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const EpiRow row = epi_row(p, m0 + wm * 64 + 16 * j + frow, coff_dev);
#pragma unroll
        for (int i = 0; i < 4; ++i) epilogue4<OutT>(p, acc[i][j], row, cols[i], vec);
    }
}

// ---------------------------------------------------------------------------------------
// Skinny GEMM (decode step, M <= 64 rows per workgroup): weight-streaming bound.
// One workgroup owns 16*NT output columns for 16*MT rows; its NW waves split K, every wave
// streams its weight slice ONCE straight from HBM into MFMA fragments (16-byte loads, no LDS
// round trip: nothing is shared between waves), reads the small activation matrix from L2,
// and the partial tiles are summed through LDS.  Grid = N / (16*NT) workgroups, so even a
// 768x768 projection spreads over 48 CUs x 4 waves instead of the 6 workgroups a 128x128
// tiling gives.
template <typename T, typename OutT, int MT, int NT, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(GemmParams p) {
    __shared__ f32x4 red[NW][MT * NT][64];
    typedef typename Mma<T>::Frag Frag;
    constexpr int UB = (NT == 1) ? 6 : (NT == 2 ? 4 : 3);  // k-steps whose loads are in flight together
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * (16 * NT);
    const int m0 = blockIdx.y * (16 * MT);
    const int ksteps_all = p.K * (int)sizeof(T) / 64;  // 64 bytes of K per MFMA fragment step
    // split-K: workgroup slice blockIdx.z of k_slices, then the NW waves of the workgroup
    const int kz = blockIdx.z, S = p.k_slices;
    const int s_per = ksteps_all / S, s_rem = ksteps_all % S;
    const int s_beg = kz * s_per + min(kz, s_rem);
    const int ksteps = s_per + (kz < s_rem ? 1 : 0);
    const int per = ksteps / NW, rem = ksteps % NW;
    const int kb = s_beg + wave * per + min(wave, rem);
    const int ke = kb + per + (wave < rem ? 1 : 0);
    const char* wp[NT];
    const char* xp[MT];
#pragma unroll
    for (int i = 0; i < NT; ++i) wp[i] = p.W + (int64_t)min(n0 + 16 * i + frow, p.N - 1) * p.ldw_b + fq * 16;
#pragma unroll
    for (int j = 0; j < MT; ++j) xp[j] = p.A + (int64_t)min(m0 + 16 * j + frow, p.M - 1) * p.lda_b + fq * 16;
    // epilogue operands (bias, device-side output offset) are fetched NOW so their memory round trip
    // overlaps the weight stream instead of following it
    const bool vec = p.vec_ok != 0;
    int64_t coff_dev = p.c_offset + (int64_t)kz * p.slab_stride;
    if (p.c_offset_dev) coff_dev += *p.c_offset_dev;
    if (kz > 0) p.bias = nullptr;  // partial slabs: the bias rides on slice 0 only
    EpiCol cols[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) cols[i] = epi_col(p, n0 + 16 * i + 4 * fq);
    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int ks = kb;
    for (; ks + UB <= ke; ks += UB) {
        Frag fw[UB][NT], fx[UB][MT];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int64_t off = (int64_t)(ks + u) * 64;
#pragma unroll
            for (int i = 0; i < NT; ++i) fw[u][i] = *reinterpret_cast<const Frag*>(wp[i] + off);
#pragma unroll
            for (int j = 0; j < MT; ++j) fx[u][j] = *reinterpret_cast<const Frag*>(xp[j] + off);
        }
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j) Mma<T>::run(fw[u][i], fx[u][j], acc[i][j]);
    }
    constexpr int UB2 = 3;  // mid-size batch so a 3..5-step slice still has all its loads in flight together
    for (; ks + UB2 <= ke; ks += UB2) {
        Frag fw[UB2][NT], fx[UB2][MT];
#pragma unroll
        for (int u = 0; u < UB2; ++u) {
            const int64_t off = (int64_t)(ks + u) * 64;
#pragma unroll
            for (int i = 0; i < NT; ++i) fw[u][i] = *reinterpret_cast<const Frag*>(wp[i] + off);
#pragma unroll
            for (int j = 0; j < MT; ++j) fx[u][j] = *reinterpret_cast<const Frag*>(xp[j] + off);
        }
#pragma unroll
        for (int u = 0; u < UB2; ++u)
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j) Mma<T>::run(fw[u][i], fx[u][j], acc[i][j]);
    }
    for (; ks < ke; ++ks) {
        const int64_t off = (int64_t)ks * 64;
        Frag fw[NT], fx[MT];
#pragma unroll
        for (int i = 0; i < NT; ++i) fw[i] = *reinterpret_cast<const Frag*>(wp[i] + off);
#pragma unroll
        for (int j = 0; j < MT; ++j) fx[j] = *reinterpret_cast<const Frag*>(xp[j] + off);
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) Mma<T>::run(fw[i], fx[j], acc[i][j]);
    }
#pragma unroll
    for (int t = 0; t < NT * MT; ++t) red[wave][t][lane] = acc[t / MT][t % MT];
    __syncthreads();
#pragma unroll
    for (int t0 = 0; t0 < NT * MT; t0 += NW) {
        const int t = t0 + wave;
        if (t < NT * MT) {
            f32x4 s = red[0][t][lane];
#pragma unroll
            for (int w = 1; w < NW; ++w) s += red[w][t][lane];
            const int i = t / MT, j = t - i * MT;
            EpiCol cc = cols[0];  // select with static register indices (i is wave dependent)
#pragma unroll
            for (int ii = 1; ii < NT; ++ii)
                if (i == ii) cc = cols[ii];
            epilogue4<OutT>(p, s, epi_row(p, m0 + 16 * j + frow, coff_dev), cc, vec);
        }
    }
}

// Wide skinny GEMM (the logits projection: M <= 64 rows, N = 51 865 columns, K <= 1024), persistent form.  gemm_skinny_kernel
// gives every 64-column workgroup its own copy of the activation matrix from L2 -- 811 workgroups x 98 KB = as many bytes as the
// weights themselves, through the same per-CU load path.  Here one workgroup per CU stages the activations ONCE into LDS
// ([64][K + 8] bf16) and its eight waves then walk 16-column tiles of the weight matrix on their own (tile = wave index + k x
// number of waves): a tile is 16 rows x K, all of its 16-byte loads in flight together, multiplied against the four 16-row
// activation tiles read from LDS; no reduction between waves.  bf16 in, any epilogue of the skinny kernel.
//
// GREEDY (wipa_logits_greedy: the logits projection of a greedy decode step, plain f32 output): the wave also keeps, per row, the
// running maximum / arg-max / sum of exponentials of the FILTERED logits (logit + suppress mask) over the columns it computes, and
// writes ONE partial per (row, wave) at the end -- the step's last launch merges 2 048 partials per row instead of reading the
// 13 MB of logits back, and with g_store = 0 the logits are not written at all (8 of this launch's 32 us were its 64-byte stores).
// Ties resolve to the lowest column like the row-scanning kernels: columns ascend within a wave, merges compare (value, index).
template <typename OutT, int KS, bool GREEDY = false>
__global__ __launch_bounds__(512) void gemm_wide_persistent_kernel(GemmParams p) {
    typedef Mma<__bf16>::Frag Frag;
    constexpr int K = KS * 32, LROW = K + 8;
    extern __shared__ __attribute__((aligned(16))) char smem_wide[];
    __bf16* sa = reinterpret_cast<__bf16*>(smem_wide);  // [64][LROW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int n_tiles = (p.N + 15) / 16;
    const int stride = gridDim.x * 8;
    // Weight fragments travel in HALF tiles (k-steps 0 .. KS/2-1 and KS/2 .. KS-1, 12 KiB per wave each): while one half is spent
    // on the MFMAs the other -- or the first half of the wave's NEXT tile -- is in flight, so the stream never stops between a
    // tile's arithmetic, its epilogue and the next tile's first load (round 4; before, all KS loads of a tile were issued, waited
    // for and spent, then the next tile started: two dependent rounds per wave).  The first tile's halves are requested BEFORE
    // the activations are staged: the weight stream (HBM) starts at once instead of after the 98 KB copy from L2 and its barrier.
    constexpr int KH = KS / 2;
    static_assert(KS % 2 == 0, "even k-step count");
    const int tile0 = blockIdx.x * 8 + wave;
    Frag fa[KH], fb[KH];
    auto wptr = [&](int tile) { return p.W + (int64_t)min(min(tile, n_tiles - 1) * 16 + frow, p.N - 1) * p.ldw_b + fq * 16; };
    {
        const char* wp0 = wptr(tile0);
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) fa[ks] = *reinterpret_cast<const Frag*>(wp0 + ks * 64);
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) fb[ks] = *reinterpret_cast<const Frag*>(wp0 + (KH + ks) * 64);
    }
    // activations -> LDS (rows past M repeat the last row; their results are dropped by the epilogue's row test)
    for (int c = tid; c < 64 * (K / 8); c += 512) {
        const int row = c / (K / 8), ch = c - row * (K / 8);
        const Frag v = *reinterpret_cast<const Frag*>(p.A + (int64_t)min(row, p.M - 1) * p.lda_b + ch * 16);
        *reinterpret_cast<Frag*>(sa + row * LROW + ch * 8) = v;
    }
    const bool vec = p.vec_ok != 0;
    int64_t coff_dev = p.c_offset;
    if (p.c_offset_dev) coff_dev += *p.c_offset_dev;
    __syncthreads();
    [[maybe_unused]] const float* gmask = nullptr;
    [[maybe_unused]] float gm[4] = {-1.0e30f, -1.0e30f, -1.0e30f, -1.0e30f}, gs[4] = {0.f, 0.f, 0.f, 0.f};
    [[maybe_unused]] int gi[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    if constexpr (GREEDY) gmask = (*p.g_pos + 1 == p.g_n_init) ? p.g_mask_first : p.g_mask_always;
    for (int tile = tile0; tile < n_tiles; tile += stride) {
        const int n0 = tile * 16;
        const bool more = tile + stride < n_tiles;
        const char* wpn = wptr(tile + stride);
        const EpiCol cc = epi_col(p, n0 + 4 * fq);
        f32x4 acc[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) {
            Frag fx[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) fx[j] = *reinterpret_cast<const Frag*>(sa + (16 * j + frow) * LROW + 32 * ks + 8 * fq);
#pragma unroll
            for (int j = 0; j < 4; ++j) Mma<__bf16>::run(fa[ks], fx[j], acc[j]);
            // keep the LDS reads of the later k-steps where they are: hoisted over the whole unrolled loop they need 384 registers
            if ((ks & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
        if (more) {  // first half of the next tile: in flight under this tile's second half and its epilogue
#pragma unroll
            for (int ks = 0; ks < KH; ++ks) fa[ks] = *reinterpret_cast<const Frag*>(wpn + ks * 64);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) {
            Frag fx[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) fx[j] = *reinterpret_cast<const Frag*>(sa + (16 * j + frow) * LROW + 32 * (KH + ks) + 8 * fq);
#pragma unroll
            for (int j = 0; j < 4; ++j) Mma<__bf16>::run(fb[ks], fx[j], acc[j]);
            if ((ks & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
        if (more) {
#pragma unroll
            for (int ks = 0; ks < KH; ++ks) fb[ks] = *reinterpret_cast<const Frag*>(wpn + (KH + ks) * 64);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (GREEDY) {
            // the lane holds rows 16 j + frow, columns n0 + 4 fq + 0..3
            const int nb = n0 + 4 * fq;
            float mk[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) mk[e] = (nb + e < p.N) ? gmask[nb + e] : -INFINITY;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[j][e] + mk[e];
                float best = gm[j];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (v[e] > best) {  // strict: the first (lowest) column of a maximum stays
                        best = v[e];
                        gi[j] = nb + e;
                    }
                gs[j] = gs[j] * __expf(gm[j] - best) + ((__expf(v[0] - best) + __expf(v[1] - best)) + (__expf(v[2] - best) + __expf(v[3] - best)));
                gm[j] = best;
            }
            if (!p.g_store) continue;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) epilogue4<OutT>(p, acc[j], epi_row(p, 16 * j + frow, coff_dev), cc, vec);
    }
    if constexpr (GREEDY) {
        // the four lanes of a row (fq = 0..3) -> one partial per (row, wave); every lane pair computes the same sums
        const int nw = gridDim.x * 8, wg = blockIdx.x * 8 + wave;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int off = 16; off <= 32; off <<= 1) {
                const float om = __shfl_xor(gm[j], off, 64), os = __shfl_xor(gs[j], off, 64);
                const int oi = __shfl_xor(gi[j], off, 64);
                const float M = fmaxf(gm[j], om);
                const float a = gs[j] * __expf(gm[j] - M), b = os * __expf(om - M);
                gs[j] = (lane & off) ? b + a : a + b;  // the same operand order in both lanes of a pair
                if (om > gm[j] || (om == gm[j] && oi < gi[j])) gi[j] = oi;
                gm[j] = M;
            }
            const int row = 16 * j + frow;
            if (fq == 0 && row < p.M) {
                float* pr = p.g_part + (int64_t)row * 3 * nw + wg;
                pr[0] = gm[j];
                pr[nw] = gs[j];
                reinterpret_cast<int*>(pr)[2 * nw] = gi[j];
            }
        }
    }
}

// Skinny GEMM with fp8 (OCP e4m3fn) weights and bf16 activations: the decode step streams every decoder matrix once per
// step, so halving the weight bytes halves that stream.  W [N, K] one byte per element, dequantised as code * w_scale[n]:
// the codes are widened to bf16 in registers (exact: e4m3 has 3 mantissa bits) and fed to the bf16 MFMA; the per-column
// scale multiplies the f32 accumulator before the usual epilogue.  A lane loads 16 weight bytes = 16 consecutive k of one
// row and spends them in two MFMAs; the activation fragments use the same k permutation (k = 64*s + 16*fq + 8*half + 0..7).
__device__ __forceinline__ void fp8x16_to_bf16(const uint4& w, bf16x8& lo, bf16x8& hi) {
    const unsigned int u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const auto a = __builtin_amdgcn_cvt_pk_f32_fp8((int)u[i], false);  // bytes 0, 1
        const auto b = __builtin_amdgcn_cvt_pk_f32_fp8((int)u[i], true);   // bytes 2, 3
        if (i < 2) {
            lo[4 * i] = (__bf16)a[0]; lo[4 * i + 1] = (__bf16)a[1]; lo[4 * i + 2] = (__bf16)b[0]; lo[4 * i + 3] = (__bf16)b[1];
        } else {
            hi[4 * (i - 2)] = (__bf16)a[0]; hi[4 * (i - 2) + 1] = (__bf16)a[1]; hi[4 * (i - 2) + 2] = (__bf16)b[0]; hi[4 * (i - 2) + 3] = (__bf16)b[1];
        }
    }
}

template <typename OutT, int MT, int NT, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_w8_kernel(GemmParams p) {
    __shared__ f32x4 red[NW][MT * NT][64];
    constexpr int UB = (NT == 1) ? 4 : 2;  // 64-k steps whose loads are in flight together
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * (16 * NT);
    const int m0 = blockIdx.y * (16 * MT);
    const int ksteps_all = p.K / 64;  // 64 k per step: 64 weight bytes, 128 activation bytes per row
    const int kz = blockIdx.z, S = p.k_slices;
    const int s_per = ksteps_all / S, s_rem = ksteps_all % S;
    const int s_beg = kz * s_per + min(kz, s_rem);
    const int ksteps = s_per + (kz < s_rem ? 1 : 0);
    const int per = ksteps / NW, rem = ksteps % NW;
    const int kb = s_beg + wave * per + min(wave, rem);
    const int ke = kb + per + (wave < rem ? 1 : 0);
    const char* wp[NT];
    const char* xp[MT];
#pragma unroll
    for (int i = 0; i < NT; ++i) wp[i] = p.W + (int64_t)min(n0 + 16 * i + frow, p.N - 1) * p.ldw_b + fq * 16;
#pragma unroll
    for (int j = 0; j < MT; ++j) xp[j] = p.A + (int64_t)min(m0 + 16 * j + frow, p.M - 1) * p.lda_b + fq * 32;
    const bool vec = p.vec_ok != 0;
    int64_t coff_dev = p.c_offset + (int64_t)kz * p.slab_stride;
    if (p.c_offset_dev) coff_dev += *p.c_offset_dev;
    if (kz > 0) p.bias = nullptr;
    EpiCol cols[NT];
    f32x4 wsc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        cols[i] = epi_col(p, n0 + 16 * i + 4 * fq);
#pragma unroll
        for (int e = 0; e < 4; ++e) wsc[i][e] = p.w_scale[min(n0 + 16 * i + 4 * fq + e, p.N - 1)];
    }
    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto step = [&](int ks0, int n) {
        uint4 fw[UB][NT];
        bf16x8 fx[UB][MT][2];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            if (u < n) {
                const int64_t ks = ks0 + u;
#pragma unroll
                for (int i = 0; i < NT; ++i) fw[u][i] = *reinterpret_cast<const uint4*>(wp[i] + ks * 64);
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    fx[u][j][0] = *reinterpret_cast<const bf16x8*>(xp[j] + ks * 128);
                    fx[u][j][1] = *reinterpret_cast<const bf16x8*>(xp[j] + ks * 128 + 16);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            if (u < n) {
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    bf16x8 wlo, whi;
                    fp8x16_to_bf16(fw[u][i], wlo, whi);
#pragma unroll
                    for (int j = 0; j < MT; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, fx[u][j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi, fx[u][j][1], acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
    };
    int ks = kb;
    for (; ks + UB <= ke; ks += UB) step(ks, UB);
    if (ks < ke) step(ks, ke - ks);
#pragma unroll
    for (int t = 0; t < NT * MT; ++t) red[wave][t][lane] = acc[t / MT][t % MT];
    __syncthreads();
#pragma unroll
    for (int t0 = 0; t0 < NT * MT; t0 += NW) {
        const int t = t0 + wave;
        if (t < NT * MT) {
            f32x4 s = red[0][t][lane];
#pragma unroll
            for (int w = 1; w < NW; ++w) s += red[w][t][lane];
            const int i = t / MT, j = t - i * MT;
            EpiCol cc = cols[0];
            f32x4 sc = wsc[0];
#pragma unroll
            for (int ii = 1; ii < NT; ++ii)
                if (i == ii) {
                    cc = cols[ii];
                    sc = wsc[ii];
                }
            s *= sc;
            epilogue4<OutT>(p, s, epi_row(p, m0 + 16 * j + frow, coff_dev), cc, vec);
        }
    }
}

template <typename OutT, int MT, int NT, int NW>
int launch_skinny_w8_cfg(const GemmParams& p, hipStream_t s) {
    dim3 grid((p.N + 16 * NT - 1) / (16 * NT), (p.M + 16 * MT - 1) / (16 * MT), p.k_slices);
    hipLaunchKernelGGL((gemm_skinny_w8_kernel<OutT, MT, NT, NW>), grid, dim3(NW * 64), 0, s, p);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

template <typename OutT, int MT>
int launch_skinny_w8_mt(const GemmParams& p, hipStream_t s) {
    const int ksteps = p.K / 64 / p.k_slices;
    if (p.N >= 8192 && MT == 4) return launch_skinny_w8_cfg<OutT, MT, 4, 4>(p, s);  // logits: 64 columns per workgroup
    if (p.N >= 8192) return launch_skinny_w8_cfg<OutT, MT, 2, 4>(p, s);
    if (ksteps >= 32) return launch_skinny_w8_cfg<OutT, MT, 1, 8>(p, s);
    return launch_skinny_w8_cfg<OutT, MT, 1, 4>(p, s);
}

template <typename OutT>
int launch_skinny_w8(const GemmParams& p, hipStream_t s) {
    if (p.M <= 16) return launch_skinny_w8_mt<OutT, 1>(p, s);
    if (p.M <= 32) return launch_skinny_w8_mt<OutT, 2>(p, s);
    // 32-row workgroups for the narrow projections, as in launch_skinny (WIPA_SKINNY_W8_ROWS=32 | 64; see there)
    static const int rows = [] { const char* e = getenv("WIPA_SKINNY_W8_ROWS"); return e ? atoi(e) : 64; }();
    if (rows == 32 && p.N < 8192 && p.M <= 128) return launch_skinny_w8_mt<OutT, 2>(p, s);
    return launch_skinny_w8_mt<OutT, 4>(p, s);
}

// Skinny GEMM with a LayerNorm prologue (decode step: mlp_ln folded into mlp1, one launch less per layer).  A workgroup
// owns 16 output columns for 16*MT rows; its 4 waves first LayerNorm those rows of the f32 residual stream (one row per
// wave at a time, the arithmetic of layernorm_kernel) into LDS as T, then split K exactly like gemm_skinny_kernel, the
// activation fragments coming from LDS instead of L2.
constexpr int LNP_NV = 5;  // K <= 1280
template <typename T, typename OutT, int MT>
__global__ __launch_bounds__(256) void gemm_skinny_ln_kernel(GemmParams p) {
    constexpr int NW = 4;
    extern __shared__ __attribute__((aligned(16))) char ln_smem[];
    __shared__ f32x4 red[NW][MT][64];
    typedef typename Mma<T>::Frag Frag;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int m0 = blockIdx.y * (16 * MT);
    const int pitch = p.K * (int)sizeof(T) + 16;
    // ---- prologue: rows m0 .. m0 + 16*MT - 1
    for (int r = wave; r < 16 * MT; r += NW) {
        const int m = min(m0 + r, p.M - 1);
        const float* xr = p.ln_x + (int64_t)m * p.ln_ldx;
        f32x4 v[LNP_NV];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < LNP_NV; ++i) {
            const int c = lane * 4 + 256 * i;
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < p.K) {
                v[i] = *reinterpret_cast<const f32x4*>(xr + c);
                sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
            }
        }
        const float mean = wave_reduce_sum(sum) / (float)p.K;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < LNP_NV; ++i) {
            const int c = lane * 4 + 256 * i;
            if (c < p.K) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dv = v[i][e] - mean;
                    sq += dv * dv;
                }
            }
        }
        const float rstd = rsqrtf(wave_reduce_sum(sq) / (float)p.K + p.ln_eps);
#pragma unroll
        for (int i = 0; i < LNP_NV; ++i) {
            const int c = lane * 4 + 256 * i;
            if (c < p.K) {
                const f32x4 ww = *reinterpret_cast<const f32x4*>(p.ln_w + c);
                const f32x4 bb = *reinterpret_cast<const f32x4*>(p.ln_b + c);
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * ww[e] + bb[e];
                T* dst = reinterpret_cast<T*>(ln_smem + r * pitch) + c;
                if constexpr (sizeof(T) == 4) {
                    *reinterpret_cast<f32x4*>(dst) = f32x4{o[0], o[1], o[2], o[3]};
                } else {
                    *reinterpret_cast<bf16x4*>(dst) = bf16x4{(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3]};
                }
            }
        }
    }
    // ---- main loop: the NW waves split K (as gemm_skinny_kernel with NT = 1, k_slices = 1)
    const int ksteps = p.K * (int)sizeof(T) / 64;
    const int per = ksteps / NW, rem = ksteps % NW;
    const int kb = wave * per + min(wave, rem);
    const int ke = kb + per + (wave < rem ? 1 : 0);
    const char* wp = p.W + (int64_t)min(n0 + frow, p.N - 1) * p.ldw_b + fq * 16;
    const bool vec = p.vec_ok != 0;
    int64_t coff_dev = p.c_offset;
    if (p.c_offset_dev) coff_dev += *p.c_offset_dev;
    const EpiCol col = epi_col(p, n0 + 4 * fq);
    f32x4 acc[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int UB = 6;
    Frag fw[UB];
    // the weight fragments do not depend on the prologue: issue the first batch before the barrier
    const int nfirst = min(UB, ke - kb);
#pragma unroll
    for (int u = 0; u < UB; ++u)
        if (u < nfirst) fw[u] = *reinterpret_cast<const Frag*>(wp + (int64_t)(kb + u) * 64);
    __syncthreads();
    const char* ap = ln_smem + frow * pitch + fq * 16;
    for (int ks = kb; ks < ke; ks += UB) {
        const int n = min(UB, ke - ks);
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            if (u < n) {
#pragma unroll
                for (int j = 0; j < MT; ++j)
                    Mma<T>::run(fw[u], *reinterpret_cast<const Frag*>(ap + (16 * j) * pitch + (ks + u) * 64), acc[j]);
            }
        }
        const int nn = min(UB, ke - (ks + UB));
#pragma unroll
        for (int u = 0; u < UB; ++u)
            if (u < nn) fw[u] = *reinterpret_cast<const Frag*>(wp + (int64_t)(ks + UB + u) * 64);
    }
#pragma unroll
    for (int j = 0; j < MT; ++j) red[wave][j][lane] = acc[j];
    __syncthreads();
    for (int j = wave; j < MT; j += NW) {
        f32x4 s = red[0][j][lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) s += red[w][j][lane];
        epilogue4<OutT>(p, s, epi_row(p, m0 + 16 * j + frow, coff_dev), col, vec);
    }
}

template <typename T, typename OutT, int MT>
int launch_skinny_ln_mt(const GemmParams& p, hipStream_t s) {
    const size_t lds = (size_t)16 * MT * (p.K * sizeof(T) + 16);  // <= 128 KiB (launch_skinny_ln); limit raised by init_attrs
    dim3 grid((p.N + 15) / 16, (p.M + 16 * MT - 1) / (16 * MT));
    hipLaunchKernelGGL((gemm_skinny_ln_kernel<T, OutT, MT>), grid, dim3(256), lds, s, p);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

template <typename T, typename OutT>
int launch_skinny_ln(const GemmParams& p, hipStream_t s) {
    // rows per workgroup: as many 16-row tiles (4, 2, 1) as fit into ~128 KiB of LDS next to the 16 KiB reduction buffer
    const size_t row = p.K * sizeof(T) + 16;
    int mt = 4;
    while (mt > 1 && (16 * mt * row > (size_t)128 * 1024 || 16 * (mt / 2) >= p.M)) mt >>= 1;
    if (mt == 4) return launch_skinny_ln_mt<T, OutT, 4>(p, s);
    if (mt == 2) return launch_skinny_ln_mt<T, OutT, 2>(p, s);
    return launch_skinny_ln_mt<T, OutT, 1>(p, s);
}

template <typename T, typename OutT, int MT, int NT, int NW>
int launch_skinny_cfg(const GemmParams& p, hipStream_t s) {
    dim3 grid((p.N + 16 * NT - 1) / (16 * NT), (p.M + 16 * MT - 1) / (16 * MT), p.k_slices);
    hipLaunchKernelGGL((gemm_skinny_kernel<T, OutT, MT, NT, NW>), grid, dim3(NW * 64), 0, s, p);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

template <typename OutT, int KS>
int launch_wide_persistent_ks(const GemmParams& p, hipStream_t s) {
    const size_t lds = (size_t)64 * (KS * 32 + 8) * 2;
    static const int wide_grid = [] { const char* e = getenv("WIPA_WIDE_GRID"); const int v = e ? atoi(e) : 256; return (v >= 32 && v <= 256) ? v : 256; }();  // A/B: 128 = half-chip launches
    hipLaunchKernelGGL((gemm_wide_persistent_kernel<OutT, KS>), dim3(wide_grid), dim3(512), lds, s, p);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}
template <typename OutT>
int launch_wide_persistent(const GemmParams& p, hipStream_t s) {
    switch (p.K) {
        case 384: return launch_wide_persistent_ks<OutT, 12>(p, s);
        case 512: return launch_wide_persistent_ks<OutT, 16>(p, s);
        case 768: return launch_wide_persistent_ks<OutT, 24>(p, s);
        default: return launch_wide_persistent_ks<OutT, 32>(p, s);
    }
}

template <typename T, typename OutT, int MT>
int launch_skinny_mt(const GemmParams& p, hipStream_t s) {
    const int ksteps = p.K * (int)sizeof(T) / 64 / p.k_slices;
    static const int wide_nt = [] { const char* e = getenv("WIPA_SKINNY_WIDE_NT"); return e ? atoi(e) : 4; }();  // A/B timing
    if constexpr (std::is_same<T, __bf16>::value && MT == 4) {
        // the persistent form for wide outputs (logits): WIPA_WIDE_PERSISTENT=0 keeps the 64-column workgroups for A/B runs
        static const bool persistent = [] { const char* e = getenv("WIPA_WIDE_PERSISTENT"); return !(e && atoi(e) == 0); }();
        if (persistent && p.N >= 8192 && p.k_slices == 1 && p.M <= 64 && !p.ln_x && !p.w_scale && !p.a_scale &&
            (p.K == 384 || p.K == 512 || p.K == 768 || p.K == 1024) && p.lda_b % 16 == 0 && p.ldw_b % 16 == 0)
            return launch_wide_persistent<OutT>(p, s);
    }
    if (p.N >= 8192 && MT == 4 && wide_nt == 4) return launch_skinny_cfg<T, OutT, MT, 4, 4>(p, s);  // logits: 64 columns per workgroup
    if (p.N >= 8192 && wide_nt == 1) return launch_skinny_cfg<T, OutT, MT, 1, 4>(p, s);
    if (p.N >= 8192) return launch_skinny_cfg<T, OutT, MT, 2, 4>(p, s);
    if (ksteps >= 64) return launch_skinny_cfg<T, OutT, MT, 1, 8>(p, s);
    return launch_skinny_cfg<T, OutT, MT, 1, 4>(p, s);
}

template <typename T, typename OutT>
int launch_skinny(const GemmParams& p, hipStream_t s) {
    if (p.M <= 16) return launch_skinny_mt<T, OutT, 1>(p, s);
    if (p.M <= 32) return launch_skinny_mt<T, OutT, 2>(p, s);
    // Narrow outputs (the decode-step projections, N < 8192): 32-row workgroups -- twice the workgroups of the 64-row form, each
    // fetching half of the activation matrix, which is 4/5 of a 64-row workgroup's bytes (98 KB against 24 KB of weights) and what
    // its single CU spends its time pulling: whisper-small, 64 rows: decode step 1.332 -> 1.311 ms, 78.1 -> 75.8 ms per pass with four
    // passes in flight; 16-row workgroups 77.3 ms.  Results are bit-identical (the same fragments in the same order).
    // WIPA_SKINNY_ROWS=64 | 16 for A/B runs.  The wide logits projection keeps 64 rows.
    static const int rows = [] { const char* e = getenv("WIPA_SKINNY_ROWS"); return e ? atoi(e) : 32; }();
    // Up to 128 rows only: every row group streams the weight slice again, and at 256 rows (whisper-medium, batch 256) eight
    // 32-row groups measured slower than four 64-row ones (decode step 8.66 vs 8.41 ms); 128 rows: 2.02 vs 2.06 ms.
    // (restricting the rule to launches that stay within one round of workgroups measured worse: 77.1 / 76.4 ms with caps of
    // 200 / 300 workgroups against 75.4 without)
    if (rows == 32 && p.N < 8192 && p.M <= 128) return launch_skinny_mt<T, OutT, 2>(p, s);
    if (rows == 16 && p.N < 8192 && p.M <= 128) return launch_skinny_mt<T, OutT, 1>(p, s);
    return launch_skinny_mt<T, OutT, 4>(p, s);
}

constexpr int SKINNY_MAX_M = 256;  // rows beyond 64 ride on grid.y: every 64-row group streams the (L2-resident) weight slice again
constexpr int SKINNY_STREAM_MAX_M = 1024;  // ... and up to here for wipa_gemm_desc.stream_weights

// ---------------------------------------------------------------------------------------
// Large-tile GEMM: 256x256 output tile, 8 waves (2 along M x 4 along N, 128x64 per wave =
// 8x4 MFMA tiles), K-step 128 bytes, LDS 2 x 64 KiB, one workgroup per CU.  Tiles are staged
// with LDS-DMA (global_load_lds, 16 bytes per lane, no VGPR round trip and no ds_write):
// the LDS image is lane-linear per 1-KiB piece (8 rows x 128 B), so the bank swizzle
// (chunk ^ ((row>>1)&7)) is applied to the per-lane SOURCE address and again on the read.
// Per K-step a wave issues 24 ds_read_b128 for 64 MFMAs (the 128x128 kernel: 16 for 32).
constexpr int LBM = 256, LBN = 256;
constexpr int LTILE = LBM * ROWB;  // 32 KiB per operand tile
constexpr int LSMEM = 4 * LTILE;   // 128 KiB

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <typename T, typename OutT, bool SPLIT>  // SPLIT: f32 products as three bf16 MFMA terms
__global__ __launch_bounds__(512) void gemm_nt256_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][W tile | A tile]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int nblocks = p.tiles_m * p.tiles_n;
    int id;
    {
        const int bid = blockIdx.x;
        const int q = nblocks >> 3, r = nblocks & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int group_size = GROUP_M * p.tiles_n;
    const int group = id / group_size;
    const int first_m = group * GROUP_M;
    const int gm = min(p.tiles_m - first_m, GROUP_M);
    const int in_group = id - group * group_size;
    const int tile_m = first_m + in_group % gm;
    const int tile_n = in_group / gm;
    const int m0 = tile_m * LBM, n0 = tile_n * LBN;

    // staging: pass i covers rows 64*i + 8*wave + (lane>>3); lane's physical chunk is lane&7.  The DMA is the BUFFER
    // form (buffer_load_dwordx4 ... lds): one resource per operand tile (SGPRs), a per-lane 32-bit byte offset computed
    // once, and the K-step as the scalar offset -- no 64-bit VALU address arithmetic per piece in the loop.
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)n0 * p.ldw_b), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)m0 * p.lda_b), 0, 0x7fffffff, 0x00020000);
    int oW[4], oA[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 64 * i + 8 * wave + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        oW[i] = (min(n0 + row, p.N - 1) - n0) * (int)p.ldw_b + c * 16;
        oA[i] = (min(m0 + row, p.M - 1) - m0) * (int)p.lda_b + c * 16;
    }
    auto stage = [&](int kt, int buf) {
        const int kb = kt * ROWB;
        char* base = smem + buf * (2 * LTILE) + wave * (8 * ROWB);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_ptr_t)(base + i * 64 * ROWB), 16, oW[i], kb, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)(base + LTILE + i * 64 * ROWB), 16, oA[i], kb, 0, 0);
        }
    };

    f32x4 acc[4][8];  // [n tile i][m tile j]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;
    const int fsw = (lane >> 1) & 7;
    const int fq = lane >> 4;
    const int nk = p.K * (int)sizeof(T) / ROWB;

    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        const char* wb = smem + (kt & 1) * (2 * LTILE) + (wn * 64 + frow) * ROWB;
        const char* ab = smem + (kt & 1) * (2 * LTILE) + LTILE + (wm * 128 + frow) * ROWB;
        bool done = false;
        if constexpr (sizeof(T) == 4 && SPLIT) {
            {
                const int c0 = (fq ^ fsw) << 4, c1 = ((fq + 4) ^ fsw) << 4;
                bf16x8 wh[4], wl[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    split_bf16x2(*reinterpret_cast<const f32x4*>(wb + i * 16 * ROWB + c0),
                                 *reinterpret_cast<const f32x4*>(wb + i * 16 * ROWB + c1), wh[i], wl[i]);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    bf16x8 xh, xl;
                    split_bf16x2(*reinterpret_cast<const f32x4*>(ab + j * 16 * ROWB + c0),
                                 *reinterpret_cast<const f32x4*>(ab + j * 16 * ROWB + c1), xh, xl);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[i], xl, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[i], xh, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[i], xh, acc[i][j], 0, 0, 0);
                    }
                }
                done = true;
            }
        }
        if (!done) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int coff = ((fq + 4 * kk) ^ fsw) << 4;
                typename Mma<T>::Frag fw[4], fx[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) fw[i] = *reinterpret_cast<const typename Mma<T>::Frag*>(wb + i * 16 * ROWB + coff);
#pragma unroll
                for (int j = 0; j < 8; ++j) fx[j] = *reinterpret_cast<const typename Mma<T>::Frag*>(ab + j * 16 * ROWB + coff);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) Mma<T>::run(fw[i], fx[j], acc[i][j]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    const bool vec = p.vec_ok != 0;
    int64_t coff_dev = p.c_offset;
    if (p.c_offset_dev) coff_dev += *p.c_offset_dev;
    if (p.stage_ok) {  // the last barrier of the K loop has retired every LDS read: reuse it as per-wave scratch
        epilogue_staged<OutT, 8>(p, acc, smem + wave * 4096, m0 + wm * 128, n0 + wn * 64, coff_dev, lane);
        return;
    }
    EpiCol cols[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cols[i] = epi_col(p, n0 + wn * 64 + 16 * i + 4 * fq);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const EpiRow row = epi_row(p, m0 + wm * 128 + 16 * j + frow, coff_dev);
#pragma unroll
        for (int i = 0; i < 4; ++i) epilogue4<OutT>(p, acc[i][j], row, cols[i], vec);
    }
}

// ---------------------------------------------------------------------------------------
// fp8 x fp8 on the block-scaled matrix instruction (BASELINE.json configs[4]: "fp8-weight inference (CDNA4 fp8 MFMA)").
// Both operands are OCP e4m3fn codes with one power-of-two scale per row: A [M, K] bytes with a_scale[M] (activations
// quantised by wipa_layernorm_fp8 / wipa_rowquant_fp8), W [N, K] bytes with w_scale[N] (Whisper.quantize_weights).  Same
// tiling, LDS image and LDS-DMA staging as gemm_nt256_kernel -- the staging moves bytes -- but a 128-byte K-step is now 128
// elements, i.e. ONE v_mfma_scale_f32_16x16x128_f8f6f4 per 16 x 16 output tile (scale operands 0: the unscaled form, which
// runs at twice the bf16 rate on gfx950): a lane (row = lane & 15, k-group = lane >> 4) feeds the 32 bytes k = 32 (lane >> 4)
// ... + 31 of its row, two ds_read_b128.  Half the staged bytes per FLOP of the bf16 kernel, which is what these tile GEMMs
// are bound by.  The per-row scales are powers of two, so acc * a_scale[m] * w_scale[n] is exact; the rest of the epilogue
// (bias, column scale, GELU, f32 residual, remaps) is the shared one.
typedef int v8i32 __attribute__((ext_vector_type(8)));
typedef int v4i32 __attribute__((ext_vector_type(4)));

template <typename OutT>
__global__ __launch_bounds__(512) void gemm_fp8_256_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][W tile | A tile]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nblocks = p.tiles_m * p.tiles_n;
    int id;
    {
        const int bid = blockIdx.x;
        const int q = nblocks >> 3, r = nblocks & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int group_size = GROUP_M * p.tiles_n;
    const int group = id / group_size;
    const int first_m = group * GROUP_M;
    const int gm = min(p.tiles_m - first_m, GROUP_M);
    const int in_group = id - group * group_size;
    const int tile_m = first_m + in_group % gm;
    const int tile_n = in_group / gm;
    const int m0 = tile_m * LBM, n0 = tile_n * LBN;

    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)n0 * p.ldw_b), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)m0 * p.lda_b), 0, 0x7fffffff, 0x00020000);
    int oW[4], oA[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 64 * i + 8 * wave + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        oW[i] = (min(n0 + row, p.N - 1) - n0) * (int)p.ldw_b + c * 16;
        oA[i] = (min(m0 + row, p.M - 1) - m0) * (int)p.lda_b + c * 16;
    }
    auto stage = [&](int kt, int buf) {
        const int kb = kt * ROWB;
        char* base = smem + buf * (2 * LTILE) + wave * (8 * ROWB);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_ptr_t)(base + i * 64 * ROWB), 16, oW[i], kb, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)(base + LTILE + i * 64 * ROWB), 16, oA[i], kb, 0, 0);
        }
    };
    f32x4 acc[4][8];  // [n tile i][m tile j]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15;
    const int fsw = (lane >> 1) & 7;
    const int fq = lane >> 4;
    const int nk = p.K / ROWB;  // one byte per element
    const int c0 = ((2 * fq) ^ fsw) << 4, c1 = ((2 * fq + 1) ^ fsw) << 4;
    auto frag = [&](const char* row) {
        const v4i32 lo = *reinterpret_cast<const v4i32*>(row + c0);
        const v4i32 hi = *reinterpret_cast<const v4i32*>(row + c1);
        return v8i32{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        const char* wb = smem + (kt & 1) * (2 * LTILE) + (wn * 64 + frow) * ROWB;
        const char* ab = smem + (kt & 1) * (2 * LTILE) + LTILE + (wm * 128 + frow) * ROWB;
        v8i32 fw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) fw[i] = frag(wb + i * 16 * ROWB);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const v8i32 fx = frag(ab + j * 16 * ROWB);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fw[i], fx, acc[i][j], 0, 0, 0, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // dequantisation: acc[i][j][e] is C[m = m0 + wm*128 + 16 j + frow][n = n0 + wn*64 + 16 i + 4 fq + e]
    {
        f32x4 ws[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + 16 * i + 4 * fq;
#pragma unroll
            for (int e = 0; e < 4; ++e) ws[i][e] = p.w_scale[min(n + e, p.N - 1)];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float as = p.a_scale[min(m0 + wm * 128 + 16 * j + frow, p.M - 1)];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] *= ws[i] * as;
        }
    }
    const bool vec = p.vec_ok != 0;
    int64_t coff_dev = p.c_offset;
    if (p.c_offset_dev) coff_dev += *p.c_offset_dev;
    if (p.stage_ok) {
        epilogue_staged<OutT, 8>(p, acc, smem + wave * 4096, m0 + wm * 128, n0 + wn * 64, coff_dev, lane);
        return;
    }
    EpiCol cols[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cols[i] = epi_col(p, n0 + wn * 64 + 16 * i + 4 * fq);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const EpiRow row = epi_row(p, m0 + wm * 128 + 16 * j + frow, coff_dev);
#pragma unroll
        for (int i = 0; i < 4; ++i) epilogue4<OutT>(p, acc[i][j], row, cols[i], vec);
    }
}

// ---------------------------------------------------------------------------------------
// 256 x 256 tile, bf16, phase-interleaved (WIPA_GEMM_TILE=2568 selects it for A/B runs).  Same LDS image as gemm_nt256_kernel
// (two buffers of [W tile | A tile], 128-byte rows, swizzled 16-byte chunks, filled by LDS-DMA) and the same wave grid
// (2 x 4, 128 x 64 per wave), but the K loop never drains the DMA queue:
//   * a K-tile (64 elements) is FOUR phases, one C quadrant (64 x 32 of the wave's 128 x 64) each: 16 MFMAs between two raw
//     s_barriers, preceded by the LDS fragment reads of that quadrant and ONE half-tile (16 KiB: 2 DMA pieces per wave) of
//     prefetch;
//   * the tile's halves are cut ACROSS the waves -- A half x = rows wm*128 + 64x .. +63 of both wave rows, W half y = columns
//     wn*64 + 32y .. +31 of all four wave columns -- so the four halves are first needed in phases 1, 1, 2, 3 and last read in
//     phases 1, 1, 2, 3 (A fragments are re-read per half, W half 0 stays in registers for phase 4): each half is re-staged
//     two phases after its last read, for the tile TWO ahead, and is read five phases after its issue;
//   * every phase ends its load segment with s_waitcnt vmcnt(8): the four youngest half-tiles stay in flight across the
//     barriers, the one issued four phases ago has landed, and after the barrier the next phase may read it;
//   * the two wave rows run half a phase apart (wave row 1 takes one barrier more before the loop, wave row 0 one more after
//     it), so on every SIMD one wave is in its MFMA segment while the other issues LDS reads and DMA.
// K must be a multiple of 128 elements (an even number of K-tiles: the loop body is two tiles so that buffer indices are
// compile-time constants).
template <typename OutT>
__global__ __launch_bounds__(512) void gemm_nt256p_kernel(GemmParams p) {
    typedef __bf16 T;
    typedef Mma<T>::Frag Frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][W tile | A tile]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nblocks = p.tiles_m * p.tiles_n;
    int id;
    {
        const int bid = blockIdx.x;
        const int q = nblocks >> 3, r = nblocks & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int group_size = GROUP_M * p.tiles_n;
    const int group = id / group_size;
    const int first_m = group * GROUP_M;
    const int gm = min(p.tiles_m - first_m, GROUP_M);
    const int in_group = id - group * group_size;
    const int tile_m = first_m + in_group % gm;
    const int tile_n = in_group / gm;
    const int m0 = tile_m * LBM, n0 = tile_n * LBN;

    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)n0 * p.ldw_b), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)m0 * p.lda_b), 0, 0x7fffffff, 0x00020000);
    // DMA pieces (8 rows x 128 B per wave instruction).  A half x, piece s: rows 128 s + 64 x + 8 wave + (lane >> 3).
    // W half y, piece s: rows 128 s + 64 (wave >> 2) + 32 y + 8 (wave & 3) + (lane >> 3).
    int oA[2][2], oW[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            const int ra = 128 * sp + 64 * h + 8 * wave + (lane >> 3);
            const int rw = 128 * sp + 64 * (wave >> 2) + 32 * h + 8 * (wave & 3) + (lane >> 3);
            oA[h][sp] = (min(m0 + ra, p.M - 1) - m0) * (int)p.lda_b + (((lane & 7) ^ ((ra >> 1) & 7)) << 4);
            oW[h][sp] = (min(n0 + rw, p.N - 1) - n0) * (int)p.ldw_b + (((lane & 7) ^ ((rw >> 1) & 7)) << 4);
        }
    const int rowA = 8 * wave, rowW = 64 * (wave >> 2) + 8 * (wave & 3);  // wave-uniform first rows of the pieces (+ 128 s + 64 x / 32 y)
    auto stage_a = [&](int h, int kt, int buf) {
        char* base = smem + buf * (2 * LTILE) + LTILE + (64 * h + rowA) * ROWB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)(base), 16, oA[h][0], kt * ROWB, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)(base + 128 * ROWB), 16, oA[h][1], kt * ROWB, 0, 0);
    };
    auto stage_w = [&](int h, int kt, int buf) {
        char* base = smem + buf * (2 * LTILE) + (32 * h + rowW) * ROWB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_ptr_t)(base), 16, oW[h][0], kt * ROWB, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_ptr_t)(base + 128 * ROWB), 16, oW[h][1], kt * ROWB, 0, 0);
    };

    f32x4 acc[4][8];  // [n tile i][m tile j]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15;
    const int fsw = (lane >> 1) & 7;
    const int fq = lane >> 4;
    const int c0 = (fq ^ fsw) << 4, c1 = ((fq + 4) ^ fsw) << 4;
    const int nk = p.K * (int)sizeof(T) / ROWB;  // even (dispatcher)
    const char* wfrag = smem + (wn * 64 + frow) * ROWB;
    const char* afrag = smem + LTILE + (wm * 128 + frow) * ROWB;
    Frag fa[4][2], fb0[2][2], fb1[2][2];
    auto read_a = [&](int x, int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const char* r = afrag + buf * (2 * LTILE) + (64 * x + 16 * j) * ROWB;
            fa[j][0] = *reinterpret_cast<const Frag*>(r + c0);
            fa[j][1] = *reinterpret_cast<const Frag*>(r + c1);
        }
    };
    auto read_b = [&](Frag (&fb)[2][2], int y, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const char* r = wfrag + buf * (2 * LTILE) + (32 * y + 16 * i) * ROWB;
            fb[i][0] = *reinterpret_cast<const Frag*>(r + c0);
            fb[i][1] = *reinterpret_cast<const Frag*>(r + c1);
        }
    };
#define WIPA_QUADRANT(X, Y, FB)                                                                  \
    do {                                                                                         \
        __builtin_amdgcn_s_barrier();                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                       \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        __builtin_amdgcn_s_setprio(1);                                                           \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                         \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                        \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                    \
                    Mma<T>::run(FB[i][kk], fa[j][kk], acc[2 * (Y) + i][4 * (X) + j]);            \
        __builtin_amdgcn_s_setprio(0);                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        __builtin_amdgcn_s_barrier();                                                            \
    } while (0)
#define WIPA_VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
    // one K-tile kt in buffer BUF.  MODE 0: steady state (all four prefetches, four half-tiles stay in flight);
    // MODE 1: tile nk-2 (prefetches of tile nk-1 only); MODE 2: tile nk-1 (nothing left to prefetch)
#define WIPA_KTILE(BUF, MODE, KT)                                                                \
    do {                                                                                         \
        read_b(fb0, 0, BUF);                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        read_a(0, BUF);                                                                          \
        if ((MODE) < 2) stage_w(1, (KT) + 1, (BUF) ^ 1);                                         \
        if ((MODE) < 2) WIPA_VMCNT(8); else WIPA_VMCNT(2);                                       \
        WIPA_QUADRANT(0, 0, fb0);                                                                \
        read_b(fb1, 1, BUF);                                                                     \
        if ((MODE) < 2) stage_a(1, (KT) + 1, (BUF) ^ 1);                                         \
        if ((MODE) < 2) WIPA_VMCNT(8); else WIPA_VMCNT(0);                                       \
        WIPA_QUADRANT(0, 1, fb1);                                                                \
        read_a(1, BUF);                                                                          \
        if ((MODE) == 0) stage_a(0, (KT) + 2, BUF);                                              \
        if ((MODE) == 0) WIPA_VMCNT(8); else if ((MODE) == 1) WIPA_VMCNT(6);                     \
        WIPA_QUADRANT(1, 1, fb1);                                                                \
        if ((MODE) == 0) stage_w(0, (KT) + 2, BUF);                                              \
        if ((MODE) == 0) WIPA_VMCNT(8); else if ((MODE) == 1) WIPA_VMCNT(4);                     \
        WIPA_QUADRANT(1, 0, fb0);                                                                \
    } while (0)

    // prologue: the six half-tiles the steady state would have issued before tile 0, in its order
    stage_a(0, 0, 0);
    stage_w(0, 0, 0);
    stage_w(1, 0, 0);
    stage_a(1, 0, 0);
    stage_a(0, 1, 1);
    stage_w(0, 1, 1);
    WIPA_VMCNT(8);  // A half 0 and W half 0 of tile 0 have landed
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();  // wave row 1 runs one barrier behind wave row 0
    int kt = 0;
    for (; kt + 3 < nk; kt += 2) {
        WIPA_KTILE(0, 0, kt);
        WIPA_KTILE(1, 0, kt + 1);
    }
    WIPA_KTILE(0, 1, kt);
    WIPA_KTILE(1, 2, kt + 1);
    if (wm == 0) __builtin_amdgcn_s_barrier();
#undef WIPA_KTILE
#undef WIPA_QUADRANT
#undef WIPA_VMCNT
    __syncthreads();  // every LDS read has retired: the epilogue reuses the buffers as per-wave scratch

    const bool vec = p.vec_ok != 0;
    int64_t coff_dev = p.c_offset;
    if (p.c_offset_dev) coff_dev += *p.c_offset_dev;
    if (p.stage_ok) {
        epilogue_staged<OutT, 8>(p, acc, smem + wave * 4096, m0 + wm * 128, n0 + wn * 64, coff_dev, lane);
        return;
    }
    EpiCol cols[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cols[i] = epi_col(p, n0 + wn * 64 + 16 * i + 4 * fq);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const EpiRow row = epi_row(p, m0 + wm * 128 + 16 * j + frow, coff_dev);
#pragma unroll
        for (int i = 0; i < 4; ++i) epilogue4<OutT>(p, acc[i][j], row, cols[i], vec);
    }
}

// ---------------------------------------------------------------------------------------
// 384 (M) x 256 (N) tile (WIPA_GEMM_TILE=384 forces it, =256 forbids it), 8 waves (2 x 4), 192 x 64 per wave = 4 x 12 MFMA tiles
// (192 accumulator registers).  Stages (256 + 384) x 128 B = 80 KiB per K-step: 1/153.6 byte per FLOP instead of 1/128, and
// N = 768 gives 750 tiles = 2.93 rounds instead of 4.39.  LDS 2 x 80 KiB = all of it.
// WN = 2: the same 384 rows against 128 columns, waves 4 x 2, 96 x 64 per wave (4 x 6 MFMA tiles).  For grids the wide tile
// quantises badly on 256 CUs: the float32 encoder of a 32-clip fine-tune batch (48 000 x 768) is 375 wide tiles = 1.46 rounds
// but 750 narrow ones = 2.93, and the f32 MFMA is slow enough that the extra staged bytes per FLOP do not show.
constexpr int XBM = 384;
constexpr int XA_TILE = XBM * ROWB;  // 48 KiB
template <int WN>
struct X384 {
    static constexpr int BN = 64 * WN;            // 256 | 128 columns
    static constexpr int MT = 12 * WN / 4;        // 16-row slices per wave: 12 | 6
    static constexpr int W_TILE = BN * ROWB;      // 32 | 16 KiB
    static constexpr int STAGE = W_TILE + XA_TILE;  // 80 | 64 KiB
    static constexpr int SMEM = 2 * STAGE;          // 160 | 128 KiB
};

template <typename T, typename OutT, bool ACT, bool SPLIT, int WN>  // SPLIT (f32 inputs only): products as three bf16 MFMA terms
__device__ __forceinline__ void gemm_nt384_body(GemmParams p) {
    typedef X384<WN> X;
    constexpr int MT = X::MT, XW_TILE = X::W_TILE, XSTAGE = X::STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][W tile BN rows | A tile 384 rows]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int nblocks = p.tiles_m * p.tiles_n;
    int id;
    {
        const int bid = blockIdx.x;
        const int q = nblocks >> 3, r = nblocks & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int group_size = GROUP_M * p.tiles_n;
    const int group = id / group_size;
    const int first_m = group * GROUP_M;
    const int gm = min(p.tiles_m - first_m, GROUP_M);
    const int in_group = id - group * group_size;
    const int tile_m = first_m + in_group % gm;
    const int tile_n = in_group / gm;
    const int m0 = tile_m * XBM, n0 = tile_n * X::BN;

    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)n0 * p.ldw_b), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)m0 * p.lda_b), 0, 0x7fffffff, 0x00020000);
    int oW[WN], oA[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int row = 64 * i + 8 * wave + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        if (i < WN) oW[i] = (min(n0 + row, p.N - 1) - n0) * (int)p.ldw_b + c * 16;
        oA[i] = (min(m0 + row, p.M - 1) - m0) * (int)p.lda_b + c * 16;
    }
    auto stage = [&](int kt, int buf) {
        const int kb = kt * ROWB;
        char* base = smem + buf * XSTAGE + wave * (8 * ROWB);
#pragma unroll
        for (int i = 0; i < WN; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_ptr_t)(base + i * 64 * ROWB), 16, oW[i], kb, 0, 0);
#pragma unroll
        for (int i = 0; i < 6; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)(base + XW_TILE + i * 64 * ROWB), 16, oA[i], kb, 0, 0);
    };
    f32x4 acc[4][MT];  // [n tile i][m tile j]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15;
    const int fsw = (lane >> 1) & 7;
    const int fq = lane >> 4;
    const int nk = p.K * (int)sizeof(T) / ROWB;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if constexpr (SPLIT && sizeof(T) == 4) {
            // The stage is converted ONCE, in place, by all 512 threads: every 16-byte chunk of four floats becomes
            // [h0 h1 h2 h3 | l0 l1 l2 l3] (bf16), so the eight waves no longer repeat the hi/lo split of the rows they share
            // and a fragment is put together from the dwords of two chunks without any arithmetic.
            char* st = smem + (kt & 1) * XSTAGE;
#pragma unroll
            for (int i = 0; i < XSTAGE / 16 / 512; ++i) {
                f32x4* cp = reinterpret_cast<f32x4*>(st + (tid + 512 * i) * 16);
                const f32x4 x = *cp;
                bf16x8 hl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const __bf16 hbits = (__bf16)x[e];
                    hl[e] = hbits;
                    hl[4 + e] = (__bf16)(x[e] - (float)hbits);
                }
                *reinterpret_cast<bf16x8*>(cp) = hl;
            }
            __syncthreads();
        }
        if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        const char* wb = smem + (kt & 1) * XSTAGE + (wn * 64 + frow) * ROWB;
        const char* ab = smem + (kt & 1) * XSTAGE + XW_TILE + (wm * (16 * MT) + frow) * ROWB;
        if constexpr (SPLIT && sizeof(T) == 4) {
            // one bf16 MFMA K-step per stage: the lane's floats 4fq..4fq+3 and 16+4fq..19+4fq of every row
            const int c0 = (fq ^ fsw) << 4, c1 = ((fq + 4) ^ fsw) << 4;
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            auto frag = [&](const char* row, bf16x8& hi, bf16x8& lo) {
                const u32x4 a = *reinterpret_cast<const u32x4*>(row + c0), b = *reinterpret_cast<const u32x4*>(row + c1);
                hi = __builtin_bit_cast(bf16x8, u32x4{a[0], a[1], b[0], b[1]});
                lo = __builtin_bit_cast(bf16x8, u32x4{a[2], a[3], b[2], b[3]});
            };
            bf16x8 wh[4], wl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) frag(wb + i * 16 * ROWB, wh[i], wl[i]);
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                bf16x8 xh, xl;
                frag(ab + j * 16 * ROWB, xh, xl);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[i], xl, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[i], xh, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[i], xh, acc[i][j], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int coff = ((fq + 4 * kk) ^ fsw) << 4;
            typename Mma<T>::Frag fw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fw[i] = *reinterpret_cast<const typename Mma<T>::Frag*>(wb + i * 16 * ROWB + coff);
            constexpr int JG = MT == 12 ? 4 : MT;  // A fragments per group: 192 accumulators leave no room for twelve at once
#pragma unroll
            for (int jg = 0; jg < MT / JG; ++jg) {
                typename Mma<T>::Frag fx[JG];
#pragma unroll
                for (int j = 0; j < JG; ++j) fx[j] = *reinterpret_cast<const typename Mma<T>::Frag*>(ab + (JG * jg + j) * 16 * ROWB + coff);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < JG; ++j) Mma<T>::run(fw[i], fx[j], acc[i][JG * jg + j]);
            }
        }
        }  // exact path
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const bool vec = p.vec_ok != 0;
    int64_t coff_dev = p.c_offset;
    if (p.c_offset_dev) coff_dev += *p.c_offset_dev;
    if (p.stage_ok) {
        if constexpr (ACT) {
            // GELU on the MFMA layout, in place and BEFORE the transposing epilogue (which is compiled without its GELU path:
            // 192 accumulators + those temporaries would spill).  A lane holds columns n0 + 64 wn + 16 i + 4 fq + (0..3).
            // The dispatcher sends an activation here only with a plain column bias and no column scale.
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n0 + wn * 64 + 16 * i + 4 * fq;
                float b4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) b4[e] = (p.bias && n + e < p.N) ? p.bias[n + e] : 0.f;
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    const f32x2 g0 = gelu_erf2(f32x2{acc[i][j][0] + b4[0], acc[i][j][1] + b4[1]});
                    const f32x2 g1 = gelu_erf2(f32x2{acc[i][j][2] + b4[2], acc[i][j][3] + b4[3]});
                    acc[i][j] = f32x4{g0.x, g0.y, g1.x, g1.y};
                    __builtin_amdgcn_sched_barrier(0);  // one tile at a time: the scheduler would otherwise overlap all 48 and spill
                }
            }
            p.bias = nullptr;
        }
        epilogue_staged<OutT, MT, 0, MT, false, !ACT>(p, acc, smem + wave * 4096, m0 + wm * (16 * MT), n0 + wn * 64, coff_dev, lane);
        return;
    }
    EpiCol cols[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cols[i] = epi_col(p, n0 + wn * 64 + 16 * i + 4 * fq);
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const EpiRow row = epi_row(p, m0 + wm * (16 * MT) + 16 * j + frow, coff_dev);
#pragma unroll
        for (int i = 0; i < 4; ++i) epilogue4<OutT>(p, acc[i][j], row, cols[i], vec);
    }
}

template <typename T, typename OutT, bool ACT, bool SPLIT>
__global__ __launch_bounds__(512) void gemm_nt384_kernel(GemmParams p) {
    gemm_nt384_body<T, OutT, ACT, SPLIT, 4>(p);
}
template <typename T, typename OutT, bool ACT, bool SPLIT>
__global__ __launch_bounds__(512) void gemm_nt384n_kernel(GemmParams p) {  // 384 x 128
    gemm_nt384_body<T, OutT, ACT, SPLIT, 2>(p);
}

// fp8 x fp8 on the 384 x 256 tile (round 4): gemm_fp8_256_kernel's arithmetic -- one
// v_mfma_scale_f32_16x16x128_f8f6f4 per 16 x 16 output tile and 128-byte K-step, a fragment = two ds_read_b128 -- in
// gemm_nt384_body's geometry (8 waves 2 x 4, 192 x 64 per wave, 192 accumulators, (256 + 384) x 128 B = 80 KiB per stage):
// 1/307 staged byte per FLOP instead of 1/256, and N = 1280 / 5120 grids quantise as for the bf16 kernel.  ACT: bias + GELU on
// the MFMA layout before the transposing epilogue (compiled without its GELU path, like the bf16 instantiation).
template <typename OutT, bool ACT>
__global__ __launch_bounds__(512) void gemm_fp8_384_kernel(GemmParams p) {
    typedef X384<4> X;
    constexpr int MT = X::MT, XW_TILE = X::W_TILE, XSTAGE = X::STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][W tile 256 rows | A tile 384 rows]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nblocks = p.tiles_m * p.tiles_n;
    int id;
    {
        const int bid = blockIdx.x;
        const int q = nblocks >> 3, r = nblocks & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int group_size = GROUP_M * p.tiles_n;
    const int group = id / group_size;
    const int first_m = group * GROUP_M;
    const int gm = min(p.tiles_m - first_m, GROUP_M);
    const int in_group = id - group * group_size;
    const int tile_m = first_m + in_group % gm;
    const int tile_n = in_group / gm;
    const int m0 = tile_m * XBM, n0 = tile_n * X::BN;

    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)n0 * p.ldw_b), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)m0 * p.lda_b), 0, 0x7fffffff, 0x00020000);
    int oW[4], oA[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int row = 64 * i + 8 * wave + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        if (i < 4) oW[i] = (min(n0 + row, p.N - 1) - n0) * (int)p.ldw_b + c * 16;
        oA[i] = (min(m0 + row, p.M - 1) - m0) * (int)p.lda_b + c * 16;
    }
    auto stage = [&](int kt, int buf) {
        const int kb = kt * ROWB;
        char* base = smem + buf * XSTAGE + wave * (8 * ROWB);
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_ptr_t)(base + i * 64 * ROWB), 16, oW[i], kb, 0, 0);
#pragma unroll
        for (int i = 0; i < 6; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)(base + XW_TILE + i * 64 * ROWB), 16, oA[i], kb, 0, 0);
    };
    f32x4 acc[4][MT];  // [n tile i][m tile j]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15;
    const int fsw = (lane >> 1) & 7;
    const int fq = lane >> 4;
    const int nk = p.K / ROWB;  // one byte per element
    const int c0 = ((2 * fq) ^ fsw) << 4, c1 = ((2 * fq + 1) ^ fsw) << 4;
    auto frag = [&](const char* row) {
        const v4i32 lo = *reinterpret_cast<const v4i32*>(row + c0);
        const v4i32 hi = *reinterpret_cast<const v4i32*>(row + c1);
        return v8i32{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        const char* wb = smem + (kt & 1) * XSTAGE + (wn * 64 + frow) * ROWB;
        const char* ab = smem + (kt & 1) * XSTAGE + XW_TILE + (wm * (16 * MT) + frow) * ROWB;
        v8i32 fw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) fw[i] = frag(wb + i * 16 * ROWB);
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const v8i32 fx = frag(ab + j * 16 * ROWB);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fw[i], fx, acc[i][j], 0, 0, 0, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // dequantisation: acc[i][j][e] is C[m = m0 + wm*192 + 16 j + frow][n = n0 + wn*64 + 16 i + 4 fq + e]
    {
        f32x4 ws[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + 16 * i + 4 * fq;
#pragma unroll
            for (int e = 0; e < 4; ++e) ws[i][e] = p.w_scale[min(n + e, p.N - 1)];
        }
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const float as = p.a_scale[min(m0 + wm * (16 * MT) + 16 * j + frow, p.M - 1)];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] *= ws[i] * as;
        }
    }
    int64_t coff_dev = p.c_offset;
    if (p.c_offset_dev) coff_dev += *p.c_offset_dev;
    // the dispatcher sends only stage_ok outputs here (and an activation only with a plain column bias and no column scale)
    if constexpr (ACT) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + 16 * i + 4 * fq;
            float b4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) b4[e] = (p.bias && n + e < p.N) ? p.bias[n + e] : 0.f;
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const f32x2 g0 = gelu_erf2(f32x2{acc[i][j][0] + b4[0], acc[i][j][1] + b4[1]});
                const f32x2 g1 = gelu_erf2(f32x2{acc[i][j][2] + b4[2], acc[i][j][3] + b4[3]});
                acc[i][j] = f32x4{g0.x, g0.y, g1.x, g1.y};
                __builtin_amdgcn_sched_barrier(0);  // one tile at a time (see gemm_nt384_body)
            }
        }
        p.bias = nullptr;
    }
    epilogue_staged<OutT, MT, 0, MT, false, !ACT>(p, acc, smem + wave * 4096, m0 + wm * (16 * MT), n0 + wn * 64, coff_dev, lane);
}

template <typename T, typename OutT>
int launch384(GemmParams p, hipStream_t s) {
    typedef X384<4> X;
    p.tiles_m = (p.M + XBM - 1) / XBM;
    p.tiles_n = (p.N + X::BN - 1) / X::BN;
    const dim3 grid(p.tiles_m * p.tiles_n);
    const bool act = p.act == 1 && p.stage_ok;
    if constexpr (sizeof(T) == 4) {
        if (p.f32_split) {
            if (act) hipLaunchKernelGGL((gemm_nt384_kernel<T, OutT, true, true>), grid, dim3(512), X::SMEM, s, p);
            else hipLaunchKernelGGL((gemm_nt384_kernel<T, OutT, false, true>), grid, dim3(512), X::SMEM, s, p);
            WIPA_LAUNCH_CHECK();
            return WIPA_OK;
        }
    }
    if (act)
        hipLaunchKernelGGL((gemm_nt384_kernel<T, OutT, true, false>), grid, dim3(512), X::SMEM, s, p);
    else
        hipLaunchKernelGGL((gemm_nt384_kernel<T, OutT, false, false>), grid, dim3(512), X::SMEM, s, p);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

template <typename T, typename OutT>
int launch384n(GemmParams p, hipStream_t s) {
    typedef X384<2> X;
    p.tiles_m = (p.M + XBM - 1) / XBM;
    p.tiles_n = (p.N + X::BN - 1) / X::BN;
    const dim3 grid(p.tiles_m * p.tiles_n);
    const bool act = p.act == 1 && p.stage_ok;
    if constexpr (sizeof(T) == 4) {
        if (p.f32_split) {
            if (act) hipLaunchKernelGGL((gemm_nt384n_kernel<T, OutT, true, true>), grid, dim3(512), X::SMEM, s, p);
            else hipLaunchKernelGGL((gemm_nt384n_kernel<T, OutT, false, true>), grid, dim3(512), X::SMEM, s, p);
            WIPA_LAUNCH_CHECK();
            return WIPA_OK;
        }
    }
    if (act)
        hipLaunchKernelGGL((gemm_nt384n_kernel<T, OutT, true, false>), grid, dim3(512), X::SMEM, s, p);
    else
        hipLaunchKernelGGL((gemm_nt384n_kernel<T, OutT, false, false>), grid, dim3(512), X::SMEM, s, p);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

template <typename OutT>
int launch256p(GemmParams p, hipStream_t s) {
    p.tiles_m = (p.M + LBM - 1) / LBM;
    p.tiles_n = (p.N + LBN - 1) / LBN;
    hipLaunchKernelGGL((gemm_nt256p_kernel<OutT>), dim3(p.tiles_m * p.tiles_n), dim3(512), LSMEM, s, p);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

template <typename T, typename OutT>
int launch256(GemmParams p, hipStream_t s) {
    p.tiles_m = (p.M + LBM - 1) / LBM;
    p.tiles_n = (p.N + LBN - 1) / LBN;
    if (sizeof(T) == 4 && p.f32_split)
        hipLaunchKernelGGL((gemm_nt256_kernel<T, OutT, sizeof(T) == 4>), dim3(p.tiles_m * p.tiles_n), dim3(512), LSMEM, s, p);
    else
        hipLaunchKernelGGL((gemm_nt256_kernel<T, OutT, false>), dim3(p.tiles_m * p.tiles_n), dim3(512), LSMEM, s, p);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

template <typename OutT>
int launch_fp8_256(GemmParams p, hipStream_t s) {
    p.tiles_m = (p.M + LBM - 1) / LBM;
    p.tiles_n = (p.N + LBN - 1) / LBN;
    hipLaunchKernelGGL((gemm_fp8_256_kernel<OutT>), dim3(p.tiles_m * p.tiles_n), dim3(512), LSMEM, s, p);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

template <typename OutT>
int launch_fp8_384(GemmParams p, hipStream_t s) {
    typedef X384<4> X;
    p.tiles_m = (p.M + XBM - 1) / XBM;
    p.tiles_n = (p.N + X::BN - 1) / X::BN;
    if (p.act == 1) hipLaunchKernelGGL((gemm_fp8_384_kernel<OutT, true>), dim3(p.tiles_m * p.tiles_n), dim3(512), X::SMEM, s, p);
    else hipLaunchKernelGGL((gemm_fp8_384_kernel<OutT, false>), dim3(p.tiles_m * p.tiles_n), dim3(512), X::SMEM, s, p);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

constexpr int SMEM_BYTES = 4 * TILE_BYTES;  // 64 KiB

// Raise the dynamic-LDS limit of every instantiation once, outside any stream capture.
int init_attrs() {
    static std::once_flag once;
    static hipError_t err = hipSuccess;
    std::call_once(once, [] {
        const void* fns[7] = {reinterpret_cast<const void*>(&gemm_nt_kernel<__bf16, __bf16>),
                              reinterpret_cast<const void*>(&gemm_nt_kernel<__bf16, float>),
                              reinterpret_cast<const void*>(&gemm_nt_kernel<float, __bf16>),
                              reinterpret_cast<const void*>(&gemm_nt_kernel<float, float>),
                              reinterpret_cast<const void*>(&gemm_nt_kernel<float, float, true, true>),
                              reinterpret_cast<const void*>(&gemm_nt_kernel<float, float, true, false>),
                              reinterpret_cast<const void*>(&gemm_nt_kernel<float, float, false, true>)};
        for (const void* f : fns) {
            const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
            if (e != hipSuccess) err = e;
        }
        const void* phased[2] = {reinterpret_cast<const void*>(&gemm_nt256p_kernel<__bf16>),
                                 reinterpret_cast<const void*>(&gemm_nt256p_kernel<float>)};
        for (const void* f : phased) {
            const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, LSMEM);
            if (e != hipSuccess) err = e;
        }
        const void* big[6] = {reinterpret_cast<const void*>(&gemm_nt256_kernel<__bf16, __bf16, false>),
                              reinterpret_cast<const void*>(&gemm_nt256_kernel<__bf16, float, false>),
                              reinterpret_cast<const void*>(&gemm_nt256_kernel<float, __bf16, false>),
                              reinterpret_cast<const void*>(&gemm_nt256_kernel<float, float, false>),
                              reinterpret_cast<const void*>(&gemm_nt256_kernel<float, __bf16, true>),
                              reinterpret_cast<const void*>(&gemm_nt256_kernel<float, float, true>)};
        for (const void* f : big) {
            const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, LSMEM);
            if (e != hipSuccess) err = e;
        }
        const void* wide[20] = {reinterpret_cast<const void*>(&gemm_nt384_kernel<__bf16, __bf16, false, false>),
                                reinterpret_cast<const void*>(&gemm_nt384_kernel<__bf16, float, false, false>),
                                reinterpret_cast<const void*>(&gemm_nt384_kernel<float, __bf16, false, false>),
                                reinterpret_cast<const void*>(&gemm_nt384_kernel<float, float, false, false>),
                                reinterpret_cast<const void*>(&gemm_nt384_kernel<__bf16, __bf16, true, false>),
                                reinterpret_cast<const void*>(&gemm_nt384_kernel<__bf16, float, true, false>),
                                reinterpret_cast<const void*>(&gemm_nt384_kernel<float, __bf16, true, false>),
                                reinterpret_cast<const void*>(&gemm_nt384_kernel<float, float, true, false>),
                                reinterpret_cast<const void*>(&gemm_nt384_kernel<float, __bf16, false, true>),
                                reinterpret_cast<const void*>(&gemm_nt384_kernel<float, float, false, true>),
                                reinterpret_cast<const void*>(&gemm_nt384_kernel<float, __bf16, true, true>),
                                reinterpret_cast<const void*>(&gemm_nt384_kernel<float, float, true, true>),
                                reinterpret_cast<const void*>(&gemm_nt384n_kernel<float, __bf16, false, false>),
                                reinterpret_cast<const void*>(&gemm_nt384n_kernel<float, float, false, false>),
                                reinterpret_cast<const void*>(&gemm_nt384n_kernel<float, __bf16, true, false>),
                                reinterpret_cast<const void*>(&gemm_nt384n_kernel<float, float, true, false>),
                                reinterpret_cast<const void*>(&gemm_nt384n_kernel<float, __bf16, false, true>),
                                reinterpret_cast<const void*>(&gemm_nt384n_kernel<float, float, false, true>),
                                reinterpret_cast<const void*>(&gemm_nt384n_kernel<float, __bf16, true, true>),
                                reinterpret_cast<const void*>(&gemm_nt384n_kernel<float, float, true, true>)};
        for (int i = 0; i < 20; ++i) {
            const hipError_t e = hipFuncSetAttribute(wide[i], hipFuncAttributeMaxDynamicSharedMemorySize, i < 12 ? X384<4>::SMEM : X384<2>::SMEM);
            if (e != hipSuccess) err = e;
        }
        const void* f8k[2] = {reinterpret_cast<const void*>(&gemm_fp8_256_kernel<__bf16>),
                              reinterpret_cast<const void*>(&gemm_fp8_256_kernel<float>)};
        for (const void* f : f8k) {
            const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, LSMEM);
            if (e != hipSuccess) err = e;
        }
        const void* f8w[4] = {reinterpret_cast<const void*>(&gemm_fp8_384_kernel<__bf16, false>),
                              reinterpret_cast<const void*>(&gemm_fp8_384_kernel<float, false>),
                              reinterpret_cast<const void*>(&gemm_fp8_384_kernel<__bf16, true>),
                              reinterpret_cast<const void*>(&gemm_fp8_384_kernel<float, true>)};
        for (const void* f : f8w) {
            const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, X384<4>::SMEM);
            if (e != hipSuccess) err = e;
        }
        const void* lnk[9] = {reinterpret_cast<const void*>(&gemm_skinny_ln_kernel<__bf16, __bf16, 1>),
                              reinterpret_cast<const void*>(&gemm_skinny_ln_kernel<__bf16, __bf16, 2>),
                              reinterpret_cast<const void*>(&gemm_skinny_ln_kernel<__bf16, __bf16, 4>),
                              reinterpret_cast<const void*>(&gemm_skinny_ln_kernel<__bf16, float, 1>),
                              reinterpret_cast<const void*>(&gemm_skinny_ln_kernel<__bf16, float, 2>),
                              reinterpret_cast<const void*>(&gemm_skinny_ln_kernel<__bf16, float, 4>),
                              reinterpret_cast<const void*>(&gemm_skinny_ln_kernel<float, float, 1>),
                              reinterpret_cast<const void*>(&gemm_skinny_ln_kernel<float, float, 2>),
                              reinterpret_cast<const void*>(&gemm_skinny_ln_kernel<float, float, 4>)};
        for (const void* f : lnk) {
            const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
            if (e != hipSuccess) err = e;
        }
        const void* widek[] = {reinterpret_cast<const void*>(&gemm_wide_persistent_kernel<float, 12>),
                               reinterpret_cast<const void*>(&gemm_wide_persistent_kernel<float, 16>),
                               reinterpret_cast<const void*>(&gemm_wide_persistent_kernel<float, 24>),
                               reinterpret_cast<const void*>(&gemm_wide_persistent_kernel<float, 32>),
                               reinterpret_cast<const void*>(&gemm_wide_persistent_kernel<__bf16, 12>),
                               reinterpret_cast<const void*>(&gemm_wide_persistent_kernel<__bf16, 16>),
                               reinterpret_cast<const void*>(&gemm_wide_persistent_kernel<__bf16, 24>),
                               reinterpret_cast<const void*>(&gemm_wide_persistent_kernel<__bf16, 32>),
                               reinterpret_cast<const void*>(&gemm_wide_persistent_kernel<float, 12, true>),
                               reinterpret_cast<const void*>(&gemm_wide_persistent_kernel<float, 16, true>),
                               reinterpret_cast<const void*>(&gemm_wide_persistent_kernel<float, 24, true>),
                               reinterpret_cast<const void*>(&gemm_wide_persistent_kernel<float, 32, true>)};
        for (const void* f : widek) {
            const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * (1024 + 8) * 2);
            if (e != hipSuccess) err = e;
        }
    });
    WIPA_CHECK_HIP(err);
    return WIPA_OK;
}

template <typename T, typename OutT>
int launch(const GemmParams& p, hipStream_t s) {
    hipLaunchKernelGGL((gemm_nt_kernel<T, OutT>), dim3(p.tiles_m * p.tiles_n, p.k_slices), dim3(256), SMEM_BYTES, s, p);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

int launch_kmajor(const GemmParams& p, hipStream_t s) {  // float32 -> float32 with K-major A and / or W
    const dim3 grid(p.tiles_m * p.tiles_n, p.k_slices);
    if (p.a_trans && p.w_trans) hipLaunchKernelGGL((gemm_nt_kernel<float, float, true, true>), grid, dim3(256), SMEM_BYTES, s, p);
    else if (p.a_trans) hipLaunchKernelGGL((gemm_nt_kernel<float, float, true, false>), grid, dim3(256), SMEM_BYTES, s, p);
    else hipLaunchKernelGGL((gemm_nt_kernel<float, float, false, true>), grid, dim3(256), SMEM_BYTES, s, p);
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

}  // namespace

int wipa_gemm_init() { return init_attrs(); }

// Dispatch census (wipa_gemm_dispatch_counts): which kernel family each wipa_gemm call went to.  A measurement / test aid --
// the parity tests of the fine-tune step assert that the shape-dependent branches they mean to cover were really taken.
namespace {
std::atomic<int64_t> g_dispatch[WIPA_GEMM_DISPATCH_CLASSES];
inline void count_dispatch(int cls) { g_dispatch[cls].fetch_add(1, std::memory_order_relaxed); }
}  // namespace

extern "C" int wipa_gemm_dispatch_counts(int64_t* out, int n, int reset) {
    WIPA_REQUIRE(n >= 0 && n <= WIPA_GEMM_DISPATCH_CLASSES && (out || n == 0), "wipa_gemm_dispatch_counts: n=%d (0..%d)", n, WIPA_GEMM_DISPATCH_CLASSES);
    for (int i = 0; i < n; ++i) out[i] = g_dispatch[i].load(std::memory_order_relaxed);
    if (reset)
        for (int i = 0; i < WIPA_GEMM_DISPATCH_CLASSES; ++i) g_dispatch[i].store(0, std::memory_order_relaxed);
    return WIPA_OK;
}

// fp8 activations x fp8 weights (wipa_gemm_desc.in_dtype = WIPA_FP8_E4M3): the block-scaled MFMA tile kernel
static int gemm_fp8(const wipa_gemm_desc* d, wipa_stream_t stream) {
    WIPA_REQUIRE(d->A && d->a_scale && d->w_scale && (d->w_dtype == 0 || d->w_dtype == WIPA_FP8_E4M3),
                 "wipa_gemm: fp8 operands need a_scale (per row of A) and w_scale (per row of W)");
    WIPA_REQUIRE(d->K % 128 == 0 && d->lda % 16 == 0 && d->ldw % 16 == 0 && ((uintptr_t)d->A % 16) == 0 && ((uintptr_t)d->W % 16) == 0,
                 "wipa_gemm: fp8 operands: K=%d must be a multiple of 128 and rows 16-byte aligned", d->K);
    WIPA_REQUIRE(!d->ln_x && !d->a_trans && !d->w_trans && d->k_slices <= 1, "wipa_gemm: fp8 operands: no LayerNorm prologue, K-major operands or k_slices");
    WIPA_REQUIRE(d->lda < (1 << 22) && d->ldw < (1 << 22), "wipa_gemm: fp8 operands: row pitch too large for 32-bit tile offsets");
    GemmParams p;
    p.A = (const char*)d->A; p.W = (const char*)d->W; p.C = (char*)d->C;
    p.bias = d->bias; p.residual = (const char*)d->residual; p.pos = d->pos; p.c_offset_dev = d->c_offset_dev;
    p.lda_b = d->lda; p.ldw_b = d->ldw; p.w_scale = d->w_scale; p.a_scale = d->a_scale;
    p.ldc = d->ldc; p.ldpos = d->ldpos; p.M = d->M; p.N = d->N; p.K = d->K;
    p.rg_in = d->rg_in > 0 ? d->rg_in : d->M;
    p.rg_valid = d->rg_in > 0 ? d->rg_valid : d->M;
    p.rg_stride = d->rg_in > 0 ? d->rg_stride : 0;
    p.cg_in = d->cg_in > 0 ? d->cg_in : d->N;
    p.cg_stride = d->cg_in > 0 ? d->cg_stride : 0;
    p.c_offset = d->c_offset; p.zero_invalid = d->zero_invalid_rows; p.bias_along_m = d->bias_along_m; p.act = d->act;
    p.col_scale_n = d->col_scale_n; p.col_scale = d->col_scale;
    p.k_slices = 1; p.slab_stride = 0;
    const int64_t osz = (int64_t)wipa_dtype_size(d->out_dtype);
    p.vec_ok = (p.ldc % 4 == 0) && (p.rg_stride % 4 == 0) && (p.cg_stride % 4 == 0) && (p.cg_in % 4 == 0) && (p.c_offset % 4 == 0) &&
               (((uintptr_t)d->C) % 16 == 0) && (!d->residual || ((uintptr_t)d->residual) % 16 == 0);
    const int64_t epc = 16 / osz;
    p.stage_ok = (p.ldc % epc == 0) && (p.rg_stride % epc == 0) && (p.cg_stride % epc == 0) && (p.cg_in % epc == 0) && (p.c_offset % epc == 0) &&
                 (p.col_scale_n % epc == 0) && !d->c_offset_dev && (((uintptr_t)d->C) % 16 == 0) &&
                 (!d->residual || ((uintptr_t)d->residual) % 16 == 0);
    const int rc = init_attrs();
    if (rc != WIPA_OK) return rc;
    count_dispatch(WIPA_GEMM_TILE_FP8);
    // 384 x 256 tile (round 4): big row counts, a staged epilogue, an activation only with a plain column bias, a grid that fills
    // the 256 CUs at least 0.95 x as well as the 256 x 256 one, and a LONG contraction -- measured on the whisper-large-v3 shapes
    // at M = 96 000 (profiles/r04_fp8_tile_ab.txt): K = 5120 1 554 -> 1 644 TF/s, but K = 1280 1 498 -> 1 475 (q|k) and
    // 1 362 -> 1 336 (mlp1 + GELU): ten K-steps do not pay for the longer prologue and the 192-accumulator epilogue.
    // WIPA_GEMM_FP8_TILE=256 / 384 forces.
    static const int force_f8 = [] { const char* e = getenv("WIPA_GEMM_FP8_TILE"); return e ? atoi(e) : 0; }();
    const bool act384_ok = d->act == 0 || (!d->bias_along_m && d->col_scale_n == 0);
    bool use384 = false;
    if (p.stage_ok && act384_ok && force_f8 != 256 && d->M >= 2 * XBM) {
        auto fill = [](int64_t tiles) { const int64_t rounds = (tiles + 255) / 256; return (double)tiles / (double)(rounds * 256); };
        const double e256 = fill((int64_t)((d->M + LBM - 1) / LBM) * ((d->N + LBN - 1) / LBN));
        const double e384 = fill((int64_t)((d->M + XBM - 1) / XBM) * ((d->N + LBN - 1) / LBN));
        use384 = force_f8 == 384 || (e384 >= 0.95 * e256 && d->K >= 2048);
    }
    if (use384) count_dispatch(WIPA_GEMM_TILE_FP8_384);
    if (use384) return d->out_dtype == WIPA_BF16 ? launch_fp8_384<__bf16>(p, (hipStream_t)stream) : launch_fp8_384<float>(p, (hipStream_t)stream);
    return d->out_dtype == WIPA_BF16 ? launch_fp8_256<__bf16>(p, (hipStream_t)stream) : launch_fp8_256<float>(p, (hipStream_t)stream);
}

extern "C" int wipa_gemm(const wipa_gemm_desc* d, wipa_stream_t stream) {
    WIPA_REQUIRE(d && (d->A || d->ln_x) && d->W && d->C, "wipa_gemm: null operand");
    WIPA_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "wipa_gemm: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
    WIPA_REQUIRE(d->in_dtype == WIPA_F32 || d->in_dtype == WIPA_BF16 || d->in_dtype == WIPA_FP8_E4M3, "wipa_gemm: bad in_dtype %d", d->in_dtype);
    WIPA_REQUIRE(d->out_dtype == WIPA_F32 || d->out_dtype == WIPA_BF16, "wipa_gemm: bad out_dtype %d", d->out_dtype);
    if (d->in_dtype == WIPA_FP8_E4M3) return gemm_fp8(d, stream);
    const int64_t esz = (int64_t)wipa_dtype_size(d->in_dtype);
    if (d->w_dtype == WIPA_FP8_E4M3) {
        // fp8 weights: weight-streaming kernel only (decode rows); larger M runs on weights dequantised to bf16 at load time
        WIPA_REQUIRE(d->in_dtype == WIPA_BF16 && d->w_scale && d->A, "wipa_gemm: fp8 weights need bf16 activations and w_scale");
        WIPA_REQUIRE(d->K % 64 == 0 && d->ldw % 16 == 0 && (d->lda * 2) % 16 == 0 && ((uintptr_t)d->A % 16) == 0 && ((uintptr_t)d->W % 16) == 0,
                     "wipa_gemm: fp8 weights: K must be a multiple of 64 and rows 16-byte aligned");
        WIPA_REQUIRE(d->M <= SKINNY_STREAM_MAX_M, "wipa_gemm: fp8 weights: M=%d exceeds the weight-streaming kernel (%d rows)", d->M, SKINNY_STREAM_MAX_M);
        WIPA_REQUIRE(!d->ln_x, "wipa_gemm: fp8 weights: no LayerNorm prologue");
    } else {
        WIPA_REQUIRE(d->w_dtype == 0, "wipa_gemm: bad w_dtype %d (0 = the input dtype, %d = fp8 e4m3)", d->w_dtype, WIPA_FP8_E4M3);
    }
    WIPA_REQUIRE(d->w_dtype == WIPA_FP8_E4M3 || (d->K * esz) % ROWB == 0, "wipa_gemm: K=%d must be a multiple of %d elements", d->K, (int)(ROWB / esz));
    WIPA_REQUIRE((d->lda * esz) % 16 == 0 && (d->ldw * esz) % 16 == 0, "wipa_gemm: lda/ldw rows must be 16-byte aligned");
    WIPA_REQUIRE(((uintptr_t)d->A % 16) == 0 && ((uintptr_t)d->W % 16) == 0, "wipa_gemm: A/W must be 16-byte aligned");
    GemmParams p;
    p.A = (const char*)d->A;
    p.W = (const char*)d->W;
    p.C = (char*)d->C;
    p.bias = d->bias;
    p.residual = (const char*)d->residual;
    p.pos = d->pos;
    p.c_offset_dev = d->c_offset_dev;
    p.lda_b = d->lda * esz;
    p.ldw_b = d->w_dtype == WIPA_FP8_E4M3 ? d->ldw : d->ldw * esz;
    p.w_scale = d->w_scale;
    p.ldc = d->ldc;
    p.ldpos = d->ldpos;
    p.M = d->M;
    p.N = d->N;
    p.K = d->K;
    p.rg_in = d->rg_in > 0 ? d->rg_in : d->M;
    p.rg_valid = d->rg_in > 0 ? d->rg_valid : d->M;
    p.rg_stride = d->rg_in > 0 ? d->rg_stride : 0;
    p.cg_in = d->cg_in > 0 ? d->cg_in : d->N;
    p.cg_stride = d->cg_in > 0 ? d->cg_stride : 0;
    p.c_offset = d->c_offset;
    p.zero_invalid = d->zero_invalid_rows;
    p.bias_along_m = d->bias_along_m;
    p.act = d->act;
    p.col_scale_n = d->col_scale_n;
    p.col_scale = d->col_scale;
    p.tiles_m = (d->M + BM - 1) / BM;
    p.tiles_n = (d->N + BN - 1) / BN;
    p.k_slices = d->k_slices > 1 ? d->k_slices : 1;
    p.slab_stride = d->slab_stride;
    if (p.k_slices > 1) {
        WIPA_REQUIRE(!d->residual && !d->pos && d->act == 0 && d->col_scale_n == 0,
                     "wipa_gemm: k_slices writes partial sums: no residual/pos/act/col_scale");
        WIPA_REQUIRE(p.k_slices <= 16 && d->slab_stride % 4 == 0, "wipa_gemm: bad k_slices / slab_stride");
    }
    const int64_t osz = (int64_t)wipa_dtype_size(d->out_dtype);
    const int64_t valign = 16 / osz == 4 ? 4 : 4;  // 4 consecutive outputs per store
    p.vec_ok = (p.ldc % valign == 0) && (p.rg_stride % valign == 0) && (p.cg_stride % valign == 0) &&
               (p.cg_in % 4 == 0) && (p.c_offset % valign == 0) && (((uintptr_t)d->C) % 16 == 0) &&
               (!d->residual || ((uintptr_t)d->residual) % 16 == 0);
    {
        const int64_t epc = 16 / osz;  // elements per 16-byte row-major store
        static const int no_stage = [] { const char* e = getenv("WIPA_GEMM_NO_STAGE"); return e ? atoi(e) : 0; }();
        p.stage_ok = !no_stage && (p.ldc % epc == 0) && (p.rg_stride % epc == 0) && (p.cg_stride % epc == 0) &&
                     (p.cg_in % epc == 0) && (p.c_offset % epc == 0) && (p.col_scale_n % epc == 0) && !d->c_offset_dev && (((uintptr_t)d->C) % 16 == 0) &&
                     (!d->residual || ((uintptr_t)d->residual) % 16 == 0);
    }
    hipStream_t s = (hipStream_t)stream;
    {
        // f32 inputs in the tile kernels: exact f32 products on the f32 MFMA unless the caller opted into three bf16 MFMA
        // terms per product (desc.f32_split: 2x the rate, ~5e-6 relative error).  The weight-streaming kernel stays exact.
        p.f32_split = (d->in_dtype == WIPA_F32 && d->f32_split) ? 1 : 0;
    }
    {
        const int rc = init_attrs();
        if (rc != WIPA_OK) return rc;
    }
    if (d->a_trans || d->w_trans) {
        WIPA_REQUIRE(d->in_dtype == WIPA_F32 && d->out_dtype == WIPA_F32 && d->w_dtype == 0 && !d->ln_x,
                     "wipa_gemm: K-major operands (a_trans / w_trans): float32 in and out only");
        WIPA_REQUIRE((!d->a_trans || (d->M % 4 == 0 && d->M >= 4)) && (!d->w_trans || (d->N % 4 == 0 && d->N >= 4)),
                     "wipa_gemm: a K-major operand needs a row count that is a multiple of 4 (M=%d N=%d)", d->M, d->N);
        p.a_trans = d->a_trans ? 1 : 0;
        p.w_trans = d->w_trans ? 1 : 0;
        count_dispatch(WIPA_GEMM_KMAJOR);
        if (p.k_slices > 1) count_dispatch(WIPA_GEMM_SPLIT_K);
        return launch_kmajor(p, s);
    }
    if (p.k_slices > 1) count_dispatch(WIPA_GEMM_SPLIT_K);
    if (d->w_dtype == WIPA_FP8_E4M3) {
        count_dispatch(WIPA_GEMM_SKINNY_FP8);
        return d->out_dtype == WIPA_BF16 ? launch_skinny_w8<__bf16>(p, s) : launch_skinny_w8<float>(p, s);
    }
    if (d->ln_x) {  // LayerNorm prologue: A is computed in the kernel from the f32 rows ln_x
        WIPA_REQUIRE(d->ln_w && d->ln_b && d->ln_ldx >= d->K && d->ln_ldx % 4 == 0 && ((uintptr_t)d->ln_x % 16) == 0,
                     "wipa_gemm: LayerNorm prologue needs ln_w, ln_b and 16-byte aligned rows of at least K floats");
        WIPA_REQUIRE(d->K % 64 == 0 && d->K <= 256 * LNP_NV && d->M <= SKINNY_STREAM_MAX_M && p.k_slices == 1,
                     "wipa_gemm: LayerNorm prologue: K=%d must be a multiple of 64 and <= %d, M <= %d, no k_slices", d->K, 256 * LNP_NV,
                     SKINNY_STREAM_MAX_M);
        WIPA_REQUIRE(!(d->in_dtype == WIPA_F32 && d->out_dtype == WIPA_BF16), "wipa_gemm: LayerNorm prologue: f32 -> bf16 is not built");
        p.ln_x = d->ln_x; p.ln_w = d->ln_w; p.ln_b = d->ln_b; p.ln_ldx = d->ln_ldx; p.ln_eps = d->ln_eps;
        count_dispatch(WIPA_GEMM_SKINNY_LN);
        if (d->in_dtype == WIPA_BF16)
            return d->out_dtype == WIPA_BF16 ? launch_skinny_ln<__bf16, __bf16>(p, s) : launch_skinny_ln<__bf16, float>(p, s);
        return launch_skinny_ln<float, float>(p, s);
    }
    // weight-streaming kernel: M <= 256 rows, or up to 1024 when the caller marks them as decode rows (prompt prefill)
    const bool skinny_shape = d->M <= SKINNY_MAX_M || (d->stream_weights && d->M <= SKINNY_STREAM_MAX_M);
    if (skinny_shape && !(d->in_dtype == WIPA_F32 && d->out_dtype == WIPA_BF16)) {
        count_dispatch(WIPA_GEMM_SKINNY);
        if (d->in_dtype == WIPA_BF16)
            return d->out_dtype == WIPA_BF16 ? launch_skinny<__bf16, __bf16>(p, s) : launch_skinny<__bf16, float>(p, s);
        return launch_skinny<float, float>(p, s);
    }
    static const int force_tile = [] {
        const char* e = getenv("WIPA_GEMM_TILE");  // debugging / A-B timing: 128 or 256
        return e ? atoi(e) : 0;
    }();
    // the 256-tile kernels address a tile with 32-bit buffer offsets: 256 rows x row pitch + K bytes must stay below 2^31
    const bool pitch_ok = p.lda_b < (1 << 22) && p.ldw_b < (1 << 22) && d->K * esz < (1 << 22);
    // A grid of 256 x 256 tiles must at least come close to filling the 256 CUs: the decoder GEMMs of a fine-tune step
    // (M = 32 clips x 64 tokens = 2048 rows, N = 768 or 3072) are 24 / 96 such tiles and ran at 7 / 51 TF/s in float32
    // (r02 trace of `bench.py --mode train`: 55 ms of a 237 ms step); they go to the 128 x 128 kernel (96 / 384 workgroups).
    static const int min_big_tiles = [] { const char* e = getenv("WIPA_GEMM_MIN_BIG_TILES"); return e ? atoi(e) : 160; }();
    const int64_t t256_grid = (int64_t)((d->M + LBM - 1) / LBM) * ((d->N + LBN - 1) / LBN);
    const bool big = pitch_ok && p.k_slices == 1 && (  // split-K beyond the skinny rows lives in the 128x128 kernel
                     force_tile >= 256 ||  // WIPA_GEMM_TILE=256: the 256x256 kernel for every shape
                                  (force_tile != 128 && d->M >= 512 && d->N >= 256 && (int64_t)d->M * d->N >= (1 << 20) &&
                                   t256_grid >= min_big_tiles));
    // 384 x 256 tile (fewer staged bytes per FLOP, 2.93 instead of 4.39 rounds at N = 768): measured faster on every encoder
    // shape without an activation (its instantiations are compiled without the GELU path, which is what keeps 192
    // accumulators + the epilogue under 256 registers), unless its grid quantises clearly worse than the 256-tile grid.
    // an activation is applied before the epilogue there, which is only equivalent for: column bias, no column scale, no
    // positional term, no residual ordering issue (act precedes pos/residual in epilogue order anyway)
    const bool act384_ok = d->act == 0 || (p.stage_ok && !d->bias_along_m && d->col_scale_n == 0);
    bool use384 = force_tile == 384 && act384_ok, narrow = false;
    if (big && force_tile == 0 && act384_ok && d->M >= 2 * XBM) {
        auto fill = [](int64_t t) { return (double)t / (double)(((t + 255) / 256) * 256); };
        const int64_t tm384 = (d->M + XBM - 1) / XBM;
        const double e256 = fill((int64_t)((d->M + LBM - 1) / LBM) * ((d->N + LBN - 1) / LBN));
        const double e384 = fill(tm384 * ((d->N + LBN - 1) / LBN));
        use384 = e384 >= 0.95 * e256;
        // 384 x 128 tiles when the wide grids leave a large part of the last round empty (48 000 x 768: 375 or 564 tiles, 0.73
        // either way, against 750 narrow ones, 0.98).  Float32 only: it runs at the f32 MFMA's pace whatever the tile (1077 ->
        // 845 us per GEMM); bf16 pays for the extra staged bytes per FLOP and measured no gain even at a fill ratio of 1.33
        // (whisper-small, 32 clips: 12.99 ms of GEMMs per pass with narrow tiles, 12.81 ms with wide ones).
        const double e384n = fill(tm384 * ((d->N + 127) / 128));
        if (d->in_dtype == WIPA_F32 && e384n >= 1.10 * (e384 > e256 ? e384 : e256)) use384 = narrow = true;
    }
    if (big && force_tile == 2568 && d->in_dtype == WIPA_BF16 && d->K % 128 == 0) {  // phase-interleaved 256 x 256 (A/B runs)
        count_dispatch(WIPA_GEMM_TILE256P);
        return d->out_dtype == WIPA_BF16 ? launch256p<__bf16>(p, s) : launch256p<float>(p, s);
    }
    if (big && use384) {
        count_dispatch(narrow ? WIPA_GEMM_TILE384N : WIPA_GEMM_TILE384);
        if (narrow) return d->out_dtype == WIPA_BF16 ? launch384n<float, __bf16>(p, s) : launch384n<float, float>(p, s);
        if (d->in_dtype == WIPA_BF16)
            return d->out_dtype == WIPA_BF16 ? launch384<__bf16, __bf16>(p, s) : launch384<__bf16, float>(p, s);
        return d->out_dtype == WIPA_BF16 ? launch384<float, __bf16>(p, s) : launch384<float, float>(p, s);
    }
    if (big) {
        count_dispatch(WIPA_GEMM_TILE256);
        if (d->in_dtype == WIPA_BF16)
            return d->out_dtype == WIPA_BF16 ? launch256<__bf16, __bf16>(p, s) : launch256<__bf16, float>(p, s);
        return d->out_dtype == WIPA_BF16 ? launch256<float, __bf16>(p, s) : launch256<float, float>(p, s);
    }
    count_dispatch(WIPA_GEMM_TILE128);
    if (d->in_dtype == WIPA_BF16) {
        return d->out_dtype == WIPA_BF16 ? launch<__bf16, __bf16>(p, s) : launch<__bf16, float>(p, s);
    }
    return d->out_dtype == WIPA_BF16 ? launch<float, __bf16>(p, s) : launch<float, float>(p, s);
}

// ---------------------------------------------------------------------------------------
// Logits projection of a greedy decode step with the arg-max / log-sum-exp partials of the filtered rows (round 4): see
// gemm_wide_persistent_kernel<.., GREEDY>.  256 workgroups x 8 waves = WIPA_GREEDY_PARTS partials per row.
extern "C" int wipa_logits_greedy_supported(int B, int V, int d, int dtype) {
    return dtype == WIPA_BF16 && B >= 1 && B <= 64 && V >= 8192 && (d == 384 || d == 512 || d == 768 || d == 1024);
}
extern "C" size_t wipa_logits_greedy_partials_bytes(int B) { return (size_t)B * 3 * WIPA_GREEDY_PARTS * 4; }

extern "C" int wipa_logits_greedy(const void* x, int64_t ldx, const void* w, int64_t ldw, float* logits, int64_t ldl, int B, int V, int d,
                                  const float* mask_first, const float* mask_always, const int32_t* pos_dev, int n_init, float* partials,
                                  size_t partials_bytes, wipa_stream_t stream) {
    WIPA_REQUIRE(x && w && mask_first && mask_always && pos_dev && partials, "wipa_logits_greedy: null pointer");
    WIPA_REQUIRE(wipa_logits_greedy_supported(B, V, d, WIPA_BF16), "wipa_logits_greedy: bf16, 1..64 rows, V >= 8192, d in {384, 512, 768, 1024} (B=%d V=%d d=%d)",
                 B, V, d);
    WIPA_REQUIRE(partials_bytes >= wipa_logits_greedy_partials_bytes(B), "wipa_logits_greedy: partials buffer too small");
    WIPA_REQUIRE((ldx * 2) % 16 == 0 && (ldw * 2) % 16 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0 &&
                     (!logits || (((uintptr_t)logits % 16) == 0 && ldl % 4 == 0 && ldl >= V)),
                 "wipa_logits_greedy: rows must be 16-byte aligned");
    GemmParams p;
    static float dummy_c[4];
    p.A = (const char*)x; p.W = (const char*)w; p.C = (char*)(logits ? logits : dummy_c);
    p.bias = nullptr; p.residual = nullptr; p.pos = nullptr; p.c_offset_dev = nullptr;
    p.lda_b = ldx * 2; p.ldw_b = ldw * 2; p.ldc = logits ? ldl : 0; p.ldpos = 0;
    p.M = B; p.N = V; p.K = d;
    p.rg_in = B; p.rg_valid = B; p.rg_stride = 0; p.cg_in = V; p.cg_stride = 0; p.c_offset = 0;
    p.zero_invalid = 0; p.bias_along_m = 0; p.act = 0; p.col_scale_n = 0; p.col_scale = 1.f;
    p.tiles_m = 1; p.tiles_n = 1; p.k_slices = 1; p.slab_stride = 0;
    p.vec_ok = 1; p.stage_ok = 0;
    p.g_mask_first = mask_first; p.g_mask_always = mask_always; p.g_pos = pos_dev; p.g_n_init = n_init; p.g_part = partials;
    p.g_store = logits ? 1 : 0;
    const int rc = init_attrs();
    if (rc != WIPA_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(WIPA_GREEDY_PARTS / 8), block(512);
    switch (d) {
        case 384: hipLaunchKernelGGL((gemm_wide_persistent_kernel<float, 12, true>), grid, block, (size_t)64 * (384 + 8) * 2, s, p); break;
        case 512: hipLaunchKernelGGL((gemm_wide_persistent_kernel<float, 16, true>), grid, block, (size_t)64 * (512 + 8) * 2, s, p); break;
        case 768: hipLaunchKernelGGL((gemm_wide_persistent_kernel<float, 24, true>), grid, block, (size_t)64 * (768 + 8) * 2, s, p); break;
        default: hipLaunchKernelGGL((gemm_wide_persistent_kernel<float, 32, true>), grid, block, (size_t)64 * (1024 + 8) * 2, s, p); break;
    }
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}
