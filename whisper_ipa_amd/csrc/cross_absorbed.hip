// Decode-step cross-attention with the key / value projections ABSORBED (round 3).
//
// mlx_whisper's MultiHeadAttention (behind DecodingTask._main_loop, scripts/transcribe_single.py:55) projects the encoder
// output xa [1500, d] of a clip to K_l = xa Wk_l^T and V_l = xa Wv_l^T + bv_l once per decoder layer l and caches them; every
// decode step then streams K_l and V_l: 2 x 1500 x d values per clip, layer and step -- 55.3 MB per clip and step for
// whisper-small, the dominant HBM stream of the whole transcription.  But both are linear images of the
// SAME xa:
//
//     scores_h = q_h K_h^T = (q_h Wk_h) xa^T            Wk_h = rows h*64 .. h*64+63 of Wk   [64, d]
//     out_h    = P_h V_h   = (P_h xa) Wv_h^T + bv_h     (sum_t P_h[t] = 1)
//
// so a step can read xa ONCE per layer for the scores and for the values: half the bytes, no K/V cache (3.5 GB -> 0.15 GB
// per 64 clips), no cross-K/V projection GEMMs (42.5 GFLOP per clip), and xa is the same for all layers.  The price is
// arithmetic -- the contraction runs over d = 768 channels instead of 64 per head, 12x the FLOPs -- which is why it only works
// on the matrix cores, with the 12 heads of a clip as one 16-wide MFMA dimension:
//
//   cross_absorb_q_kernel    Qp[b][h][:] = scale * q_h[b] Wk_h              [B, 16, d] bf16 (heads 12..15 stay zero)
//   cross_absorbed_v2_kernel (d <= 768, the default) one workgroup per (clip, frame split) of three INDEPENDENT waves: a wave owns
//                            16-frame groups of xa in two private LDS slots filled by LDS-DMA (no barrier in the loop), computes
//                            S^T[16 frames, 16 heads] over all channels against the absorbed queries held in registers, a
//                            softmax against a fixed per-head reference, and O'[16 heads, d] += P group with the SAME group read
//                            column-wise by ds_read_b64_tr_b16; the three waves' (m, l, O') merge through LDS at the end
//   cross_absorbed_kernel    (d = 1024, and WIPA_ABS_KERNEL=1 for A/B runs) the first form: 32-frame tiles shared by 4 waves
//                            that split the CHANNELS, partial scores exchanged through LDS in a fixed order, online softmax
//   cross_merge_proj_kernel  merges the splits (fixed order), out_h = (O'_h / l_h) Wv_h^T + bv_h -> [B, d] bf16
//
// Rounding points differ from the cached-K/V path (K and V are never rounded to bf16 here; Qp and O' are): the results
// are equal up to bf16 noise, and closer to the f32 arithmetic.  bf16 models with <= 16 heads and d in {384, 512, 768, 1024}.
#include <cstdlib>
#include <mutex>

#include "wipa_common.h"

namespace {

constexpr float NEG_BIG = -1.0e30f;
constexpr int FT = 32;  // frames per tile
typedef __attribute__((address_space(3))) void* lds_ptr_a;
typedef int v2i32 __attribute__((ext_vector_type(2)));

struct AbsParams {
    const __bf16* qp;     // [B][16][D]
    const __bf16* xa;     // [B][Tk][D]
    float* part_m;        // [B][S][16]
    float* part_l;        // [B][S][16]
    float* part_o;        // [B][S][16][D]
    int Tk, n_splits, tiles_per_split, H;
};

// chunk swizzle of the LDS tile image: 16-byte chunk c of frame row r sits at chunk c ^ ((r & 7) << 1).  The four rows of a
// transposed-read block (and the two blocks of a 32-lane half, 4 rows apart) then hit disjoint banks.
__device__ __forceinline__ int swz(int row) { return (row & 7) << 1; }

// NW waves split the channels (scores: k-steps; O': column tiles).  NBUF tile buffers: with three, two tiles are in flight
// while one is being spent and a tile costs two workgroup barriers; with two (d = 1024: 64 KiB tiles) a third barrier guards
// the buffer that is re-staged.
template <int D>
struct AbsCfg {
    static constexpr int NW = 4;  // (8 waves and two / three buffers were measured for d = 768: no faster)
    static constexpr int ROWB = D * 2;
    static constexpr int TILE = FT * ROWB;
    static constexpr int SX = NW * 2 * 64 * 4 * (int)sizeof(float);
    static constexpr int NBUF = (3 * TILE + SX <= 160 * 1024) ? 3 : 2;
    static constexpr int SMEM = NBUF * TILE + SX;
};

template <int D>
__global__ __launch_bounds__(512, 1) void cross_absorbed_kernel(AbsParams p) {
    typedef AbsCfg<D> X;
    constexpr int NW = X::NW, ROWB = X::ROWB, TILE = X::TILE, NBUF = X::NBUF;
    constexpr int NDMA = TILE / 1024 / NW;    // LDS-DMA transfers per wave and tile (1 KiB each)
    constexpr int KS = D / NW / 32;           // k-steps (32 channels) of the score product per wave
    constexpr int CT = D / NW / 16;           // 16-channel column tiles of O' per wave
    static_assert(D % (32 * NW) == 0 && TILE % (1024 * NW) == 0, "width");
    extern __shared__ __attribute__((aligned(1024))) char smem[];  // [NBUF tiles][FT rows][ROWB] | sx [NW][2][64][4] f32
    float* sx = reinterpret_cast<float*>(smem + NBUF * TILE);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int split = blockIdx.x, b = blockIdx.y;
    const int tile0 = split * p.tiles_per_split;
    const int n_tiles_total = (p.Tk + FT - 1) / FT;
    const int nt = max(0, min(p.tiles_per_split, n_tiles_total - tile0));
    const __bf16* xb = p.xa + (int64_t)b * p.Tk * D;

    // Qp fragments of this wave's channels: B operand of the score product, lane (head = l15, k-group g)
    bf16x8 qf[KS];
    {
        const __bf16* qr = p.qp + ((int64_t)b * 16 + l15) * D + wave * (D / NW) + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qr + 32 * ks);
    }
    // LDS-DMA: transfer i of this wave covers image bytes [1024 (NDMA wave + i), +1024); a lane's 16 bytes sit at image
    // (row, chunk') and come from source chunk chunk' ^ swz(row) of that frame
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, 0x7fffffff, 0x00020000);
    int drow[NDMA], dch[NDMA];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        const int off = 1024 * (NDMA * wave + i) + 16 * lane;
        drow[i] = off / ROWB;
        dch[i] = ((off % ROWB) >> 4) ^ swz(off / ROWB);
    }
    auto stage = [&](int t, int buf) {
        const int f0 = (tile0 + t) * FT;
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int voff = min(f0 + drow[i], p.Tk - 1) * ROWB + dch[i] * 16;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_ptr_a)(smem + buf * TILE + 1024 * (NDMA * wave + i)), 16, voff, 0, 0, 0);
        }
    };
    auto wait_dma = [&](bool one_behind) {  // this wave's transfers of the current tile have landed (one younger tile may fly on)
        if (one_behind) {
            if constexpr (NDMA == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if constexpr (NDMA == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if constexpr (NDMA == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if constexpr (NDMA == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (NDMA == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if constexpr (NDMA == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    f32x4 acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = NEG_BIG, l_run = 0.f;  // of head l15 (replicated over the four lane groups)

    if (nt > 0) stage(0, 0);
    if (nt > 1) stage(1, 1);
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_a)smem;
    for (int t = 0; t < nt; ++t) {
        const int buf = t % NBUF;
        wait_dma(t + 1 < nt);
        __builtin_amdgcn_s_barrier();  // every wave's share of tile t has landed; every wave is done with tile t - 1
        asm volatile("" ::: "memory");
        if constexpr (NBUF == 3) {
            if (t + 2 < nt) stage(t + 2, (t + 2) % 3);  // into the buffer tile t - 1 has just left
        }
        const char* tb = smem + buf * TILE;
        // ---- partial scores over this wave's channels: S^T[frame 16 ft + 4g + r][head l15]
        f32x4 s[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
            const int row = 16 * ft + l15;
            const char* rp = tb + row * ROWB;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int ch = (4 * (KS * wave + ks) + g) ^ swz(row);
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(rp + 16 * ch);
                s[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[ks], s[ft], 0, 0, 0);
            }
        }
        *reinterpret_cast<f32x4*>(sx + ((wave * 2 + 0) * 64 + lane) * 4) = s[0];
        *reinterpret_cast<f32x4*>(sx + ((wave * 2 + 1) * 64 + lane) * 4) = s[1];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
            f32x4 tot = *reinterpret_cast<const f32x4*>(sx + ((0 * 2 + ft) * 64 + lane) * 4);
#pragma unroll
            for (int w = 1; w < NW; ++w) tot += *reinterpret_cast<const f32x4*>(sx + ((w * 2 + ft) * 64 + lane) * 4);  // fixed order
            s[ft] = tot;
        }
        // ---- online softmax of head l15 over the tile's 32 frames (8 in this lane, the rest in lanes l15 + 16 k)
        const int f0 = (tile0 + t) * FT;
        float mx = NEG_BIG;
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (f0 + 16 * ft + 4 * g + r >= p.Tk) s[ft][r] = NEG_BIG;
                mx = fmaxf(mx, s[ft][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        float pr[8], ls = 0.f;
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = (s[ft][r] <= -1.0e29f) ? 0.f : __expf(s[ft][r] - m_new);
                pr[4 * ft + r] = e;
                ls += e;
            }
        ls += __shfl_xor(ls, 16, 64);
        ls += __shfl_xor(ls, 32, 64);
        l_run = l_run * alpha + ls;
        m_run = m_new;
        // the accumulators hold heads 4g + r in their rows: fetch those heads' rescale factors from the lanes that own them
        float al[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) al[r] = __shfl(alpha, 4 * g + r, 64);
        bf16x8 pf;  // A operand of P x tile: row = head l15, k-slot (g, j) = frame (j < 4 ? 4g + j : 16 + 4g + j - 4)
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (__bf16)pr[j];
        // ---- O'[head][channel] += P x tile: the tile read column-wise (transposed LDS reads), 16 channels per column tile
        {
            const int q = l15 >> 2, pp = l15 & 3;
            const int r1 = 4 * g + q, r2 = 16 + 4 * g + q;
            const unsigned a1 = lds_base + buf * TILE + r1 * ROWB + 8 * (pp & 1);
            const unsigned a2 = lds_base + buf * TILE + r2 * ROWB + 8 * (pp & 1);
            const int cbase = (D / NW / 8) * wave + (pp >> 1);  // logical 16-byte chunk of column tile 0: (2 wave D/NW + 8 pp) / 16
            typedef int v4i32x __attribute__((ext_vector_type(4)));
            // Transposed reads and the wait that completes them live in ONE asm statement: the compiler believes an asm's
            // outputs are written when the statement ends, so with the wait in a later statement it was free to copy a
            // destination register before the LDS data had arrived -- harmless at idle-LDS latencies, wrong as soon as a
            // co-resident workgroup of another stream slowed the LDS down (found as run-to-run id differences with four
            // passes in flight).  Three column tiles (six reads) per statement.
            static_assert(CT % 3 == 0 || CT % 2 == 0, "column tiles per wave");
            constexpr int GB = (CT % 3 == 0) ? 3 : 2;
#pragma unroll
            for (int c0 = 0; c0 < CT; c0 += GB) {
                v2i32 lo[3], hi[3];
                unsigned ad[6];
#pragma unroll
                for (int c = 0; c < GB; ++c) {
                    const int lc = cbase + 2 * (c0 + c);
                    ad[2 * c] = a1 + 16 * (lc ^ swz(r1));
                    ad[2 * c + 1] = a2 + 16 * (lc ^ swz(r2));
                }
                if constexpr (GB == 3) {
                    asm volatile(
                        "ds_read_b64_tr_b16 %0, %6\n\tds_read_b64_tr_b16 %1, %7\n\tds_read_b64_tr_b16 %2, %8\n\t"
                        "ds_read_b64_tr_b16 %3, %9\n\tds_read_b64_tr_b16 %4, %10\n\tds_read_b64_tr_b16 %5, %11\n\t"
                        "s_waitcnt lgkmcnt(0)"
                        : "=&v"(lo[0]), "=&v"(hi[0]), "=&v"(lo[1]), "=&v"(hi[1]), "=&v"(lo[2]), "=&v"(hi[2])
                        : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5])
                        : "memory");
                } else {
                    asm volatile(
                        "ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %5\n\tds_read_b64_tr_b16 %2, %6\n\t"
                        "ds_read_b64_tr_b16 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                        : "=&v"(lo[0]), "=&v"(hi[0]), "=&v"(lo[1]), "=&v"(hi[1])
                        : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3])
                        : "memory");
                }
#pragma unroll
                for (int c = 0; c < GB; ++c) {
                    const bf16x8 xf = __builtin_bit_cast(bf16x8, v4i32x{lo[c][0], lo[c][1], hi[c][0], hi[c][1]});
                    f32x4 o = acc[c0 + c];
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] *= al[r];
                    acc[c0 + c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, xf, o, 0, 0, 0);
                }
            }
        }
        if constexpr (NBUF == 2) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // all waves are done with tile t: its buffer may be re-staged
            asm volatile("" ::: "memory");
            if (t + 2 < nt) stage(t + 2, buf);
        }
    }
    // ---- partial results of this split
    const int64_t ps = (int64_t)b * p.n_splits + split;
    if (wave == 0 && g == 0) {
        p.part_m[ps * 16 + l15] = m_run;
        p.part_l[ps * 16 + l15] = l_run;
    }
    float* po = p.part_o + ps * 16 * D;
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) po[(int64_t)(4 * g + r) * D + wave * (D / NW) + 16 * c + l15] = acc[c][r];
}

// ---------------------------------------------------------------------------------------------------------------------
// Second form of the streaming kernel: INDEPENDENT WAVES.  The kernel above splits a tile's channels over the waves, which
// costs two workgroup barriers, an exchange of partial scores through LDS and a redundant softmax per 32 frames -- a serial
// chain of ~1.8 us per tile on one wave per SIMD, as long as the tile's DMA itself, and the two do not overlap well (measured:
// stream alone 26.7 us, arithmetic alone 21.7 us, together 31-41 us per launch).  Here every wave is a flash-decoding worker
// of its own: it stages its OWN 16-frame groups (two private 24 KiB slots, LDS-DMA, its own vmcnt waits -- no barrier in the
// loop), computes the scores of its 16 frames over all 768 channels (24 MFMAs, the absorbed queries of all k-steps held in
// registers), the softmax of 16 frames per head, and O'[16 heads x 768] += P x group on v_mfma_f32_16x16x16_bf16 with the
// group read column-wise by ONE transposed read per column tile (a lane's four probabilities ARE the A fragment).  Three waves
// per workgroup (6 x 24 KiB = 144 KiB of LDS); their (m, l, O') are merged through LDS once, at the end.
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));

template <int D, int NW_>
struct AbsCfg2 {
    static constexpr int NWV = NW_;                 // independent waves per workgroup (3: 144 KiB of LDS; 2: 96 KiB, room for co-residents)
    static constexpr int GF = 16;                   // frames per group
    static constexpr int ROWB = D * 2;
    static constexpr int SLOT = GF * ROWB;          // 24 KiB for d = 768
    static constexpr int SMEM_LOOP = NWV * 2 * SLOT;
    static constexpr int MROW = D + 4;              // padded row of the final merge image
    static constexpr int SMEM_MERGE = NWV * 16 * MROW * 4 + NWV * 32 * 4;
    static constexpr int SMEM = SMEM_LOOP > SMEM_MERGE ? SMEM_LOOP : SMEM_MERGE;
};

// eight transposed reads (column tiles 8 blk .. 8 blk + 7: per-lane addresses A[0..7] + one immediate) and their wait in ONE
// asm statement (see the note on the first kernel: the outputs are only valid once the wait has run)
#define WIPA_TR8(X, A, IMM)                                                                                                          \
    asm volatile("ds_read_b64_tr_b16 %0, %8 offset:" #IMM "\n\tds_read_b64_tr_b16 %1, %9 offset:" #IMM                              \
                 "\n\tds_read_b64_tr_b16 %2, %10 offset:" #IMM "\n\tds_read_b64_tr_b16 %3, %11 offset:" #IMM                          \
                 "\n\tds_read_b64_tr_b16 %4, %12 offset:" #IMM "\n\tds_read_b64_tr_b16 %5, %13 offset:" #IMM                          \
                 "\n\tds_read_b64_tr_b16 %6, %14 offset:" #IMM "\n\tds_read_b64_tr_b16 %7, %15 offset:" #IMM "\n\ts_waitcnt lgkmcnt(0)" \
                 : "=&v"(X[0]), "=&v"(X[1]), "=&v"(X[2]), "=&v"(X[3]), "=&v"(X[4]), "=&v"(X[5]), "=&v"(X[6]), "=&v"(X[7])           \
                 : "v"(A[0]), "v"(A[1]), "v"(A[2]), "v"(A[3]), "v"(A[4]), "v"(A[5]), "v"(A[6]), "v"(A[7])                           \
                 : "memory")

template <int D, int NW_>
__global__ __launch_bounds__(64 * NW_, 1) void cross_absorbed_v2_kernel(AbsParams p) {
    typedef AbsCfg2<D, NW_> X;
    constexpr int NWV = X::NWV, GF = X::GF, ROWB = X::ROWB, SLOT = X::SLOT;
    constexpr int NDMA = SLOT / 1024;  // LDS-DMA transfers per group (24 for d = 768)
    constexpr int KS = D / 32;         // k-steps of the score product
    constexpr int CT = D / 16;         // column tiles of O'
    static_assert(SLOT % 1024 == 0 && NDMA <= 32 && CT % 8 == 0 && KS % 4 == 0 && CT / 8 <= 8 && D % 128 == 0, "width");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int split = blockIdx.x, b = blockIdx.y;
    const int groups_total = (p.Tk + GF - 1) / GF;
    const int gps = 2 * p.tiles_per_split;  // 16-frame groups per split (tiles_per_split counts 32-frame tiles)
    const int g0 = split * gps;
    const int ng = max(0, min(gps, groups_total - g0));            // groups of this split
    const int my_n = ng > wave ? (ng - wave + NWV - 1) / NWV : 0;  // this wave takes groups wave, wave + 3, ...
    const __bf16* xb = p.xa + (int64_t)b * p.Tk * D;
    char* my = smem + wave * (2 * SLOT);

    bf16x8 qf[KS];  // absorbed queries, all k-steps: B operand of the scores, lane (head l15, k-group g)
    {
        const __bf16* qr = p.qp + ((int64_t)b * 16 + l15) * D + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qr + 32 * ks);
    }
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, 0x7fffffff, 0x00020000);
    // LDS-DMA: transfer i covers image bytes [1024 i, +1024) of a slot; a lane's 16 bytes sit at image (row, chunk') and come
    // from source chunk chunk' ^ swz(row).  1024-byte transfers against ROWB-byte rows repeat every PER transfers = QR rows
    // (3 : 2 for d = 768): transfer PER j + t puts the lane on row QR j + r_t, chunk c_t, so its source offset is
    //   (f0 + QR j) ROWB  [scalar]  +  r_t ROWB + ((c_t ^ 2 r_t) << 4 ^ S_j << 4)   with S_j = ((QR j) & 7) << 1 a constant:
    // one v_xad per transfer from two registers per t.  (Recomputing row and chunk per transfer -- 17 instructions, two of
    // them quarter rate -- was ~1 us of issue time per 24 KiB group and wave, a quarter of the kernel.)  A group that hangs
    // over the end of the clip (the last one) clamps its rows the slow way.
    constexpr int G1K = (ROWB % 1024 == 0) ? 1024 : ((ROWB % 512 == 0) ? 512 : 256);  // gcd(1024, ROWB)
    constexpr int PER = ROWB / G1K, QR = 1024 / G1K;
    static_assert(NDMA % PER == 0 && PER <= 3 && (QR == 1 || QR == 2 || QR == 4), "transfer / row period");
    int dma_a[PER], dma_b[PER];
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int off = 1024 * t + 16 * lane;
        const int r = off / ROWB, c = (off - r * ROWB) >> 4;
        dma_a[t] = (c ^ (r << 1)) << 4;
        dma_b[t] = r * ROWB;
    }
    int lane_v = lane;
    auto stage = [&](int i_local, int slot) {
        const int f0 = (g0 + wave + NWV * i_local) * GF;
        if (f0 + GF <= p.Tk) {
#pragma unroll
            for (int i = 0; i < NDMA; ++i) {
                const int j = i / PER, t = i % PER;
                const int sj = (((QR * j) & 7) << 1) << 4;
                const int voff = (dma_a[t] ^ sj) + dma_b[t];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_ptr_a)(my + slot * SLOT + 1024 * i), 16, voff, (f0 + QR * j) * ROWB, 0, 0);
            }
        } else {
            asm volatile("" : "+v"(lane_v));
#pragma unroll
            for (int i = 0; i < NDMA; ++i) {
                const int off = 1024 * i + 16 * lane_v;
                const int row = off / ROWB;
                const int ch = ((off - row * ROWB) >> 4) ^ swz(row);
                const int voff = min(f0 + row, p.Tk - 1) * ROWB + ch * 16;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_ptr_a)(my + slot * SLOT + 1024 * i), 16, voff, 0, 0, 0);
            }
        }
    };
    const unsigned lds_my = (unsigned)(uintptr_t)(lds_ptr_a)my;
    // per-lane LDS addresses (slot 0).  Scores: row l15, k-step ks -> 256 (ks >> 2) + 64 ((ks & 3) ^ t) + 16 (g ^ (sw & 3)): four
    // bases + an immediate.  Transposed reads: rows 4g + q, column tile 8 blk + cc -> 256 blk + 32 (cc ^ k) + 16 (pp >> 1) + 8 (pp & 1):
    // eight bases + an immediate.
    const char* sb[4];
    {
        const int sw = swz(l15), t = (sw >> 2) & 3;
#pragma unroll
        for (int u = 0; u < 4; ++u) sb[u] = my + l15 * ROWB + 64 * (u ^ t) + 16 * (g ^ (sw & 3));
    }
    unsigned ta[8];
    {
        const int q = l15 >> 2, pp = l15 & 3, r1 = 4 * g + q, k = r1 & 7;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) ta[cc] = lds_my + r1 * ROWB + 32 * (cc ^ k) + 16 * (pp >> 1) + 8 * (pp & 1);
    }
    auto wait_slot = [&](bool more_in_flight) {
        if (more_in_flight) {
            if constexpr (NDMA == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            else if constexpr (NDMA == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else if constexpr (NDMA == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if constexpr (NDMA == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    // scores S^T[frame 4g + r][head l15] of one group over all channels (four interleaved accumulation chains), frames past the
    // end masked; mx = the head's maximum over the group's 16 frames
    auto scores = [&](int slot, int i_local, float& mx) -> f32x4 {
        f32x4 s4[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        const int so = slot * SLOT;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(sb[ks & 3] + so + 256 * (ks >> 2));
            s4[ks & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[ks], s4[ks & 3], 0, 0, 0);
        }
        f32x4 s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        const int f0 = (g0 + wave + NWV * i_local) * GF;
        mx = NEG_BIG;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (f0 + 4 * g + r >= p.Tk) s[r] = NEG_BIG;
            mx = fmaxf(mx, s[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        return s;
    };
    // Softmax against a FIXED per-head reference m_ref (the maximum over the wave's first group) instead of a running maximum:
    // p = exp(s - m_ref) may exceed 1 -- bf16 and f32 carry the range -- and O' is never rescaled.  The 192 accumulators live in
    // the accumulation registers; any conditional multiply of them makes the compiler copy all of them to vector registers at
    // the top of every iteration and spill the queries.  If a later group's maximum climbs more than DRIFT above the reference
    // (e^40 x 1500 frames x |xa| stays far inside f32), the wave starts over: one scores-only sweep finds its exact maximum,
    // then the stream is redone against that.  Same input -> same path -> same bits.
    constexpr float DRIFT = 40.f;
    f32x4 acc[CT];
    float m_ref = NEG_BIG, l_run = 0.f;
    bool exact = false;
    for (;;) {
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        l_run = 0.f;
        bool drifted = false;
        if (my_n > 0) stage(0, 0);
        if (my_n > 1) stage(1, 1);
        for (int i = 0; i < my_n; ++i) {
            const int slot = i & 1;
            wait_slot(i + 1 < my_n);
            float mx;
            const f32x4 s = scores(slot, i, mx);
            if (i == 0 && !exact) m_ref = mx;
            if (__builtin_amdgcn_ballot_w64(mx > m_ref + DRIFT) != 0) {
                drifted = true;
                break;
            }
            float ls = 0.f;
            bf16x4v pf;  // A operand of P x group (16x16x16): row = head l15, k = frames 4g .. 4g + 3 -- the lane's own values
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = (s[r] <= -1.0e29f) ? 0.f : __expf(s[r] - m_ref);
                pf[r] = (__bf16)e;
                ls += e;
            }
            ls += __shfl_xor(ls, 16, 64);
            ls += __shfl_xor(ls, 32, 64);
            l_run += ls;
            // ---- O'[head][channel] += P x group: ONE transposed read per column tile (rows = frames 4g + q, 16 channels)
            {
                unsigned tas[8];
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) tas[cc] = ta[cc] + slot * SLOT;
#define WIPA_PV_BLK(BLK, IMM)                                                                                                       \
    if constexpr (CT / 8 > BLK) {                                                                                                   \
        v2i32 x[8];                                                                                                                 \
        WIPA_TR8(x, tas, IMM);                                                                                                      \
        _Pragma("unroll") for (int cc = 0; cc < 8; ++cc) acc[8 * BLK + cc] =                                                       \
            __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pf, __builtin_bit_cast(bf16x4v, x[cc]), acc[8 * BLK + cc], 0, 0, 0);         \
    }
                WIPA_PV_BLK(0, 0)
                WIPA_PV_BLK(1, 256)
                WIPA_PV_BLK(2, 512)
                WIPA_PV_BLK(3, 768)
                WIPA_PV_BLK(4, 1024)
                WIPA_PV_BLK(5, 1280)
                WIPA_PV_BLK(6, 1536)
                WIPA_PV_BLK(7, 1792)
#undef WIPA_PV_BLK
            }
            // every read of this slot has completed (score fragments are spent, the transposed reads were waited for): re-stage
            asm volatile("" ::: "memory");
            if (i + 2 < my_n) stage(i + 2, slot);
        }
        if (!drifted) break;
        // rare: exact maximum of this wave's frames, scores only
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float mt = NEG_BIG;
        stage(0, 0);
        if (my_n > 1) stage(1, 1);
        for (int i = 0; i < my_n; ++i) {
            const int slot = i & 1;
            wait_slot(i + 1 < my_n);
            float mx;
            (void)scores(slot, i, mx);
            mt = fmaxf(mt, mx);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (i + 2 < my_n) stage(i + 2, slot);
        }
        m_ref = mt;
        exact = true;
    }
    const float m_run = m_ref;
    // ---- merge the three waves' (m, l, O') through LDS: one partial per workgroup
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // rows padded by 4 floats: the four lane groups of a wave write rows 4 apart, (D + 4) * 16 bytes = 64 mod 256 spreads them over
    // the banks; one float4 column per thread and one head per sweep in the sum (NWV * 64 = D / 4 threads)
    constexpr int RS = X::MROW;
    float* so = reinterpret_cast<float*>(smem);                   // [NWV][16 heads][RS]
    float* sm = so + NWV * 16 * RS;                               // [NWV][16] m, then [NWV][16] l
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) so[(wave * 16 + 4 * g + r) * RS + 16 * c + l15] = acc[c][r];
    if (g == 0) {
        sm[wave * 16 + l15] = m_run;
        sm[NWV * 16 + wave * 16 + l15] = l_run;
    }
    __syncthreads();
    const int64_t ps = (int64_t)b * p.n_splits + split;
    float* po = p.part_o + ps * 16 * D;
    for (int hd = 0; hd < p.H; ++hd) {  // rows of the padded heads are never read by the merge
        float mv[NWV], M = NEG_BIG;
#pragma unroll
        for (int v = 0; v < NWV; ++v) {
            mv[v] = sm[v * 16 + hd];
            M = fmaxf(M, mv[v]);
        }
        for (int c4 = tid; 4 * c4 < D; c4 += NWV * 64) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int v = 0; v < NWV; ++v)  // fixed order
                o += __expf(mv[v] - M) * *reinterpret_cast<const f32x4*>(so + (v * 16 + hd) * RS + 4 * c4);
            *reinterpret_cast<f32x4*>(po + hd * D + 4 * c4) = o;
        }
    }
    if (tid < 16) {
        float M = NEG_BIG, Lsum = 0.f;
#pragma unroll
        for (int v = 0; v < NWV; ++v) M = fmaxf(M, sm[v * 16 + tid]);
#pragma unroll
        for (int v = 0; v < NWV; ++v) Lsum += __expf(sm[v * 16 + tid] - M) * sm[NWV * 16 + v * 16 + tid];
        p.part_m[ps * 16 + tid] = M;
        p.part_l[ps * 16 + tid] = Lsum;
    }
}
#undef WIPA_TR8

// Qp[b][h][c] = scale * sum_j q[b][h*64 + j] WkT[c][h*64 + j]     (WkT = Wk^T, [d_in][d_out]: 16-byte loads along j)
// one WAVE per (head 0..15, 16 clips, channel quarter): 192 short independent chains instead of 48 workgroups; heads >= H are
// the zero rows of the 16-wide MFMA dimension
template <int D>
__global__ __launch_bounds__(64) void cross_absorb_q_kernel(const __bf16* __restrict__ q, int64_t q_rs, const __bf16* __restrict__ wkT,
                                                            __bf16* __restrict__ qp, int B, int H, float scale) {
    constexpr int CT = D / 4 / 16;
    const int lane = threadIdx.x & 63, wave = blockIdx.z;
    const int l15 = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, b0 = blockIdx.y * 16;
    if (h >= H) {
        for (int e = lane; e < 16 * (D / 4); e += 64) {
            const int bb = b0 + e / (D / 4);
            if (bb < B) qp[((int64_t)bb * 16 + h) * D + wave * (D / 4) + e % (D / 4)] = (__bf16)0.f;
        }
        return;
    }
    const int bq = min(b0 + l15, B - 1);
    bf16x8 qa[2];  // A operand: row = clip l15, k = j
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qa[ks] = *reinterpret_cast<const bf16x8*>(q + (int64_t)bq * q_rs + h * 64 + 32 * ks + 8 * g);
    bf16x8 wb[CT][2];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            wb[i][ks] = *reinterpret_cast<const bf16x8*>(wkT + (int64_t)(wave * (D / 4) + 16 * i + l15) * D + h * 64 + 32 * ks + 8 * g);
#pragma unroll
    for (int i = 0; i < CT; ++i) {
        const int c0 = wave * (D / 4) + 16 * i;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[ks], wb[i][ks], acc, 0, 0, 0);
        // acc[r] = Qp[clip b0 + 4g + r][channel c0 + l15]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int bb = b0 + 4 * g + r;
            if (bb < B) qp[((int64_t)bb * 16 + h) * D + c0 + l15] = (__bf16)(acc[r] * scale);
        }
    }
}

// merge of the frame splits (fixed order) and the absorbed value projection: out[b][h*64 + j] = (O'_h / l_h) . Wv[h*64 + j] + bv.
// One workgroup per (head, 4 clips): the clips are rows of the MFMA tile, the head's 64 outputs four column tiles, and the waves
// each take a 96-channel slice of the reduction dimension: a lane merges exactly the 8-channel pieces of its clip that its A
// fragments need (4 splits x 2 float4 per k-step, rounded through bf16, the activation dtype), the weight fragments come straight
// from L2, and the slice sums meet in LDS in a fixed order.  Measured alternatives at 64 clips: one workgroup per (head, clip)
// reading the head's 98 KB of weights each -- 75 MB of L2 reads, 8.7 us; one per (head, 16 clips) -- 48 workgroups pull the
// 9.4 MB of partials through 48 CUs, 11.7 us.  Compile-time variants of this form (-DWIPA_MERGE_KQ / -DWIPA_MERGE_CL): 192 channels
// per wave 10.9 us, 2 clips per workgroup 13.8, 8 clips 10.8, 8 clips x 192 channels 11.5 -- all within a microsecond of the 10.4 us
// of (96, 4): the kernel sits on its chain of dependent round trips (weights and partials in, LDS reduction, out), not on its shape.
template <int D>
struct MergeCfg {
#ifndef WIPA_MERGE_KQ
#define WIPA_MERGE_KQ 96
#endif
#ifndef WIPA_MERGE_CL
#define WIPA_MERGE_CL 4
#endif
    static constexpr int KQ = (D % 96 == 0) ? WIPA_MERGE_KQ : (D == 512 ? 64 : 128);  // channels per wave
    static constexpr int NWM = D / KQ;                                       // waves: 4 (d = 384), 8 (512, 768, 1024)
    static constexpr int KS = KQ / 32;
    static constexpr int CL = WIPA_MERGE_CL;  // clips per workgroup (1, 2, 4, 8 or 16)
};

// Every lane computes the split weights of its own clip from the statistics it loaded itself.  BANNED here (DESIGN.md section
// 8.1): one thread loading the statistics and computing the weights of the workgroup's clips into LDS for everybody to read
// behind a barrier -- in this kernel that form gave whole wrong (head, 4-clip) workgroups on bit-identical inputs (round 3 / 4;
// its seven diagnostic variants and their ISA are in the history at commit 65cd433, the audit in profiles/r05_merge_isa_audit.txt;
// no mechanism was found, so the form stays out).
// OUTP (round 4): the cross-attention OUT projection of the head rides in the same launch.  The head's value vector
// v_h[clip][64] (bf16, with bv: exactly what the plain kernel stores) stays in LDS and is multiplied by Wo[:, h*64 .. h*64+63]^T on
// MFMA (48 column tiles over the waves, the fragments requested with the first loads of the kernel): slab_h[clip][0..d) in f32,
// one slab per head at slabs_out + h * slab_stride (head 0 carries the out-projection bias).  The next LayerNorm sums x + the H
// slabs in head order (deterministic; add_slabs_layernorm takes up to 16).  One launch instead of two per layer.
template <int D, bool OUTP = false>
__global__ __launch_bounds__(64 * MergeCfg<D>::NWM) void cross_merge_proj_kernel(
    const float* __restrict__ part_m, const float* __restrict__ part_l, const float* __restrict__ part_o, int n_splits,
    const __bf16* __restrict__ wv, const float* __restrict__ bv, __bf16* __restrict__ out, int64_t o_rs, int B,
    const __bf16* __restrict__ wo = nullptr, const float* __restrict__ bo = nullptr, float* __restrict__ slabs_out = nullptr,
    int64_t slab_stride = 0) {
    constexpr int KQ = MergeCfg<D>::KQ, KS = MergeCfg<D>::KS, NWM = MergeCfg<D>::NWM;
    static_assert(KQ % 32 == 0 && NWM * KQ == D, "width");
    constexpr int CL = MergeCfg<D>::CL;
    constexpr int NTO = D / 16;                       // column tiles of the out projection
    constexpr int NTOW = (NTO + NWM - 1) / NWM;       // per wave (6 for d = 768)
    __shared__ __attribute__((aligned(16))) float red[NWM][CL][64 + 4];
    __shared__ __attribute__((aligned(16))) __bf16 vh[OUTP ? CL : 1][64 + 8];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, b0 = blockIdx.y * CL;
    // this lane's clip (A rows): only CL of the tile's 16 rows are distinct clips -- the others repeat them (same addresses, one
    // fetch) and their results are dropped; rows past B are clamped the same way
    const int bc = min(b0 + (l15 & (CL - 1)), B - 1);
    // every load of the kernel is requested before the first use (the partials were just written by other XCDs: each miss is a
    // trip to memory): weight fragments -- column tile nt, k-step ks -> Wv[h*64 + 16 nt + l15][w KQ + 32 ks + 8 g ..] ...
    bf16x8 wq[4][KS];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            wq[nt][ks] = *reinterpret_cast<const bf16x8*>(wv + (int64_t)(h * 64 + 16 * nt + l15) * D + w * KQ + 32 * ks + 8 * g);
    // (OUTP) Wo fragments of this wave's column tiles: tile nt, k-step ks -> Wo[16 nt + l15][h*64 + 32 ks + 8 g ..]
    bf16x8 wof[OUTP ? NTOW : 1][2];
    if constexpr (OUTP) {
#pragma unroll
        for (int i = 0; i < NTOW; ++i) {
            const int nt = min(w * NTOW + i, NTO - 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                wof[i][ks] = *reinterpret_cast<const bf16x8*>(wo + (int64_t)(16 * nt + l15) * D + h * 64 + 32 * ks + 8 * g);
        }
    }
    // ... the split statistics of this lane's clip ...
    float pm[4], pl[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int sc = min(s, n_splits - 1);
        pm[s] = part_m[((int64_t)bc * n_splits + sc) * 16 + h];
        pl[s] = part_l[((int64_t)bc * n_splits + sc) * 16 + h];
    }
    // ... and the 8-channel pieces of the partial rows its A fragments are made of
    f32x4 po[KS][4][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int sc = min(s, n_splits - 1);
            const float* pr = part_o + (((int64_t)bc * n_splits + sc) * 16 + h) * D + w * KQ + 32 * ks + 8 * g;
            po[ks][s][0] = *reinterpret_cast<const f32x4*>(pr);
            po[ks][s][1] = *reinterpret_cast<const f32x4*>(pr + 4);
        }
    __builtin_amdgcn_sched_barrier(0);
    // split weights exp(m_s - M) / L: every lane computes them for itself (no shared array, no single-thread section: see the
    // determinism test)
    float ws[4];
    {
        float M = NEG_BIG;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            if (s < n_splits) M = fmaxf(M, pm[s]);
        float Lsum = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            ws[s] = s < n_splits ? __expf(pm[s] - M) : 0.f;
            Lsum += ws[s] * pl[s];
        }
        const float inv = 1.0f / Lsum;
#pragma unroll
        for (int s = 0; s < 4; ++s) ws[s] *= inv;
    }
    f32x4 acc[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {  // fixed order; splits past n_splits carry weight 0
            lo += ws[s] * po[ks][s][0];
            hi += ws[s] * po[ks][s][1];
        }
        bf16x8 a;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[e] = (__bf16)lo[e];
            a[4 + e] = (__bf16)hi[e];
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wq[nt][ks], acc[nt], 0, 0, 0);
    }
    // acc[nt][r] = this wave's share of the sum for tile row 4g + r (clip b0 + r in lane group 0), output 16 nt + l15
    if (4 * g < CL) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * g + r < CL) red[w][4 * g + r][16 * nt + l15] = acc[nt][r];
    }
    __syncthreads();
    for (int e = tid; e < CL * 64; e += 64 * NWM) {
        const int row = e >> 6, j = e & 63;
        float o = 0.f;
#pragma unroll
        for (int v = 0; v < NWM; ++v) o += red[v][row][j];  // fixed order
        const __bf16 val = (__bf16)(o + bv[h * 64 + j]);
        if constexpr (OUTP) vh[row][j] = val;
        else if (b0 + row < B) out[(int64_t)(b0 + row) * o_rs + h * 64 + j] = val;
    }
    if constexpr (OUTP) {
        __syncthreads();
        // A rows = clips (row l15 & (CL - 1): the tile's other rows repeat them, their results are dropped), k = the head's 64 values
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&vh[l15 & (CL - 1)][8 * g]);
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&vh[l15 & (CL - 1)][32 + 8 * g]);
#pragma unroll
        for (int i = 0; i < NTOW; ++i) {
            const int nt = w * NTOW + i;
            if (nt < NTO) {
                f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, wof[i][0], acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, wof[i][1], acc2, 0, 0, 0);
                // acc2[r] = slab_h[clip b0 + 4 g + r][16 nt + l15]: lane group 0 holds the CL real rows
                if (4 * g < CL) {
                    const float bias = (h == 0) ? bo[16 * nt + l15] : 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * g + r < CL && b0 + 4 * g + r < B)
                            slabs_out[(int64_t)h * slab_stride + (int64_t)(b0 + 4 * g + r) * D + 16 * nt + l15] = acc2[r] + bias;
                }
            }
        }
    }
}

// Prologue of the absorbed cross block in ONE launch (instead of add_slabs_layernorm + query GEMM + cross_absorb_q): one
// workgroup per (head, 16 clips), eight waves.
//   1. rows: x = x_in + slabs (fixed order) -> LayerNorm -> bf16 in LDS; the h = 0 workgroup writes x to x_out (the other
//      residual buffer: the 12 head workgroups of a clip group all read the row one of them rewrites) and zero-fills the padded
//      heads' rows of Qp;
//   2. q_h[16 clips][64] = LN(x) Wq_h^T + bq_h, times the query's share of the score scale: MFMA, the waves split K (24 k-steps),
//      partial tiles meet in LDS in a fixed order, rounded to bf16 like the separate query GEMM's output;
//   3. Qp_h[16 clips][d] = k_scale * q_h Wk_h: MFMA, the waves split the d / 16 column tiles; WkT rows are the B fragments.
// Weight fragments of steps 2 and 3 are requested before step 1 (196 KB per workgroup, from L2).
struct AbsPrologueParams {
    const float* x_in;
    float* x_out;
    const float* slabs;
    const float* ln_w;
    const float* ln_b;
    const __bf16* wq;    // [d][d]  ([out][in])
    const float* bq;     // [d]
    const __bf16* wkT;   // [d][d]  ([channel][out])
    __bf16* qp;          // [B][16][d]
    int64_t slab_stride;
    int n_slabs, B, H;
    float eps, q_scale, k_scale;
};

template <int D, int CG>  // CG clips per workgroup: 16 (every MFMA row a clip) or 8 (twice the workgroups, half the rows each)
__global__ __launch_bounds__(512) void cross_absorb_prologue_kernel(AbsPrologueParams p) {
    constexpr int KS = D / 32;          // k-steps of the query projection
    constexpr int KSW = (KS + 7) / 8;   // per wave
    constexpr int NT = D / 16;          // column tiles of Qp
    constexpr int NTW = (NT + 7) / 8;   // per wave
    constexpr int LROW = D + 8;         // padded LDS row of the normalised activations (bf16)
    constexpr int V4 = D / 256;         // float4 per lane of a row (D / 4 / 64)
    static_assert(D % 256 == 0 || D == 384, "width");
    __shared__ __attribute__((aligned(16))) __bf16 lnx[16][LROW];
    __shared__ __attribute__((aligned(16))) float red[8][16][64 + 4];
    __shared__ __attribute__((aligned(16))) __bf16 qh[16][64 + 8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, b0 = blockIdx.y * CG;
    // ---- weight fragments first: Wq rows h*64 + 16 nt + l15 at this wave's k-steps; WkT rows (channels) of this wave's column tiles
    bf16x8 wqf[4][KSW];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int i = 0; i < KSW; ++i) {
            const int ks = min(wave * KSW + i, KS - 1);
            wqf[nt][i] = *reinterpret_cast<const bf16x8*>(p.wq + (int64_t)(h * 64 + 16 * nt + l15) * D + 32 * ks + 8 * g);
        }
    bf16x8 wkf[NTW][2];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int nt = min(wave * NTW + i, NT - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            wkf[i][ks] = *reinterpret_cast<const bf16x8*>(p.wkT + (int64_t)(16 * nt + l15) * D + h * 64 + 32 * ks + 8 * g);
    }
    // ---- 1. slab sum + LayerNorm of rows 2 wave, 2 wave + 1 (one wave per row: 4 consecutive floats per lane and 256-column block)
    constexpr int NV = (D + 255) / 256;
#pragma unroll
    for (int rr = 0; rr < CG / 8; ++rr) {
        const int row = (CG / 8) * wave + rr;
        const int b = min(b0 + row, p.B - 1);
        f32x4 v[NV], ww[NV], bb[NV];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = 4 * lane + 256 * i;
            v[i] = ww[i] = bb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < D) {
                v[i] = *reinterpret_cast<const f32x4*>(p.x_in + (int64_t)b * D + c);
                f32x4 sl[4];
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    sl[s] = (s < p.n_slabs) ? *reinterpret_cast<const f32x4*>(p.slabs + (int64_t)s * p.slab_stride + (int64_t)b * D + c)
                                            : f32x4{0.f, 0.f, 0.f, 0.f};
                ww[i] = *reinterpret_cast<const f32x4*>(p.ln_w + c);
                bb[i] = *reinterpret_cast<const f32x4*>(p.ln_b + c);
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    if (s < p.n_slabs) v[i] += sl[s];  // fixed order
                sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
                if (h == 0 && b0 + row < p.B) *reinterpret_cast<f32x4*>(p.x_out + (int64_t)b * D + c) = v[i];
            }
        }
        const float mean = wave_reduce_sum(sum) / (float)D;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (4 * lane + 256 * i < D) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dv = v[i][e] - mean;
                    sq += dv * dv;
                }
            }
        const float rstd = rsqrtf(wave_reduce_sum(sq) / (float)D + p.eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = 4 * lane + 256 * i;
            if (c < D) {
#pragma unroll
                for (int e = 0; e < 4; ++e) lnx[row][c + e] = (__bf16)((v[i][e] - mean) * rstd * ww[i][e] + bb[i][e]);
            }
        }
    }
    if (h == 0) {  // padded heads of Qp: zero (the streaming kernel multiplies all 16 rows of the head dimension)
        for (int e = tid; e < CG * (16 - p.H) * (D / 8); e += 512) {
            const int row = e / ((16 - p.H) * (D / 8)), rem = e - row * ((16 - p.H) * (D / 8));
            const int hp = p.H + rem / (D / 8), c8 = rem % (D / 8);
            if (b0 + row < p.B) *reinterpret_cast<bf16x8*>(p.qp + ((int64_t)(b0 + row) * 16 + hp) * D + 8 * c8) = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    __syncthreads();
    // ---- 2. query of the head: partial over this wave's k-steps
    {
        f32x4 acc[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int i = 0; i < KSW; ++i) {
            const int ks = wave * KSW + i;
            if (ks < KS) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(&lnx[l15][32 * ks + 8 * g]);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wqf[nt][i], acc[nt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][4 * g + r][16 * nt + l15] = acc[nt][r];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < CG / 8; ++i) {
        const int e = tid + 512 * i, row = e >> 6, j = e & 63;
        float q = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) q += red[w][row][j];  // fixed order
        qh[row][j] = (__bf16)((q + p.bq[h * 64 + j]) * p.q_scale);
    }
    if constexpr (CG < 16) {  // the unused MFMA rows: defined values (their products are dropped)
        for (int e = tid; e < (16 - CG) * 64; e += 512) qh[CG + (e >> 6)][e & 63] = (__bf16)0.f;
    }
    __syncthreads();
    // ---- 3. absorbed query: this wave's column tiles
    {
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&qh[l15][8 * g]);
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&qh[l15][32 + 8 * g]);
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const int nt = wave * NTW + i;
            if (nt < NT) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, wkf[i][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, wkf[i][1], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int bb_ = b0 + 4 * g + r;
                    if (4 * g + r < CG && bb_ < p.B) p.qp[((int64_t)bb_ * 16 + h) * D + 16 * nt + l15] = (__bf16)(acc[r] * p.k_scale);
                }
            }
        }
    }
}

// WIPA_ABS_KERNEL=1 keeps the channel-split kernel (A/B runs); default: the independent-wave kernel
template <int D>
int launch_attn(const AbsParams& p_in, int B, hipStream_t s) {
    const char* e = getenv("WIPA_ABS_KERNEL");
    const AbsParams& p = p_in;
    // d = 1024: 64 column tiles are all 256 accumulation registers and a 16-frame group is 32 KiB, so the independent-wave form fits
    // with TWO waves only -- measured slower than the channel-split form there (whisper-medium, 256 clips: 191.5 vs 171.7 us per
    // launch, 845 vs 821 ms per pass); WIPA_ABS_KERNEL=2 selects it for A/B runs
    if ((e && atoi(e) == 1) || (D > 768 && !(e && atoi(e) == 2)))
        hipLaunchKernelGGL((cross_absorbed_kernel<D>), dim3(p.n_splits, B), dim3(64 * AbsCfg<D>::NW), AbsCfg<D>::SMEM, s, p);
    else if constexpr (D > 768)
        hipLaunchKernelGGL((cross_absorbed_v2_kernel<D, 2>), dim3(p.n_splits, B), dim3(128), (AbsCfg2<D, 2>::SMEM), s, p);
    else {
        static const int waves = [] { const char* w = getenv("WIPA_ABS_WAVES"); return w ? atoi(w) : 3; }();  // A/B: 2 leaves 64 KiB of LDS and two SIMDs to other kernels
        if (waves == 2) hipLaunchKernelGGL((cross_absorbed_v2_kernel<D, 2>), dim3(p.n_splits, B), dim3(128), (AbsCfg2<D, 2>::SMEM), s, p);
        else hipLaunchKernelGGL((cross_absorbed_v2_kernel<D, 3>), dim3(p.n_splits, B), dim3(192), (AbsCfg2<D, 3>::SMEM), s, p);
    }
    return WIPA_OK;
}

}  // namespace

// Frame splits per clip: a clip's result must not depend on the batch it rides in (the partition of the frames fixes the order of
// the softmax merges), so the count is a property of the CALL, never of B.  FOUR by default: a workgroup streams 375 frames at the
// same per-CU rate whether 1 or 256 clips are decoded, and 64 clips x 4 splits are exactly one round on the 256 CUs (the kernel
// holds a CU: 145 KiB of LDS).  Fewer splits = fewer, longer workgroups: measured r04 (whisper-small, 64 clips; lone launch /
// lone step / pass with 4 passes in flight): 4: 32.4 us / 1.253 ms / 75.3 ms; 3: 32.7 / 1.261 / 73.5 (192 CUs already reach the
// HBM-side limit of this access path); 2: 40.4 / 1.352 / 72.3 (128 CUs run at their own 34 GB/s each, and the streaming kernels
// of two passes run side by side instead of queueing for the whole chip); 1: 66.1 / 1.674 / 72.1.  Short inputs: at least two
// 32-frame tiles per split.  WIPA_ABS_SPLITS overrides the DEFAULT (want = 0) for A/B runs.
extern "C" int wipa_cross_absorbed_splits(int want, int Tk) {
    const int tiles = (Tk + FT - 1) / FT;
    static const int dflt = [] { const char* e = getenv("WIPA_ABS_SPLITS"); const int v = e ? atoi(e) : 4; return (v >= 1 && v <= 4) ? v : 4; }();
    int s = (want >= 1 && want <= 4) ? want : dflt;
    if (s > tiles / 2) s = tiles / 2;
    return s < 1 ? 1 : s;
}

// sized for four splits whatever a call uses: one state blob serves every dec_cross_splits
extern "C" size_t wipa_cross_absorbed_scratch_bytes(int B, int d, int Tk) {
    (void)Tk;
    return (size_t)B * 16 * d * 2 + (size_t)B * 4 * 16 * (2 + (size_t)d) * 4 + 1024;
}

extern "C" int wipa_cross_absorbed_init(int d) {
    // raise the dynamic-LDS limit outside any stream capture; once per width and process (the driver calls are not free and a
    // caller may sit inside a capture: after the first call this function issues nothing)
    static std::mutex mu;
    static bool done[4] = {false, false, false, false};
    const int wi = d == 384 ? 0 : d == 512 ? 1 : d == 768 ? 2 : d == 1024 ? 3 : -1;
    if (wi < 0) return WIPA_ERR_ARG;
    std::lock_guard<std::mutex> lk(mu);
    if (done[wi]) return WIPA_OK;
    hipError_t e = hipSuccess;
#define ABS_ATTR(D)                                                                                                                        \
    {                                                                                                                                      \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cross_absorbed_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize,      \
                                AbsCfg<D>::SMEM);                                                                                          \
        if (e == hipSuccess && D <= 768)                                                                                                   \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cross_absorbed_v2_kernel<(D <= 768 ? D : 768), 3>),                     \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, AbsCfg2<(D <= 768 ? D : 768), 3>::SMEM);                   \
        if (e == hipSuccess)                                                                                                               \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cross_absorbed_v2_kernel<D, 2>),                                        \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, AbsCfg2<D, 2>::SMEM);                                      \
    }
    if (d == 384) ABS_ATTR(384)
    else if (d == 512) ABS_ATTR(512)
    else if (d == 768) ABS_ATTR(768)
    else if (d == 1024) ABS_ATTR(1024)
    else return WIPA_ERR_ARG;
#undef ABS_ATTR
    WIPA_CHECK_HIP(e);
    done[wi] = true;
    return WIPA_OK;
}

// the streaming kernel alone on a scratch whose Qp a wipa_cross_absorbed_attention call has filled (measurement aid: bench.py
// times the dominant kernel of the decode step by itself)
extern "C" int wipa_cross_absorbed_stream(const void* xa, void* scratch, size_t scratch_bytes, int B, int H, int d, int Tk, int n_splits,
                                          wipa_stream_t stream) {
    WIPA_REQUIRE(xa && scratch && B > 0 && H >= 1 && H <= 16 && d == H * 64 && (d == 384 || d == 512 || d == 768 || d == 1024) && Tk >= 1 &&
                     n_splits >= 0 && n_splits <= 4,
                 "wipa_cross_absorbed_stream: bad arguments");
    WIPA_REQUIRE(scratch_bytes >= wipa_cross_absorbed_scratch_bytes(B, d, Tk), "wipa_cross_absorbed_stream: scratch too small");
    const int rc0 = wipa_cross_absorbed_init(d);
    if (rc0 != WIPA_OK) return rc0;
    const int S = wipa_cross_absorbed_splits(n_splits, Tk);
    char* sc = (char*)scratch;
    AbsParams p = {};
    p.qp = (const __bf16*)sc; p.xa = (const __bf16*)xa;
    p.part_m = (float*)(sc + (size_t)B * 16 * d * 2);
    p.part_l = p.part_m + (size_t)B * S * 16;
    p.part_o = p.part_l + (size_t)B * S * 16;
    p.Tk = Tk; p.n_splits = S; p.H = H;
    const int tiles = (Tk + FT - 1) / FT;
    p.tiles_per_split = (tiles + S - 1) / S;
    int rc;
    if (d == 384) rc = launch_attn<384>(p, B, (hipStream_t)stream);
    else if (d == 512) rc = launch_attn<512>(p, B, (hipStream_t)stream);
    else if (d == 768) rc = launch_attn<768>(p, B, (hipStream_t)stream);
    else rc = launch_attn<1024>(p, B, (hipStream_t)stream);
    if (rc != WIPA_OK) return rc;
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

extern "C" int wipa_cross_absorbed_attention(const void* q, int64_t q_row_stride, const void* wkT, const void* xa, const void* wv,
                                             const float* bv, void* out, int64_t out_row_stride, void* scratch, size_t scratch_bytes,
                                             int B, int H, int d, int Tk, float k_scale, int n_splits, wipa_stream_t stream) {
    WIPA_REQUIRE(q && wkT && xa && wv && bv && out && scratch, "wipa_cross_absorbed_attention: null pointer");
    WIPA_REQUIRE(n_splits >= 0 && n_splits <= 4, "wipa_cross_absorbed_attention: n_splits %d (0 = default, 1..4)", n_splits);
    WIPA_REQUIRE(B > 0 && B <= 65535 && H >= 1 && H <= 16 && d == H * 64 && (d == 384 || d == 512 || d == 768 || d == 1024) && Tk >= 1,
                 "wipa_cross_absorbed_attention: bf16, <= 16 heads of 64, d in {384, 512, 768, 1024} (got H=%d d=%d)", H, d);
    WIPA_REQUIRE(q_row_stride % 8 == 0 && out_row_stride >= d && ((uintptr_t)q % 16) == 0 && ((uintptr_t)xa % 16) == 0 &&
                     ((uintptr_t)wkT % 16) == 0 && ((uintptr_t)wv % 16) == 0 && ((uintptr_t)scratch % 16) == 0,
                 "wipa_cross_absorbed_attention: operands must be 16-byte aligned");
    WIPA_REQUIRE((int64_t)Tk * d * 2 < ((int64_t)1 << 31), "wipa_cross_absorbed_attention: clip too long for 32-bit tile offsets");
    WIPA_REQUIRE(scratch_bytes >= wipa_cross_absorbed_scratch_bytes(B, d, Tk), "wipa_cross_absorbed_attention: scratch too small");
    hipStream_t s = (hipStream_t)stream;
    {
        const int rc0 = wipa_cross_absorbed_init(d);  // once per width
        if (rc0 != WIPA_OK) return rc0;
    }
    const int S = wipa_cross_absorbed_splits(n_splits, Tk);
    char* sc = (char*)scratch;
    __bf16* qp = (__bf16*)sc;
    float* part_m = (float*)(sc + (size_t)B * 16 * d * 2);
    float* part_l = part_m + (size_t)B * S * 16;
    float* part_o = part_l + (size_t)B * S * 16;
    AbsParams p = {};
    p.qp = qp; p.xa = (const __bf16*)xa; p.part_m = part_m; p.part_l = part_l; p.part_o = part_o;
    p.Tk = Tk; p.n_splits = S; p.H = H;
    const int tiles = (Tk + FT - 1) / FT;
    p.tiles_per_split = (tiles + S - 1) / S;
    const dim3 gq(16, (B + 15) / 16, 4), gm(H, (B + WIPA_MERGE_CL - 1) / WIPA_MERGE_CL);
    int rc = WIPA_OK;
    const char* st_env = getenv("WIPA_ABS_STAGES");  // debugging: bit 0 absorb-q, bit 1 stream, bit 2 merge (default all)
    const int stages = st_env ? atoi(st_env) : 7;
#define ABS_RUN(D)                                                                                                                         \
    do {                                                                                                                                   \
        if (stages & 1)                                                                                                                    \
            hipLaunchKernelGGL((cross_absorb_q_kernel<D>), gq, dim3(64), 0, s, (const __bf16*)q, q_row_stride, (const __bf16*)wkT, qp, B, H, k_scale); \
        if (stages & 2) rc = launch_attn<D>(p, B, s);                                                                                      \
        if (rc == WIPA_OK && (stages & 4)) {                                                                                               \
            hipLaunchKernelGGL((cross_merge_proj_kernel<D>), gm, dim3(64 * MergeCfg<D>::NWM), 0, s, part_m, part_l, part_o, S,            \
                               (const __bf16*)wv, bv, (__bf16*)out, out_row_stride, B);                                                    \
        }                                                                                                                                  \
    } while (0)
    if (d == 384) ABS_RUN(384);
    else if (d == 512) ABS_RUN(512);
    else if (d == 768) ABS_RUN(768);
    else ABS_RUN(1024);
#undef ABS_RUN
    if (rc != WIPA_OK) return rc;
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}

// The whole absorbed cross block of a decode step in three launches: [slab sum + residual + cross_attn_ln + cross query + absorbed
// query] -> streaming kernel -> [merge + value projection].  c->kv is the encoder output xa [B][Tk][d]; c->x_out must not alias
// c->x_in (the caller swaps its two residual buffers afterwards, as with wipa_decode_cross_block).
static int absorbed_block(const wipa_cross_block_desc* c, const void* wkT, const void* wv, const float* bv, const void* wo, const float* bo,
                          float* slabs_out, int64_t slab_stride, void* scratch, size_t scratch_bytes, wipa_stream_t stream);

extern "C" int wipa_decode_cross_absorbed_block(const wipa_cross_block_desc* c, const void* wkT, const void* wv, const float* bv, void* scratch,
                                                size_t scratch_bytes, wipa_stream_t stream) {
    WIPA_REQUIRE(c && c->out, "wipa_decode_cross_absorbed_block: null pointer");
    return absorbed_block(c, wkT, wv, bv, nullptr, nullptr, nullptr, 0, scratch, scratch_bytes, stream);
}

// ... and with the cross-attention OUT projection in the third launch: instead of c->out, H split-K slabs
// slabs_out[h * slab_stride + b * d + n] = v_h[b] . Wo[n][h*64 .. h*64+63] (+ bo[n] in slab 0), to be summed onto the residual rows
// by the next wipa_add_slabs_layernorm(n_slabs = H).  slabs_out may be the buffer c->slabs points to (the prologue has read it).
extern "C" int wipa_decode_cross_absorbed_block_out(const wipa_cross_block_desc* c, const void* wkT, const void* wv, const float* bv,
                                                    const void* wo, const float* bo, float* slabs_out, int64_t slab_stride, void* scratch,
                                                    size_t scratch_bytes, wipa_stream_t stream) {
    WIPA_REQUIRE(c && wo && bo && slabs_out && slab_stride >= (int64_t)c->B * c->d && ((uintptr_t)wo % 16) == 0,
                 "wipa_decode_cross_absorbed_block_out: null pointer / short slab stride");
    return absorbed_block(c, wkT, wv, bv, wo, bo, slabs_out, slab_stride, scratch, scratch_bytes, stream);
}

static int absorbed_block(const wipa_cross_block_desc* c, const void* wkT, const void* wv, const float* bv, const void* wo, const float* bo,
                          float* slabs_out, int64_t slab_stride, void* scratch, size_t scratch_bytes, wipa_stream_t stream) {
    WIPA_REQUIRE(c && c->x_in && c->x_out && c->ln_w && c->ln_b && c->wq && c->bq && c->kv && (c->out || wo) && wkT && wv && bv && scratch,
                 "wipa_decode_cross_absorbed_block: null pointer");
    WIPA_REQUIRE(c->x_in != c->x_out, "wipa_decode_cross_absorbed_block: x_out must not alias x_in");
    const int B = c->B, H = c->H, d = c->d, Tk = c->Tk;
    WIPA_REQUIRE(c->dtype == WIPA_BF16 && B > 0 && B <= 65535 * 16 && H >= 1 && H <= 16 && d == H * 64 &&
                     (d == 384 || d == 512 || d == 768 || d == 1024) && Tk >= 1 && c->n_slabs >= 0 && c->n_slabs <= 4 && !c->bias_o &&
                     c->cross_splits >= 0 && c->cross_splits <= 4,
                 "wipa_decode_cross_absorbed_block: bf16, <= 16 heads of 64, d in {384, 512, 768, 1024}, <= 4 slabs, no separate bias, "
                 "cross_splits 0..4");
    WIPA_REQUIRE(c->n_slabs == 0 || c->slabs, "wipa_decode_cross_absorbed_block: slabs missing");
    WIPA_REQUIRE((int64_t)Tk * d * 2 < ((int64_t)1 << 31), "wipa_decode_cross_absorbed_block: clip too long for 32-bit tile offsets");
    WIPA_REQUIRE(scratch_bytes >= wipa_cross_absorbed_scratch_bytes(B, d, Tk), "wipa_decode_cross_absorbed_block: scratch too small");
    hipStream_t s = (hipStream_t)stream;
    const int S = wipa_cross_absorbed_splits(c->cross_splits, Tk);
    char* sc = (char*)scratch;
    __bf16* qp = (__bf16*)sc;
    float* part_m = (float*)(sc + (size_t)B * 16 * d * 2);
    float* part_l = part_m + (size_t)B * S * 16;
    float* part_o = part_l + (size_t)B * S * 16;
    AbsPrologueParams q = {};
    q.x_in = c->x_in; q.x_out = c->x_out; q.slabs = c->slabs; q.ln_w = c->ln_w; q.ln_b = c->ln_b;
    q.wq = (const __bf16*)c->wq; q.bq = c->bq; q.wkT = (const __bf16*)wkT; q.qp = qp;
    q.slab_stride = c->slab_stride; q.n_slabs = c->n_slabs; q.B = B; q.H = H;
    q.eps = c->eps; q.q_scale = c->qk_scale; q.k_scale = c->qk_scale;
    AbsParams p = {};
    p.qp = qp; p.xa = (const __bf16*)c->kv; p.part_m = part_m; p.part_l = part_l; p.part_o = part_o;
    p.Tk = Tk; p.n_splits = S; p.H = H;
    const int tiles = (Tk + FT - 1) / FT;
    p.tiles_per_split = (tiles + S - 1) / S;
    // clips per prologue workgroup: 8 (default since round 4: H x B / 8 = 96 workgroups at 64 clips; a lone decode step 1.304 vs
    // 1.338 ms, the pipelined pass unchanged) or 16 (WIPA_ABS_PROLOGUE_CLIPS=16: every MFMA row a clip, 48 workgroups)
    static const int cg = [] { const char* e = getenv("WIPA_ABS_PROLOGUE_CLIPS"); return e ? atoi(e) : 8; }();
    const dim3 gp(H, cg == 8 ? (B + 7) / 8 : (B + 15) / 16), gm(H, (B + WIPA_MERGE_CL - 1) / WIPA_MERGE_CL);
    int rc = WIPA_OK;
#define ABS_BLOCK(D)                                                                                                                       \
    do {                                                                                                                                   \
        if (cg == 8) hipLaunchKernelGGL((cross_absorb_prologue_kernel<D, 8>), gp, dim3(512), 0, s, q);                                     \
        else hipLaunchKernelGGL((cross_absorb_prologue_kernel<D, 16>), gp, dim3(512), 0, s, q);                                            \
        rc = launch_attn<D>(p, B, s);                                                                                                      \
        if (rc == WIPA_OK && wo)                                                                                                           \
            hipLaunchKernelGGL((cross_merge_proj_kernel<D, true>), gm, dim3(64 * MergeCfg<D>::NWM), 0, s, part_m, part_l, part_o,  \
                               S, (const __bf16*)wv, bv, (__bf16*)nullptr, (int64_t)d, B, (const __bf16*)wo, bo, slabs_out, slab_stride);  \
        else if (rc == WIPA_OK)                                                                                                            \
            hipLaunchKernelGGL((cross_merge_proj_kernel<D>), gm, dim3(64 * MergeCfg<D>::NWM), 0, s, part_m, part_l, part_o, S,             \
                               (const __bf16*)wv, bv, (__bf16*)c->out, (int64_t)d, B);                                                     \
    } while (0)
    if (d == 384) ABS_BLOCK(384);
    else if (d == 512) ABS_BLOCK(512);
    else if (d == 768) ABS_BLOCK(768);
    else ABS_BLOCK(1024);
#undef ABS_BLOCK
    if (rc != WIPA_OK) return rc;
    WIPA_LAUNCH_CHECK();
    return WIPA_OK;
}
