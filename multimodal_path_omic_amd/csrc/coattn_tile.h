// LDS tile image + MFMA fragment plumbing shared by the forward and backward
// long-bag cross-attention kernels (K1 MCAT / K2 NaCAGaT; SURVEY.md section 2 kernel table).
//
// One wave owns one tile of 32 patch rows x E bf16 (E = 128/256/512) at a time.
//
// Image: row-major, E*2 bytes per row, 16-byte chunk c of row r stored at chunk
//   c ^ (2*(r&7)).  That one XOR makes BOTH kinds of read conflict-free
//   (MI355X_MICROARCH.md, LDS table):
//   * row reads  ds_read_b128: lane (p = l&15, g = l>>4) reads row p, chunk 4s+g
//     -> the MFMA 16x16x32 operand "row p, k = 32s + 8g .. +7";
//   * transposed reads ds_read_b64_tr_b16: a 16-lane group reads 4 rows x 16 columns
//     and receives them column-major -> the operand "row = column d, k = patch".
// How the image is filled depends on the kernel: the forward pass of a bf16 bag (E <= 256) sends its tiles
// global -> LDS directly (global_load_lds_dwordx4 from inline asm with hand-counted waits, coattn_fwd.hip: the
// chunk swizzle is applied to the GLOBAL chunk a lane fetches, the LDS side stays linear); every other pass
// stages through registers with the Stage struct below (global_load_dwordx4 issued a tile ahead, ds_write after
// the previous tile's MFMA work: the T14 issue-early / write-late split of cdna_hip_programming.md).  A bf16
// bag is copied as is; an fp32 bag is split into a bf16 hi tile and a bf16 lo tile.
//
// Orientation of every MFMA: the query index is always the MFMA column (lane & 15),
// so a lane's running max / sum / rescale factor belong to the same query as all of
// its accumulators, and the exponentiated scores feed the second product straight
// from registers (no LDS round trip, no shuffles apart from the 2-step row max).
#pragma once
#include "mpo_common.h"

constexpr int kTileRows = 32;

// ---------------------------------------------------------------- work plan of a bag pass over a ragged window
// A bag pass gives every workgroup one contiguous row range of one slide.
//   uniform (wg_start == nullptr): grid (splits, n_slides); slide b is cut into `splits` equal ranges.
//   planned (wg_start != nullptr): grid (n_wg); every workgroup gets `rows_per_wg` rows (multiple of 32), slide b owns
//     workgroups wg_start[b] .. wg_start[b+1]-1, i.e. ceil(M_b / rows_per_wg) of them: splits proportional to the
//     bag length, so a window of 2k..30k-patch bags is balanced (the uniform cut makes the longest slide set the time).
// Partials are indexed by workgroup: b * splits + split (uniform) or the workgroup id (planned).
struct BagPlan {
    const int* wg_start = nullptr;   // device, n_slides + 1 entries
    int n_slides = 0;
    int splits = 1;
    int rows_per_wg = 0;
    int n_wg = 0;
};
inline dim3 plan_grid(const BagPlan& p) { return p.wg_start ? dim3(p.n_wg, 1) : dim3(p.splits, p.n_slides); }
inline size_t plan_parts(const BagPlan& p) { return p.wg_start ? (size_t)p.n_wg : (size_t)p.splits * p.n_slides; }
__device__ __forceinline__ size_t plan_parts_dev(const BagPlan& p) { return p.wg_start ? (size_t)p.n_wg : (size_t)p.splits * p.n_slides; }

struct WgGeom {
    int b, row_begin, m_rows, r0, r1, ntiles;
    size_t part;                      // index of this workgroup's partial
};
__device__ __forceinline__ WgGeom wg_geom(const int* __restrict__ cu, const BagPlan& pl) {
    WgGeom g;
    int split, rps;
    if (pl.wg_start != nullptr) {
        const int wg = blockIdx.x;
        int lo = 0, hi = pl.n_slides;                       // wg_start[lo] <= wg < wg_start[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pl.wg_start[mid] <= wg) lo = mid; else hi = mid;
        }
        g.b = lo;
        split = wg - pl.wg_start[lo];
        rps = pl.rows_per_wg;
        g.part = wg;
    } else {
        g.b = blockIdx.y;
        split = blockIdx.x;
        rps = 0;
        g.part = (size_t)g.b * pl.splits + split;
    }
    g.row_begin = cu[g.b];
    g.m_rows = cu[g.b + 1] - g.row_begin;
    if (pl.wg_start == nullptr) rps = ((g.m_rows + pl.splits - 1) / pl.splits + kTileRows - 1) / kTileRows * kTileRows;
    g.r0 = split * rps;
    g.r1 = min(g.m_rows, g.r0 + rps);
    g.ntiles = g.r1 > g.r0 ? (g.r1 - g.r0 + kTileRows - 1) / kTileRows : 0;
    return g;
}
// partial index range [s0, s1) of slide b
__device__ __forceinline__ void slide_parts(const BagPlan& pl, int b, int& s0, int& s1) {
    if (pl.wg_start != nullptr) { s0 = pl.wg_start[b]; s1 = pl.wg_start[b + 1]; }
    else { s0 = b * pl.splits; s1 = s0 + pl.splits; }
}

template <int E_>
struct TileGeom {
    static constexpr int ROWB = E_ * 2;                 // bytes per image row
    static constexpr int TILEB = kTileRows * ROWB;      // 8/16/32 KiB
    static constexpr int KS = E_ / 32;                  // k-steps of a row-operand product
    static constexpr int DT = E_ / 16;                  // 16-column tiles across E
};

__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------- fill: through registers (T14 split)
// A Stage holds 16 x 16 bytes per lane in flight: a whole 32-row tile of a bf16 bag, or 16 rows
// (`half` = 0/1) of an fp32 bag.  load() only issues the global loads (1 KiB contiguous per
// wave-instruction); store() converts and writes the swizzled image, so the loads of the next
// tile stay in flight under the MFMA work of the current one.  Rows >= m_rows are clamped to
// the last valid row (finite data; the caller masks them).
template <int E_, bool F32BAG>
struct Stage {
    static constexpr int ROWS = F32BAG ? 16 : 32;
    static constexpr int CH_PER_ROW = F32BAG ? E_ / 4 : E_ / 8;      // 16-byte chunks per bag row
    static constexpr int N = ROWS * CH_PER_ROW / 64;
    f32x4 v[N];

    __device__ __forceinline__ void load(const void* slide_, int row0, int m_rows, int half, int lane) {
        const char* slide = reinterpret_cast<const char*>(slide_);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int ci = i * 64 + lane;
            int grow = row0 + ROWS * half + ci / CH_PER_ROW;
            grow = grow < m_rows ? grow : m_rows - 1;
            v[i] = *reinterpret_cast<const f32x4*>(slide + ((size_t)grow * CH_PER_ROW + ci % CH_PER_ROW) * 16);
        }
    }
    __device__ __forceinline__ void store(char* thi, char* tlo, int half, int lane) const {
        using G = TileGeom<E_>;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int ci = i * 64 + lane;
            const int r = ROWS * half + ci / CH_PER_ROW;
            const int cc = ci % CH_PER_ROW;
            if constexpr (!F32BAG) {
                *reinterpret_cast<f32x4*>(thi + r * G::ROWB + ((cc ^ ((r & 7) << 1)) << 4)) = v[i];
            } else {
                const int off = r * G::ROWB + ((((cc >> 1) ^ ((r & 7) << 1))) << 4) + 8 * (cc & 1);
                bf16x4 hi, lo;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    __bf16 h, l;
                    split_bf16(v[i][j], h, l);
                    hi[j] = h;
                    lo[j] = l;
                }
                *reinterpret_cast<bf16x4*>(thi + off) = hi;
                *reinterpret_cast<bf16x4*>(tlo + off) = lo;
            }
        }
    }
};

// ---------------------------------------------------------------- operand reads
// Row operand: rows 16*pt + (lane&15), k = 32*s + 8*(lane>>4) .. +7.
template <int E_>
__device__ __forceinline__ bf16x8 row_frag(const char* tile, int pt, int s, int lane) {
    using G = TileGeom<E_>;
    const int p = lane & 15, g = lane >> 4;
    const int c = (4 * s + g) ^ ((lane & 7) << 1);
    return *reinterpret_cast<const bf16x8*>(tile + (16 * pt + p) * G::ROWB + (c << 4));
}

// Transposed operand for column tile t: lane (i = l&15, g = l>>4) receives, for column
// d = 16t + i, the 8 patches  k-order(g, j) = j < 4 ? 4g + j : 16 + 4g + (j - 4).
// That is exactly the patch order in which the score accumulators of lane group g sit
// (tile rows 4g+r from p-tile 0, 16+4g+r from p-tile 1), so P needs no permutation.
template <int E_>
__device__ __forceinline__ bf16x8 col_frag(const char* tile, int t, int lane) {
    using G = TileGeom<E_>;
    const int i = lane & 15, g = lane >> 4;
    const int q4 = i >> 2, p4 = i & 3;
    const int r0 = 4 * g + q4;                       // rows r0 and 16 + r0 share (r & 7)
    const int c = (2 * t + (p4 >> 1)) ^ ((r0 & 7) << 1);
    const int off = r0 * G::ROWB + (c << 4) + 8 * (p4 & 1);
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + off));
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + off + 16 * G::ROWB));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// Query-side operand (the MFMA "B" matrix, column = query): lane (q = l&15, g) holds
// x[q][32s + 8g + j], split hi/lo.  Rows q >= n_q are zero.
template <int E_>
__device__ __forceinline__ void load_query_frags(const float* x /* [n_q][E] */, int n_q, int lane,
                                                 bf16x8 (&hi)[TileGeom<E_>::KS], bf16x8 (&lo)[TileGeom<E_>::KS]) {
    const int q = lane & 15, g = lane >> 4;
    const float live = q < n_q ? 1.0f : 0.0f;
    const float* row = x + (q < n_q ? q : n_q - 1) * E_ + 8 * g;
#pragma unroll
    for (int s = 0; s < TileGeom<E_>::KS; ++s) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(row + 32 * s);
        const f32x4 b = *reinterpret_cast<const f32x4*>(row + 32 * s + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = (j < 4 ? a[j] : b[j - 4]) * live;
            __bf16 h, l;
            split_bf16(v, h, l);
            hi[s][j] = h;
            lo[s][j] = l;
        }
    }
}

// The same operand in THREE bf16 terms (hi + mid + lo = all 24 mantissa bits of x): for the score product of the forward
// pass, where a logit of magnitude L moves the attention map by L x (operand error) relative -- two terms (2^-17) left a
// peaky fixture (|logit| ~ 130) at 1.1e-3 on the map.
template <int E_>
__device__ __forceinline__ void load_query_frags3(const float* x /* [n_q][E] */, int n_q, int lane,
                                                  bf16x8 (&hi)[TileGeom<E_>::KS], bf16x8 (&mid)[TileGeom<E_>::KS],
                                                  bf16x8 (&lo)[TileGeom<E_>::KS]) {
    const int q = lane & 15, g = lane >> 4;
    const float live = q < n_q ? 1.0f : 0.0f;
    const float* row = x + (q < n_q ? q : n_q - 1) * E_ + 8 * g;
#pragma unroll
    for (int s = 0; s < TileGeom<E_>::KS; ++s) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(row + 32 * s);
        const f32x4 b = *reinterpret_cast<const f32x4*>(row + 32 * s + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = (j < 4 ? a[j] : b[j - 4]) * live;
            const __bf16 h = (__bf16)v;
            const float r1 = v - (float)h;
            const __bf16 m = (__bf16)r1;
            hi[s][j] = h;
            mid[s][j] = m;
            lo[s][j] = (__bf16)(r1 - (float)m);
        }
    }
}
// scores^T with the three-term query operand; an fp32 bag (hi + lo images) adds lo x (hi + mid)
template <int E_, int NT>
__device__ __forceinline__ void tile_dot_rows3(const char* thi, const char* tlo, const bf16x8 (&xh)[TileGeom<E_>::KS],
                                               const bf16x8 (&xm)[TileGeom<E_>::KS], const bf16x8 (&xl)[TileGeom<E_>::KS],
                                               f32x4& s0, f32x4& s1, int lane) {
#pragma unroll
    for (int s = 0; s < TileGeom<E_>::KS; ++s) {
        const bf16x8 a0 = row_frag<E_>(thi, 0, s, lane);
        const bf16x8 a1 = row_frag<E_>(thi, 1, s, lane);
        s0 = mfma_bf16(a0, xh[s], s0);
        s1 = mfma_bf16(a1, xh[s], s1);
        s0 = mfma_bf16(a0, xm[s], s0);
        s1 = mfma_bf16(a1, xm[s], s1);
        s0 = mfma_bf16(a0, xl[s], s0);
        s1 = mfma_bf16(a1, xl[s], s1);
        if (NT == 2) {
            const bf16x8 b0 = row_frag<E_>(tlo, 0, s, lane);
            const bf16x8 b1 = row_frag<E_>(tlo, 1, s, lane);
            s0 = mfma_bf16(b0, xh[s], s0);
            s1 = mfma_bf16(b1, xh[s], s1);
            s0 = mfma_bf16(b0, xm[s], s0);
            s1 = mfma_bf16(b1, xm[s], s1);
        }
    }
}

// scores^T for both 16-row halves of the tile:  s[pt][r] = sum_k tile[16pt + 4g + r][k] * x[q][k]
template <int E_, int NT>
__device__ __forceinline__ void tile_dot_rows(const char* thi, const char* tlo,
                                              const bf16x8 (&xh)[TileGeom<E_>::KS], const bf16x8 (&xl)[TileGeom<E_>::KS],
                                              f32x4& s0, f32x4& s1, int lane) {
#pragma unroll
    for (int s = 0; s < TileGeom<E_>::KS; ++s) {
        const bf16x8 a0 = row_frag<E_>(thi, 0, s, lane);
        const bf16x8 a1 = row_frag<E_>(thi, 1, s, lane);
        s0 = mfma_bf16(a0, xh[s], s0);
        s1 = mfma_bf16(a1, xh[s], s1);
        s0 = mfma_bf16(a0, xl[s], s0);
        s1 = mfma_bf16(a1, xl[s], s1);
        if (NT == 2) {
            const bf16x8 b0 = row_frag<E_>(tlo, 0, s, lane);
            const bf16x8 b1 = row_frag<E_>(tlo, 1, s, lane);
            s0 = mfma_bf16(b0, xh[s], s0);
            s1 = mfma_bf16(b1, xh[s], s1);
        }
    }
}

// acc[t][r] += sum_p tile[p][16t + 4g + r] * w[p][q]   (w given in the k-order of col_frag, hi/lo)
template <int E_, int NT>
__device__ __forceinline__ void tile_accum_cols(const char* thi, const char* tlo, bf16x8 wh, bf16x8 wl,
                                                f32x4 (&acc)[TileGeom<E_>::DT], int lane) {
#pragma unroll
    for (int t = 0; t < TileGeom<E_>::DT; ++t) {
        const bf16x8 h = col_frag<E_>(thi, t, lane);
        acc[t] = mfma_bf16(h, wh, acc[t]);
        acc[t] = mfma_bf16(h, wl, acc[t]);
        if (NT == 2) {
            const bf16x8 l = col_frag<E_>(tlo, t, lane);
            acc[t] = mfma_bf16(l, wh, acc[t]);
        }
    }
}

__device__ __forceinline__ void pack_hi_lo(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        __bf16 h, l;
        split_bf16(v[j], h, l);
        hi[j] = h;
        lo[j] = l;
    }
}
