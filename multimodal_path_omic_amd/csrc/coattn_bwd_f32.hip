// K1 backward for an fp32-stored bag (the reference's own storage; BASELINE configuration 5: 100 000-patch fp32 slides),
// embed 256, N <= 8 queries, with or without a gradient arriving on the map (the `cesar` loss's ||A||_2 term,
// models/loss.py:88-101): the arithmetic of coattn_bwd.hip
//   S[n][m] = qk[n].H[m],  A = exp2(S - lse),  dA[n][m] = dctx[n].H[m] (+ da_map[n][m]),  dS = A (dA - delta),
//   dqk[n] = sum_m dS[n][m] H[m],   dH[m] = sum_n A[n][m] dctx[n] + dS[n][m] qk[n]
// on the VECTOR ALUs in plain fp32 (models/mcat/mcat.py:97 differentiated; the forward is coattn_fwd.hip).
//
// Why not the matrix pipe: with N = 6 queries every product here is skinny -- 12 + 12 + 6 FMAs per element of H.  The general
// kernel splits the fp32 tile into two bf16 images, runs both row products in both orientations as three-term MFMAs with 26
// of 32 k-slots empty, stages dH through the tile's LDS image and spills (1 KB of scratch per lane): 642 us per 8 x 100 000
// window = 0.32 of HBM peak.  Here a lane keeps 4 columns of a row (a wave reads whole 1 KB rows, one float4 per lane), the
// two row products are 4 FMAs per (query, lane) and one TRANSPOSED wave reduction of the 2 N values per row (pairs share a
// v_permlane32_swap, pairs of pairs a v_permlane16_swap, four DPP steps finish four values at once; the scalars every lane
// needs come back by v_readlane), exp2 once per (row, query) on the folded registers, dH and dqk by FMAs against the lane's
// columns of qk / dctx held in registers.  No LDS image, no fragments, no
// splitting: plain fp32 products (the MFMA path carries the split's 2^-17).  8 waves per workgroup; the rows of the next
// 16-row step are requested into the register a row has just left (bag_key_grad_kernel's ring).
#include <type_traits>

#include "coattn_tile.h"
#include "mpo_common.h"
#include "mpo_kernels.h"

namespace {

constexpr int F_E = 256;
constexpr int F_WAVES = 8;
constexpr int F_HR = 16;                                   // rows per step (the register ring)

// sum over the 64 lanes, result in every lane (prologue only)
__device__ __forceinline__ float wave_allsum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
// a | b  ->  lanes 0..31: a's two halves added lane by lane, lanes 32..63: b's   (v_permlane32_swap: the upper half of the
// first operand changes places with the lower half of the second)
__device__ __forceinline__ float fold32(float a, float b) {
    const auto s = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    return __builtin_bit_cast(float, (unsigned)s[0]) + __builtin_bit_cast(float, (unsigned)s[1]);
}
// x = (a | b), y = (c | d) as fold32 leaves them  ->  16-lane rows (a, c, b, d), each value's lanes added pairwise once more
// (v_permlane16_swap: the odd rows of the first operand change places with the even rows of the second)
__device__ __forceinline__ float fold16(float x, float y) {
    const auto s = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
    return __builtin_bit_cast(float, (unsigned)s[0]) + __builtin_bit_cast(float, (unsigned)s[1]);
}
// sum inside each 16-lane row, result in every lane of the row: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ float row_allsum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));
    return v;
}

// lane I of every 16-lane row to all lanes of its row (DPP row_newbcast: one vector instruction)
template <int I>
__device__ __forceinline__ float row_bcast(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + I, 0xF, 0xF, false));
}

// DMAP: a gradient arrives on the map (da_map, ragged [n_q][M_b] per slide; delta then already holds rowsum(A da_map) too).
// Its values enter on the FOLDED registers (one lane row per query): a step's 16 rows x (up to) 8 queries are fetched as two
// coalesced loads -- lane (16 row-of-queries + i) takes (query, row i) -- and row i's are broadcast along the lane rows.
template <int NQA, bool DMAP>
__global__ __launch_bounds__(F_WAVES * 64, 1)
void coattn_bwd_f32_kernel(const float* __restrict__ bag, const int* __restrict__ cu,
                           const float* __restrict__ qk2,      // [n_slides][n_q][256] log2 units
                           const float* __restrict__ lse2,     // [n_slides][n_q]      log2 units
                           const float* __restrict__ dctx,     // [n_slides][n_q][256]
                           const float* __restrict__ delta,    // [n_slides][n_q] or NULL: rowsum(dctx * ctx) computed here
                           const float* __restrict__ ctx,      // [n_slides][n_q][256], read when delta == NULL
                           const float* __restrict__ da_map,   // DMAP: ragged [n_q][M_b] per slide
                           float* __restrict__ dbag,           // [total_rows][256]
                           float* __restrict__ part_dqk,       // [parts][n_q][256] (natural units)
                           float* __restrict__ part_colsum,    // nullable [parts][256]
                           int n_q, BagPlan plan) {
    constexpr int RED = 8 * F_E + F_E;                       // per wave: dqk [8][256] + column sums
    __shared__ __attribute__((aligned(16))) float lds[F_WAVES * RED];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const WgGeom wg = wg_geom(cu, plan);
    const int b = wg.b, m_rows = wg.m_rows, r0 = wg.r0, r1 = wg.r1;

    // the lane's 4 columns of qk (log2 units) and dctx; lse and delta per query (rows n >= n_q: zero operands, A = 0)
    f32x4 qk[NQA], dc[NQA];
    float ls[8], dl[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) { ls[n] = INFINITY; dl[n] = 0.f; }     // exp2(s - inf) = 0: a dead query contributes nothing
#pragma unroll
    for (int n = 0; n < NQA; ++n) {
        const bool live = n < n_q;
        const size_t at = ((size_t)b * n_q + (live ? n : 0)) * F_E + 4 * lane;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        qk[n] = live ? *reinterpret_cast<const f32x4*>(qk2 + at) : z;
        dc[n] = live ? *reinterpret_cast<const f32x4*>(dctx + at) : z;
        if (live) ls[n] = lse2[(size_t)b * n_q + n];
        if (delta != nullptr) {
            if (live) dl[n] = delta[(size_t)b * n_q + n];
        } else {
            const f32x4 c4 = live ? *reinterpret_cast<const f32x4*>(ctx + at) : z;
            dl[n] = wave_allsum((dc[n][0] * c4[0] + dc[n][1] * c4[1]) + (dc[n][2] * c4[2] + dc[n][3] * c4[3]));
        }
    }
    f32x4 acc[NQA];
#pragma unroll
    for (int n = 0; n < NQA; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};
    const float* slide = bag + (size_t)wg.row_begin * F_E;
    float* dslide = dbag + (size_t)wg.row_begin * F_E;
    // 16-row steps dealt to the waves one by one (59 tiles over 8 waves idle five waves for a whole tile in the last round)
    const int n_units = r1 > r0 ? (r1 - r0 + F_HR - 1) / F_HR : 0;
    auto step_row = [&](int st) { return r0 + F_HR * (wave + st * F_WAVES); };
    const int n_steps = wave < n_units ? (n_units - wave + F_WAVES - 1) / F_WAVES : 0;

    f32x4 hv[F_HR];
    auto fetch_row = [&](int st, int i) {
        st = st < n_steps ? st : n_steps - 1;
        int grow = step_row(st) + i;
        grow = grow < m_rows ? grow : m_rows - 1;
        hv[i] = *reinterpret_cast<const f32x4*>(slide + (size_t)grow * F_E + 4 * lane);
    };
    // Lane constants of the folded layout below: 16-lane row r of a folded register holds query  perm(r) = (0, 2, 1, 3)[r]
    // (+ 4 for the second register)
    const int frow = lane >> 4;
    const int fq = ((frow & 1) << 1) | (frow >> 1);
    float ls_lo = ls[0], ls_hi = ls[4], dl_lo = dl[0], dl_hi = dl[4];
#pragma unroll
    for (int n = 1; n < 4; ++n)
        if (fq == n) { ls_lo = ls[n]; dl_lo = dl[n]; ls_hi = ls[4 + n]; dl_hi = dl[4 + n]; }
    // one row: h = the lane's 4 columns of H[m].  The 2 N row products are reduced over the wave TRANSPOSED: two values
    // share a v_permlane32_swap (each keeps one half of the wave), two such pairs a v_permlane16_swap (each value one
    // 16-lane row), then four DPP steps finish all four at once -- 10 swaps + 16 DPP adds per row for 12..16 values instead
    // of 8 operations per value; exp2 and dS run on the folded registers (one lane row per query), and the 2 N scalars
    // every lane needs come back as v_readlane broadcasts.
    const float* da_b = DMAP ? da_map + (size_t)n_q * wg.row_begin : nullptr;
    const bool lo_live = fq < n_q, hi_live = NQA > 4 && 4 + fq < n_q;
    auto row_math = [&](const f32x4& h, f32x4& o, float dm_lo, float dm_hi) {
        float sp[8], dp[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            if (n < NQA) {
                const float s2x = h[0] * qk[n][0] + h[2] * qk[n][2], s2y = h[1] * qk[n][1] + h[3] * qk[n][3];
                const float d2x = h[0] * dc[n][0] + h[2] * dc[n][2], d2y = h[1] * dc[n][1] + h[3] * dc[n][3];
                sp[n] = s2x + s2y;
                dp[n] = d2x + d2y;
            } else {
                sp[n] = 0.f;
                dp[n] = 0.f;
            }
        }
        const float s_lo = row_allsum(fold16(fold32(sp[0], sp[1]), fold32(sp[2], sp[3])));   // rows: queries 0, 2, 1, 3
        float d_lo = row_allsum(fold16(fold32(dp[0], dp[1]), fold32(dp[2], dp[3])));
        if constexpr (DMAP) d_lo += dm_lo;
        const float a_lo = __builtin_amdgcn_exp2f(s_lo - ls_lo);
        const float ds_lo = a_lo * (d_lo - dl_lo);
        float a_hi = 0.f, ds_hi = 0.f;
        if constexpr (NQA > 4) {
            const float s_hi = row_allsum(fold16(fold32(sp[4], sp[5]), fold32(sp[6], sp[7])));   // rows: queries 4, 6, 5, 7
            float d_hi = row_allsum(fold16(fold32(dp[4], dp[5]), fold32(dp[6], dp[7])));
            if constexpr (DMAP) d_hi += dm_hi;
            a_hi = __builtin_amdgcn_exp2f(s_hi - ls_hi);
            ds_hi = a_hi * (d_hi - dl_hi);
        }
        o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < NQA; ++n) {
            const int src = 16 * ((((n & 3) & 1) << 1) | ((n & 3) >> 1));         // the lane row that holds query n
            const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, n < 4 ? a_lo : a_hi), src));
            const float ds = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, n < 4 ? ds_lo : ds_hi), src));
            o += dc[n] * a;
            o += qk[n] * (ds * kLn2);                            // qk in natural units = qk2 * ln 2
            acc[n] += h * ds;
        }
    };
    auto step = [&](int st) {                                    // all 16 rows exist: nothing is predicated
        const int row0 = step_row(st);
        float dmv_lo = 0.f, dmv_hi = 0.f;                         // DMAP: (this lane row's query, row row0 + (lane & 15))
        if constexpr (DMAP) {
            if (lo_live) dmv_lo = da_b[(size_t)fq * m_rows + row0 + (lane & 15)];
            if (hi_live) dmv_hi = da_b[(size_t)(4 + fq) * m_rows + row0 + (lane & 15)];
        }
        auto one = [&](auto tag) {
            constexpr int i = decltype(tag)::value;
            const f32x4 h = hv[i];
            fetch_row(st + 1, i);                                 // the register is free again: next step's row i
            f32x4 o;
            row_math(h, o, DMAP ? row_bcast<i>(dmv_lo) : 0.f, DMAP ? row_bcast<i>(dmv_hi) : 0.f);
            *reinterpret_cast<f32x4*>(dslide + (size_t)(row0 + i) * F_E + 4 * lane) = o;
            csum += o;
            __builtin_amdgcn_sched_barrier(0);                    // one row's working set at a time
        };
        using std::integral_constant;
        one(integral_constant<int, 0>{}); one(integral_constant<int, 1>{}); one(integral_constant<int, 2>{}); one(integral_constant<int, 3>{});
        one(integral_constant<int, 4>{}); one(integral_constant<int, 5>{}); one(integral_constant<int, 6>{}); one(integral_constant<int, 7>{});
        one(integral_constant<int, 8>{}); one(integral_constant<int, 9>{}); one(integral_constant<int, 10>{}); one(integral_constant<int, 11>{});
        one(integral_constant<int, 12>{}); one(integral_constant<int, 13>{}); one(integral_constant<int, 14>{}); one(integral_constant<int, 15>{});
        static_assert(F_HR == 16, "one() per row of a step");
    };
    auto ragged_step = [&](int st) {                             // the last step(s) of a range: the rows that exist, loaded where used
        const int row0 = step_row(st);
        const int nvalid = max(0, min(F_HR, r1 - row0));
        for (int i = 0; i < nvalid; ++i) {
            const f32x4 h = *reinterpret_cast<const f32x4*>(slide + (size_t)(row0 + i) * F_E + 4 * lane);
            f32x4 o;
            float dm_lo = 0.f, dm_hi = 0.f;
            if constexpr (DMAP) {                                 // (every lane of a lane row reads the same value)
                if (lo_live) dm_lo = da_b[(size_t)fq * m_rows + row0 + i];
                if (hi_live) dm_hi = da_b[(size_t)(4 + fq) * m_rows + row0 + i];
            }
            row_math(h, o, dm_lo, dm_hi);
            *reinterpret_cast<f32x4*>(dslide + (size_t)(row0 + i) * F_E + 4 * lane) = o;
            csum += o;
        }
    };
    if (n_steps > 0) {
        int n_full = 0;
        while (n_full < n_steps && step_row(n_full) + F_HR <= r1) ++n_full;
#pragma unroll
        for (int i = 0; i < F_HR; ++i) fetch_row(0, i);
        int st = 0;
        for (; st < n_full; ++st) step(st);
        for (; st < n_steps; ++st) ragged_step(st);
    }
    // the eight waves through LDS, summed in a fixed order
    float* red = lds + wave * RED;
#pragma unroll
    for (int n = 0; n < NQA; ++n) *reinterpret_cast<f32x4*>(red + n * F_E + 4 * lane) = acc[n];
    *reinterpret_cast<f32x4*>(red + 8 * F_E + 4 * lane) = csum;
    __syncthreads();
    for (int idx = threadIdx.x; idx < n_q * F_E; idx += F_WAVES * 64) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < F_WAVES; ++w) s += lds[w * RED + idx];
        part_dqk[wg.part * n_q * F_E + idx] = s;
    }
    if (part_colsum != nullptr)
        for (int idx = threadIdx.x; idx < F_E; idx += F_WAVES * 64) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < F_WAVES; ++w) s += lds[w * RED + 8 * F_E + idx];
            part_colsum[wg.part * F_E + idx] = s;
        }
}

bool g_bwd_f32_enabled = true;

}  // namespace

int mpo_coattn_bwd_f32_enable(int enabled) {
    const int was = g_bwd_f32_enabled ? 1 : 0;
    g_bwd_f32_enabled = enabled != 0;
    return was;
}
bool mpo_coattn_bwd_f32_covers(int bag_f32, int embed, int n_q, const float* da_map) {
    (void)da_map;                                             // (with or without a gradient on the map)
    return g_bwd_f32_enabled && bag_f32 && embed == F_E && n_q >= 1 && n_q <= 8;
}

int mpo_launch_coattn_bwd_f32(const void* bag, const int* cu, const float* qk2, const float* lse2, const float* dctx,
                              const float* delta, const float* ctx, const float* da_map, void* dbag, float* part_dqk,
                              float* part_colsum, int n_q, const BagPlan& plan, hipStream_t stream) {
    MPO_CHECK(n_q >= 1 && n_q <= 8, "coattn backward (fp32 bag): 1..8 queries (got %d)", n_q);
    MPO_CHECK(delta || ctx, "coattn backward: delta or ctx");
    MPO_CHECK(da_map == nullptr || delta != nullptr, "coattn backward (fp32 bag): a gradient on the map needs delta (with its share in)");
    MPO_CHECK(((reinterpret_cast<uintptr_t>(bag) | reinterpret_cast<uintptr_t>(dbag) | reinterpret_cast<uintptr_t>(qk2) |
                reinterpret_cast<uintptr_t>(dctx) | reinterpret_cast<uintptr_t>(ctx)) & 15) == 0,
              "coattn backward (fp32 bag): operands must be 16-byte aligned");
    const dim3 grid = plan_grid(plan);
#define MPO_F32_BWD(NQ_, DM_) coattn_bwd_f32_kernel<NQ_, DM_><<<grid, F_WAVES * 64, 0, stream>>>(static_cast<const float*>(bag), cu, qk2, lse2, dctx, \
        delta, ctx, da_map, static_cast<float*>(dbag), part_dqk, part_colsum, n_q, plan)
    if (n_q <= 6) { if (da_map) MPO_F32_BWD(6, true); else MPO_F32_BWD(6, false); }
    else          { if (da_map) MPO_F32_BWD(8, true); else MPO_F32_BWD(8, false); }
#undef MPO_F32_BWD
    MPO_LAUNCH_CHECK();
    return 0;
}
