// K1 backward: one pass over the bag recomputes the folded co-attention from the saved
// log-sum-exp and emits  dH (gradient of the bag, written once, coalesced)  and split-M
// partials of  dqk (gradient of the folded query qk = (q/sqrt(E)) W_k).
//
// With  S[n][m] = qk[n].H[m],  A = softmax(S),  ctx = A H  (coattn_fwd.hip) and upstream
// gradients dctx (N x E) and, optionally, dA_ext on the attention map itself (the NaCAGaT
// 'cesar' loss back-propagates into the map: models/loss.py:97, models/nacagat/main.py:49-50):
//   dA[n][m]  = dctx[n].H[m] + dA_ext[n][m]
//   dS[n][m]  = A[n][m] (dA[n][m] - delta[n]),   delta[n] = dctx[n].ctx[n] + sum_m A dA_ext
//   dqk[n]    = sum_m dS[n][m] H[m]
//   dH[m]     = sum_n A[n][m] dctx[n] + dS[n][m] qk[n]
// Per 32-row tile the MFMA runs the two row products (scores, dA) in BOTH orientations -- query
// on the lane for the dqk accumulation (as the forward), patch on the lane for dH -- by swapping
// the operand roles of the same registers; nothing is transposed through memory.  dH is formed
// as  dH^T[d][p] = Z^T[d][k] W^T[k][p]  with k running over (A rows | dS rows), so each lane owns
// 4 consecutive d of one patch; the tile's own LDS image is overwritten with dH and copied out
// in whole rows.
#include <type_traits>

#include "coattn_tile.h"
#include "mpo_kernels.h"

namespace {

template <int E_, bool F32BAG>
struct BwdCfg {
    static constexpr int NT = F32BAG ? 2 : 1;
    static constexpr int WAVES = E_ == 512 ? (F32BAG ? 1 : 2) : 4;
    static constexpr int WAVE_LDS = NT * TileGeom<E_>::TILEB;     // also holds the dH image (E*2*32 or E*4*32 bytes)
    static constexpr int TILES_BYTES = WAVES * WAVE_LDS;
    // the per-slide Z^T operand of the dH product lives in LDS (one copy per workgroup): as 64-128 registers per lane
    // it pushed the kernel far past the register file (1000+ accvgpr spill moves per pass over the loop)
    static constexpr int Z_BYTES = NT * TileGeom<E_>::DT * 64 * 16;
    static constexpr int DST_LD = kTileRows + 4;                  // row stride of the pad in floats: 36 keeps the four
                                                                  // lane groups of a write on different banks (32 = 4-way)
    static constexpr int DST_BYTES = F32BAG ? 0 : WAVES * 16 * DST_LD * 4;      // per-wave [16 queries][32 patches] pad
    static constexpr int LDS_BYTES = TILES_BYTES + Z_BYTES + DST_BYTES;
};

// rows-on-lane orientation: out[pt][r] = sum_k x[4g + r][k] * tile[16pt + (lane&15)][k]
template <int E_, int NT>
__device__ __forceinline__ void tile_dot_rows_T(const char* thi, const char* tlo,
                                                const bf16x8 (&xh)[TileGeom<E_>::KS], const bf16x8 (&xl)[TileGeom<E_>::KS],
                                                f32x4& s0, f32x4& s1, int lane) {
#pragma unroll
    for (int s = 0; s < TileGeom<E_>::KS; ++s) {
        const bf16x8 a0 = row_frag<E_>(thi, 0, s, lane);
        const bf16x8 a1 = row_frag<E_>(thi, 1, s, lane);
        s0 = mfma_bf16(xh[s], a0, s0);
        s1 = mfma_bf16(xh[s], a1, s1);
        s0 = mfma_bf16(xl[s], a0, s0);
        s1 = mfma_bf16(xl[s], a1, s1);
        if (NT == 2) {
            const bf16x8 b0 = row_frag<E_>(tlo, 0, s, lane);
            const bf16x8 b1 = row_frag<E_>(tlo, 1, s, lane);
            s0 = mfma_bf16(xh[s], b0, s0);
            s1 = mfma_bf16(xh[s], b1, s1);
        }
    }
}

template <int E_, bool F32BAG>
__global__ __launch_bounds__((BwdCfg<E_, F32BAG>::WAVES * 64), 1)
void coattn_bwd_kernel(const void* __restrict__ bag_, const int* __restrict__ cu,
                       const float* __restrict__ qk2,      // [n_slides][n_q][E] log2 units
                       const float* __restrict__ lse2,     // [n_slides][n_q]    log2 units
                       const float* __restrict__ dctx,     // [n_slides][n_q][E]
                       const float* __restrict__ delta,    // [n_slides][n_q] = rowsum(dctx * ctx) (+ map term); NULL: computed here from ctx
                       const float* __restrict__ ctx,      // [n_slides][n_q][E], read when delta == NULL
                       const float* __restrict__ da_map,   // nullable, ragged [n_q][M_b] per slide
                       void* __restrict__ dbag_,           // [total_rows][E], bag dtype
                       float* __restrict__ part_dqk,       // [n_slides][splits][n_q][E] (natural units)
                       float* __restrict__ part_colsum,    // nullable [parts][E]: column sums of the dH rows written here
                       int n_q, BagPlan plan,
                       float relu_gate /* 0: off; else 1/(1-p): dH *= (H > 0 ? relu_gate : 0), bf16 bag only */) {
    using G = TileGeom<E_>;
    using C = BwdCfg<E_, F32BAG>;
    constexpr int WAVES = C::WAVES;
    constexpr int NT = C::NT;
    constexpr int EB = F32BAG ? 4 : 2;                            // bytes per bag element
    __shared__ __attribute__((aligned(16))) char lds[C::LDS_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const WgGeom wg = wg_geom(cu, plan);
    const int b = wg.b, row_begin = wg.row_begin, m_rows = wg.m_rows, r0 = wg.r0, r1 = wg.r1, ntiles = wg.ntiles;
    const int n_my = wave < ntiles ? (ntiles - wave + WAVES - 1) / WAVES : 0;

    char* thi = lds + wave * C::WAVE_LDS;
    char* tlo = thi + (NT - 1) * G::TILEB;
    const int c16 = lane & 15, g = lane >> 4;
    const float* qk_b = qk2 + (size_t)b * n_q * E_;
    const float* dc_b = dctx + (size_t)b * n_q * E_;

    // operands indexed by the query (hi/lo): used as MFMA B (query on lane) and as MFMA A (patch on lane)
    bf16x8 qh[G::KS], ql[G::KS], dch[G::KS], dcl[G::KS];
    load_query_frags<E_>(qk_b, n_q, lane, qh, ql);
    load_query_frags<E_>(dc_b, n_q, lane, dch, dcl);

    // Z^T operand of the dH product, per 16-column tile t: lane (d = 16t + c16, g) element j:
    //   j < 4 : dctx[4g + j][d]          j >= 4 : qk_nat[4g + j - 4][d]      (zero for query rows >= n_q)
    // kept in LDS as [hi | lo][t][lane] 16-byte fragments; the waves build a quarter each
    bf16x8* zbuf = reinterpret_cast<bf16x8*>(lds + C::TILES_BYTES);
    for (int t = wave; t < G::DT; t += WAVES) {
        float z[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int qq = 4 * g + j;
            const int qc = qq < n_q ? qq : n_q - 1;
            const float live = qq < n_q ? 1.0f : 0.0f;
            z[j] = dc_b[qc * E_ + 16 * t + c16] * live;
            z[4 + j] = qk_b[qc * E_ + 16 * t + c16] * (live * kLn2);
        }
        bf16x8 h, l;
        pack_hi_lo(z, h, l);
        zbuf[t * 64 + lane] = h;
        if (NT == 2) zbuf[(G::DT + t) * 64 + lane] = l;
    }
    __syncthreads();

    // per-lane row constants in both orientations; +inf lse switches padded query rows off (A = 0)
    const float lse_q = c16 < n_q ? lse2[(size_t)b * n_q + c16] : INFINITY;
    float del_q, lse_p[4], del_p[4];
    if (delta != nullptr) {
        del_q = c16 < n_q ? delta[(size_t)b * n_q + c16] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) del_p[r] = 4 * g + r < n_q ? delta[(size_t)b * n_q + 4 * g + r] : 0.f;
    } else {
        // delta[q] = dctx[q] . ctx[q]: lane (q = c16, g) sums a quarter of the row, every wave on its own (no exchange
        // through LDS; this used to be a launch of its own in front of the bag pass)
        const float* cx_b = ctx + (size_t)b * n_q * E_;
        float acc = 0.f;
        if (c16 < n_q) {
#pragma unroll 4
            for (int e = g * (E_ / 4); e < (g + 1) * (E_ / 4); e += 4) {
                const f32x4 u = *reinterpret_cast<const f32x4*>(dc_b + c16 * E_ + e);
                const f32x4 v = *reinterpret_cast<const f32x4*>(cx_b + c16 * E_ + e);
                acc += (u[0] * v[0] + u[1] * v[1]) + (u[2] * v[2] + u[3] * v[3]);
            }
        }
        acc += __shfl_xor(acc, 16, 64);
        acc += __shfl_xor(acc, 32, 64);
        del_q = c16 < n_q ? acc : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float t = __shfl(acc, 4 * g + r, 64);               // lane q of the wave holds row q's total
            del_p[r] = 4 * g + r < n_q ? t : 0.f;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) lse_p[r] = 4 * g + r < n_q ? lse2[(size_t)b * n_q + 4 * g + r] : INFINITY;
    const float* da_b = da_map ? da_map + (size_t)n_q * row_begin : nullptr;

    f32x4 accq[G::DT];
#pragma unroll
    for (int t = 0; t < G::DT; ++t) accq[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const char* slide = reinterpret_cast<const char*>(bag_) + (size_t)row_begin * E_ * EB;
    char* dslide = reinterpret_cast<char*>(dbag_) + (size_t)row_begin * E_ * EB;

    // column sums of the emitted dH (= the bias gradient of the layer that produced the bag when the ReLU gate is on):
    // in the copy-out loop a lane always handles the same 16-byte column chunk, so it keeps that chunk's sums
    constexpr int CS_N = 16 / EB;                                 // elements per 16-byte chunk
    constexpr int CS_CPL = (E_ * EB / 16 + 63) / 64;              // chunk columns a lane alternates between (2 at E=512 fp32)
    float csum[CS_CPL * CS_N];
#pragma unroll
    for (int j = 0; j < CS_CPL * CS_N; ++j) csum[j] = 0.f;

    Stage<E_, F32BAG> st0;
    Stage<E_, F32BAG> st1;
    // One tile.  gfx950 counts loads and stores in ONE in-order counter and hipcc's wait counts must hold on every path
    // into a wait, so the step is written for constant counts: FULL tiles (all 32 rows exist) have no predicated load or
    // store at all, every global load of a step -- the map-gradient values first, the next tile's rows after them -- is
    // issued before the compute with clamped addresses (rows are clamped to the slide, so the prefetch past the last
    // tile re-reads valid rows), and the caller peels the first step so that the loop's entry and back edges agree.
    // Before: every step drained the counter (vmcnt(0)), i.e. waited for the previous tile's dH stores to be
    // acknowledged before staging the next rows.
    auto tile_step = [&](int it, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int trow = r0 + kTileRows * (wave + it * WAVES);
        const int nvalid = FULL ? kTileRows : min(kTileRows, r1 - trow);
        st0.store(thi, tlo, 0, lane);
        if constexpr (F32BAG) st1.store(thi, tlo, 1, lane);
        // map gradient of this tile's rows in the patch-on-lane orientation (and query-on-lane for the fp32 bag)
        float dap[2][4], daq[2][4];
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dap[pt][r] = daq[pt][r] = 0.f;
        if (da_b != nullptr) {
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    {
                        const int row = 16 * pt + c16, qq = 4 * g + r;
                        const int rc = min(trow + row, m_rows - 1), qc = min(qq, n_q - 1);
                        const float v = da_b[(size_t)qc * m_rows + rc];
                        dap[pt][r] = ((FULL || row < nvalid) && qq < n_q) ? v : 0.f;
                    }
                    if constexpr (F32BAG) {
                        const int row = 16 * pt + 4 * g + r;
                        const int rc = min(trow + row, m_rows - 1), qc = min(c16, n_q - 1);
                        const float v = da_b[(size_t)qc * m_rows + rc];
                        daq[pt][r] = ((FULL || row < nvalid) && c16 < n_q) ? v : 0.f;
                    }
                }
            }
        }
        st0.load(slide, trow + kTileRows * WAVES, m_rows, 0, lane);
        if constexpr (F32BAG) st1.load(slide, trow + kTileRows * WAVES, m_rows, 1, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        bf16x8 wph[2], wpl[2];
        if constexpr (!F32BAG) {
            // ---------------- patch on the lane: A, dS with the query index in the registers
            // (The scores are computed in THIS orientation only.  The dqk accumulation below needs dS with the query on
            //  the lane: it is transposed through a 2 KB LDS pad -- 8 scalar writes, 2 x 16-byte reads per lane -- instead of
            //  recomputing both row products in the other orientation: 64 MFMAs and 32 tile reads less per tile.)
            float* dsT = reinterpret_cast<float*>(lds + C::TILES_BYTES + C::Z_BYTES) + wave * (16 * C::DST_LD);   // [q][patch]
            {
                f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, d0 = s0, d1 = s0;
                tile_dot_rows_T<E_, NT>(thi, tlo, qh, ql, s0, s1, lane);
                tile_dot_rows_T<E_, NT>(thi, tlo, dch, dcl, d0, d1, lane);
    #pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
                    const int row = 16 * pt + c16;
                    const bool ok = FULL || row < nvalid;
                    float w[8];
    #pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int qq = 4 * g + r;
                        const float da = (pt == 0 ? d0[r] : d1[r]) + dap[pt][r];
                        const float a = ok ? __builtin_amdgcn_exp2f((pt == 0 ? s0[r] : s1[r]) - lse_p[r]) : 0.f;
                        w[r] = a;
                        w[4 + r] = a * (da - del_p[r]);
                        dsT[qq * C::DST_LD + row] = w[4 + r];
                    }
                    pack_hi_lo(w, wph[pt], wpl[pt]);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();

            // ---------------- query on the lane: dS^T -> dqk accumulation (reads the H image: before dH overwrites it)
            {
                // lane (q = c16, g) takes patches 4g..4g+3 and 16+4g..16+4g+3: the k-order of col_frag
                const f32x4 lo4 = *reinterpret_cast<const f32x4*>(dsT + c16 * C::DST_LD + 4 * g);
                const f32x4 hi4 = *reinterpret_cast<const f32x4*>(dsT + c16 * C::DST_LD + 16 + 4 * g);
                const float ds[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                bf16x8 wh, wl;
                pack_hi_lo(ds, wh, wl);
                tile_accum_cols<E_, NT>(thi, tlo, wh, wl, accq, lane);
            }
        } else {
            // fp32 bag: both images + Z fill the 160 KB of LDS, no room for the pad -- both orientations are recomputed
            // ---------------- query on the lane: dS^T -> dqk accumulation
            {
                f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, d0 = s0, d1 = s0;
                tile_dot_rows<E_, NT>(thi, tlo, qh, ql, s0, s1, lane);
                tile_dot_rows<E_, NT>(thi, tlo, dch, dcl, d0, d1, lane);
                float ds[8];
    #pragma unroll
                for (int r = 0; r < 4; ++r) {
    #pragma unroll
                    for (int pt = 0; pt < 2; ++pt) {
                        const int row = 16 * pt + 4 * g + r;
                        const bool ok = FULL || row < nvalid;
                        const float da = (pt == 0 ? d0[r] : d1[r]) + daq[pt][r];
                        const float a = ok ? __builtin_amdgcn_exp2f((pt == 0 ? s0[r] : s1[r]) - lse_q) : 0.f;
                        ds[4 * pt + r] = a * (da - del_q);
                    }
                }
                bf16x8 wh, wl;
                pack_hi_lo(ds, wh, wl);
                tile_accum_cols<E_, NT>(thi, tlo, wh, wl, accq, lane);
            }

            // ---------------- patch on the lane: A, dS with the query index in the registers
            {
                f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, d0 = s0, d1 = s0;
                tile_dot_rows_T<E_, NT>(thi, tlo, qh, ql, s0, s1, lane);
                tile_dot_rows_T<E_, NT>(thi, tlo, dch, dcl, d0, d1, lane);
    #pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
                    const int row = 16 * pt + c16;
                    const bool ok = FULL || row < nvalid;
                    float w[8];
    #pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float da = (pt == 0 ? d0[r] : d1[r]) + dap[pt][r];
                        const float a = ok ? __builtin_amdgcn_exp2f((pt == 0 ? s0[r] : s1[r]) - lse_p[r]) : 0.f;
                        w[r] = a;
                        w[4 + r] = a * (da - del_p[r]);
                    }
                    pack_hi_lo(w, wph[pt], wpl[pt]);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ---------------- dH^T[d][p] = Z^T W^T, written over the (now dead) tile image
#pragma unroll
        for (int t = 0; t < G::DT; ++t) {
            const bf16x8 zh = zbuf[t * 64 + lane];
            const bf16x8 zl = NT == 2 ? zbuf[(G::DT + t) * 64 + lane] : zh;
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const int row = 16 * pt + c16;
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
                o = mfma_bf16(zh, wph[pt], o);
                if (NT == 2) {
                    o = mfma_bf16(zh, wpl[pt], o);
                    o = mfma_bf16(zl, wph[pt], o);
                }
                // lane holds dH[row][16t + 4g .. +3]
                if constexpr (!F32BAG) {
                    const int c = (2 * t + (g >> 1)) ^ ((row & 7) << 1);
                    bf16x4* slot = reinterpret_cast<bf16x4*>(thi + row * G::ROWB + (c << 4) + 8 * (g & 1));
                    if (relu_gate != 0.f) {
                        // the bag is H = drop(relu(pre)): the slot about to receive dH still holds H itself
                        const bf16x4 hv = *slot;
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] *= (float)hv[j] > 0.f ? relu_gate : 0.f;
                    }
                    bf16x4 ob = {f2bf(o[0]), f2bf(o[1]), f2bf(o[2]), f2bf(o[3])};
                    *slot = ob;
                } else {
                    const int c = (4 * t + g) ^ ((row & 7) << 1);
                    *reinterpret_cast<f32x4*>(thi + row * (E_ * 4) + (c << 4)) = o;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // ---------------- copy the dH image out in whole rows (1 KiB per wave-instruction)
        {
            constexpr int CH_PER_ROW = E_ * EB / 16;
            constexpr int NCH = kTileRows * CH_PER_ROW / 64;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int ci = i * 64 + lane;
                const int r = ci / CH_PER_ROW, cc = ci % CH_PER_ROW;
                const f32x4 v = *reinterpret_cast<const f32x4*>(thi + r * (E_ * EB) + ((cc ^ ((r & 7) << 1)) << 4));
                if (FULL || r < nvalid) {
                    *reinterpret_cast<f32x4*>(dslide + ((size_t)(trow + r) * CH_PER_ROW + cc) * 16) = v;
                    if (part_colsum != nullptr) {
                        if constexpr (F32BAG) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) csum[(i % CS_CPL) * CS_N + j] += v[j];
                        } else {
                            const bf16x8 hv = __builtin_bit_cast(bf16x8, v);
#pragma unroll
                            for (int j = 0; j < 8; ++j) csum[(i % CS_CPL) * CS_N + j] += (float)hv[j];
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    if (n_my > 0) {
        using Full = std::integral_constant<bool, true>;
        using Guarded = std::integral_constant<bool, false>;
        int n_full = 0;
        while (n_full < n_my && r0 + kTileRows * (wave + n_full * WAVES) + kTileRows <= r1) ++n_full;
        st0.load(slide, r0 + kTileRows * wave, m_rows, 0, lane);
        if constexpr (F32BAG) st1.load(slide, r0 + kTileRows * wave, m_rows, 1, lane);
        int it = 0;
        if (n_full > 0) {
            tile_step(0, Full());
            for (it = 1; it < n_full; ++it) tile_step(it, Full());
        }
        for (; it < n_my; ++it) tile_step(it, Guarded());
    }

    // ---- merge the waves' dqk through LDS and write this workgroup's partial
    __syncthreads();
    {
        float* wq = reinterpret_cast<float*>(thi);                // [16][E] floats
#pragma unroll
        for (int t = 0; t < G::DT; ++t)
            *reinterpret_cast<f32x4*>(wq + c16 * E_ + 16 * t + 4 * g) = accq[t];
    }
    __syncthreads();
    const size_t pbase = wg.part;
    for (int idx = threadIdx.x; idx < n_q * E_; idx += WAVES * 64) {
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) a += reinterpret_cast<const float*>(lds + w * C::WAVE_LDS)[idx];
        part_dqk[pbase * n_q * E_ + idx] = a;
    }
    if (part_colsum != nullptr) {                                 // same exchange for the column sums
        __syncthreads();
        constexpr int CH_PER_ROW = E_ * EB / 16;
        static_assert(CH_PER_ROW % 64 == 0 || 64 % CH_PER_ROW == 0, "a lane must keep fixed column chunks across the copy-out");
#pragma unroll
        for (int o = CH_PER_ROW; o < 64; o <<= 1) {               // lanes l, l + CH_PER_ROW, ... share a chunk
#pragma unroll
            for (int j = 0; j < CS_CPL * CS_N; ++j) csum[j] += __shfl_xor(csum[j], o);
        }
        float* wc = reinterpret_cast<float*>(thi);
        if (lane < CH_PER_ROW) {
#pragma unroll
            for (int k = 0; k < CS_CPL; ++k)
#pragma unroll
                for (int j = 0; j < CS_N; ++j) wc[(lane % CH_PER_ROW + 64 * k) * CS_N + j] = csum[k * CS_N + j];
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < E_; idx += WAVES * 64) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) a += reinterpret_cast<const float*>(lds + w * C::WAVE_LDS)[idx];
            part_colsum[pbase * E_ + idx] = a;
        }
    }
}

// The small work that follows a split-M bag pass, in ONE launch (each used to be its own dependent launch):
//   rows y < n_slides * n_red : out_k[b][i] = sum_s part_k[s][i] over slide b's workgroups (64 float4 columns per
//                               workgroup, the 4 waves split the partials);
//   the extra row y == n_slides * n_red : column sums over ALL partials (colsum[c] = sum_s part_cs[s][c], the bias
//                               gradient of the layer that produced the bag) and zero-fills of up to two regions.
__global__ __launch_bounds__(256)
void bag_finish_kernel(BagFinish f, int n_slides, int per_slide, BagPlan plan) {
    __shared__ __attribute__((aligned(16))) float red[4][256];
    const int c = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i4 = blockIdx.x * 64 + c;                       // float4 index inside the row's block
    const int rows_red = n_slides * f.n_red;
    const bool extra = (int)blockIdx.y >= rows_red;
    if (extra) {
        const int tid = blockIdx.x * 256 + threadIdx.x, nthr = gridDim.x * 256;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (f.zero[k])
                for (int i = tid; i < f.n_zero[k]; i += nthr) f.zero[k][i] = 0.f;
        // column sums over ALL partials: a workgroup takes 8 float4 columns, its 32 thread groups stride the partials
        // (8 independent loads per thread at 256 partials), one exchange through LDS
        const int c8 = threadIdx.x & 7, pl = threadIdx.x >> 3;
        const int col4 = blockIdx.x * 8 + c8;
        if (f.part_cs == nullptr || 4 * (int)blockIdx.x * 8 >= f.cs_cols) return;
        const int parts = (int)plan_parts_dev(plan);
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (4 * col4 < f.cs_cols) {
#pragma unroll 8
            for (int s = pl; s < parts; s += 32) a += *reinterpret_cast<const f32x4*>(f.part_cs + (size_t)s * f.cs_cols + 4 * col4);
        }
        f32x4* red4 = reinterpret_cast<f32x4*>(&red[0][0]);
        red4[threadIdx.x] = a;
        __syncthreads();
        f32x4 t = {0.f, 0.f, 0.f, 0.f};                            // 4 x 8 partial sums per column, then one thread adds the four
        if (pl < 4) {
#pragma unroll
            for (int k = 0; k < 8; ++k) t += red4[(pl * 8 + k) * 8 + c8];
        }
        __syncthreads();                                           // every read of the first round is done before slots are reused
        if (pl < 4) red4[threadIdx.x] = t;
        __syncthreads();
        if (pl == 0 && 4 * col4 < f.cs_cols) {
            const f32x4 tot = (red4[c8] + red4[8 + c8]) + (red4[16 + c8] + red4[24 + c8]);
            *reinterpret_cast<f32x4*>(f.colsum + 4 * col4) = tot;
        }
        return;
    }
    const int which = (int)blockIdx.y / n_slides, b = (int)blockIdx.y % n_slides;
    const float* part = f.part[which];
    float* out = f.out[which] + (size_t)b * per_slide;
    const int width = per_slide;
    int s0, s1;
    slide_parts(plan, b, s0, s1);
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (4 * i4 < width)
        for (int s = s0 + w; s < s1; s += 4)
            a += *reinterpret_cast<const f32x4*>(part + (size_t)s * width + 4 * i4);
    *reinterpret_cast<f32x4*>(&red[w][4 * c]) = a;
    __syncthreads();
    if (w == 0 && 4 * i4 < width) {
        f32x4 t = *reinterpret_cast<const f32x4*>(&red[0][4 * c]);
        t += *reinterpret_cast<const f32x4*>(&red[1][4 * c]);
        t += *reinterpret_cast<const f32x4*>(&red[2][4 * c]);
        t += *reinterpret_cast<const f32x4*>(&red[3][4 * c]);
        *reinterpret_cast<f32x4*>(out + 4 * i4) = t;
    }
}

// out[r] = a[r] . b[r]
__global__ void rowdot_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int rows, int cols) {
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= rows) return;
    float s = 0.f;
    for (int c = threadIdx.x & 63; c < cols; c += 64) s += a[(size_t)r * cols + c] * b[(size_t)r * cols + c];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) out[r] = s;
}

// delta[b][q] (+)= sum_m A[q][m] dA[q][m] over one slide's ragged [n_q][M_b] block
__global__ void map_rowdot_kernel(const float* __restrict__ a_map, const float* __restrict__ da_map, const int* __restrict__ cu,
                                  float* __restrict__ delta, int n_q, int accumulate) {
    const int q = blockIdx.x, b = blockIdx.y;
    const int row_begin = cu[b], m_rows = cu[b + 1] - row_begin;
    const size_t base = (size_t)n_q * row_begin + (size_t)q * m_rows;
    float s = 0.f;
    for (int m = threadIdx.x; m < m_rows; m += blockDim.x) s += a_map[base + m] * da_map[base + m];
    __shared__ float red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = red[0] + red[1] + red[2] + red[3];
        float* d = delta + (size_t)b * n_q + q;
        *d = accumulate ? *d + t : t;
    }
}

// out[q-row of slide b][m] = scale[b] * a[..][m]  over a slide's ragged [n_q][M_b] block (backward of a per-slide map norm)
__global__ void map_block_scale_kernel(const float* __restrict__ a_map, const float* __restrict__ scale, const int* __restrict__ cu,
                                       float* __restrict__ out, int n_q) {
    const int q = blockIdx.x, b = blockIdx.y;
    const int row_begin = cu[b], m_rows = cu[b + 1] - row_begin;
    const size_t base = (size_t)n_q * row_begin + (size_t)q * m_rows;
    const float sc = scale[b];
    for (int m = threadIdx.x; m < m_rows; m += blockDim.x) out[base + m] = sc * a_map[base + m];
}

}  // namespace

int mpo_launch_map_block_scale(const float* a_map, const float* scale, const int* cu, float* out, int n_slides, int n_q,
                               hipStream_t stream) {
    dim3 grid(n_q, n_slides);
    map_block_scale_kernel<<<grid, 256, 0, stream>>>(a_map, scale, cu, out, n_q);
    MPO_LAUNCH_CHECK();
    return 0;
}

int mpo_launch_coattn_bwd(const void* bag, int bag_f32, const int* cu, int n_slides, int embed,
                          const float* qk2, const float* lse2, const float* dctx, const float* delta, const float* ctx,
                          const float* a_map, const float* da_map,
                          void* dbag, float* part_dqk, float* part_colsum, int n_q, const BagPlan& plan, float relu_gate,
                          hipStream_t stream) {
    (void)n_slides;
    (void)a_map;
    MPO_CHECK(relu_gate == 0.f || !bag_f32, "coattn backward: the fused relu/dropout gate needs a bf16 bag");
    MPO_CHECK(delta || ctx, "coattn backward: delta or ctx");
    if (mpo_coattn_bwd8_covers(bag_f32, embed, n_q, da_map))
        return mpo_launch_coattn_bwd8(bag, cu, qk2, lse2, dctx, delta, ctx, dbag, part_dqk, part_colsum, n_q, plan, relu_gate, stream);
    if (mpo_coattn_bwd_f32_covers(bag_f32, embed, n_q, da_map))
        return mpo_launch_coattn_bwd_f32(bag, cu, qk2, lse2, dctx, delta, ctx, da_map, dbag, part_dqk, part_colsum, n_q, plan, stream);
    dim3 grid = plan_grid(plan);
#define MPO_BWD_CASE(EV)                                                                                     \
    case EV:                                                                                                 \
        if (bag_f32)                                                                                         \
            coattn_bwd_kernel<EV, true><<<grid, BwdCfg<EV, true>::WAVES * 64, 0, stream>>>(                  \
                bag, cu, qk2, lse2, dctx, delta, ctx, da_map, dbag, part_dqk, part_colsum, n_q, plan, relu_gate);  \
        else                                                                                                 \
            coattn_bwd_kernel<EV, false><<<grid, BwdCfg<EV, false>::WAVES * 64, 0, stream>>>(                \
                bag, cu, qk2, lse2, dctx, delta, ctx, da_map, dbag, part_dqk, part_colsum, n_q, plan, relu_gate);  \
        break;
    switch (embed) {
        MPO_BWD_CASE(128)
        MPO_BWD_CASE(256)
        MPO_BWD_CASE(512)
        default:
            mpo_set_error("coattn backward: embed_dim %d not in {128,256,512}", embed);
            return 1;
    }
#undef MPO_BWD_CASE
    MPO_LAUNCH_CHECK();
    return 0;
}

int mpo_launch_bag_finish(const BagFinish& f, int n_slides, int n_q, int embed, const BagPlan& plan, hipStream_t stream) {
    MPO_CHECK(f.n_red >= 0 && f.n_red <= 2, "bag finish: 0..2 reductions (got %d)", f.n_red);
    MPO_CHECK((embed & 3) == 0 && (f.cs_cols & 3) == 0, "bag finish: widths must be multiples of 4");
    const int per = n_q * embed;
    int bx = f.n_red ? (per / 4 + 63) / 64 : 1;
    if (f.part_cs) bx = max(bx, (f.cs_cols / 4 + 7) / 8);
    if (f.zero[0] || f.zero[1]) bx = max(bx, 8);
    dim3 grid(bx, n_slides * f.n_red + 1);
    bag_finish_kernel<<<grid, 256, 0, stream>>>(f, n_slides, per, plan);
    MPO_LAUNCH_CHECK();
    return 0;
}
int mpo_launch_coattn_bwd_reduce(const float* part_dqk, float* dqk, int n_slides, int n_q, int embed, const BagPlan& plan,
                                 hipStream_t stream) {
    BagFinish f{};
    f.part[0] = part_dqk; f.out[0] = dqk; f.n_red = 1;
    return mpo_launch_bag_finish(f, n_slides, n_q, embed, plan, stream);
}

int mpo_launch_rowdot(const float* a, const float* b, float* out, int rows, int cols, hipStream_t stream) {
    if (rows <= 0) return 0;
    rowdot_kernel<<<(rows + 3) / 4, 256, 0, stream>>>(a, b, out, rows, cols);
    MPO_LAUNCH_CHECK();
    return 0;
}

int mpo_launch_map_rowdot(const float* a_map, const float* da_map, const int* cu, float* delta, int n_slides, int n_q,
                          int accumulate, hipStream_t stream) {
    dim3 grid(n_q, n_slides);
    map_rowdot_kernel<<<grid, 256, 0, stream>>>(a_map, da_map, cu, delta, n_q, accumulate);
    MPO_LAUNCH_CHECK();
    return 0;
}
