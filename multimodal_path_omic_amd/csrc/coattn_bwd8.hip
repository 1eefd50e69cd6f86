// K1 backward for the benchmarked geometry -- bf16 bag, embed_dim 256, at most 8 omic queries, no gradient on the map --
// with TWO waves per SIMD (8 waves per workgroup).  Same mathematics and the same outputs as coattn_bwd_kernel<256, false>
// (coattn_bwd.hip: dH written once, split-M partials of dqk, column sums of the emitted dH); the general kernel keeps every
// other geometry (fp32 bags, E = 128 / 512, 9..16 queries, a gradient arriving on the map).
//
// Why a second kernel.  The general kernel runs ONE wave per SIMD at ~500 registers (query operands, a staged tile and the
// dqk accumulators all live in the register file): per 32-row tile a wave spends ~17 k cycles on 128 MFMAs (2 k cycles),
// ~1.5 k vector and ~150 LDS instructions, i.e. it mostly waits for its own dependent instructions, and nothing else is
// resident on the SIMD to issue meanwhile (r02: 0.44-0.47 of HBM peak inside the step; the same pass with tiles direct to LDS
// and 186 + 256 registers did not move, NOTES.md).  Here a wave fits 256 registers, so two waves share a SIMD and one
// computes while the other waits for its tile:
//   * query-side operands (qk hi/lo, dctx hi/lo) are 16-byte MFMA fragments in LDS, compact [array][k-step][lane group][8 query
//     slots] (16 KiB per workgroup instead of 128 registers per lane);
//   * ONE tile image per wave, filled global -> LDS directly (global_load_lds_dwordx4, the image's chunk swizzle applied to the
//     GLOBAL chunk a lane fetches); no staging registers.  A wave requests its next tile right after the copy-out of the
//     current one has read the image and then waits for it -- the sibling wave of the SIMD is in its compute phase meanwhile;
//   * both orientations of the two row products (scores, dA) come out of ONE pass over the fragments: every k-step issues the
//     tile fragment once as MFMA B operand (patch on the lane -> A, dS in the k-order of the dH product) and once as A operand
//     (query on the lane -> dS^T in the k-order of the dqk accumulation).  64 more MFMAs per tile on a matrix pipe that idles
//     anyway, and the 2-KiB-per-wave transposition pad of the general kernel (8 scalar LDS writes per lane, a wave barrier,
//     two reads) is gone;
//   * dH leaves the MFMA as 4 consecutive embed columns of one patch row per lane (8 bytes).  The general kernel stores those
//     8-byte slots (4-way bank conflicts: 47 % of its LDS cycles, r01 SQ counters).  Here lane groups g and g ^ 1 exchange
//     halves (v_permlane16_swap_b32: a lane then owns ONE 16-byte chunk of one row), so the ReLU/dropout gate reads H and the
//     store writes dH as ds_read_b128 / ds_write_b128 (reads conflict-free with the image swizzle below; the stores stay
//     2-way -- eight consecutive rows per store group -- at a cost below the instruction's own 13 cycles);
//   * the Z^T operand of the dH product is kept for lane groups 0 / 1 only (queries 0..7; 8 KiB): the A / dS values of dead
//     queries are exact zeros, so what their k-slots multiply is irrelevant as long as it is finite.  The gate's scale
//     1 / (1 - p) is folded into Z.
// LDS: 8 x 16 KiB images + 8 KiB Z + 16 KiB query fragments = 152 KiB.
//
// Roofline: HBM.  Algorithmic bytes per patch row 2 x 512 (H_bag read, dH written); 32 x 15 000 rows: 491.52 MB per launch.
#include "coattn_tile.h"
#include "mpo_kernels.h"

namespace {

constexpr int E8 = 256;
constexpr int W8 = 8;                                     // waves per workgroup: two per SIMD
constexpr int QS = 8;                                     // query slots of a fragment row (dead slots hold zeros)
using G8 = TileGeom<E8>;
constexpr int OFF_Z8 = W8 * G8::TILEB;                    // 131 072
constexpr int Z8_BYTES = G8::DT * 32 * 16;                // [t][lane group 0 / 1][16 lanes] fragments: 8 KiB
constexpr int OFF_QF8 = OFF_Z8 + Z8_BYTES;                // 139 264
constexpr int QF8_ARR = G8::KS * 4 * QS * 16;             // one operand array [k-step][lane group][slot]: 4 KiB
constexpr int LDS8 = OFF_QF8 + 4 * QF8_ARR;               // 155 648
static_assert(LDS8 <= 160 * 1024, "LDS budget");

// Image layout: row-major 512-byte rows, 16-byte chunk c of row r at  c ^ sw8(r),  sw8(r) = 2 (r & 7) ^ (r >> 4): K1's
// swizzle (coattn_tile.h) with the 16-row half of the tile folded into bit 0.  Row-operand reads (one 16-row half per
// instruction) and transposed reads are conflict-free as with K1's; the extra bit makes the per-lane 16-byte dH slots
// conflict-free too: a ds_read_b128 lane group mixes lane groups g and g + 1 (MI355X_MICROARCH.md, LDS), which own the SAME
// chunk of rows 16 apart -- without the bit they would sit on the same banks (measured: 38 % of the kernel's LDS cycles).
__device__ __forceinline__ int sw8(int r) { return ((r & 7) << 1) ^ ((r >> 4) & 1); }
__device__ __forceinline__ bf16x8 row_frag8(const char* tile, int pt, int s, int lane) {
    const int p = lane & 15, g = lane >> 4;
    const int c = (4 * s + g) ^ ((lane & 7) << 1) ^ pt;
    return *reinterpret_cast<const bf16x8*>(tile + (16 * pt + p) * G8::ROWB + (c << 4));
}
__device__ __forceinline__ bf16x8 col_frag8(const char* tile, int t, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const int q4 = i >> 2, p4 = i & 3;
    const int r0 = 4 * g + q4;                       // rows r0 (half 0) and 16 + r0 (half 1: bit 0 of the chunk flipped)
    const int c = (2 * t + (p4 >> 1)) ^ ((r0 & 7) << 1);
    const int off = r0 * G8::ROWB + 8 * (p4 & 1);
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + off + (c << 4)));
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + off + ((c ^ 1) << 4) + 16 * G8::ROWB));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ unsigned lds_addr8(const char* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
// one LDS-DMA wave-instruction: lane l's 16 bytes at `src` land at lds_dst + 16 l (lds_dst wave-uniform)
__device__ __forceinline__ void glds16_8(const char* src, unsigned lds_dst) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"       // m0 is "reserved": nothing else in this kernel uses it
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(lds_dst) : "memory", "m0");
#pragma clang diagnostic pop
}

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__global__ __launch_bounds__(W8 * 64, 1)
void coattn_bwd8_kernel(const __bf16* __restrict__ bag, const int* __restrict__ cu,
                        const float* __restrict__ qk2,      // [n_slides][n_q][256] log2 units
                        const float* __restrict__ lse2,     // [n_slides][n_q]      log2 units
                        const float* __restrict__ dctx,     // [n_slides][n_q][256]
                        const float* __restrict__ delta,    // [n_slides][n_q] or NULL: rowsum(dctx * ctx) computed here
                        const float* __restrict__ ctx,      // [n_slides][n_q][256], read when delta == NULL
                        __bf16* __restrict__ dbag,          // [total_rows][256]
                        float* __restrict__ part_dqk,       // [parts][n_q][256] (natural units)
                        float* __restrict__ part_colsum,    // nullable [parts][256]: column sums of the dH rows written here
                        int n_q, BagPlan plan,
                        float relu_gate /* 0: off; else 1/(1-p): dH *= (H > 0 ? relu_gate : 0) */) {
    __shared__ __attribute__((aligned(1024))) char lds[LDS8];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const WgGeom wg = wg_geom(cu, plan);
    const int b = wg.b, row_begin = wg.row_begin, m_rows = wg.m_rows, r0 = wg.r0, r1 = wg.r1, ntiles = wg.ntiles;
    const int n_my = wave < ntiles ? (ntiles - wave + W8 - 1) / W8 : 0;
    const int c16 = lane & 15, g = lane >> 4;
    const float* qk_b = qk2 + (size_t)b * n_q * E8;
    const float* dc_b = dctx + (size_t)b * n_q * E8;
    char* img = lds + wave * G8::TILEB;
    const unsigned img_lds = lds_addr8(img);
    const char* slide = reinterpret_cast<const char*>(bag) + (size_t)row_begin * G8::ROWB;
    char* dslide = reinterpret_cast<char*>(dbag) + (size_t)row_begin * G8::ROWB;

    // rows [trow, trow + 32) of the slide -> this wave's image: instruction i carries rows 2 i, 2 i + 1 (1 KiB, linear on the
    // LDS side); lane (row = lane >> 5, position = lane & 31) fetches the global chunk position ^ sw8(row), which is the
    // chunk the image keeps at that position.  Rows past the slide are clamped (finite data; masked below).
    auto issue_tile = [&](int trow) {
        int el = lane;
        asm volatile("" : "+v"(el));                              // (per call: the address parts are not hoisted and kept)
        const int rl = el >> 5, cpos = el & 31;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = 2 * i + rl;
            const int c = cpos ^ sw8(r);
            const int grow = min(trow + r, m_rows - 1);
            glds16_8(slide + (size_t)grow * G8::ROWB + (c << 4), img_lds + i * 1024);
        }
    };
    if (n_my > 0) issue_tile(r0 + kTileRows * wave);             // in flight under the prologue

    // ---- prologue: query-side fragments and Z^T into LDS
    {
        // pair 0: qk (log2 units), pair 1: dctx; entry (pair, s, g', slot): x[slot][32 s + 8 g' .. + 7] split hi / lo
        for (int e = tid; e < 2 * G8::KS * 4 * QS; e += W8 * 64) {
            const int slot = e % QS;
            int rest = e / QS;
            const int gg = rest & 3;
            rest >>= 2;
            const int s = rest & (G8::KS - 1), pair = rest >> 3;
            const float* row = (pair ? dc_b : qk_b) + (slot < n_q ? slot : 0) * E8 + 32 * s + 8 * gg;
            const float live = slot < n_q ? 1.0f : 0.0f;
            const f32x4 a = *reinterpret_cast<const f32x4*>(row), c = *reinterpret_cast<const f32x4*>(row + 4);
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = a[j] * live;
                v[4 + j] = c[j] * live;
            }
            bf16x8 h, l;
            pack_hi_lo(v, h, l);
            char* dst = lds + OFF_QF8 + (2 * pair) * QF8_ARR + (((s * 4 + gg) * QS + slot) << 4);
            *reinterpret_cast<bf16x8*>(dst) = h;
            *reinterpret_cast<bf16x8*>(dst + QF8_ARR) = l;
        }
        // Z^T fragment of column tile t for lane (d = 16 t + i, g' < 2): j < 4: dctx[4 g' + j][d], j >= 4: qk_nat[4 g' + j - 4][d],
        // times the gate's scale
        const float zs = relu_gate != 0.f ? relu_gate : 1.0f;
        for (int e = tid; e < G8::DT * 32; e += W8 * 64) {
            const int i = e & 15, gg = (e >> 4) & 1, t = e >> 5;
            float z[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int qq = 4 * gg + j;
                const int qc = qq < n_q ? qq : n_q - 1;
                const float live = qq < n_q ? zs : 0.0f;
                z[j] = dc_b[qc * E8 + 16 * t + i] * live;
                z[4 + j] = qk_b[qc * E8 + 16 * t + i] * (live * kLn2);
            }
            bf16x8 h, l;
            pack_hi_lo(z, h, l);
            *reinterpret_cast<bf16x8*>(lds + OFF_Z8 + (e << 4)) = h;
        }
    }
    // per-lane row constants in both orientations; +inf lse switches dead query rows off (A = 0)
    const float lse_q = c16 < n_q ? lse2[(size_t)b * n_q + c16] : INFINITY;
    float del_q, lse_p[4], del_p[4];
    if (delta != nullptr) {
        del_q = c16 < n_q ? delta[(size_t)b * n_q + c16] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) del_p[r] = 4 * g + r < n_q ? delta[(size_t)b * n_q + 4 * g + r] : 0.f;
    } else {
        // delta[q] = dctx[q] . ctx[q]: lane (q = c16, g) sums a quarter of the row, every wave on its own
        const float* cx_b = ctx + (size_t)b * n_q * E8;
        float acc = 0.f;
        if (c16 < n_q) {
#pragma unroll 4
            for (int e = g * (E8 / 4); e < (g + 1) * (E8 / 4); e += 4) {
                const f32x4 u = *reinterpret_cast<const f32x4*>(dc_b + c16 * E8 + e);
                const f32x4 v = *reinterpret_cast<const f32x4*>(cx_b + c16 * E8 + e);
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
    __syncthreads();                                              // fragments and Z visible

    f32x4 accq[G8::DT];
#pragma unroll
    for (int t = 0; t < G8::DT; ++t) accq[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float csum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) csum[j] = 0.f;
    const bool gate_on = relu_gate != 0.f;

    for (int it = 0; it < n_my; ++it) {
        const int trow = r0 + kTileRows * (wave + it * W8);
        const int nvalid = min(kTileRows, r1 - trow);
        int el = lane;
        asm volatile("" : "+v"(el));                              // (per tile: LDS addresses are formed here, not carried)
        const int ec = el & 15, eg = el >> 4;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the tile has landed (and the previous tile's stores are out)

        bf16x8 wph[2], wh, wl;
        {
            // ---------------- the two row products, both orientations, one pass over the fragments
            f32x4 sT[2], dT[2], sN[2], dN[2];
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) sT[pt] = dT[pt] = sN[pt] = dN[pt] = f32x4{0.f, 0.f, 0.f, 0.f};
            // (dead MFMA rows / columns -- queries n_q..15 -- read slot ec & 7: zeros, or a live query's finite fragment whose
            //  products the +inf log-sum-exp below turns into exact zeros; lanes ec and ec + 8 share an address: a broadcast)
            const char* qf = lds + OFF_QF8 + ((eg * QS + (ec & 7)) << 4);
#pragma unroll
            for (int s = 0; s < G8::KS; ++s) {
                const bf16x8 a0 = row_frag8(img, 0, s, el);
                const bf16x8 a1 = row_frag8(img, 1, s, el);
                const char* qs = qf + s * (4 * QS * 16);
                const bf16x8 qh = *reinterpret_cast<const bf16x8*>(qs);
                const bf16x8 ql = *reinterpret_cast<const bf16x8*>(qs + QF8_ARR);
                const bf16x8 dh = *reinterpret_cast<const bf16x8*>(qs + 2 * QF8_ARR);
                const bf16x8 dl = *reinterpret_cast<const bf16x8*>(qs + 3 * QF8_ARR);
                // patch on the lane: out[r] = x[4 g + r] . tile[16 pt + (lane & 15)]
                sT[0] = mfma_bf16(qh, a0, sT[0]);
                sT[1] = mfma_bf16(qh, a1, sT[1]);
                dT[0] = mfma_bf16(dh, a0, dT[0]);
                dT[1] = mfma_bf16(dh, a1, dT[1]);
                // query on the lane: out[r] = tile[16 pt + 4 g + r] . x[lane & 15]
                sN[0] = mfma_bf16(a0, qh, sN[0]);
                sN[1] = mfma_bf16(a1, qh, sN[1]);
                dN[0] = mfma_bf16(a0, dh, dN[0]);
                dN[1] = mfma_bf16(a1, dh, dN[1]);
                sT[0] = mfma_bf16(ql, a0, sT[0]);
                sT[1] = mfma_bf16(ql, a1, sT[1]);
                dT[0] = mfma_bf16(dl, a0, dT[0]);
                dT[1] = mfma_bf16(dl, a1, dT[1]);
                sN[0] = mfma_bf16(a0, ql, sN[0]);
                sN[1] = mfma_bf16(a1, ql, sN[1]);
                dN[0] = mfma_bf16(a0, dl, dN[0]);
                dN[1] = mfma_bf16(a1, dl, dN[1]);
            }
            // patch on the lane -> B operand of the dH product: k-slot (g, j < 4) = A[4 g + j][p], (g, j >= 4) = dS[4 g + j - 4][p]
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const bool ok = 16 * pt + ec < nvalid;
                float w[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = ok ? __builtin_amdgcn_exp2f(sT[pt][r] - lse_p[r]) : 0.f;
                    w[r] = a;
                    w[4 + r] = a * (dT[pt][r] - del_p[r]);
                }
                bf16x8 lo_unused;
                pack_hi_lo(w, wph[pt], lo_unused);
            }
            // query on the lane -> dS^T in the k-order of col_frag (patches 4 g + j, 16 + 4 g + j)
            float ds[8];
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = 16 * pt + 4 * eg + r < nvalid;
                    const float a = ok ? __builtin_amdgcn_exp2f(sN[pt][r] - lse_q) : 0.f;
                    ds[4 * pt + r] = a * (dN[pt][r] - del_q);
                }
            pack_hi_lo(ds, wh, wl);
        }
        // ---------------- dqk^T[d][q] += H^T[d][p] dS^T[p][q]   (reads the H image: before dH overwrites it)
#pragma unroll
        for (int t = 0; t < G8::DT; ++t) {
            const bf16x8 hf = col_frag8(img, t, el);
            accq[t] = mfma_bf16(hf, wh, accq[t]);
            accq[t] = mfma_bf16(hf, wl, accq[t]);
        }

        // ---------------- dH^T[d][p] = Z^T W^T; lane groups g, g ^ 1 exchange halves, the lane then owns the 16-byte chunk
        // 2 t + (g >> 1) of row 16 (g & 1) + (lane & 15): gate against H read from that very slot, dH written over it
        {
            const int row = 16 * (eg & 1) + ec;
            char* rowp = img + row * G8::ROWB;
            const int swz = sw8(row), chi = eg >> 1;
            const char* zf = lds + OFF_Z8 + ((((eg & 1) << 4) + ec) << 4);
#pragma unroll
            for (int t = 0; t < G8::DT; ++t) {
                const bf16x8 zh = *reinterpret_cast<const bf16x8*>(zf + t * 512);
                const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                const f32x4 o0 = mfma_bf16(zh, wph[0], z4);       // rows d = 16 t + 4 g + r of patch row (lane & 15)
                const f32x4 o1 = mfma_bf16(zh, wph[1], z4);       // ... of patch row 16 + (lane & 15)
                const bf16x4 b0 = {f2bf(o0[0]), f2bf(o0[1]), f2bf(o0[2]), f2bf(o0[3])};
                const bf16x4 b1 = {f2bf(o1[0]), f2bf(o1[1]), f2bf(o1[2]), f2bf(o1[3])};
                typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                const u32x2 x = __builtin_bit_cast(u32x2, b0), y = __builtin_bit_cast(u32x2, b1);
                // even lane groups keep their pt-0 half and receive the partner's pt-0 half; odd groups receive the partner's
                // pt-1 half in front of their own (v_permlane16_swap_b32: odd rows of the first <-> even rows of the second)
                const auto s0 = __builtin_amdgcn_permlane16_swap(x[0], y[0], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(x[1], y[1], false, false);
                u32x4 out = {s0[0], s1[0], s0[1], s1[1]};
                u32x4* slot = reinterpret_cast<u32x4*>(rowp + (((2 * t + chi) ^ swz) << 4));
                if (gate_on) {
                    // H = drop(relu(.)) >= 0 in bf16: a 16-bit field f is non-zero iff bit 15 of (f & 0x7FFF) + 0x7FFF is set
                    const u32x4 hv = *slot;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned u = ((hv[k] & 0x7FFF7FFFu) + 0x7FFF7FFFu) & 0x80008000u;
                        out[k] &= __umul24(u >> 15, 0xFFFFu);           // 0x0001'0001 pattern x 0xFFFF: v_mul_u32_u24, full rate
                    }
                }
                *slot = out;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // ---------------- copy the dH image out in whole rows (1 KiB per wave-instruction) + its column sums
        {
            const int rl = el >> 5, cc = el & 31;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int r = 2 * i + rl;
                const f32x4 v = *reinterpret_cast<const f32x4*>(img + r * G8::ROWB + ((cc ^ sw8(r)) << 4));
                if (r < nvalid) *reinterpret_cast<f32x4*>(dslide + ((size_t)(trow + r) * 32 + cc) * 16) = v;
                if (part_colsum != nullptr) {                     // (rows past the range carry exact zeros: A = 0 there)
                    const bf16x8 hv = __builtin_bit_cast(bf16x8, v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) csum[j] += (float)hv[j];
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // every read of the image is done: it may be refilled
        if (it + 1 < n_my) issue_tile(trow + kTileRows * W8);
    }

    // ---- merge the waves' dqk through LDS and write this workgroup's partial
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
        float* wq = reinterpret_cast<float*>(img);                // [16][256] floats = this wave's image
#pragma unroll
        for (int t = 0; t < G8::DT; ++t)
            *reinterpret_cast<f32x4*>(wq + c16 * E8 + 16 * t + 4 * g) = accq[t];
    }
    __syncthreads();
    const size_t pbase = wg.part;
    for (int idx = tid; idx < n_q * E8; idx += W8 * 64) {
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < W8; ++w) a += reinterpret_cast<const float*>(lds + w * G8::TILEB)[idx];
        part_dqk[pbase * n_q * E8 + idx] = a;
    }
    if (part_colsum != nullptr) {                                 // same exchange for the column sums
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) csum[j] += __shfl_xor(csum[j], 32);       // lanes l, l + 32 share a column chunk
        float* wc = reinterpret_cast<float*>(img);
        if (lane < 32) {
#pragma unroll
            for (int j = 0; j < 8; ++j) wc[lane * 8 + j] = csum[j];
        }
        __syncthreads();
        for (int idx = tid; idx < E8; idx += W8 * 64) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < W8; ++w) a += reinterpret_cast<const float*>(lds + w * G8::TILEB)[idx];
            part_colsum[pbase * E8 + idx] = a;
        }
    }
}

int g_bwd8_enabled = 1;

}  // namespace

// 1: geometries the two-waves-per-SIMD kernel covers go through it (default); 0: everything on the general kernel (the
// tests' cross-check).  Returns the previous setting.
int mpo_coattn_bwd8_enable(int enabled) {
    const int prev = g_bwd8_enabled;
    g_bwd8_enabled = enabled ? 1 : 0;
    return prev;
}

bool mpo_coattn_bwd8_covers(int bag_f32, int embed, int n_q, const float* da_map) {
    return g_bwd8_enabled && !bag_f32 && embed == E8 && n_q >= 1 && n_q <= 8 && da_map == nullptr;
}

int mpo_launch_coattn_bwd8(const void* bag, const int* cu, const float* qk2, const float* lse2, const float* dctx,
                           const float* delta, const float* ctx, void* dbag, float* part_dqk, float* part_colsum, int n_q,
                           const BagPlan& plan, float relu_gate, hipStream_t stream) {
    MPO_CHECK(n_q >= 1 && n_q <= 8, "coattn backward (8 waves): 1..8 queries (got %d)", n_q);
    MPO_CHECK(delta || ctx, "coattn backward: delta or ctx");
    coattn_bwd8_kernel<<<plan_grid(plan), W8 * 64, 0, stream>>>(
        reinterpret_cast<const __bf16*>(bag), cu, qk2, lse2, dctx, delta, ctx, reinterpret_cast<__bf16*>(dbag), part_dqk,
        part_colsum, n_q, plan, relu_gate);
    MPO_LAUNCH_CHECK();
    return 0;
}
