// Patch-side gradient of K2 (NaCAGaT narrow-gated co-attention, models/blocks.py:151-206 backward) for a bf16 bag, ONE pass
// with the product back through the key projection inside:
//     d_bag[m][e'] = ( sum_e dK[m][e] W_k[e][e']  +  sum_q A_drop[q][m] dctx[q][e'] ) * (H[m][e'] > 0 ? gate : 0)
// dK = the key gradient (bf16, from bag_outer_gated), K = H W_k^T + b_k the caller-side key projection, A_drop the ragged
// post-dropout map, dctx the gradient of the value-side context, H the bag itself: the sign of H = dropout(relu(.)) is the
// ReLU / dropout derivative of the patch layer that produced it, gate = 1 / (1 - p).  Column sums of the emitted rows (= that
// layer's bias gradient) on the way out.
//
// Replaces a library bf16 GEMM (dK W_k: 133 us of hipBLASLt per 32 x 15 000-row window, r02) followed by
// bag_outer_gate_kernel (150 us: re-reads the GEMM's output and H, adds the rank-n_q outer product, gates, rewrites):
// reads dK and H once, writes d_bag once: 3 x 512 B per patch row.
//
// A workgroup of 8 waves owns a row range of ONE slide (the window's work plan) and walks it in 32-row tiles; wave w owns
// output columns 32 w .. + 31 and keeps ITS slice of W_k as MFMA A fragments in registers for the whole kernel (2 column
// tiles x 8 k-steps, one bf16 term: the result is emitted in bf16), plus the slide's dctx columns (hi / lo) for the outer
// product, which is one more MFMA k-step with the queries as k.  The product is taken transposed -- D[e'][patch] -- so a
// lane ends with four consecutive output columns of one patch row; lane groups g, g ^ 1 exchange halves
// (v_permlane16_swap_b32) and the lane then owns ONE 16-byte chunk: gate against H read from that slot of the H image, result
// written over it (conflict-free b128 accesses), then the image leaves in whole rows.  dK and H tiles are staged
// cooperatively global -> registers two tiles ahead -> double-buffered LDS images (keyproj.hip's pipeline).
#include "coattn_tile.h"
#include "mpo_kernels.h"

namespace {

constexpr int PG_E = 256;
constexpr int PG_WAVES = 8;
constexpr int PG_THREADS = PG_WAVES * 64;
using PGG = TileGeom<PG_E>;                                  // ROWB 512, TILEB 16 KiB, KS 8
constexpr int PG_MAPB = kTileRows * 16 * 4;                  // map values of a tile: [32 rows][16 queries] floats = 2 KiB
constexpr int PG_BUF = 2 * PGG::TILEB + PG_MAPB;             // dK image | H image (becomes the output image) | map values
typedef unsigned int pg_u32x4 __attribute__((ext_vector_type(4)));
// chunk swizzle of the H / output image: K1's 2 (r & 7) with the 16-row half folded into bit 0 -- a ds_read_b128 lane group
// mixes lane groups g and g + 1, which own the same chunk of rows 16 apart (see coattn_bwd8.hip)
__device__ __forceinline__ int pg_sw(int r) { return ((r & 7) << 1) ^ ((r >> 4) & 1); }

__global__ __launch_bounds__(PG_THREADS, 1)
void k2_patch_grad_kernel(const int* __restrict__ cu, const pg_u32x4* __restrict__ dk, const float* __restrict__ w_k,
                          const float* __restrict__ amap, const float* __restrict__ dctx, const pg_u32x4* __restrict__ hbag,
                          pg_u32x4* __restrict__ out, float gate /* 0: no gating */, float* __restrict__ part_colsum /* nullable */,
                          int n_q, BagPlan plan) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * PG_BUF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    const WgGeom wg = wg_geom(cu, plan);
    const int b = wg.b, m_rows = wg.m_rows, r0 = wg.r0, r1 = wg.r1, ntiles = wg.ntiles;
    const int n0 = 32 * wave;

    // W_k slice as MFMA A fragments: lane (i = output column e' = n0 + 16 ct + c16, g) holds W_k[e = 32 s + 8 g + j][e'], j = 0..7
    bf16x8 wf[2][PGG::KS];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int s = 0; s < PGG::KS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) wf[ct][s][j] = (__bf16)w_k[(size_t)(32 * s + 8 * g + j) * PG_E + n0 + 16 * ct + c16];
    // dctx columns of this slide as the A operand of the outer product: k-slot (g, j < 4) = query 4 g + j, slots j >= 4 zero
    bf16x8 zh[2], zl[2];
    {
        const float* dcb = dctx + (size_t)b * n_q * PG_E;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            float z[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int qq = 4 * g + j;
                z[j] = qq < n_q ? dcb[qq * PG_E + n0 + 16 * ct + c16] : 0.f;
                z[4 + j] = 0.f;
            }
            pack_hi_lo(z, zh[ct], zl[ct]);
        }
    }
    const size_t slide_ch = (size_t)wg.row_begin * 32;          // 16-byte chunks before this slide
    const pg_u32x4* dks = dk + slide_ch;
    const pg_u32x4* hs = hbag + slide_ch;
    pg_u32x4* os = out + slide_ch;
    const float* amb = amap + (size_t)n_q * wg.row_begin;      // slide block [n_q][m_rows]

    // staging: thread t moves chunks t and t + 512 of the dK tile and of the H tile, and one map value (row t & 31, query t >> 5)
    pg_u32x4 ka[2], ha[2], kb[2], hb[2];
    float ma, mb;
    const int mq = tid >> 5, mr = tid & 31;
    auto fetch = [&](pg_u32x4 (&kk)[2], pg_u32x4 (&hh)[2], float& mm, int it) {
        it = it < ntiles ? it : ntiles - 1;                       // (clamped: uniform control flow, a few redundant loads at the tail)
        const int row0 = r0 + kTileRows * it;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ci = tid + i * PG_THREADS;
            const int row = min(row0 + (ci >> 5), m_rows - 1);
            kk[i] = dks[(size_t)row * 32 + (ci & 31)];
            hh[i] = hs[(size_t)row * 32 + (ci & 31)];
        }
        const int row = min(row0 + mr, m_rows - 1);
        mm = amb[(size_t)min(mq, n_q - 1) * m_rows + row];
    };
    auto stage = [&](const pg_u32x4 (&kk)[2], const pg_u32x4 (&hh)[2], float mm, char* buf, int it) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ci = tid + i * PG_THREADS;
            const int r = ci >> 5, cc = ci & 31;
            *reinterpret_cast<pg_u32x4*>(buf + r * PGG::ROWB + ((cc ^ ((r & 7) << 1)) << 4)) = kk[i];
            *reinterpret_cast<pg_u32x4*>(buf + PGG::TILEB + r * PGG::ROWB + ((cc ^ pg_sw(r)) << 4)) = hh[i];
        }
        const bool live = mq < n_q && r0 + kTileRows * it + mr < r1;           // dead queries / rows past the range: zero
        reinterpret_cast<float*>(buf + 2 * PGG::TILEB)[mr * 16 + mq] = live ? mm : 0.f;
    };
    float csum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) csum[j] = 0.f;

    auto tile = [&](char* buf, int it) {
        const int row0 = r0 + kTileRows * it;
        const int nvalid = min(kTileRows, r1 - row0);
        const char* imk = buf;
        char* imh = buf + PGG::TILEB;
        const float* wm = reinterpret_cast<const float*>(buf + 2 * PGG::TILEB);
        f32x4 acc[2][2];                                          // [pt][ct]: D[e' = n0 + 16 ct + 4 g + r][patch 16 pt + c16]
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[pt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < PGG::KS; ++s) {
            const bf16x8 a0 = row_frag<PG_E>(imk, 0, s, lane);
            const bf16x8 a1 = row_frag<PG_E>(imk, 1, s, lane);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                acc[0][ct] = mfma_bf16(wf[ct][s], a0, acc[0][ct]);
                acc[1][ct] = mfma_bf16(wf[ct][s], a1, acc[1][ct]);
            }
        }
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {                           // + A_drop^T dctx: k = queries
            const f32x4 m4 = *reinterpret_cast<const f32x4*>(wm + (16 * pt + c16) * 16 + 4 * g);
            const float w[8] = {m4[0], m4[1], m4[2], m4[3], 0.f, 0.f, 0.f, 0.f};
            bf16x8 wh, wl;
            pack_hi_lo(w, wh, wl);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                acc[pt][ct] = mfma_bf16(zh[ct], wh, acc[pt][ct]);
                acc[pt][ct] = mfma_bf16(zh[ct], wl, acc[pt][ct]);
                acc[pt][ct] = mfma_bf16(zl[ct], wh, acc[pt][ct]);
            }
        }
        // lane groups g, g ^ 1 exchange halves: the lane then owns chunk 4 wave + 2 ct + (g >> 1) of row 16 (g & 1) + c16
        const int row = 16 * (g & 1) + c16;
        char* rowp = imh + row * PGG::ROWB;
        const int swz = pg_sw(row);
        const float sc = gate != 0.f ? gate : 1.0f;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const f32x4 o0 = acc[0][ct] * sc, o1 = acc[1][ct] * sc;
            const bf16x4 b0 = {f2bf(o0[0]), f2bf(o0[1]), f2bf(o0[2]), f2bf(o0[3])};
            const bf16x4 b1 = {f2bf(o1[0]), f2bf(o1[1]), f2bf(o1[2]), f2bf(o1[3])};
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
            const u32x2 x = __builtin_bit_cast(u32x2, b0), y = __builtin_bit_cast(u32x2, b1);
            const auto s0 = __builtin_amdgcn_permlane16_swap(x[0], y[0], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(x[1], y[1], false, false);
            pg_u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
            pg_u32x4* slot = reinterpret_cast<pg_u32x4*>(rowp + (((4 * wave + 2 * ct + (g >> 1)) ^ swz) << 4));
            if (gate != 0.f) {
                // H = drop(relu(.)) >= 0 in bf16: a 16-bit field f is non-zero iff bit 15 of (f & 0x7FFF) + 0x7FFF is set
                const pg_u32x4 hv = *slot;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned u = ((hv[k] & 0x7FFF7FFFu) + 0x7FFF7FFFu) & 0x80008000u;
                    o[k] &= __umul24(u >> 15, 0xFFFFu);
                }
            }
            *slot = o;
        }
        __syncthreads();                                           // the output image is complete
#pragma unroll
        for (int i = 0; i < 2; ++i) {                              // whole rows out + their column sums
            const int ci = tid + i * PG_THREADS;
            const int r = ci >> 5, cc = ci & 31;
            const pg_u32x4 v = *reinterpret_cast<const pg_u32x4*>(imh + r * PGG::ROWB + ((cc ^ pg_sw(r)) << 4));
            if (r < nvalid) {
                os[(size_t)(row0 + r) * 32 + cc] = v;
                if (part_colsum != nullptr) {
                    const bf16x8 hv = __builtin_bit_cast(bf16x8, v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) csum[j] += (float)hv[j];
                }
            }
        }
    };

    if (ntiles > 0) {
        fetch(ka, ha, ma, 0);
        fetch(kb, hb, mb, 1);
        for (int it = 0; it < ntiles; it += 2) {
            stage(ka, ha, ma, lds, it);
            fetch(ka, ha, ma, it + 2);
            __syncthreads();
            tile(lds, it);
            if (it + 1 < ntiles) {                                 // (workgroup-uniform)
                stage(kb, hb, mb, lds + PG_BUF, it + 1);
                fetch(kb, hb, mb, it + 3);
                __syncthreads();
                tile(lds + PG_BUF, it + 1);
            }
        }
    }
    if (part_colsum != nullptr) {
        // thread (chunk column cc = tid & 31; rows tid >> 5 and 16 + (tid >> 5) of every tile) holds the sums of 8 columns
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds);               // [16 row groups][256 columns]
#pragma unroll
        for (int j = 0; j < 8; ++j) red[(tid >> 5) * PG_E + 8 * (tid & 31) + j] = csum[j];
        __syncthreads();
        if (tid < PG_E) {
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) a += red[r * PG_E + tid];
            part_colsum[(size_t)wg.part * PG_E + tid] = a;
        }
    }
}

}  // namespace

int mpo_launch_k2_patch_grad(const int* cu, const void* dk_bf16, const float* w_k, const float* amap, const float* dctx,
                             const void* hbag_bf16, void* out_bf16, float gate, float* part_colsum, int n_q, int embed,
                             const BagPlan& plan, hipStream_t stream) {
    MPO_CHECK(embed == PG_E, "K2 patch-side gradient kernel: embed_dim %d not built (256 only)", embed);
    MPO_CHECK(n_q >= 1 && n_q <= 16, "K2 patch-side gradient: 1..16 queries (got %d)", n_q);
    k2_patch_grad_kernel<<<plan_grid(plan), PG_THREADS, 0, stream>>>(
        cu, static_cast<const pg_u32x4*>(dk_bf16), w_k, amap, dctx, static_cast<const pg_u32x4*>(hbag_bf16),
        static_cast<pg_u32x4*>(out_bf16), gate, part_colsum, n_q, plan);
    MPO_LAUNCH_CHECK();
    return 0;
}
