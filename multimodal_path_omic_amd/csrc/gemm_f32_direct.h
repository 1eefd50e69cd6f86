// Direct (no LDS staging) small-row fp32 GEMM kernels: templates only.  The instantiations live in four translation
// units (gemm_f32_direct_{single,group}{4,8}.hip) so that they compile in parallel -- in one file they were 2 minutes of
// the 2.4-minute clean build.
#pragma once
#include "gemm_f32_gate.h"

namespace {

// ------------------------------------------------------------------------------------------------ direct variant
// The token tail's products are ~25 MFLOP each and sit in dependent chains of ~100 launches per window: what
// counts is the latency of ONE product, not throughput.  This variant drops the LDS staging round trip:
//   * 16 x 16 outputs per workgroup (3-4x more workgroups than the staged kernel: every CU gets one),
//   * the four waves split K; each lane loads its MFMA fragments straight from global/L2 (a float4 per four
//     MFMAs when k is contiguous, four strided scalars otherwise), up to 12 k-blocks in flight at once,
//   * one LDS exchange at the end sums the four partial tiles; thread t then owns output element t.
// Measured r01 in a HIP graph: 192x256x256 6.8 us (staged) -> see DESIGN.md.
constexpr int DB = 16;                 // tile edge
constexpr int DMAXB = 8;               // k-blocks (of 16) a wave keeps in flight at most (12 made the compiler serialise the loads)

template <bool KC>
__device__ __forceinline__ f32x4 direct_frag(const float* __restrict__ p, int ld, int mn, int mn_lim, int k0, int k_lim, bool vec_ok) {
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
    if (mn >= mn_lim || k0 >= k_lim) return r;
    if (KC) {
        const float* q = p + (size_t)mn * ld + k0;
        if (vec_ok && k0 + 3 < k_lim) return *reinterpret_cast<const f32x4*>(q);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (k0 + j < k_lim) r[j] = q[j];
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (k0 + j < k_lim) r[j] = p[(size_t)(k0 + j) * ld + mn];
    }
    return r;
}
// gate factors for the four elements of one fragment; gv holds the gate tensor's values there (loaded WITH the
// fragment, so the gate never adds a dependent memory round trip)
template <bool KC>
__device__ __forceinline__ void direct_gate(f32x4& v, const f32x4& gv, const GateFn& gf, int ld, int mn, int mn_lim, int k0, int k_lim) {
    if (mn >= mn_lim || k0 >= k_lim) return;
    if (KC) {
        const size_t idx0 = (size_t)mn * ld + k0;
        if (gf.draws() && (idx0 & 3) == 0) {
            const uint64_t ctr = gf.off + (idx0 >> 2);
            const uint4 r = draw4x32((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)gf.seed, (uint32_t)(gf.seed >> 32));
            const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) if (k0 + j < k_lim) v[j] *= gf.with_word(gv[j], w[j]);
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) if (k0 + j < k_lim) v[j] *= gf(gv[j], idx0 + j);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (k0 + j < k_lim) v[j] *= gf(gv[j], (size_t)(k0 + j) * ld + mn);
    }
}
// Row-contiguous operand with a drawing gate: element (k0 + j, mn) has index (k0 + j) * ld + mn, so the four lanes of
// a quad (mn = 4q .. 4q+3) share ONE dropout counter per j.  Lane s of the quad draws the counter of j = s and the
// quad transposes the 4 x 4 words with four quad shuffles: one draw per lane and fragment instead of four.
// Every lane of the wave must call this (no early exit before the shuffles).
__device__ __forceinline__ void direct_gate_rows_quad(f32x4& v, const f32x4& gv, const GateFn& gf, int ld, int mn, int mn_lim,
                                                      int k0, int k_lim, int lane) {
    const int lq = lane & 3;
    const size_t idx_own = (size_t)(k0 + lq) * ld + (mn & ~3);           // first element of the quad's group for j = lq
    const uint64_t ctr = gf.off + (idx_own >> 2);
    const uint4 r = draw4x32((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)gf.seed, (uint32_t)(gf.seed >> 32));
    const uint32_t own[4] = {r.x, r.y, r.z, r.w};
    uint32_t w[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int pick = lq ^ t;                                         // word wanted by the lane this value goes to
        const uint32_t send = pick == 0 ? own[0] : pick == 1 ? own[1] : pick == 2 ? own[2] : own[3];
        const uint32_t got = (uint32_t)__shfl_xor((int)send, t);        // from lane lq ^ t: its word for lane lq
#pragma unroll
        for (int j = 0; j < 4; ++j) if (j == (lq ^ t)) w[j] = got;       // that lane drew element j = lq ^ t
    }
    if (mn >= mn_lim) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) if (k0 + j < k_lim) v[j] *= gf.with_word(gv[j], w[j]);
}

// NB k-blocks of one wave: every load is issued before the first MFMA (out-of-range fragments are zero-filled,
// so the count can be a compile-time constant and nothing is predicated)
template <bool A_KC, bool B_KC>
struct DirectCtx {
    const GemmArgs& g;
    const GateFn& gf;
    int m, n, kq, kend, lane;
    bool a_vec, b_vec, gated, want_bsum;
    template <int NB>
    __device__ __forceinline__ void chunk(int kbase, f32x4& acc0, f32x4& acc1, float& bsum) const {
        f32x4 a[NB], b[NB], gv[NB];
        const bool gate_tensor = gated && gf.g != nullptr;
        const bool quad_draw = gf.draws() && (g.lda & 3) == 0;           // (m0 is a multiple of 16: quads are aligned)
        const bool g_vec = a_vec && (reinterpret_cast<uintptr_t>(gf.g) & 15) == 0;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int k0 = kbase + 16 * u + 4 * kq;
            a[u] = direct_frag<A_KC>(g.A, g.lda, m, g.M, k0, kend, a_vec);
            b[u] = direct_frag<B_KC>(g.B, g.ldb, n, g.N, k0, kend, b_vec);
            gv[u] = gate_tensor ? direct_frag<A_KC>(gf.g, g.lda, m, g.M, k0, kend, g_vec) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            if (gated) {
                if (!A_KC && quad_draw) direct_gate_rows_quad(a[u], gv[u], gf, g.lda, m, g.M, kbase + 16 * u + 4 * kq, kend, lane);
                else direct_gate<A_KC>(a[u], gv[u], gf, g.lda, m, g.M, kbase + 16 * u + 4 * kq, kend);
            }
            if (want_bsum) bsum += (a[u][0] + a[u][1]) + (a[u][2] + a[u][3]);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][0], b[u][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][1], b[u][1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][2], b[u][2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][3], b[u][3], acc1, 0, 0, 0);
        }
    }
};

struct DirectLds {
    float part[4][256];
    float bsum[4][16];
};

template <bool A_KC, bool B_KC, int NBMAX>
__device__ __forceinline__ void gemm_f32_direct_body(const GemmArgs& g, DirectLds& lds) {
    if ((int)blockIdx.y * DB >= g.M || (int)blockIdx.x * DB >= g.N) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.y * DB, n0 = blockIdx.x * DB;
    const int i16 = lane & 15, kq = lane >> 4;
    const bool a_vec = (g.lda & 3) == 0 && (reinterpret_cast<uintptr_t>(g.A) & 15) == 0;
    const bool b_vec = (g.ldb & 3) == 0 && (reinterpret_cast<uintptr_t>(g.B) & 15) == 0;
    GateFn gf;
    gf.g = g.gate; gf.mode = g.gate_mode; gf.p = g.gate_p; gf.seed = g.gate_seed;
    gf.off = epoch_offset(g.gate_off, g.rng_epoch);
    gf.inv_keep = g.gate_p > 0.f ? 1.0f / (1.0f - g.gate_p) : 1.0f;
    const bool gated = g.gate_mode != MPO_GATE_NONE;
    const bool want_bsum = g.bias_grad != nullptr && blockIdx.x == 0;

    // epilogue operands of the output element this thread will own (thread t <-> element t of the tile): requested
    // FIRST, so that their latency hides behind the fragment loads instead of adding a dependent round trip (bias, then
    // mask, residual, old C) after the exchange -- these launches are latency chains, not throughput problems
    const int erow = tid >> 4, ecol = tid & 15;
    const int em = m0 + erow, en = n0 + ecol;
    const bool e_in = em < g.M && en < g.N;
    const size_t eo = (size_t)em * g.ldc + en;
    const float e_bias = (e_in && g.bias) ? g.bias[en] : 0.f;
    const float e_mask = (e_in && g.mask) ? g.mask[eo] : 1.0f;
    const float e_res = (e_in && g.residual) ? g.residual[eo] : 0.f;
    const float e_old = (e_in && g.accumulate) ? g.C[eo] : 0.f;
    // ... and the element's dropout draw (the counter hash) while those loads are in flight
    float e_keep = 1.0f;                                        // plain dropout: keep / (1 - p) or 0;  alpha dropout: 1 or 0
    if (g.drop_p > 0.f) {
        const unsigned long long doff = epoch_offset(g.drop_off, g.rng_epoch);
        e_keep = dropout_keep(g.drop_seed, doff, eo, g.drop_p, g.alpha_dropout ? 1.0f : 1.0f / (1.0f - g.drop_p));
    }

    const int kw = ((g.K + 63) >> 6) << 4;                      // k per wave, a multiple of 16
    const int kbeg = wave * kw, kend = min(g.K, kbeg + kw);
    const int nkb = kend > kbeg ? (kend - kbeg + 15) >> 4 : 0;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    float bsum = 0.f;
    const int nkb_all = kw >> 4;                                // wave-uniform AND workgroup-uniform block count
    DirectCtx<A_KC, B_KC> cx{g, gf, m0 + i16, n0 + i16, kq, kend, lane, a_vec, b_vec, gated, want_bsum};
    // NBMAX is picked by the host from the largest K of the launch (4: K <= 256, 8: K <= 512, else 12): the register
    // footprint -- hence how many workgroups a CU holds -- follows the blocks kept in flight
    for (int kb0 = 0; kb0 < nkb_all; kb0 += NBMAX) cx.template chunk<NBMAX>(kbeg + 16 * kb0, acc0, acc1, bsum);
    (void)nkb;
    // ---- exchange: lane holds D[row = 4*kq + r][col = i16]
#pragma unroll
    for (int r = 0; r < 4; ++r) lds.part[wave][(4 * kq + r) * 16 + i16] = acc0[r] + acc1[r];
    if (want_bsum) {
        bsum += __shfl_xor(bsum, 16);
        bsum += __shfl_xor(bsum, 32);
        if (lane < 16) lds.bsum[wave][lane] = bsum;
    }
    __syncthreads();
    if (e_in) {
        float v = (lds.part[0][tid] + lds.part[1][tid]) + (lds.part[2][tid] + lds.part[3][tid]);
        v = (v + e_bias) * g.alpha;
        v = apply_act(v, g.act);
        if (g.drop_p > 0.f) {
            if (g.alpha_dropout) v = alpha_drop_a(g.drop_p) * (e_keep != 0.f ? v : kAlphaPrime) + alpha_drop_b(g.drop_p);
            else v *= e_keep;
        }
        v = v * e_mask + e_res + e_old;
        g.C[eo] = v;
    }
    if (want_bsum && tid < DB && m0 + tid < g.M)
        g.bias_grad[m0 + tid] = (lds.bsum[0][tid] + lds.bsum[1][tid]) + (lds.bsum[2][tid] + lds.bsum[3][tid]);
}

template <int NBMAX>
__global__ __launch_bounds__(256)
void gemm_f32_direct_kernel(GemmGroup grp) {
    __shared__ DirectLds lds;
    const GemmArgs& g = grp.g[blockIdx.z];
    switch (g.layout) {
        case 3: gemm_f32_direct_body<true, true, NBMAX>(g, lds); break;
        case 2: gemm_f32_direct_body<true, false, NBMAX>(g, lds); break;
        case 1: gemm_f32_direct_body<false, true, NBMAX>(g, lds); break;
        default: gemm_f32_direct_body<false, false, NBMAX>(g, lds); break;
    }
}
template <bool A_KC, bool B_KC, int NBMAX>
__global__ __launch_bounds__(256)
void gemm_f32_direct_single_kernel(GemmArgs g) {
    __shared__ DirectLds lds;
    gemm_f32_direct_body<A_KC, B_KC, NBMAX>(g, lds);
}

template <int NBMAX>
inline void direct_launch_single(const GemmArgs& g, int layout, dim3 grid, hipStream_t stream) {
    switch (layout) {
        case 3: gemm_f32_direct_single_kernel<true, true, NBMAX><<<grid, 256, 0, stream>>>(g); break;
        case 2: gemm_f32_direct_single_kernel<true, false, NBMAX><<<grid, 256, 0, stream>>>(g); break;
        case 1: gemm_f32_direct_single_kernel<false, true, NBMAX><<<grid, 256, 0, stream>>>(g); break;
        default: gemm_f32_direct_single_kernel<false, false, NBMAX><<<grid, 256, 0, stream>>>(g); break;
    }
}

}  // namespace
