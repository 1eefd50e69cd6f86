// Branch-free fast path of the direct small-row fp32 GEMM (gemm_f32_direct.h): templates only.
//
// The general body guards every fragment element (row / column / k limits, alignment, gate kind): ~7 500 instructions
// and ~600 branches per layout, and a grouped launch carries four layouts -- the token tail's 35 GEMM launches per
// window step spent most of their 8-19 us walking that code (instruction fetch, not loads or MFMAs: weights hot in L2
// made them 10 % faster, no more).  Nearly all of those products are regular:
//     M % 16 == 0, N % 16 == 0, K % 16 == 0, leading dimensions % 4 == 0, 16-byte aligned operands,
//     gate one of {none, a value gate (ReLU / ELU / tanh / sigmoid derivative, multiply), regenerated dropout,
//     the AlphaDropout + ELU derivative of the omic SNNs (value and random word)}.
// For those this body has no per-element predicate at all: whole-fragment float4 loads (or four strided scalars when k is
// not the contiguous index), the gate class fixed at compile time, a k loop of full chunks plus one remainder chunk.
// Arithmetic, summation order and random streams are those of the general body (results are bit-identical); launches
// with an irregular member (the 256 -> 1 scorer, the 4-class classifier) keep using it.
#pragma once
#include "gemm_f32_gate.h"

namespace {

constexpr int FB = 16;                 // tile edge

struct FastLds {
    float part[4][256];
    float bsum[4][16];
};

// gate class of a launch member: 0 none, 1 value gate (no random numbers), 2 MPO_GATE_RNG, 3 MPO_GATE_ELU_ADROP

template <bool KC>
__device__ __forceinline__ f32x4 fast_frag(const float* __restrict__ p, int ld, int mn, int k0) {
    if (KC) return *reinterpret_cast<const f32x4*>(p + (size_t)mn * ld + k0);
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = p[(size_t)(k0 + j) * ld + mn];
    return r;
}

template <bool A_KC, bool B_KC, int GC>
struct FastCtx {
    const GemmArgs& g;
    const GateFn& gf;
    int m, n, kq, lane;
    bool want_bsum;
    template <int NB>
    __device__ __forceinline__ void chunk(int kbase, f32x4& acc0, f32x4& acc1, float& bsum) const {
        f32x4 a[NB], b[NB], gv[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int k0 = kbase + 16 * u + 4 * kq;
            a[u] = fast_frag<A_KC>(g.A, g.lda, m, k0);
            b[u] = fast_frag<B_KC>(g.B, g.ldb, n, k0);
            if (GC == 1 || GC == 3) gv[u] = fast_frag<A_KC>(gf.g, g.lda, m, k0);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int k0 = kbase + 16 * u + 4 * kq;
            if (GC == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) a[u][j] *= gf(gv[u][j], 0);
            } else if (GC >= 2) {
                if (A_KC) {
                    // element (m, k0 + j) has index m * lda + k0 + j; lda % 4 == 0 and k0 % 4 == 0: one counter per fragment
                    const uint64_t ctr = gf.off + (((size_t)m * g.lda + k0) >> 2);
                    const uint4 r = draw4x32((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)gf.seed, (uint32_t)(gf.seed >> 32));
                    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[u][j] *= gf.with_word(GC == 3 ? gv[u][j] : 0.f, w[j]);
                } else {
                    // element (k0 + j, m) has index (k0 + j) * lda + m: the lanes of a quad (m = 4q .. 4q+3) share one
                    // counter per j; lane s of the quad draws j = s, four quad exchanges transpose the words
                    const int lq = lane & 3;
                    const size_t idx_own = (size_t)(k0 + lq) * g.lda + (m & ~3);
                    const uint64_t ctr = gf.off + (idx_own >> 2);
                    const uint4 r = draw4x32((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)gf.seed, (uint32_t)(gf.seed >> 32));
                    const uint32_t own[4] = {r.x, r.y, r.z, r.w};
                    uint32_t w[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int pick = lq ^ t;
                        const uint32_t send = pick == 0 ? own[0] : pick == 1 ? own[1] : pick == 2 ? own[2] : own[3];
                        const uint32_t got = (uint32_t)__shfl_xor((int)send, t);
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (j == (lq ^ t)) w[j] = got;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[u][j] *= gf.with_word(GC == 3 ? gv[u][j] : 0.f, w[j]);
                }
            }
            if (want_bsum) bsum += (a[u][0] + a[u][1]) + (a[u][2] + a[u][3]);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][0], b[u][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][1], b[u][1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][2], b[u][2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][3], b[u][3], acc1, 0, 0, 0);
        }
    }
};

template <bool A_KC, bool B_KC, int GC, int NBMAX>
__device__ __forceinline__ void gemm_f32_fast_body(const GemmArgs& g, FastLds& lds) {
    if ((int)blockIdx.y * FB >= g.M || (int)blockIdx.x * FB >= g.N) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.y * FB, n0 = blockIdx.x * FB;
    const int i16 = lane & 15, kq = lane >> 4;
    GateFn gf;
    gf.g = g.gate; gf.mode = g.gate_mode; gf.p = g.gate_p; gf.seed = g.gate_seed;
    gf.off = epoch_offset(g.gate_off, g.rng_epoch);
    gf.inv_keep = g.gate_p > 0.f ? 1.0f / (1.0f - g.gate_p) : 1.0f;
    const bool want_bsum = g.bias_grad != nullptr && blockIdx.x == 0;

    // epilogue operands of the element this thread will own, requested first (their latency hides behind the fragments)
    const int erow = tid >> 4, ecol = tid & 15;
    const size_t eo = (size_t)(m0 + erow) * g.ldc + n0 + ecol;
    const float e_bias = g.bias ? g.bias[n0 + ecol] : 0.f;
    const float e_mask = g.mask ? g.mask[eo] : 1.0f;
    const float e_res = g.residual ? g.residual[eo] : 0.f;
    const float e_old = g.accumulate ? g.C[eo] : 0.f;
    float e_keep = 1.0f;
    if (g.drop_p > 0.f) {
        const unsigned long long doff = epoch_offset(g.drop_off, g.rng_epoch);
        e_keep = dropout_keep(g.drop_seed, doff, eo, g.drop_p, g.alpha_dropout ? 1.0f : 1.0f / (1.0f - g.drop_p));
    }

    const int kw = ((g.K + 63) >> 6) << 4;                      // k per wave, a multiple of 16 (as in the general body)
    const int kbeg = wave * kw;
    const int nkb = max(0, min(kw, g.K - kbeg)) >> 4;           // K % 16 == 0: whole blocks; wave-uniform
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    float bsum = 0.f;
    FastCtx<A_KC, B_KC, GC> cx{g, gf, m0 + i16, n0 + i16, kq, lane, want_bsum};
    int kb = 0;
    for (; kb + NBMAX <= nkb; kb += NBMAX) cx.template chunk<NBMAX>(kbeg + 16 * kb, acc0, acc1, bsum);
    // remainder of the wave's k range: one chunk of 1 .. NBMAX-1 blocks (wave-uniform)
    const int rem = nkb - kb;
    if (rem >= 4) { cx.template chunk<4>(kbeg + 16 * kb, acc0, acc1, bsum); kb += 4; }
    switch (nkb - kb) {
        case 3: cx.template chunk<3>(kbeg + 16 * kb, acc0, acc1, bsum); break;
        case 2: cx.template chunk<2>(kbeg + 16 * kb, acc0, acc1, bsum); break;
        case 1: cx.template chunk<1>(kbeg + 16 * kb, acc0, acc1, bsum); break;
        default: break;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) lds.part[wave][(4 * kq + r) * 16 + i16] = acc0[r] + acc1[r];
    if (want_bsum) {
        bsum += __shfl_xor(bsum, 16);
        bsum += __shfl_xor(bsum, 32);
        if (lane < 16) lds.bsum[wave][lane] = bsum;
    }
    __syncthreads();
    {
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
    if (want_bsum && tid < FB)
        g.bias_grad[m0 + tid] = (lds.bsum[0][tid] + lds.bsum[1][tid]) + (lds.bsum[2][tid] + lds.bsum[3][tid]);
}

// GCL: the gate class a launch may contain besides "none" -- 1: value gates, 2: regenerated dropout, 3: AlphaDropout + ELU
// derivative.  (One kernel with several classes crosses a size at which the compiler copies the by-value GemmGroup
// into scratch.)
template <bool A_KC, bool B_KC, int GCL, int NBMAX>
__device__ __forceinline__ void gemm_f32_fast_member(const GemmArgs& g, FastLds& lds) {
    if (g.gate_mode == MPO_GATE_NONE) gemm_f32_fast_body<A_KC, B_KC, 0, NBMAX>(g, lds);     // uniform over the workgroup
    else gemm_f32_fast_body<A_KC, B_KC, GCL, NBMAX>(g, lds);
}

// (the regenerated-dropout kernel at four blocks in flight fits 80 registers without spilling when asked to: six
//  workgroups per CU instead of four, and its launches are the ones with 1 400 - 2 000 workgroups)
template <int GCL, int NBMAX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((GCL == 2 && NBMAX == 4) ? 6 : 1, 8)))
void gemm_f32_fast_group_kernel(GemmGroup grp) {
    __shared__ FastLds lds;
    const GemmArgs& g = grp.g[blockIdx.z];
    switch (g.layout) {
        case 3: gemm_f32_fast_member<true, true, GCL, NBMAX>(g, lds); break;
        case 2: gemm_f32_fast_member<true, false, GCL, NBMAX>(g, lds); break;
        case 1: gemm_f32_fast_member<false, true, GCL, NBMAX>(g, lds); break;
        default: gemm_f32_fast_member<false, false, GCL, NBMAX>(g, lds); break;
    }
}
template <bool A_KC, bool B_KC, int GCL, int NBMAX>
__global__ __launch_bounds__(256)
void gemm_f32_fast_single_kernel(GemmArgs g) {
    __shared__ FastLds lds;
    gemm_f32_fast_member<A_KC, B_KC, GCL, NBMAX>(g, lds);
}

}  // namespace
