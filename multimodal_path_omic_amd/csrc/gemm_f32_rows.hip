// fp32 GEMM for products with MANY rows (row f3: the set-Transformer / pooling products over the 15 000 rows of a bag).
// The token-tail kernels (gemm_f32_fast.h / gemm_f32_direct.h) give a workgroup ONE 16 x 16 tile: right for 192 rows, but at
// 15 000 rows a [M x 768] product is 45 000 workgroups that each wait one load latency for 16 MFMAs per wave -- 44 rounds of
// workgroups, ~0.2 ms per launch, 7.4 ms of the 25 ms step.  Here a workgroup owns 32 rows x 64 columns: every A fragment
// feeds four column tiles and every B fragment two row tiles (6 fragment loads for 8 tiles' MFMAs instead of 16), K is still
// split over the four waves, the partial tiles meet in LDS and the same epilogue (bias, activation, dropout, mask, residual,
// accumulate) and A-operand gates (activation derivatives, regenerated dropout, same random streams) apply.  K is cut over the
// waves as in the small kernels; each tile has one accumulator per wave here (two there), so results agree to fp32 rounding,
// not bitwise.  Rows past M are clamped on load and not stored.
// Taken for: A k-contiguous (forward and input-gradient layouts), M >= 512, N % 64 == 0, K % 64 == 0, no bias-gradient output.
#include "gemm_f32_gate.h"

namespace {

constexpr int RB = 32, CB = 64;                  // rows / columns of C per workgroup
constexpr int RT = RB / 16, CT = CB / 16;

struct RowsLds { float part[4][RT * CT][256]; };             // 32 KiB

template <bool B_KC, int GC>
__global__ __launch_bounds__(256)
void gemm_f32_rows_kernel(GemmArgs g) {
    __shared__ RowsLds lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.y * RB, n0 = blockIdx.x * CB;
    GateFn gf;
    gf.g = g.gate; gf.mode = g.gate_mode; gf.p = g.gate_p; gf.seed = g.gate_seed;
    gf.off = epoch_offset(g.gate_off, g.rng_epoch);
    gf.inv_keep = g.gate_p > 0.f ? 1.0f / (1.0f - g.gate_p) : 1.0f;

    const int kw = g.K >> 2;                                  // k per wave (K % 64 == 0: whole 16-blocks)
    const int kbeg = wave * kw, nkb = kw >> 4;
    int arow[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) arow[rt] = min(m0 + 16 * rt + i16, g.M - 1);
    f32x4 acc[RT][CT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kb = 0; kb < nkb; kb += 2) {                     // two 16-blocks of k per batch (nkb is even: K % 128 == 0, or one block left)
        f32x4 a[2][RT], gv[2][RT], b[2][CT];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k0 = kbeg + 16 * min(kb + u, nkb - 1) + 4 * kq;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                a[u][rt] = *reinterpret_cast<const f32x4*>(g.A + (size_t)arow[rt] * g.lda + k0);
                if (GC == 1 || GC == 3) gv[u][rt] = *reinterpret_cast<const f32x4*>(gf.g + (size_t)arow[rt] * g.lda + k0);
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int n = n0 + 16 * ct + i16;
                if (B_KC) {
                    b[u][ct] = *reinterpret_cast<const f32x4*>(g.B + (size_t)n * g.ldb + k0);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) b[u][ct][j] = g.B[(size_t)(k0 + j) * g.ldb + n];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (kb + u >= nkb) break;                         // (odd block count: the second half of the last batch is a repeat)
            const int k0 = kbeg + 16 * (kb + u) + 4 * kq;
            if (GC == 1) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[u][rt][j] *= gf(gv[u][rt][j], 0);
            } else if (GC >= 2) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    // element (m, k0 + j) has index m * lda + k0 + j; lda % 4 == 0 and k0 % 4 == 0: one counter per fragment
                    const uint64_t ctr = gf.off + (((size_t)arow[rt] * g.lda + k0) >> 2);
                    const uint4 r = draw4x32((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)gf.seed, (uint32_t)(gf.seed >> 32));
                    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[u][rt][j] *= gf.with_word(GC == 3 ? gv[u][rt][j] : 0.f, w[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][rt][j], b[u][ct][j], acc[rt][ct], 0, 0, 0);
        }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) lds.part[wave][rt * CT + ct][(4 * kq + r) * 16 + i16] = acc[rt][ct][r];
    __syncthreads();
    const int erow = tid >> 4, ecol = tid & 15;
    const unsigned long long doff = epoch_offset(g.drop_off, g.rng_epoch);
#pragma unroll
    for (int t = 0; t < RT * CT; ++t) {
        const int row = m0 + 16 * (t / CT) + erow, col = n0 + 16 * (t % CT) + ecol;
        if (row >= g.M) continue;
        const size_t at = (size_t)row * g.ldc + col;
        float v = (lds.part[0][t][tid] + lds.part[1][t][tid]) + (lds.part[2][t][tid] + lds.part[3][t][tid]);
        v = (v + (g.bias ? g.bias[col] : 0.f)) * g.alpha;
        v = apply_act(v, g.act);
        if (g.drop_p > 0.f) {
            const float keep = dropout_keep(g.drop_seed, doff, at, g.drop_p, g.alpha_dropout ? 1.0f : 1.0f / (1.0f - g.drop_p));
            if (g.alpha_dropout) v = alpha_drop_a(g.drop_p) * (keep != 0.f ? v : kAlphaPrime) + alpha_drop_b(g.drop_p);
            else v *= keep;
        }
        v = v * (g.mask ? g.mask[at] : 1.0f) + (g.residual ? g.residual[at] : 0.f) + (g.accumulate ? g.C[at] : 0.f);
        g.C[at] = v;
    }
}

template <int GC>
void launch_rows(const GemmArgs& g, bool b_kc, hipStream_t stream) {
    const dim3 grid(g.N / CB, (g.M + RB - 1) / RB);
    if (b_kc) gemm_f32_rows_kernel<true, GC><<<grid, 256, 0, stream>>>(g);
    else gemm_f32_rows_kernel<false, GC><<<grid, 256, 0, stream>>>(g);
}

}  // namespace

// gate_class: 0 none, 1 value gate, 2 regenerated dropout, 3 AlphaDropout + ELU derivative (as in gemm_f32_fast.h)
void mpo_rows_single(const GemmArgs& g, int layout, int gate_class, hipStream_t stream) {
    const bool b_kc = (layout & 1) != 0;
    switch (gate_class) {
        case 3: launch_rows<3>(g, b_kc, stream); break;
        case 2: launch_rows<2>(g, b_kc, stream); break;
        case 1: launch_rows<1>(g, b_kc, stream); break;
        default: launch_rows<0>(g, b_kc, stream); break;
    }
}
