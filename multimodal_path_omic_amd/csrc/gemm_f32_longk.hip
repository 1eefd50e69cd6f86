// fp32 weight-gradient products with a LONG inner dimension (row f3: dW = dY^T X over the 15 000 rows of a bag).
// The token-tail kernels give one workgroup a 16 x 16 tile over ALL of K: with K = 15 000 that is a chain of ~60 dependent load
// batches per wave on a few hundred workgroups -- 0.23-0.36 ms per launch, eleven launches per step.  Here K is cut into
// slices of 512 rows over grid.z; a workgroup computes a 32 x 64 block of the product for its slice (four waves x 128 rows,
// every fragment reused as in gemm_f32_rows.hip) and ADDS it into C (and its share of the bias gradient) atomically; the
// launcher zeroes C first unless the product accumulates.  The order in which slices arrive is not fixed: sums over the 15 000
// rows agree between runs to fp32 rounding, not bitwise (the small kernels' order is fixed).
// Taken for: both operands k-strided (layout 0), K >= 2048, M % 32 == 0, N % 64 == 0, plain epilogue (alpha only).
#include "gemm_f32_gate.h"

namespace {

constexpr int LO = 32, LI = 64, LKS = 512;       // rows / columns of C per workgroup, k rows per slice
constexpr int LRT = LO / 16, LCT = LI / 16;

struct LongKLds {
    float part[4][LRT * LCT][256];
    float bsum[4][LO];
};

template <int GC>
__global__ __launch_bounds__(256)
void gemm_f32_longk_kernel(GemmArgs g) {
    __shared__ LongKLds lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.y * LO, n0 = blockIdx.x * LI;
    GateFn gf;
    gf.g = g.gate; gf.mode = g.gate_mode; gf.p = g.gate_p; gf.seed = g.gate_seed;
    gf.off = epoch_offset(g.gate_off, g.rng_epoch);
    gf.inv_keep = g.gate_p > 0.f ? 1.0f / (1.0f - g.gate_p) : 1.0f;
    const bool want_bsum = g.bias_grad != nullptr && blockIdx.x == 0;

    f32x4 acc[LRT][LCT];
#pragma unroll
    for (int rt = 0; rt < LRT; ++rt)
#pragma unroll
        for (int ct = 0; ct < LCT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[LRT] = {0.f, 0.f};
    const int kbase = blockIdx.z * LKS + wave * (LKS / 4);
    for (int c = 0; c < LKS / 64; c += 2) {                  // the wave's 128 rows: eight 16-blocks, two per batch
        f32x4 a[2][LRT], gv[2][LRT], b[2][LCT];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k0 = kbase + 16 * (c + u) + 4 * kq;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const size_t kc = (size_t)min(k0 + j, g.K - 1);                    // rows past the end: clamped, zeroed below
#pragma unroll
                for (int rt = 0; rt < LRT; ++rt) {
                    a[u][rt][j] = g.A[kc * g.lda + m0 + 16 * rt + i16];
                    if (GC == 1 || GC == 3) gv[u][rt][j] = gf.g[kc * g.lda + m0 + 16 * rt + i16];
                }
#pragma unroll
                for (int ct = 0; ct < LCT; ++ct) b[u][ct][j] = g.B[kc * g.ldb + n0 + 16 * ct + i16];
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k0 = kbase + 16 * (c + u) + 4 * kq;
#pragma unroll
            for (int rt = 0; rt < LRT; ++rt) {
                const int m = m0 + 16 * rt + i16;
                if (GC == 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[u][rt][j] *= gf(gv[u][rt][j], 0);
                } else if (GC >= 2) {
                    // element (k0 + j, m) has index (k0 + j) * lda + m: the lanes of a quad (m = 4q .. 4q+3) share one counter
                    // per j; lane s of the quad draws j = s, four quad exchanges transpose the words (as in gemm_f32_fast.h)
                    const int lq = lane & 3;
                    const size_t idx_own = (size_t)min(k0 + lq, g.K - 1) * g.lda + (m & ~3);
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
                    for (int j = 0; j < 4; ++j) a[u][rt][j] *= gf.with_word(GC == 3 ? gv[u][rt][j] : 0.f, w[j]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (k0 + j >= g.K) a[u][rt][j] = 0.f;
                    bsum[rt] += a[u][rt][j];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rt = 0; rt < LRT; ++rt)
#pragma unroll
                    for (int ct = 0; ct < LCT; ++ct)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][rt][j], b[u][ct][j], acc[rt][ct], 0, 0, 0);
        }
    }
#pragma unroll
    for (int rt = 0; rt < LRT; ++rt)
#pragma unroll
        for (int ct = 0; ct < LCT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) lds.part[wave][rt * LCT + ct][(4 * kq + r) * 16 + i16] = acc[rt][ct][r];
    if (want_bsum) {
#pragma unroll
        for (int rt = 0; rt < LRT; ++rt) {
            float s = bsum[rt];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (lane < 16) lds.bsum[wave][16 * rt + lane] = s;
        }
    }
    __syncthreads();
    const int erow = tid >> 4, ecol = tid & 15;
#pragma unroll
    for (int t = 0; t < LRT * LCT; ++t) {
        const float v = (lds.part[0][t][tid] + lds.part[1][t][tid]) + (lds.part[2][t][tid] + lds.part[3][t][tid]);
        atomicAdd(g.C + (size_t)(m0 + 16 * (t / LCT) + erow) * g.ldc + n0 + 16 * (t % LCT) + ecol, v * g.alpha);
    }
    if (want_bsum && tid < LO)
        atomicAdd(g.bias_grad + m0 + tid, (lds.bsum[0][tid] + lds.bsum[1][tid]) + (lds.bsum[2][tid] + lds.bsum[3][tid]));
}

}  // namespace

// gate_class as in gemm_f32_fast.h.  Returns a hip error code (the zero-fills are stream operations).
int mpo_longk_single(const GemmArgs& g, int gate_class, hipStream_t stream) {
    if (!g.accumulate) {
        if (g.ldc == g.N) {
            if (hipError_t e = hipMemsetAsync(g.C, 0, (size_t)g.M * g.N * sizeof(float), stream)) return (int)e;
        } else {
            if (hipError_t e = hipMemset2DAsync(g.C, (size_t)g.ldc * sizeof(float), 0, (size_t)g.N * sizeof(float), g.M, stream)) return (int)e;
        }
    }
    if (g.bias_grad)
        if (hipError_t e = hipMemsetAsync(g.bias_grad, 0, (size_t)g.M * sizeof(float), stream)) return (int)e;
    const dim3 grid(g.N / LI, g.M / LO, (g.K + LKS - 1) / LKS);
    switch (gate_class) {
        case 3: gemm_f32_longk_kernel<3><<<grid, 256, 0, stream>>>(g); break;
        case 2: gemm_f32_longk_kernel<2><<<grid, 256, 0, stream>>>(g); break;
        case 1: gemm_f32_longk_kernel<1><<<grid, 256, 0, stream>>>(g); break;
        default: gemm_f32_longk_kernel<0><<<grid, 256, 0, stream>>>(g); break;
    }
    return 0;
}
