// Small-row fp32 GEMM on the fp32-input MFMA (v_mfma_f32_16x16x4_f32: exact fp32, k-ordered fma chain).
//
// Everything after the co-attention runs on N x d = 6 x 256 tokens per slide:
// the q/k-fold/v-unfold/out projections of K1/K2, the four linears of the Contextual Attention Gate,
// the set-Transformer projections and FFN, the gated-MIL branches, rho, fusion and classifier, and all
// of their backward products.  With a window of slides batched the row count is 6 x n_slides.
//
//   C[m][n] = epilogue( alpha * ( sum_k A(m,k) * B(n,k) + bias[n] ) )
//   A(m,k) = A_KC ? A[m*lda + k] : A[k*lda + m]       B(n,k) = B_KC ? B[n*ldb + k] : B[k*ldb + n]
// which covers  y = x W^T + b   (A_KC, B_KC),  dx = dy W  (A_KC, !B_KC)  and  dW = dy^T x  (!A_KC, !B_KC).
// Epilogue: activation, optional keep-mask multiply (dropout), optional residual add, optional
// accumulate into C (beta = 1).
//
// Tile: 16 rows x 128 columns per 256-thread workgroup (4 waves x 16x32), K step 32, operands staged in
// LDS as [row][k] with a 34-float row stride (conflict-free ds_read_b32 for the 16x16x4 operand map:
// lane l reads row l&15, k = 4*kk + (l>>4)).
#include "mpo_common.h"
#include "mpo_kernels.h"

namespace {

constexpr int BM = 16, BN = 128, BK = 32, LDT = BK + 2;

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case MPO_ACT_RELU: return fmaxf(v, 0.f);
        case MPO_ACT_ELU: return v > 0.f ? v : expm1f(v);
        case MPO_ACT_TANH: return tanhf(v);
        case MPO_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
        default: return v;
    }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256)
void gemm_f32_kernel(GemmArgs g) {
    __shared__ float As[BM * LDT];
    __shared__ float Bs[BN * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const float* __restrict__ A = g.A;
    const float* __restrict__ B = g.B;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < g.K; k0 += BK) {
        // ---- stage A tile (16 x 32)
        for (int e = tid; e < BM * BK; e += 256) {
            int m, k;
            if (A_KC) { m = e / BK; k = e % BK; } else { k = e / BM; m = e % BM; }
            const int gm = m0 + m, gk = k0 + k;
            float v = 0.f;
            if (gm < g.M && gk < g.K) v = A_KC ? A[(size_t)gm * g.lda + gk] : A[(size_t)gk * g.lda + gm];
            As[m * LDT + k] = v;
        }
        // ---- stage B tile (128 x 32)
        for (int e = tid; e < BN * BK; e += 256) {
            int n, k;
            if (B_KC) { n = e / BK; k = e % BK; } else { k = e / BN; n = e % BN; }
            const int gn = n0 + n, gk = k0 + k;
            float v = 0.f;
            if (gn < g.N && gk < g.K) v = B_KC ? B[(size_t)gn * g.ldb + gk] : B[(size_t)gk * g.ldb + gn];
            Bs[n * LDT + k] = v;
        }
        __syncthreads();
        const float* ap = As + (lane & 15) * LDT + (lane >> 4);
        const float* bp = Bs + (wave * 32 + (lane & 15)) * LDT + (lane >> 4);
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            const float a = ap[4 * kk];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bp[4 * kk], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bp[16 * LDT + 4 * kk], acc1, 0, 0, 0);
        }
        __syncthreads();
    }
    // ---- epilogue: D col = lane&15, row = 4*(lane>>4) + r
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int n = n0 + wave * 32 + 16 * c + (lane & 15);
        if (n >= g.N) continue;
        const float bias = g.bias ? g.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 4 * (lane >> 4) + r;
            if (m >= g.M) continue;
            float v = ((c == 0 ? acc0[r] : acc1[r]) + bias) * g.alpha;
            v = apply_act(v, g.act);
            const size_t o = (size_t)m * g.ldc + n;
            if (g.mask) v *= g.mask[o];
            if (g.residual) v += g.residual[o];
            if (g.accumulate) v += g.C[o];
            g.C[o] = v;
        }
    }
}

// colsum[n] (+)= sum_m X[m][n]   (bias gradients)
__global__ void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, int M, int N, int ld, int accumulate) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += x[(size_t)m * ld + n];
    out[n] = accumulate ? out[n] + s : s;
}

}  // namespace

int mpo_launch_gemm(const GemmArgs& g, int a_kc, int b_kc, hipStream_t stream) {
    if (g.M <= 0 || g.N <= 0) return 0;
    MPO_CHECK(g.K > 0, "gemm: K must be positive (got %d)", g.K);
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM);
    if (a_kc && b_kc) gemm_f32_kernel<true, true><<<grid, 256, 0, stream>>>(g);
    else if (a_kc && !b_kc) gemm_f32_kernel<true, false><<<grid, 256, 0, stream>>>(g);
    else if (!a_kc && b_kc) gemm_f32_kernel<false, true><<<grid, 256, 0, stream>>>(g);
    else gemm_f32_kernel<false, false><<<grid, 256, 0, stream>>>(g);
    MPO_LAUNCH_CHECK();
    return 0;
}

int mpo_launch_colsum(const float* x, float* out, int M, int N, int ld, int accumulate, hipStream_t stream) {
    if (N <= 0) return 0;
    colsum_kernel<<<(N + 255) / 256, 256, 0, stream>>>(x, out, M, N, ld, accumulate);
    MPO_LAUNCH_CHECK();
    return 0;
}
