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
// The kernels live in gemm_f32_direct.h (16 x 16 outputs per workgroup, the four waves split K, fragments
// straight from L2); this file holds the launchers and the bias-gradient column sum.
#include "mpo_common.h"
#include "mpo_kernels.h"
#include "gemm_f32_gate.h"

namespace {

constexpr int DB = 16;
constexpr int DMAXB = 8;
}  // namespace
void mpo_direct_single_nb4(const GemmArgs& g, int layout, dim3 grid, hipStream_t stream);
void mpo_direct_single_nb8(const GemmArgs& g, int layout, dim3 grid, hipStream_t stream);
void mpo_direct_group_nb4(const GemmGroup& grp, dim3 grid, hipStream_t stream);
void mpo_direct_group_nb8(const GemmGroup& grp, dim3 grid, hipStream_t stream);
void mpo_fast_group(const GemmGroup& grp, int gate_classes, int nbmax, dim3 grid, hipStream_t stream);
void mpo_fast_single(const GemmArgs& g, int layout, int gate_classes, int nbmax, dim3 grid, hipStream_t stream);
void mpo_rows_single(const GemmArgs& g, int layout, int gate_class, hipStream_t stream);
int mpo_longk_single(const GemmArgs& g, int gate_class, hipStream_t stream);
namespace {
inline int direct_nbmax(int k) { return k <= 256 ? 4 : DMAXB; }
// A product the branch-free body (gemm_f32_fast.h) can run: whole tiles, whole k-blocks per wave, vector-loadable
// operands, a gate it knows at compile time.  Everything else goes through the general body.
bool g_fast_path = true;                   // mpo_set_gemm_fast_path(): verification hook, on in production
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// -> gate class of the member (0 none, 1 value gate, 2 regenerated dropout, 3 AlphaDropout + ELU derivative), or -1: not
// for the fast body
inline int fast_class(const GemmArgs& g) {
    if (!g_fast_path) return -1;
    if (g.M <= 0 || g.N <= 0 || (g.M & 15) || (g.N & 15) || g.K < 16 || (g.K & 15)) return -1;
    if ((g.lda & 3) || (g.ldb & 3) || !aligned16(g.A) || !aligned16(g.B)) return -1;
    switch (g.gate_mode) {
        case MPO_GATE_NONE: return 0;
        case MPO_GATE_RNG: return 2;
        case MPO_GATE_RELU: case MPO_GATE_ELU: case MPO_GATE_TANH: case MPO_GATE_SIGMOID: case MPO_GATE_MUL:
            return (g.gate != nullptr && aligned16(g.gate)) ? 1 : -1;
        case MPO_GATE_ELU_ADROP: return (g.gate != nullptr && aligned16(g.gate)) ? 3 : -1;
        default: return -1;
    }
}
// -> gate class, or -1: a product with many rows for gemm_f32_rows.hip (32 x 64 tiles): A k-contiguous, whole column blocks
// and k-blocks, vector-loadable operands, no bias-gradient output
inline int rows_class(const GemmArgs& g, int layout) {
    if (!g_fast_path || !(layout & 2) || g.M < 512 || (g.N & 63) || g.K < 64 || (g.K & 63) || g.bias_grad != nullptr) return -1;
    if ((g.lda & 3) || !aligned16(g.A)) return -1;
    if ((layout & 1) && ((g.ldb & 3) || !aligned16(g.B))) return -1;
    switch (g.gate_mode) {
        case MPO_GATE_NONE: return 0;
        case MPO_GATE_RNG: return 2;
        case MPO_GATE_RELU: case MPO_GATE_ELU: case MPO_GATE_TANH: case MPO_GATE_SIGMOID: case MPO_GATE_MUL:
            return (g.gate != nullptr && aligned16(g.gate)) ? 1 : -1;
        case MPO_GATE_ELU_ADROP: return (g.gate != nullptr && aligned16(g.gate)) ? 3 : -1;
        default: return -1;
    }
}
// -> gate class, or -1: a weight-gradient product over a long row axis for gemm_f32_longk.hip (K cut into slices over the grid,
// partial blocks added atomically): both operands k-strided, whole 32 x 64 blocks, nothing but alpha in the epilogue
inline int longk_class(const GemmArgs& g, int layout) {
    if (!g_fast_path || layout != 0 || g.K < 2048 || (g.M & 31) || (g.N & 63) || (g.lda & 3)) return -1;
    if (g.bias || g.mask || g.residual || g.act != MPO_ACT_NONE || g.drop_p > 0.f) return -1;
    switch (g.gate_mode) {
        case MPO_GATE_NONE: return 0;
        case MPO_GATE_RNG: return 2;
        case MPO_GATE_RELU: case MPO_GATE_ELU: case MPO_GATE_TANH: case MPO_GATE_SIGMOID: case MPO_GATE_MUL:
            return g.gate != nullptr ? 1 : -1;
        case MPO_GATE_ELU_ADROP: return g.gate != nullptr ? 3 : -1;
        default: return -1;
    }
}
inline void launch_longk(const GemmArgs& g, int gate_class, hipStream_t stream) {
    if (int e = mpo_longk_single(g, gate_class, stream)) mpo_set_error("long-K weight gradient: zero-fill failed (hip error %d)", e);
}
void launch_direct_single(const GemmArgs& g, int layout, dim3 grid, hipStream_t stream) {
    const int rc = rows_class(g, layout);
    if (rc >= 0) { mpo_rows_single(g, layout, rc, stream); return; }
    const int lc = longk_class(g, layout);
    if (lc >= 0) { launch_longk(g, lc, stream); return; }
    const int fc = fast_class(g);
    if (fc >= 0) { mpo_fast_single(g, layout, fc, direct_nbmax(g.K), grid, stream); return; }
    if (direct_nbmax(g.K) == 4) mpo_direct_single_nb4(g, layout, grid, stream);
    else mpo_direct_single_nb8(g, layout, grid, stream);
}
void launch_direct_group(const GemmGroup& all, dim3 grid, hipStream_t stream) {
    // members with many rows (gemm_f32_rows.hip) or a long inner dimension (gemm_f32_longk.hip) leave the group for their own
    // launches; the members of a group are independent products, so the order of the launches does not matter
    GemmGroup grp;
    grp.n = 0;
    for (int i = 0; i < all.n; ++i) {
        const int rc = rows_class(all.g[i], all.g[i].layout), lc = longk_class(all.g[i], all.g[i].layout);
        if (rc >= 0) mpo_rows_single(all.g[i], all.g[i].layout, rc, stream);
        else if (lc >= 0) launch_longk(all.g[i], lc, stream);
        else grp.g[grp.n++] = all.g[i];
    }
    if (grp.n == 0) return;
    if (grp.n != all.n) {
        int mx = 0, nx = 0;
        for (int i = 0; i < grp.n; ++i) { mx = std::max(mx, grp.g[i].M); nx = std::max(nx, grp.g[i].N); }
        grid = dim3((nx + DB - 1) / DB, (mx + DB - 1) / DB, grp.n);
    }
    int kmax = 0, classes = 0;                                    // classes: bit c set when a member has gate class c
    bool fast = true;
    for (int i = 0; i < grp.n; ++i) {
        kmax = grp.g[i].K > kmax ? grp.g[i].K : kmax;
        const int fc = fast_class(grp.g[i]);
        fast = fast && fc >= 0;
        if (fc > 0) classes |= 1 << fc;
    }
    fast = fast && (classes & (classes - 1)) == 0;                // two gate classes in one launch: general body
    // a launch that fills the chip several times over is throughput-bound: the 4-block variant's smaller register
    // footprint (more workgroups per CU) then beats having all of K in flight at once
    const size_t wgs = (size_t)grid.x * grid.y * grid.z;
    const int nb = (wgs > 1024 || direct_nbmax(kmax) == 4) ? 4 : DMAXB;
    if (fast) { mpo_fast_group(grp, classes >> 3 ? 3 : classes >> 2 ? 2 : 1, nb, grid, stream); return; }
    if (nb == 4) mpo_direct_group_nb4(grp, grid, stream);
    else mpo_direct_group_nb8(grp, grid, stream);
}

// colsum[n] (+)= sum_m X[m][n]   (bias gradients, merging per-workgroup partials): 16 columns per workgroup, 16 row
// groups of 16 lanes (64 columns x 4 row groups left a [1024][256] merge on 4 workgroups: 62 us)
__global__ __launch_bounds__(256)
void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, int M, int N, int ld, int accumulate) {
    __shared__ float red[16][17];
    const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int n = blockIdx.x * 16 + c;
    float s = 0.f;
    if (n < N)
        for (int m = rg; m < M; m += 16) s += x[(size_t)m * ld + n];
    red[rg][c] = s;
    __syncthreads();
    if (rg == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][c];
        out[n] = accumulate ? out[n] + t : t;
    }
}

}  // namespace

int mpo_gemm_fast_path(int enabled) {
    const int was = g_fast_path ? 1 : 0;
    g_fast_path = enabled != 0;
    return was;
}

int mpo_launch_gemm(const GemmArgs& g, int a_kc, int b_kc, hipStream_t stream) {
    if (g.M <= 0 || g.N <= 0) return 0;
    MPO_CHECK(g.K > 0, "gemm: K must be positive (got %d)", g.K);
    dim3 dgrid((g.N + DB - 1) / DB, (g.M + DB - 1) / DB);
    launch_direct_single(g, 2 * (a_kc ? 1 : 0) + (b_kc ? 1 : 0), dgrid, stream);
    MPO_LAUNCH_CHECK();
    return 0;
}

int mpo_launch_gemm_group(const GemmGroup& grp, int a_kc, int b_kc, hipStream_t stream) {
    MPO_CHECK(grp.n >= 1 && grp.n <= 8, "grouped gemm: 1..8 members (got %d)", grp.n);
    int mx = 0, nx = 0;
    for (int i = 0; i < grp.n; ++i) {
        MPO_CHECK(grp.g[i].K > 0, "grouped gemm: member %d has K = %d", i, grp.g[i].K);
        if (grp.g[i].M > mx) mx = grp.g[i].M;
        if (grp.g[i].N > nx) nx = grp.g[i].N;
    }
    if (mx <= 0 || nx <= 0) return 0;
    GemmGroup tagged = grp;
    for (int i = 0; i < tagged.n; ++i) tagged.g[i].layout = 2 * (a_kc ? 1 : 0) + (b_kc ? 1 : 0);
    dim3 dgrid((nx + DB - 1) / DB, (mx + DB - 1) / DB, grp.n);
    launch_direct_group(tagged, dgrid, stream);
    MPO_LAUNCH_CHECK();
    return 0;
}

int mpo_launch_gemm_mixed(const GemmGroup& grp, hipStream_t stream) {
    MPO_CHECK(grp.n >= 1 && grp.n <= 8, "mixed grouped gemm: 1..8 members (got %d)", grp.n);
    int mx = 0, nx = 0;
    for (int i = 0; i < grp.n; ++i) {
        MPO_CHECK(grp.g[i].K > 0, "mixed grouped gemm: member %d has K = %d", i, grp.g[i].K);
        MPO_CHECK(grp.g[i].layout >= 0 && grp.g[i].layout <= 3, "mixed grouped gemm: member %d has layout %d", i, grp.g[i].layout);
        if (grp.g[i].M > mx) mx = grp.g[i].M;
        if (grp.g[i].N > nx) nx = grp.g[i].N;
    }
    if (mx <= 0 || nx <= 0) return 0;
    dim3 dgrid((nx + DB - 1) / DB, (mx + DB - 1) / DB, grp.n);
    launch_direct_group(grp, dgrid, stream);
    MPO_LAUNCH_CHECK();
    return 0;
}

int mpo_launch_colsum(const float* x, float* out, int M, int N, int ld, int accumulate, hipStream_t stream) {
    if (N <= 0) return 0;
    colsum_kernel<<<(N + 15) / 16, 256, 0, stream>>>(x, out, M, N, ld, accumulate);
    MPO_LAUNCH_CHECK();
    return 0;
}
