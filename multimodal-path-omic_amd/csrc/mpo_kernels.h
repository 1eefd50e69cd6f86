// Internal launcher prototypes shared by the translation units of libmpo_hip.so.
// The public C ABI is include/mpo_hip.h; nothing here is exported.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

enum { MPO_ACT_NONE = 0, MPO_ACT_RELU = 1, MPO_ACT_ELU = 2, MPO_ACT_TANH = 3, MPO_ACT_SIGMOID = 4 };

struct GemmArgs {
    const float* A = nullptr;
    const float* B = nullptr;
    float* C = nullptr;
    const float* bias = nullptr;       // [N] or null
    const float* residual = nullptr;   // [M][ldc] or null, added after the activation
    const float* mask = nullptr;       // [M][ldc] or null, multiplied after the activation (dropout keep-scale)
    int M = 0, N = 0, K = 0;
    int lda = 0, ldb = 0, ldc = 0;
    float alpha = 1.0f;
    int act = MPO_ACT_NONE;
    int accumulate = 0;                // C += result
};

int mpo_launch_gemm(const GemmArgs& g, int a_kc, int b_kc, hipStream_t stream);
int mpo_launch_colsum(const float* x, float* out, int M, int N, int ld, int accumulate, hipStream_t stream);

// y[R][O] = act(alpha * (x[R][I] W[O][I]^T + b))
inline int mpo_linear_fwd(const float* x, const float* w, const float* b, float* y, int R, int I, int O,
                          float alpha, int act, hipStream_t s) {
    GemmArgs g;
    g.A = x; g.B = w; g.C = y; g.bias = b;
    g.M = R; g.N = O; g.K = I; g.lda = I; g.ldb = I; g.ldc = O; g.alpha = alpha; g.act = act;
    return mpo_launch_gemm(g, 1, 1, s);
}
// dx[R][I] (+)= alpha * dy[R][O] W[O][I]
inline int mpo_linear_bwd_input(const float* dy, const float* w, float* dx, int R, int I, int O, float alpha,
                                int accumulate, hipStream_t s) {
    GemmArgs g;
    g.A = dy; g.B = w; g.C = dx;
    g.M = R; g.N = I; g.K = O; g.lda = O; g.ldb = I; g.ldc = I; g.alpha = alpha; g.accumulate = accumulate;
    return mpo_launch_gemm(g, 1, 0, s);
}
// dW[O][I] = alpha * dy[R][O]^T x[R][I]   (ldw = row stride of dW),   db[O] = alpha-less column sums of dy
inline int mpo_linear_bwd_weight(const float* dy, const float* x, float* dw, float* db, int R, int I, int O,
                                 float alpha, hipStream_t s) {
    GemmArgs g;
    g.A = dy; g.B = x; g.C = dw;
    g.M = O; g.N = I; g.K = R; g.lda = O; g.ldb = I; g.ldc = I; g.alpha = alpha;
    int rc = mpo_launch_gemm(g, 0, 0, s);
    if (rc) return rc;
    if (db) return mpo_launch_colsum(dy, db, R, O, O, 0, s);
    return 0;
}

// ---- K1/K2 long-bag cross-attention (coattn_fwd.hip / coattn_bwd.hip)
extern "C" int mpo_coattn_splits(int n_slides, int max_rows);
int mpo_launch_coattn_fwd_partial(const void* bag, int bag_f32, const int* cu, int n_slides, int embed,
                                  const float* qk2, float* part_ml, float* part_ctx, float* s_out,
                                  int n_q, int splits, hipStream_t stream);
int mpo_launch_coattn_combine(const float* part_ml, const float* part_ctx, float* ctx, float* lse2,
                              int n_slides, int n_q, int embed, int splits, hipStream_t stream);
int mpo_launch_coattn_normalize(float* a, const float* lse2, const int* cu, int n_slides, int n_q, int max_rows,
                                float drop_p, unsigned long long seed, unsigned long long offset, hipStream_t stream);
int mpo_launch_coattn_bwd(const void* bag, int bag_f32, const int* cu, int n_slides, int embed,
                          const float* qk2, const float* lse2, const float* dctx, const float* delta,
                          const float* a_map, const float* da_map,
                          void* dbag, float* part_dqk, int n_q, int splits, hipStream_t stream);
int mpo_launch_coattn_bwd_reduce(const float* part_dqk, float* dqk, int n_slides, int n_q, int embed, int splits,
                                 hipStream_t stream);
int mpo_launch_rowdot(const float* a, const float* b, float* out, int rows, int cols, hipStream_t stream);
int mpo_launch_map_rowdot(const float* a_map, const float* da_map, const int* cu, float* delta, int n_slides, int n_q,
                          int accumulate, hipStream_t stream);

// ---- generic bag / map kernels (bagops.hip): the modular form of K2
int mpo_launch_bag_rowdot(const void* bag, int bag_f32, const int* cu, int n_slides, int embed, const float* r,
                          float* map, float alpha, int n_q, int splits, hipStream_t stream);
int mpo_launch_bag_colacc(const void* bag, int bag_f32, const int* cu, int n_slides, int embed, const float* wmap,
                          float* part, int n_q, int splits, hipStream_t stream);
int mpo_launch_bag_outer(const int* cu, int n_slides, int embed, const float* w1, const float* z1, const float* w2,
                         const float* z2, void* dx, int out_f32, int n_q, int splits, hipStream_t stream);
int mpo_launch_gated_softmax_fwd(const float* amap_a, const float* gmap, const int* cu, float* out_map, float* lse2,
                                 float* asum, int n_slides, int n_q, float drop_p, unsigned long long seed,
                                 unsigned long long offset, hipStream_t stream);
int mpo_launch_gated_softmax_bwd(const float* amap_a, const float* gmap, const int* cu, const float* lse2,
                                 const float* dasum, const float* d_ext, float* da_map, float* dg_map, int n_slides,
                                 int n_q, float drop_p, unsigned long long seed, unsigned long long offset,
                                 hipStream_t stream);
int mpo_launch_bag_tanh_fwd(const void* x, void* y, size_t n, int f32, hipStream_t stream);
int mpo_launch_bag_tanh_bwd(const void* y, const void* dy, void* dx, size_t n, int f32, hipStream_t stream);
int mpo_launch_qprep(const float* q, float* qt, float* qs2, float* tq, int n, float c_nat, hipStream_t stream);
int mpo_launch_qprep_bwd(const float* dqt, const float* dtq, const float* tq, const float* d_ext, float* dq, int n,
                         float c_nat, hipStream_t stream);
int mpo_launch_row_scaled_bias(float* y, const float* s, const float* bias, int rows, int cols, hipStream_t stream);
