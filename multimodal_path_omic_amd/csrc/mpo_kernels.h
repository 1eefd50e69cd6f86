// Internal launcher prototypes shared by the translation units of libmpo_hip.so.
// The public C ABI is include/mpo_hip.h; nothing here is exported.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include <stdint.h>
enum { MPO_ACT_NONE = 0, MPO_ACT_RELU = 1, MPO_ACT_ELU = 2, MPO_ACT_TANH = 3, MPO_ACT_SIGMOID = 4 };
// Gate applied to the A operand while it is staged: A(m,k) *= gate_fn(G(m,k)), G laid out like A.
// It folds the derivative of the activation (and of a following dropout) of the layer that PRODUCED
// the saved tensor G into the backward GEMMs dx = (dy*gate) W and dW = (dy*gate)^T x:
//   RELU    G = drop(relu(pre)):    G > 0 ? 1/(1-p) : 0
//   ELU     G = elu(pre):           G > 0 ? 1 : G + 1
//   TANH    G = drop(tanh(pre)):    t = G(1-p);  G != 0 ? (1 - t^2)/(1-p) : (p > 0 ? 0 : 1)
//   SIGMOID G = drop(sigmoid(pre)): s = G(1-p);  G != 0 ? s(1-s)/(1-p) : 0
//   RNG     no tensor: the keep-scale of the dropout stream (gate_seed, gate_off) at the element's index
//   MUL     plain element-wise factor G
//   ELU_ADROP G = alpha_dropout_p(elu(pre)) (nn.AlphaDropout): keep = stream bit; u = (G - b)/a; keep ? a * elu'(u) : 0
enum { MPO_GATE_NONE = 0, MPO_GATE_RELU = 1, MPO_GATE_ELU = 2, MPO_GATE_TANH = 3, MPO_GATE_SIGMOID = 4,
       MPO_GATE_RNG = 5, MPO_GATE_MUL = 6, MPO_GATE_ELU_ADROP = 7 };

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
    // epilogue dropout (after the activation): C *= keep-scale of stream (drop_seed, drop_off) at index m*ldc+n;
    // alpha_dropout != 0 selects nn.AlphaDropout's form instead: C = a * (keep ? C : alpha') + b
    int alpha_dropout = 0;
    float drop_p = 0.f;
    uint64_t drop_seed = 0, drop_off = 0;
    // A-operand gate (see MPO_GATE_*)
    const float* gate = nullptr;
    int gate_mode = MPO_GATE_NONE;
    float gate_p = 0.f;
    uint64_t gate_seed = 0, gate_off = 0;
    // dW GEMMs: bias_grad[m] = sum_k A(m,k) (after the gate) -- the bias gradient, for free
    float* bias_grad = nullptr;
    // optional device-resident epoch added (x 2^40) to both dropout offsets (graph replay; mpo_common.h)
    const unsigned long long* rng_epoch = nullptr;
    // operand layout for mpo_launch_gemm_mixed: 2 * a_kc + b_kc
    int layout = 3;
};

struct DropSpec {                      // one dropout stream: p = 0 means "no dropout"
    float p = 0.f;
    uint64_t seed = 0, off = 0;
    const unsigned long long* epoch = nullptr;
};
struct GateSpec {
    const float* g = nullptr;
    int mode = MPO_GATE_NONE;
    float p = 0.f;
    uint64_t seed = 0, off = 0;
    const unsigned long long* epoch = nullptr;
};

int mpo_launch_gemm(const GemmArgs& g, int a_kc, int b_kc, hipStream_t stream);
// up to 8 independent GEMMs of the same operand layout in ONE launch (blockIdx.z picks the member)
struct GemmGroup {
    GemmArgs g[8];
    int n = 0;
};
int mpo_launch_gemm_group(const GemmGroup& grp, int a_kc, int b_kc, hipStream_t stream);
// members may differ in operand layout (GemmArgs::layout)
int mpo_launch_gemm_mixed(const GemmGroup& grp, hipStream_t stream);
int mpo_launch_colsum(const float* x, float* out, int M, int N, int ld, int accumulate, hipStream_t stream);

// y[R][O] = drop(act(alpha * (x[R][I] W[O][I]^T + b))) [+ residual]
inline GemmArgs mpo_args_fwd(const float* x, const float* w, const float* b, float* y, int R, int I, int O,
                             float alpha, int act, const float* residual = nullptr, DropSpec drop = DropSpec()) {
    GemmArgs g;
    g.A = x; g.B = w; g.C = y; g.bias = b; g.residual = residual;
    g.M = R; g.N = O; g.K = I; g.lda = I; g.ldb = I; g.ldc = O; g.alpha = alpha; g.act = act;
    g.drop_p = drop.p; g.drop_seed = drop.seed; g.drop_off = drop.off; g.rng_epoch = drop.epoch;
    g.layout = 3;
    return g;
}
inline int mpo_linear_fwd(const float* x, const float* w, const float* b, float* y, int R, int I, int O,
                          float alpha, int act, hipStream_t s, const float* residual = nullptr,
                          DropSpec drop = DropSpec()) {
    return mpo_launch_gemm(mpo_args_fwd(x, w, b, y, R, I, O, alpha, act, residual, drop), 1, 1, s);
}
// dx[R][I] (+)= alpha * (dy*gate)[R][O] W[O][I]
inline GemmArgs mpo_args_bwd_input(const float* dy, const float* w, float* dx, int R, int I, int O, float alpha,
                                   int accumulate, GateSpec gate = GateSpec()) {
    GemmArgs g;
    g.A = dy; g.B = w; g.C = dx;
    g.M = R; g.N = I; g.K = O; g.lda = O; g.ldb = I; g.ldc = I; g.alpha = alpha; g.accumulate = accumulate;
    g.gate = gate.g; g.gate_mode = gate.mode; g.gate_p = gate.p; g.gate_seed = gate.seed; g.gate_off = gate.off;
    g.rng_epoch = gate.epoch;
    g.layout = 2;
    return g;
}
inline int mpo_linear_bwd_input(const float* dy, const float* w, float* dx, int R, int I, int O, float alpha,
                                int accumulate, hipStream_t s, GateSpec gate = GateSpec()) {
    return mpo_launch_gemm(mpo_args_bwd_input(dy, w, dx, R, I, O, alpha, accumulate, gate), 1, 0, s);
}
// dW[O][I] = alpha * (dy*gate)[R][O]^T x[R][I],   db[O] = column sums of dy*gate (nullable)
inline GemmArgs mpo_args_bwd_weight(const float* dy, const float* x, float* dw, float* db, int R, int I, int O,
                                    float alpha, GateSpec gate = GateSpec()) {
    GemmArgs g;
    g.A = dy; g.B = x; g.C = dw;
    g.M = O; g.N = I; g.K = R; g.lda = O; g.ldb = I; g.ldc = I; g.alpha = alpha;
    g.gate = gate.g; g.gate_mode = gate.mode; g.gate_p = gate.p; g.gate_seed = gate.seed; g.gate_off = gate.off;
    g.rng_epoch = gate.epoch;
    g.bias_grad = db;
    g.layout = 0;
    return g;
}
inline int mpo_linear_bwd_weight(const float* dy, const float* x, float* dw, float* db, int R, int I, int O,
                                 float alpha, hipStream_t s, GateSpec gate = GateSpec()) {
    return mpo_launch_gemm(mpo_args_bwd_weight(dy, x, dw, db, R, I, O, alpha, gate), 0, 0, s);
}
// a layer's input- and weight-gradient products in ONE launch (both read the same dy)
inline int mpo_linear_bwd_pair(const GemmArgs& dx, const GemmArgs& dw, hipStream_t s) {
    GemmGroup grp;
    grp.g[0] = dx; grp.g[1] = dw; grp.n = 2;
    return mpo_launch_gemm_mixed(grp, s);
}
// up to four independent products of any layout in ONE launch
inline int mpo_gemm_together(hipStream_t s, const GemmArgs& a, const GemmArgs& b, const GemmArgs* c = nullptr,
                             const GemmArgs* d = nullptr) {
    GemmGroup grp;
    grp.g[0] = a; grp.g[1] = b; grp.n = 2;
    if (c) grp.g[grp.n++] = *c;
    if (d) grp.g[grp.n++] = *d;
    return mpo_launch_gemm_mixed(grp, s);
}

// ---- K1/K2 long-bag cross-attention (coattn_fwd.hip / coattn_bwd.hip)
struct BagPlan;
extern "C" int mpo_coattn_splits(int n_slides, int max_rows);
// rows H2 / f1: the patch layer of a bf16 window, 1024 -> 256, one pass over the raw patch matrix (patch_fc_fwd.hip)
int mpo_launch_pack_patch_weight(const float* w, void* out, int embed, int patch_dim, hipStream_t stream);
int mpo_launch_patch_fc_fwd(const void* x, const void* w_packed, const float* bias, const int* cu, void* h_out, int embed, float drop_p,
                            unsigned long long seed, unsigned long long offset, const unsigned long long* epoch,
                            const BagPlan& plan, hipStream_t stream);
int mpo_launch_coattn_fwd_partial(const void* bag, int bag_f32, const int* cu, int n_slides, int embed,
                                  const float* qk2, float* part_ml, float* part_ctx, float* s_out,
                                  int n_q, const BagPlan& plan, hipStream_t stream);
int mpo_launch_coattn_combine(const float* part_ml, const float* part_ctx, float* ctx, float* lse2,
                              int n_slides, int n_q, int embed, const BagPlan& plan, hipStream_t stream);
int mpo_launch_coattn_normalize(float* a, const float* lse2, const int* cu, int n_slides, int n_q, int max_rows,
                                float drop_p, unsigned long long seed, unsigned long long offset, hipStream_t stream);
int mpo_launch_coattn_bwd(const void* bag, int bag_f32, const int* cu, int n_slides, int embed,
                          const float* qk2, const float* lse2, const float* dctx, const float* delta /* NULL: from ctx */,
                          const float* ctx, const float* a_map, const float* da_map,
                          void* dbag, float* part_dqk, float* part_colsum /* nullable [parts][E] */, int n_q, const BagPlan& plan,
                          float relu_gate, hipStream_t stream);
// the same pass for a bf16 bag at embed 256, <= 8 queries, no map gradient, two waves per SIMD (coattn_bwd8.hip);
// mpo_launch_coattn_bwd routes to it when mpo_coattn_bwd8_covers()
bool mpo_coattn_bwd8_covers(int bag_f32, int embed, int n_q, const float* da_map);
// K1 backward of an fp32 bag on the vector ALUs (coattn_bwd_f32.hip): embed 256, n_q <= 8, with or without a map gradient
int mpo_coattn_bwd_f32_enable(int enabled);
bool mpo_coattn_bwd_f32_covers(int bag_f32, int embed, int n_q, const float* da_map);
int mpo_launch_coattn_bwd_f32(const void* bag, const int* cu, const float* qk2, const float* lse2, const float* dctx,
                              const float* delta, const float* ctx, const float* da_map, void* dbag, float* part_dqk,
                              float* part_colsum, int n_q, const BagPlan& plan, hipStream_t stream);
int mpo_coattn_bwd8_enable(int enabled);   // returns the previous setting
int mpo_launch_coattn_bwd8(const void* bag, const int* cu, const float* qk2, const float* lse2, const float* dctx,
                           const float* delta, const float* ctx, void* dbag, float* part_dqk, float* part_colsum, int n_q,
                           const BagPlan& plan, float relu_gate, hipStream_t stream);
// fp32-stored window: the patch layer as three-term bf16 products (patch_fc_f32.hip); ws = the *_workspace_floats() below
size_t mpo_patch_fc_f32_workspace_floats();
size_t mpo_patch_wgrad_f32_workspace_floats();
int mpo_launch_patch_fc_f32(const float* x, const float* w, const float* bias, float* h, long long total_rows, int embed,
                            int patch_dim, float drop_p, unsigned long long seed, unsigned long long offset,
                            const unsigned long long* epoch, float x_scale, float* ws, hipStream_t stream);
int mpo_launch_patch_wgrad_f32(const float* dh, const float* hbag, const float* x, long long total_rows, int embed, int patch_dim,
                               float gate, float* d_weight, float* d_bias, float* ws, hipStream_t stream);
// K2's patch-side gradient with the product back through the key projection inside the pass (k2_patchgrad.hip)
int mpo_launch_k2_patch_grad(const int* cu, const void* dk_bf16, const float* w_k, const float* amap, const float* dctx,
                             const void* hbag_bf16, void* out_bf16, float gate, float* part_colsum, int n_q, int embed,
                             const BagPlan& plan, hipStream_t stream);
int mpo_gemm_fast_path(int enabled);   // gemm_f32.hip: returns the previous setting
// dW_H = g^T X of the patch layer, hand-written (patch_wgrad.hip): part = mpo_patch_wgrad_partial_floats() floats
size_t mpo_patch_wgrad_partial_floats(int embed, int patch_dim);
int mpo_launch_patch_wgrad(const void* g_bf16, const void* x_bf16, int64_t total_rows, int embed, int patch_dim, float* part,
                           float* d_weight, int workgroups, hipStream_t stream);
// what follows a split-M bag pass, one launch: up to two per-slide reductions of [parts][n_q*E] partials, the column
// sums over all partials of a [parts][cs_cols] array, zero-fills of up to two regions (coattn_bwd.hip: bag_finish_kernel)
struct BagFinish {
    const float* part[2];
    float* out[2];
    int n_red;
    const float* part_cs;
    float* colsum;
    int cs_cols;
    float* zero[2];
    int n_zero[2];
};
int mpo_launch_bag_finish(const BagFinish& f, int n_slides, int n_q, int embed, const BagPlan& plan, hipStream_t stream);
int mpo_launch_coattn_bwd_reduce(const float* part_dqk, float* dqk, int n_slides, int n_q, int embed, const BagPlan& plan,
                                 hipStream_t stream);
int mpo_launch_rowdot(const float* a, const float* b, float* out, int rows, int cols, hipStream_t stream);
int mpo_launch_map_block_scale(const float* a_map, const float* scale, const int* cu, float* out, int n_slides, int n_q,
                               hipStream_t stream);
int mpo_launch_map_rowdot(const float* a_map, const float* da_map, const int* cu, float* delta, int n_slides, int n_q,
                          int accumulate, hipStream_t stream);

// ---- generic bag / map kernels (bagops.hip): the modular form of K2
int mpo_launch_bag_rowdot(const void* bag, int bag_f32, const int* cu, int n_slides, int embed, const float* r,
                          float* map, float alpha, int n_q, const BagPlan& plan, hipStream_t stream);
int mpo_launch_bag_colacc(const void* bag, int bag_f32, const int* cu, int n_slides, int embed, const float* wmap,
                          float* part, int n_q, const BagPlan& plan, hipStream_t stream);
int mpo_launch_bag_outer(const int* cu, int n_slides, int embed, const float* w1, const float* z1, const float* w2,
                         const float* z2, void* dx, int out_f32, int n_q, const BagPlan& plan, hipStream_t stream);
int mpo_launch_gated_softmax_fwd(const float* amap_a, const float* gmap, const int* cu, float* out_map, float* lse2,
                                 float* asum, int n_slides, int n_q, float drop_p, unsigned long long seed,
                                 unsigned long long offset, const unsigned long long* epoch, hipStream_t stream);
int mpo_launch_gated_softmax_bwd(const float* amap_a, const float* gmap, const int* cu, const float* lse2,
                                 const float* dasum, const float* d_ext, float* da_map, float* dg_map, int n_slides,
                                 int n_q, float drop_p, unsigned long long seed, unsigned long long offset,
                                 const unsigned long long* epoch, hipStream_t stream);
int mpo_launch_bag_tanh_fwd(const void* x, void* y, size_t n, int f32, hipStream_t stream);
int mpo_launch_bag_tanh_bwd(const void* y, const void* dy, void* dx, size_t n, int f32, hipStream_t stream);
int mpo_launch_qprep(const float* q, float* qt, float* qs2, float* tq, int n, float c_nat, hipStream_t stream);
int mpo_launch_qprep_bwd(const float* dqt, const float* dtq, const float* tq, const float* d_ext, float* dq, int n,
                         float c_nat, hipStream_t stream);
int mpo_launch_row_scaled_bias(float* y, const float* s, const float* bias, int rows, int cols, hipStream_t stream);

// ---- tail kernels (tail.hip)
// LayerNorm over rows that belong to up to kMaxBranches independent modules (branch = row / rows_per_branch)
constexpr int kMaxBranches = 4;
struct LnBranches {
    const float* w[kMaxBranches] = {};
    const float* b[kMaxBranches] = {};
    float* dw[kMaxBranches] = {};
    float* db[kMaxBranches] = {};
    int rows_per_branch = 0, n = 0;
};
int mpo_launch_ln_fwd_br(const float* x, const LnBranches& p, float* y, float* stats, int rows, int d, float eps, hipStream_t s);
// what: 1 = dx, 2 = parameter gradients, 3 = both in one launch
int mpo_launch_ln_bwd_br(const float* dy, const float* x, const float* stats, const LnBranches& p, float* dx, int rows, int d,
                         int accumulate, int what, hipStream_t s);
int mpo_launch_ln_fwd(const float* x, const float* w, const float* b, float* y, float* stats, int rows, int d, float eps,
                      hipStream_t s);
int mpo_launch_ln_bwd(const float* dy, const float* x, const float* stats, const float* w, float* dx, float* dw, float* db,
                      int rows, int d, int accumulate, hipStream_t s);
int mpo_launch_ln_bwd_params_only(const float* dy, const float* x, const float* stats, float* dw, float* db, int rows, int d,
                                  hipStream_t s);
// longest token axis the LDS-resident attention kernels of tail.hip take; longer axes (the rows of a bag) run bag_selfattn.hip
constexpr int kSmallAttnMaxT = 16;
// self-attention over the M rows of a bag (bag_selfattn.hip): qkv [n_seq][M][3 d] -> o [n_seq][M][d]; saved:
// mpo_bag_sa_saved_floats() floats (log-sum-exps + the three-term bf16 path's operand forms); map (optional, H == 1)
// [n_seq][M][M]; backward: dqkv [n_seq][M][3 d], scratch of mpo_bag_sa_bwd_floats() floats
int mpo_bag_sa_supported_head_dim(int hd);
int mpo_bag_sa_set_bf16x3(int enabled);
size_t mpo_bag_sa_saved_floats(int n_seq, int M, int d, int H);
size_t mpo_bag_sa_bwd_floats(int n_seq, int M, int d, int H);
int mpo_launch_bag_sa_fwd(const float* qkv, int n_seq, int M, int d, int H, float drop_p, unsigned long long seed,
                          unsigned long long offset, const unsigned long long* epoch, float* o, float* saved, float* map, hipStream_t s);
int mpo_launch_bag_sa_bwd(const float* qkv, const float* o, const float* saved, const float* d_o, int n_seq, int M, int d, int H,
                          float drop_p, unsigned long long seed, unsigned long long offset, const unsigned long long* epoch,
                          float* dqkv, float* scratch, hipStream_t s);
int mpo_launch_mha_small_fwd(const float* qkv, float* o, float* p_save, int B, int T, int d, int H, float drop_p,
                             unsigned long long seed, unsigned long long offset, const unsigned long long* epoch,
                             hipStream_t s);
int mpo_launch_mha_small_bwd(const float* qkv, const float* p_save, const float* d_o, float* dqkv, int B, int T, int d, int H,
                             hipStream_t s);
// the pooling head's scorer (attention_c: d -> 1) per branch, for the kernels that fold it into the pooling (token tail)
struct PoolScorer {
    const float* wc[2] = {nullptr, nullptr};      // attention_c.weight [1][d]
    const float* bc[2] = {nullptr, nullptr};      // attention_c.bias [1]
    float* dwc[2] = {nullptr, nullptr};           // their gradients (backward)
    float* dbc[2] = {nullptr, nullptr};
    int n_slides = 1;                             // slides per branch
};
int mpo_launch_pool_score_fwd(const float* a, const float* b, const float* x, const PoolScorer& ps, float* scores, float* w,
                              float* h, int B, int L, int d, hipStream_t s);
int mpo_launch_pool_score_bwd(const float* dh, const float* x, const float* w, const float* d_ext, const float* a, const float* b,
                              const PoolScorer& ps, float* d_scores, float* dx, float* da, float* db, int B, int L, int d,
                              hipStream_t s);
int mpo_launch_pool_fwd(const float* scores, const float* x, float* w, float* h, int B, int L, int d, hipStream_t s);
int mpo_launch_pool_bwd(const float* dh, const float* x, const float* w, const float* d_ext, float* d_scores, float* dx,
                        int B, int L, int d, hipStream_t s);
int mpo_launch_head_fwd(const float* logits, float* hazards, float* survs, float* y, int B, int C, hipStream_t s);
int mpo_launch_head_loss(const float* logits, const long long* label, const float* cens, const float* w, float* hazards,
                         float* survs, float* y, float* loss, float* risk, float* dlogits, int B, int C, float alpha,
                         float eps, hipStream_t s);
int mpo_launch_counters_bump(unsigned long long* epoch, int* step, hipStream_t s);
int mpo_launch_head_bwd(const float* hazards, const float* survs, const float* y, const float* dhz, const float* dsv,
                        const float* dy, float* dlogits, int B, int C, hipStream_t s);
int mpo_launch_ew_add(float* acc, const float* b, size_t n, hipStream_t s);      // acc += b
int mpo_launch_ew_mul(const float* a, const float* b, float* out, int n, hipStream_t s);
int mpo_launch_ew_mul2(const float* x, const float* p, const float* q, float* xp, float* xq, int n, hipStream_t s);
int mpo_launch_ew_add(const float* a, const float* b, float* out, int n, hipStream_t s);
int mpo_launch_cag_mid_fwd(const float* u1, const float* u2, const float* u3, const float* gw, const float* gb, const float* ew,
                           const float* eb, float* t1, float* t3, float* gout, float* eout, float* m, float* stats_g, float* stats_e,
                           int rows, int d, float eps, hipStream_t s);
int mpo_launch_cag_mid_bwd(const float* dm, const float* t1, const float* t3, const float* gout, const float* eout,
                           const float* gw, const float* ew, const float* stats_g, const float* stats_e, float* dG, float* dE,
                           float* ds12, float* ds3, int rows, int d, hipStream_t s);

// h = drop(relu(h + bias)) in place on a bf16 [rows][cols] tensor (the patch layer's epilogue)
int mpo_launch_bias_relu_dropout_bf16(void* h, const float* bias, size_t rows, int cols, float drop_p,
                                      unsigned long long seed, unsigned long long offset, const unsigned long long* epoch,
                                      hipStream_t stream);
// g = dy * (h > 0 ? 1/(1-p) : 0) on bf16 tensors (derivative of the same epilogue)
int mpo_relu_dropout_bwd_blocks(size_t n, int with_colsum);
int mpo_launch_relu_dropout_bwd_bf16(const void* h, const void* dy, void* g, size_t n, float drop_p, int cols,
                                     float* part_colsum /* nullable [blocks][cols] */, hipStream_t stream);

int mpo_launch_adam_flat(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                         float wd, int step, const int* step_dev, hipStream_t stream);
int mpo_launch_colsum_bf16(const void* x, float* out, size_t rows, int cols, hipStream_t stream);

// gated (tanh on the fly) single-pass variants for K2 (bagops.hip)
int mpo_launch_bag_rowdot_gated(const void* bag, int bag_f32, const int* cu, int n_slides, int embed, const float* r1,
                                const float* r2, float* a_map, float* g_map, int n_q, const BagPlan& plan, hipStream_t stream);
int mpo_launch_bag_colacc_gated(const void* bag, int bag_f32, const int* cu, int n_slides, int embed, const float* w1map,
                                const float* w2map, float* part1, float* part2, int n_q, const BagPlan& plan,
                                hipStream_t stream);
// G = (W1^T Z1 + ADD) * relu/dropout gate of H, bf16 rows, + column sums (bagops.hip)
int mpo_launch_bag_outer_gate(const int* cu, int n_slides, int embed, const float* w1, const float* z1, const void* addend,
                              const void* hbag, void* out, float gate, float* part_colsum, int n_q, const BagPlan& plan,
                              hipStream_t stream);
int mpo_launch_bag_key_grad(const float* kbag, const int* cu, int n_slides, int embed, const float* w1, const float* z1,
                            const float* w2, const float* z2, void* dk, int dk_f32, float* part_colsum, float* part1,
                            float* part2, int n_q, const BagPlan& plan, hipStream_t stream);
int mpo_launch_bag_outer_gated(const float* kbag, const int* cu, int n_slides, int embed, const float* w1, const float* z1,
                               const float* w2, const float* z2, void* dk, int dk_f32, float* part_colsum /* nullable */,
                               int n_q, const BagPlan& plan, hipStream_t stream);

// 'ces' survival loss (tail.hip)
int mpo_launch_ces_loss_fwd(const float* hazards, const float* survs, const long long* label, const float* cens, float* loss,
                            float* risk, int B, int C, float alpha, float eps, hipStream_t s);
int mpo_launch_ces_loss_bwd(const float* hazards, const float* survs, const long long* label, const float* cens,
                            const float* d_loss, int d_loss_scalar, float* d_hazards, float* d_survs, int B, int C,
                            float alpha, float eps, hipStream_t s);

// K2 key projection from a bf16 bag with hi/lo-split fp32 weights (keyproj.hip)
int mpo_launch_key_proj(const void* hbag_bf16, const float* w, const float* bias, float* kout, int rows, int embed,
                        hipStream_t stream);
