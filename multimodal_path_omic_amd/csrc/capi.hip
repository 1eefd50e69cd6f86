// extern "C" entry points of libmpo_hip.so (declared in include/mpo_hip.h) and the host-side
// orchestration of each one: a fixed sequence of kernel launches on the caller's stream, working
// only in caller-provided buffers.  No allocation, no synchronisation, graph-capturable.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/mpo_hip.h"
#include "coattn_tile.h"
#include "mpo_common.h"
#include "mpo_kernels.h"

static thread_local char g_err[512] = "";

void mpo_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// bump allocator over the caller's workspace
struct Arena {
    char* base;
    size_t size, off = 0;
    Arena(void* p, size_t n) : base(static_cast<char*>(p)), size(n) {}
    float* floats(size_t n) {
        const size_t o = align_up(off, 256);
        if (o + n * 4 > size) return nullptr;
        off = o + n * 4;
        return reinterpret_cast<float*>(base + o);
    }
};
static inline size_t arena_need(size_t acc, size_t n_floats) { return align_up(acc, 256) + n_floats * 4; }

namespace {
bool g_k2_one_pass_key = true;     // K2 backward: the whole key gradient in one pass over K (bag_key_grad_kernel; mpo_set_nacagat_one_pass_key_grad)
}

// ONE workgroup per CU over the window (256 CUs): long row ranges amortise the per-workgroup prologue
// (query fragments) and epilogue (LDS merge, partial write); measured r01 on 32 x 15k bf16:
// 256 WGs 51.6 us, 512 59.0, 1024 73.0, 2048 98.9.
extern "C" int mpo_coattn_target_workgroups(void) {
    return 256;
}

extern "C" int mpo_coattn_splits(int n_slides, int max_rows) {
    // uniform cut: the same number of row ranges for every slide; at least one 32-row tile per wave of a workgroup
    const int target = mpo_coattn_target_workgroups();
    int s = (target + n_slides - 1) / n_slides;
    if (s > 512) s = 512;
    const int cap = (max_rows + 127) / 128;
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    return s;
}

// C-ABI plan -> kernel plan.  NULL (or a NULL wg_start) selects the uniform cut.
static BagPlan make_plan(const mpo_bag_plan* p, int n_slides, int max_rows) {
    BagPlan pl;
    pl.n_slides = n_slides;
    if (p != nullptr && p->wg_start != nullptr) {
        pl.wg_start = p->wg_start;
        pl.n_wg = p->n_wg;
        pl.rows_per_wg = p->rows_per_wg;
    } else {
        pl.splits = mpo_coattn_splits(n_slides, max_rows);
    }
    return pl;
}
static int check_plan(const BagPlan& pl, int n_slides) {
    if (pl.wg_start == nullptr) return 0;
    MPO_CHECK(pl.rows_per_wg >= kTileRows && pl.rows_per_wg % kTileRows == 0, "bag plan: rows_per_wg %d must be a positive multiple of %d",
              pl.rows_per_wg, kTileRows);
    MPO_CHECK(pl.n_wg >= n_slides && pl.n_wg <= mpo_coattn_target_workgroups() + n_slides,
              "bag plan: n_wg %d outside [n_slides, target + n_slides]", pl.n_wg);
    return 0;
}
// upper bound on the number of split-M partials of any plan for this window
static size_t max_parts(int n_slides) { return (size_t)mpo_coattn_target_workgroups() + (size_t)n_slides; }

extern "C" {

int mpo_abi_version(void) { return 14; }
const char* mpo_last_error(void) { return g_err; }

int mpo_linear_forward(const float* x, const float* weight, const float* bias, float* y, int rows, int in_features,
                       int out_features, float alpha, int act, mpo_stream_t stream) {
    return mpo_linear_fwd(x, weight, bias, y, rows, in_features, out_features, alpha, act, stream);
}
int mpo_linear_backward_input(const float* dy, const float* weight, float* dx, int rows, int in_features,
                              int out_features, float alpha, int accumulate, mpo_stream_t stream) {
    return mpo_linear_bwd_input(dy, weight, dx, rows, in_features, out_features, alpha, accumulate, stream);
}
int mpo_linear_backward_weight(const float* dy, const float* x, float* dweight, float* dbias, int rows,
                               int in_features, int out_features, float alpha, mpo_stream_t stream) {
    return mpo_linear_bwd_weight(dy, x, dweight, dbias, rows, in_features, out_features, alpha, stream);
}

// ------------------------------------------------------------------------------------------- K1
// saved layout (floats), R = n_slides * n_q:   qs [R,E] | qk2 [R,E] | ctx [R,E] | attn [R,E] | lse2 [R]
size_t mpo_coattn_saved_floats(int n_slides, int n_q, int embed) {
    const size_t R = (size_t)n_slides * n_q;
    return 4 * R * embed + R;
}

size_t mpo_coattn_workspace_bytes(int n_slides, int n_q, int embed, int max_rows) {
    const size_t R = (size_t)n_slides * n_q;
    const size_t parts = max_parts(n_slides);
    (void)max_rows;
    size_t a = 0;
    // forward: part_ml, part_ctx.  backward: dattn, dctx, dqk, dq_pre, delta, part_dqk  (take the larger)
    size_t f = 0;
    f = arena_need(f, parts * 32);
    f = arena_need(f, parts * n_q * embed);
    size_t b = 0;
    for (int i = 0; i < 4; ++i) b = arena_need(b, R * embed);
    b = arena_need(b, R);
    b = arena_need(b, parts * n_q * embed);
    b = arena_need(b, parts * embed);
    a = f > b ? f : b;
    return a + 256;
}

static int check_common(int bag_dtype, int n_slides, int total_rows, int max_rows, int n_q, int embed) {
    MPO_CHECK(bag_dtype == MPO_F32 || bag_dtype == MPO_BF16, "bag dtype %d is neither MPO_F32 nor MPO_BF16", bag_dtype);
    MPO_CHECK(n_slides >= 1, "n_slides must be >= 1 (got %d)", n_slides);
    MPO_CHECK(n_q >= 1 && n_q <= 16, "number of omic queries must be in 1..16 (got %d)", n_q);
    MPO_CHECK(embed == 128 || embed == 256 || embed == 512, "embed_dim %d not in {128,256,512}", embed);
    MPO_CHECK(max_rows >= 1 && total_rows >= n_slides, "every slide needs at least one patch (total_rows %d, max_rows %d)",
              total_rows, max_rows);
    return 0;
}

int mpo_coattn_mcat_forward(const void* bag, int bag_dtype, const int32_t* cu_rows, int n_slides, int total_rows,
                            int max_rows, const float* query, int n_q, int embed, const float* in_w,
                            const float* in_b, const float* out_w, const float* out_b, float* out, float* attn_map,
                            float* saved, const mpo_bag_plan* plan_, void* workspace, size_t workspace_bytes,
                            mpo_stream_t stream) {
    if (int rc = check_common(bag_dtype, n_slides, total_rows, max_rows, n_q, embed)) return rc;
    const int E = embed, R = n_slides * n_q;
    const BagPlan plan = make_plan(plan_, n_slides, max_rows);
    if (int rc = check_plan(plan, n_slides)) return rc;
    Arena ws(workspace, workspace_bytes);
    float* part_ml = ws.floats(plan_parts(plan) * 32);
    float* part_ctx = ws.floats(plan_parts(plan) * n_q * E);
    MPO_CHECK(part_ml && part_ctx, "coattn forward: workspace too small (%zu bytes)", workspace_bytes);
    float* qs = saved;
    float* qk2 = qs + (size_t)R * E;
    float* ctx = qk2 + (size_t)R * E;
    float* attn = ctx + (size_t)R * E;
    float* lse2 = attn + (size_t)R * E;
    const float scale = 1.0f / sqrtf((float)E);
    int rc;
    // qs = (query W_q^T + b_q) / sqrt(E)
    if ((rc = mpo_linear_fwd(query, in_w, in_b, qs, R, E, E, scale, MPO_ACT_NONE, stream))) return rc;
    // qk2 = log2(e) * qs W_k     (fold of the key projection into the query; key bias cancels in softmax)
    if ((rc = mpo_linear_bwd_input(qs, in_w + (size_t)E * E, qk2, R, E, E, kLog2e, 0, stream))) return rc;
    if ((rc = mpo_launch_coattn_fwd_partial(bag, bag_dtype == MPO_F32, cu_rows, n_slides, E, qk2, part_ml, part_ctx,
                                            attn_map, n_q, plan, stream))) return rc;
    if ((rc = mpo_launch_coattn_combine(part_ml, part_ctx, ctx, lse2, n_slides, n_q, E, plan, stream))) return rc;
    // attn = ctx W_v^T + b_v   (rows of A sum to one);  out = attn W_o^T + b_o
    if ((rc = mpo_linear_fwd(ctx, in_w + (size_t)2 * E * E, in_b + 2 * E, attn, R, E, E, 1.0f, MPO_ACT_NONE, stream))) return rc;
    if ((rc = mpo_linear_fwd(attn, out_w, out_b, out, R, E, E, 1.0f, MPO_ACT_NONE, stream))) return rc;
    if (attn_map)
        if ((rc = mpo_launch_coattn_normalize(attn_map, lse2, cu_rows, n_slides, n_q, max_rows, 0.f, 0, 0, stream))) return rc;
    return 0;
}

// ------------------------------------------------------------------------------------------- row f1: patch layer + K1
size_t mpo_patch_coattn_workspace_bytes(int n_slides, int n_q, int embed, int patch_dim) {
    size_t f = 0;
    f = arena_need(f, max_parts(n_slides) * 32);
    f = arena_need(f, max_parts(n_slides) * n_q * embed);
    f = arena_need(f, (size_t)embed * patch_dim / 2);           // W_H as bf16 (2 bytes per element)
    return f + 256;
}

int mpo_patch_coattn_mcat_forward(const void* patches, const int32_t* cu_rows, int n_slides, int total_rows, int max_rows,
                                  int patch_dim, const float* patch_weight, const float* patch_bias, float drop_p,
                                  uint64_t seed, uint64_t offset, const uint64_t* rng_epoch,
                                  const float* query, int n_q, int embed, const float* in_w, const float* in_b,
                                  const float* out_w, const float* out_b, void* h_bag, float* out, float* attn_map,
                                  float* saved, const mpo_bag_plan* plan_, void* workspace, size_t workspace_bytes,
                                  mpo_stream_t stream) {
    if (int rc = check_common(MPO_BF16, n_slides, total_rows, max_rows, n_q, embed)) return rc;
    MPO_CHECK(embed == 256 && patch_dim == 1024, "fused patch layer + co-attention is built for 1024 -> 256 (got %d -> %d)",
              patch_dim, embed);
    const int E = embed, R = n_slides * n_q;
    const BagPlan plan = make_plan(plan_, n_slides, max_rows);
    if (int rc = check_plan(plan, n_slides)) return rc;
    Arena ws(workspace, workspace_bytes);
    float* part_ml = ws.floats(plan_parts(plan) * 32);
    float* part_ctx = ws.floats(plan_parts(plan) * n_q * E);
    float* w_bf16 = ws.floats((size_t)E * patch_dim / 2);
    MPO_CHECK(part_ml && part_ctx && w_bf16, "fused patch layer + co-attention: workspace too small (%zu bytes)", workspace_bytes);
    float* qs = saved;                                     // same saved layout as mpo_coattn_mcat_forward: its backward applies
    float* qk2 = qs + (size_t)R * E;
    float* ctx = qk2 + (size_t)R * E;
    float* attn = ctx + (size_t)R * E;
    float* lse2 = attn + (size_t)R * E;
    const float scale = 1.0f / sqrtf((float)E);
    int rc;
    if ((rc = mpo_launch_pack_patch_weight(patch_weight, w_bf16, E, patch_dim, stream))) return rc;
    if ((rc = mpo_linear_fwd(query, in_w, in_b, qs, R, E, E, scale, MPO_ACT_NONE, stream))) return rc;
    if ((rc = mpo_linear_bwd_input(qs, in_w + (size_t)E * E, qk2, R, E, E, kLog2e, 0, stream))) return rc;
    if ((rc = mpo_launch_patch_fc_fwd(patches, w_bf16, patch_bias, cu_rows, h_bag, E, drop_p, seed, offset,
                                      reinterpret_cast<const unsigned long long*>(rng_epoch), plan, stream))) return rc;
    if ((rc = mpo_launch_coattn_fwd_partial(h_bag, 0, cu_rows, n_slides, E, qk2, part_ml, part_ctx, attn_map, n_q, plan, stream))) return rc;
    if ((rc = mpo_launch_coattn_combine(part_ml, part_ctx, ctx, lse2, n_slides, n_q, E, plan, stream))) return rc;
    if ((rc = mpo_linear_fwd(ctx, in_w + (size_t)2 * E * E, in_b + 2 * E, attn, R, E, E, 1.0f, MPO_ACT_NONE, stream))) return rc;
    if ((rc = mpo_linear_fwd(attn, out_w, out_b, out, R, E, E, 1.0f, MPO_ACT_NONE, stream))) return rc;
    if (attn_map)
        if ((rc = mpo_launch_coattn_normalize(attn_map, lse2, cu_rows, n_slides, n_q, max_rows, 0.f, 0, 0, stream))) return rc;
    return 0;
}

// The patch layer alone, H_bag = dropout(relu(X W_H^T + b_H)) (models/mcat/mcat.py:24-29,87), as ONE pass of the same
// kernel with its co-attention slices switched off: for the models whose co-attention needs more than H_bag (NaCAGaT's
// key projection) and for MCAT outside the fused configuration.  Workspace: the packed bf16 copy of the weight.
size_t mpo_patch_fc_workspace_bytes(int embed, int patch_dim) { return (size_t)(embed < 256 ? 256 : embed) * patch_dim * 2 + 256; }
int mpo_patch_fc_forward(const void* patches, const int32_t* cu_rows, int n_slides, int total_rows, int max_rows, int patch_dim,
                         const float* patch_weight, const float* patch_bias, int embed, float drop_p, uint64_t seed,
                         uint64_t offset, const uint64_t* rng_epoch, void* h_bag, const mpo_bag_plan* plan_, void* workspace,
                         size_t workspace_bytes, mpo_stream_t stream) {
    if (int rc = check_common(MPO_BF16, n_slides, total_rows, max_rows, 1, embed)) return rc;
    MPO_CHECK((embed == 128 || embed == 256 || embed == 512) && patch_dim == 1024,
              "patch layer kernel is built for 1024 -> 128, 256 or 512 (got %d -> %d)", patch_dim, embed);
    MPO_CHECK((int64_t)max_rows * embed * 2 < ((int64_t)1 << 31), "patch layer: a slide's H_bag of 2 GiB or more (%d rows)", max_rows);
    const BagPlan plan = make_plan(plan_, n_slides, max_rows);
    if (int rc = check_plan(plan, n_slides)) return rc;
    Arena ws(workspace, workspace_bytes);
    float* w_bf16 = ws.floats((size_t)(embed < 256 ? 256 : embed) * patch_dim / 2);
    MPO_CHECK(w_bf16, "patch layer: workspace too small (%zu bytes)", workspace_bytes);
    int rc;
    if ((rc = mpo_launch_pack_patch_weight(patch_weight, w_bf16, embed, patch_dim, stream))) return rc;
    return mpo_launch_patch_fc_fwd(patches, w_bf16, patch_bias, cu_rows, h_bag, embed, drop_p, seed, offset,
                                   reinterpret_cast<const unsigned long long*>(rng_epoch), plan, stream);
}

// fp32-stored window: the patch layer on patch_fc_f32.hip
size_t mpo_patch_fc_f32_workspace_bytes(int backward) {
    return 256 + 4 * (backward ? mpo_patch_wgrad_f32_workspace_floats() : mpo_patch_fc_f32_workspace_floats());
}
int mpo_patch_fc_f32_forward(const float* patches, int64_t total_rows, int patch_dim, const float* patch_weight,
                             const float* patch_bias, int embed, float drop_p, uint64_t seed, uint64_t offset,
                             const uint64_t* rng_epoch, float x_scale, float* h_bag, void* workspace, size_t workspace_bytes,
                             mpo_stream_t stream) {
    MPO_CHECK(patches && patch_weight && patch_bias && h_bag, "fp32 patch layer: null operand");
    MPO_CHECK(total_rows >= 1, "fp32 patch layer: total_rows %lld", (long long)total_rows);
    Arena ws(workspace, workspace_bytes);
    float* wpk = ws.floats(mpo_patch_fc_f32_workspace_floats());
    MPO_CHECK(wpk, "fp32 patch layer: workspace too small (%zu bytes)", workspace_bytes);
    return mpo_launch_patch_fc_f32(patches, patch_weight, patch_bias, h_bag, total_rows, embed, patch_dim, drop_p, seed, offset,
                                   reinterpret_cast<const unsigned long long*>(rng_epoch), x_scale, wpk, stream);
}
int mpo_patch_fc_f32_backward(const float* d_h_bag, const float* h_bag, const float* patches, int64_t total_rows, int embed,
                              int patch_dim, float gate, float* d_weight, float* d_bias, void* workspace, size_t workspace_bytes,
                              mpo_stream_t stream) {
    MPO_CHECK(d_h_bag && patches && d_weight, "fp32 patch layer backward: null operand");
    MPO_CHECK(total_rows >= 1, "fp32 patch layer backward: total_rows %lld", (long long)total_rows);
    Arena ws(workspace, workspace_bytes);
    float* part = ws.floats(mpo_patch_wgrad_f32_workspace_floats());
    MPO_CHECK(part, "fp32 patch layer backward: workspace too small (%zu bytes)", workspace_bytes);
    return mpo_launch_patch_wgrad_f32(d_h_bag, h_bag, patches, total_rows, embed, patch_dim, gate, d_weight, d_bias, part, stream);
}

// the fused bag pass alone (bench.py's roofline leg, profiling workloads)
int mpo_patch_coattn_fwd_bagpass(const void* patches, const void* w_packed, const float* bias, const int32_t* cu_rows, int n_slides,
                                 const float* qk2, void* h_bag, float* part_ml, float* part_ctx, int n_q, int max_rows,
                                 float drop_p, uint64_t seed, uint64_t offset, const mpo_bag_plan* plan_, mpo_stream_t stream) {
    const BagPlan plan = make_plan(plan_, n_slides, max_rows);
    if (int rc = check_plan(plan, n_slides)) return rc;
    if (int rc = mpo_launch_patch_fc_fwd(patches, w_packed, bias, cu_rows, h_bag, 256, drop_p, seed, offset, nullptr, plan, stream)) return rc;
    if (qk2 == nullptr) return 0;
    return mpo_launch_coattn_fwd_partial(h_bag, 0, cu_rows, n_slides, 256, qk2, part_ml, part_ctx, nullptr, n_q, plan, stream);
}
int mpo_pack_patch_weight(const float* weight, void* packed, int embed, int patch_dim, mpo_stream_t stream) {
    return mpo_launch_pack_patch_weight(weight, packed, embed, patch_dim, stream);
}

int mpo_coattn_mcat_backward(const void* bag, int bag_dtype, const int32_t* cu_rows, int n_slides, int total_rows,
                             int max_rows, const float* query, int n_q, int embed, const float* in_w,
                             const float* out_w, const float* saved, const float* attn_map, const float* d_out,
                             const float* d_attn_map, float* d_query, int d_query_accumulate, void* d_bag,
                             float* d_bag_colsum, float* d_in_w,
                             float* d_in_b, float* d_out_w, float* d_out_b, float bag_relu_gate, const mpo_bag_plan* plan_,
                             void* workspace, size_t workspace_bytes, mpo_stream_t stream) {
    if (int rc = check_common(bag_dtype, n_slides, total_rows, max_rows, n_q, embed)) return rc;
    MPO_CHECK(!d_attn_map || attn_map, "coattn backward: a gradient on the attention map needs the forward's map");
    const int E = embed, R = n_slides * n_q;
    const BagPlan plan = make_plan(plan_, n_slides, max_rows);
    if (int rc = check_plan(plan, n_slides)) return rc;
    Arena ws(workspace, workspace_bytes);
    float* dattn = ws.floats((size_t)R * E);
    float* dctx = ws.floats((size_t)R * E);
    float* dqk = ws.floats((size_t)R * E);
    float* dq_pre = ws.floats((size_t)R * E);
    float* delta = ws.floats(R);
    float* part_dqk = ws.floats(plan_parts(plan) * n_q * E);
    float* part_cs = d_bag_colsum ? ws.floats(plan_parts(plan) * E) : nullptr;
    MPO_CHECK(dattn && dctx && dqk && dq_pre && delta && part_dqk && (part_cs || !d_bag_colsum),
              "coattn backward: workspace too small (%zu bytes)", workspace_bytes);
    const float* qs = saved;
    const float* qk2 = qs + (size_t)R * E;
    const float* ctx = qk2 + (size_t)R * E;
    const float* attn = ctx + (size_t)R * E;
    const float* lse2 = attn + (size_t)R * E;
    const float* w_q = in_w;
    const float* w_k = in_w + (size_t)E * E;
    const float* w_v = in_w + (size_t)2 * E * E;
    const float scale = 1.0f / sqrtf((float)E);
    int rc;
    // out = attn W_o^T + b_o
    if ((rc = mpo_linear_bwd_pair(mpo_args_bwd_input(d_out, out_w, dattn, R, E, E, 1.0f, 0),
                                  mpo_args_bwd_weight(d_out, attn, d_out_w, d_out_b, R, E, E, 1.0f), stream))) return rc;
    // attn = ctx W_v^T + b_v
    if ((rc = mpo_linear_bwd_pair(mpo_args_bwd_input(dattn, w_v, dctx, R, E, E, 1.0f, 0),
                                  mpo_args_bwd_weight(dattn, ctx, d_in_w + (size_t)2 * E * E, d_in_b + 2 * E, R, E, E, 1.0f),
                                  stream))) return rc;
    // delta = rowsum(dctx * ctx) [+ rowsum(A * dA_ext)]: inside the bag pass unless a map gradient adds its term
    if (d_attn_map) {
        if ((rc = mpo_launch_rowdot(dctx, ctx, delta, R, E, stream))) return rc;
        if ((rc = mpo_launch_map_rowdot(attn_map, d_attn_map, cu_rows, delta, n_slides, n_q, 1, stream))) return rc;
    }
    // the bag pass
    if ((rc = mpo_launch_coattn_bwd(bag, bag_dtype == MPO_F32, cu_rows, n_slides, E, qk2, lse2, dctx,
                                    d_attn_map ? delta : nullptr, ctx, attn_map, d_attn_map, d_bag, part_dqk, part_cs, n_q,
                                    plan, bag_relu_gate, stream))) return rc;
    {   // one launch: dqk = sum of the split-M partials, the bag's column sums, db_k = 0 (softmax is shift-invariant)
        BagFinish f{};
        f.part[0] = part_dqk; f.out[0] = dqk; f.n_red = 1;
        f.part_cs = part_cs; f.colsum = d_bag_colsum; f.cs_cols = E;
        f.zero[0] = d_in_b + E; f.n_zero[0] = E;
        if ((rc = mpo_launch_bag_finish(f, n_slides, n_q, E, plan, stream))) return rc;
    }
    // qk = qs W_k :  dqs = dqk W_k^T (folded with the 1/sqrt(E) of qs = scale * (...)),  dW_k = qs^T dqk
    if ((rc = mpo_gemm_together(stream, mpo_args_fwd(dqk, w_k, nullptr, dq_pre, R, E, E, scale, MPO_ACT_NONE),
                                mpo_args_bwd_weight(qs, dqk, d_in_w + (size_t)E * E, nullptr, R, E, E, 1.0f)))) return rc;
    // q_pre = query W_q^T + b_q
    if ((rc = mpo_linear_bwd_pair(mpo_args_bwd_input(dq_pre, w_q, d_query, R, E, E, 1.0f, d_query_accumulate ? 1 : 0),
                                  mpo_args_bwd_weight(dq_pre, query, d_in_w, d_in_b, R, E, E, 1.0f), stream))) return rc;
    return 0;
}

// ------------------------------------------------------------------------------------------- K2
// saved layout (floats), R = n_slides * n_q:  qt | qs2 | tq | ctx | attn  (each [R,E])  | lse2 [R] | asum [R]
size_t mpo_nacagat_saved_floats(int n_slides, int n_q, int embed) {
    const size_t R = (size_t)n_slides * n_q;
    return 5 * R * embed + 2 * R;
}
size_t mpo_nacagat_workspace_bytes(int n_slides, int n_q, int embed, int max_rows, int total_rows) {
    const size_t R = (size_t)n_slides * n_q;
    (void)max_rows;
    size_t b = 0;
    for (int i = 0; i < 6; ++i) b = arena_need(b, R * embed);
    b = arena_need(b, R);
    b = arena_need(b, max_parts(n_slides) * n_q * embed);
    b = arena_need(b, max_parts(n_slides) * n_q * embed);
    b = arena_need(b, max_parts(n_slides) * embed);
    b = arena_need(b, (size_t)n_q * total_rows);
    b = arena_need(b, (size_t)n_q * total_rows);
    if (embed == 512) {                                            // the column-half passes of the big model (see below)
        for (int i = 0; i < 6; ++i) b = arena_need(b, R * (embed / 2));
        b = arena_need(b, (size_t)2 * n_q * total_rows);
    }
    return b + 256;
}

// embed_dim 512 ('big', models/nacagat/nacagat.py:17-18): the bag kernels are built for embed <= 256, so the two bags travel in
// the SPLIT-HALVES layout [2][total_rows][256] (columns 0..255 | 256..511) and every bag pass runs once per column half on
// the 256-wide kernels: the score / gradient maps are sums over the halves (linear in the embed index), the column-indexed
// results (context, query-side sums, dK, dH) are the halves side by side.  The 6 x 512 query-side tensors keep their natural
// layout; their halves are strided copies.  Functional, not tuned (four small copies and one map-sized add per pass).
static int copy_cols(float* dst, size_t dst_ld, const float* src, size_t src_ld, int rows, int cols, hipStream_t s) {
    MPO_HIP(hipMemcpy2DAsync(dst, dst_ld * 4, src, src_ld * 4, (size_t)cols * 4, rows, hipMemcpyDeviceToDevice, s));
    return 0;
}

int mpo_coattn_nacagat_forward(const void* kbag, int k_dtype, const void* hbag, int bag_dtype, const int32_t* cu_rows, int n_slides,
                               int total_rows, int max_rows, const float* query, int n_q, int embed,
                               const float* in_w, const float* in_b, const float* out_w, const float* out_b,
                               float drop_p, uint64_t seed, uint64_t offset, const uint64_t* rng_epoch,
                               float* q_proj, float* out, float* attn_map, float* score_maps,
                               float* saved, const mpo_bag_plan* plan_, void* workspace, size_t workspace_bytes,
                               mpo_stream_t stream) {
    if (int rc = check_common(bag_dtype, n_slides, total_rows, max_rows, n_q, embed)) return rc;
    MPO_CHECK(drop_p >= 0.f && drop_p < 1.f, "attention dropout p must be in [0,1) (got %f)", (double)drop_p);
    MPO_CHECK(k_dtype == MPO_F32, "nacagat co-attention: K must be fp32 (k_dtype %d): the narrow gate amplifies key rounding", k_dtype);
    const int E = embed, R = n_slides * n_q, f32 = bag_dtype == MPO_F32;
    const int NH = E == 512 ? 2 : 1, EH = E / NH;                          // column halves (split-halves bag layout at 512)
    const size_t half_k = (size_t)total_rows * EH * 4, half_h = (size_t)total_rows * EH * (f32 ? 4 : 2);
    const BagPlan splits = make_plan(plan_, n_slides, max_rows);          // (named `splits`: it replaces the old count)
    if (int rc = check_plan(splits, n_slides)) return rc;
    Arena ws(workspace, workspace_bytes);
    float* part = ws.floats(plan_parts(splits) * n_q * E);
    MPO_CHECK(part, "nacagat forward: workspace too small (%zu bytes)", workspace_bytes);
    float *hq1 = nullptr, *hq2 = nullptr, *hctx = nullptr, *tmp_maps = nullptr;
    if (NH > 1) {
        hq1 = ws.floats((size_t)R * EH); hq2 = ws.floats((size_t)R * EH); hctx = ws.floats((size_t)R * EH);
        tmp_maps = ws.floats((size_t)2 * n_q * total_rows);
        MPO_CHECK(hq1 && hq2 && hctx && tmp_maps, "nacagat forward: workspace too small (%zu bytes)", workspace_bytes);
    }
    float* qt = saved;
    float* qs2 = qt + (size_t)R * E;
    float* tq = qs2 + (size_t)R * E;
    float* ctx = tq + (size_t)R * E;
    float* attn = ctx + (size_t)R * E;
    float* lse2 = attn + (size_t)R * E;
    float* asum = lse2 + R;
    float* a_map = score_maps;
    float* g_map = score_maps + (size_t)n_q * total_rows;
    int rc;
    // q = query W_q^T + b_q  (returned: the reference hands it to the CAG, models/blocks.py:110,206)
    if ((rc = mpo_linear_fwd(query, in_w, in_b, q_proj, R, E, E, 1.0f, MPO_ACT_NONE, stream))) return rc;
    if ((rc = mpo_launch_qprep(q_proj, qt, qs2, tq, R * E, 1.0f / sqrtf((float)E), stream))) return rc;
    // one pass over K: a = qs2 . K and g = tanh(q) . tanh(K)   (tanh(K) is never materialised)
    for (int h = 0; h < NH; ++h) {
        const float *r1 = qs2, *r2 = tq;
        if (NH > 1) {
            if ((rc = copy_cols(hq1, EH, qs2 + h * EH, E, R, EH, stream))) return rc;
            if ((rc = copy_cols(hq2, EH, tq + h * EH, E, R, EH, stream))) return rc;
            r1 = hq1; r2 = hq2;
        }
        float* am = h == 0 ? a_map : tmp_maps;
        float* gm = h == 0 ? g_map : tmp_maps + (size_t)n_q * total_rows;
        if ((rc = mpo_launch_bag_rowdot_gated(static_cast<const char*>(kbag) + h * half_k, 1, cu_rows, n_slides, EH, r1, r2, am, gm,
                                              n_q, splits, stream))) return rc;
        if (h > 0)
            if ((rc = mpo_launch_ew_add(score_maps, tmp_maps, (size_t)2 * n_q * total_rows, stream))) return rc;
    }
    if ((rc = mpo_launch_gated_softmax_fwd(a_map, g_map, cu_rows, attn_map, lse2, asum, n_slides, n_q, drop_p, seed, offset,
                                           reinterpret_cast<const unsigned long long*>(rng_epoch), stream))) return rc;
    for (int h = 0; h < NH; ++h) {
        if ((rc = mpo_launch_bag_colacc(static_cast<const char*>(hbag) + h * half_h, f32, cu_rows, n_slides, EH, attn_map, part, n_q,
                                        splits, stream))) return rc;
        if ((rc = mpo_launch_coattn_bwd_reduce(part, NH > 1 ? hctx : ctx, n_slides, n_q, EH, splits, stream))) return rc;
        if (NH > 1)
            if ((rc = copy_cols(ctx + h * EH, E, hctx, EH, R, EH, stream))) return rc;
    }
    // attn = ctx W_v^T + (sum_m A_drop) b_v ;  out = attn W_o^T + b_o
    if ((rc = mpo_linear_fwd(ctx, in_w + (size_t)2 * E * E, nullptr, attn, R, E, E, 1.0f, MPO_ACT_NONE, stream))) return rc;
    if ((rc = mpo_launch_row_scaled_bias(attn, asum, in_b + 2 * E, R, E, stream))) return rc;
    if ((rc = mpo_linear_fwd(attn, out_w, out_b, out, R, E, E, 1.0f, MPO_ACT_NONE, stream))) return rc;
    return 0;
}

int mpo_coattn_nacagat_backward(const void* kbag, int k_dtype, const void* hbag, int bag_dtype,
                                const int32_t* cu_rows, int n_slides, int total_rows, int max_rows,
                                const float* query, int n_q, int embed, const float* in_w, const float* in_b,
                                const float* out_w, float drop_p, uint64_t seed, uint64_t offset, const uint64_t* rng_epoch,
                                const float* saved, const float* score_maps, const float* attn_map,
                                const float* d_out, const float* d_attn_map, const float* d_q_proj,
                                float* d_query, int d_query_accumulate, void* d_kbag, int dk_dtype, float* d_kbag_colsum, void* d_hbag,
                                float* d_ctx, float* d_in_w, float* d_in_b, float* d_out_w, float* d_out_b,
                                const mpo_bag_plan* plan_, void* workspace, size_t workspace_bytes, mpo_stream_t stream) {
    if (int rc = check_common(bag_dtype, n_slides, total_rows, max_rows, n_q, embed)) return rc;
    MPO_CHECK(k_dtype == MPO_F32, "nacagat co-attention: K must be fp32 (k_dtype %d): the narrow gate amplifies key rounding", k_dtype);
    MPO_CHECK(dk_dtype == MPO_F32 || dk_dtype == MPO_BF16, "d_kbag dtype %d is neither MPO_F32 nor MPO_BF16", dk_dtype);
    const int E = embed, R = n_slides * n_q, f32 = bag_dtype == MPO_F32;
    const int NH = E == 512 ? 2 : 1, EH = E / NH;                          // column halves (split-halves bag layout at 512)
    const size_t half_k = (size_t)total_rows * EH * 4, half_h = (size_t)total_rows * EH * (f32 ? 4 : 2);
    const size_t half_dk = (size_t)total_rows * EH * (dk_dtype == MPO_F32 ? 4 : 2);
    const BagPlan splits = make_plan(plan_, n_slides, max_rows);
    if (int rc = check_plan(splits, n_slides)) return rc;
    Arena ws(workspace, workspace_bytes);
    float* dattn = ws.floats((size_t)R * E);
    float* dctx_ws = ws.floats((size_t)R * E);
    float* dctx = d_ctx ? d_ctx : dctx_ws;                 // a caller that finishes dH itself keeps dL/dctx
    MPO_CHECK(d_ctx != nullptr || d_hbag != nullptr, "nacagat backward: neither d_hbag nor d_ctx given");
    float* dqt = ws.floats((size_t)R * E);
    float* dtq = ws.floats((size_t)R * E);
    float* dq = ws.floats((size_t)R * E);
    float* spare = ws.floats((size_t)R * E);
    float* dasum = ws.floats(R);
    float* part = ws.floats(plan_parts(splits) * n_q * E);
    float* part2 = ws.floats(plan_parts(splits) * n_q * E);
    float* part_cs = d_kbag_colsum ? ws.floats(plan_parts(splits) * E) : nullptr;
    MPO_CHECK(part_cs || !d_kbag_colsum, "nacagat backward: workspace too small (%zu bytes)", workspace_bytes);
    float* ds1_map = ws.floats((size_t)n_q * total_rows);
    float* dg_map = ws.floats((size_t)n_q * total_rows);
    MPO_CHECK(dattn && dctx_ws && dqt && dtq && dq && spare && dasum && part && part2 && ds1_map && dg_map,
              "nacagat backward: workspace too small (%zu bytes)", workspace_bytes);
    float* hb[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};      // [R][EH] scratch of the column-half passes
    float* tmp_map = nullptr;
    if (NH > 1) {
        for (int i = 0; i < 6; ++i) hb[i] = ws.floats((size_t)R * EH);
        tmp_map = ws.floats((size_t)2 * n_q * total_rows);
        MPO_CHECK(hb[5] && tmp_map, "nacagat backward: workspace too small (%zu bytes)", workspace_bytes);
    }
    const float* qt = saved;
    const float* qs2 = qt + (size_t)R * E;
    const float* tq = qs2 + (size_t)R * E;
    const float* ctx = tq + (size_t)R * E;
    const float* attn = ctx + (size_t)R * E;
    const float* lse2 = attn + (size_t)R * E;
    const float* asum = lse2 + R;
    const float* a_map = score_maps;
    const float* g_map = score_maps + (size_t)n_q * total_rows;
    const float* w_q = in_w;
    const float* w_v = in_w + (size_t)2 * E * E;
    const float* b_v = in_b + 2 * E;
    (void)qs2;
    int rc;
    // out = attn W_o^T + b_o
    if ((rc = mpo_linear_bwd_pair(mpo_args_bwd_input(d_out, out_w, dattn, R, E, E, 1.0f, 0),
                                  mpo_args_bwd_weight(d_out, attn, d_out_w, d_out_b, R, E, E, 1.0f), stream))) return rc;
    // attn = ctx W_v^T + asum (x) b_v
    {
        const GemmArgs dbv = mpo_args_bwd_weight(asum, dattn, d_in_b + 2 * E, nullptr, R, E, 1, 1.0f);       // db_v = asum^T dattn
        const GemmArgs das = mpo_args_fwd(dattn, b_v, nullptr, dasum, R, E, 1, 1.0f, MPO_ACT_NONE);           // dasum = dattn b_v
        if ((rc = mpo_gemm_together(stream, mpo_args_bwd_input(dattn, w_v, dctx, R, E, E, 1.0f, 0),
                                    mpo_args_bwd_weight(dattn, ctx, d_in_w + (size_t)2 * E * E, nullptr, R, E, E, 1.0f),
                                    &dbv, &das))) return rc;
    }
    // map side: dA = dctx . H^T (summed over the column halves at 512; hb[0], hb[1] keep the two halves of dctx)
    for (int h = 0; h < NH; ++h) {
        const float* dch = dctx;
        if (NH > 1) {
            if ((rc = copy_cols(hb[h], EH, dctx + h * EH, E, R, EH, stream))) return rc;
            dch = hb[h];
        }
        if ((rc = mpo_launch_bag_rowdot(static_cast<const char*>(hbag) + h * half_h, f32, cu_rows, n_slides, EH, dch,
                                        h == 0 ? ds1_map : tmp_map, 1.0f, n_q, splits, stream))) return rc;
        if (h > 0)
            if ((rc = mpo_launch_ew_add(ds1_map, tmp_map, (size_t)n_q * total_rows, stream))) return rc;
    }
    if ((rc = mpo_launch_gated_softmax_bwd(a_map, g_map, cu_rows, lse2, dasum, d_attn_map, ds1_map, dg_map, n_slides, n_q,
                                           drop_p, seed, offset, reinterpret_cast<const unsigned long long*>(rng_epoch), stream))) return rc;
    // query side: dq~ = ds1 K, dtq = dg TK
    // (one pass over K, tanh on the fly) -- or, for N <= 6 at embed <= 256, out of the bag-side pass below: both need K and
    // the two maps and nothing of each other, so K (491 MB per 32 x 15 000 window) is read once for the two
    const bool one_pass = g_k2_one_pass_key && NH == 1 && n_q <= 6;
    for (int h = 0; h < NH && !one_pass; ++h) {
        if ((rc = mpo_launch_bag_colacc_gated(static_cast<const char*>(kbag) + h * half_k, 1, cu_rows, n_slides, EH, ds1_map, dg_map,
                                              part, part2, n_q, splits, stream))) return rc;
        BagFinish f{};
        f.part[0] = part; f.out[0] = NH > 1 ? hb[2] : dqt; f.part[1] = part2; f.out[1] = NH > 1 ? hb[3] : dtq; f.n_red = 2;
        if ((rc = mpo_launch_bag_finish(f, n_slides, n_q, EH, splits, stream))) return rc;
        if (NH > 1) {
            if ((rc = copy_cols(dqt + h * EH, E, hb[2], EH, R, EH, stream))) return rc;
            if ((rc = copy_cols(dtq + h * EH, E, hb[3], EH, R, EH, stream))) return rc;
        }
    }
    auto query_side = [&]() -> int {
        if (int r = mpo_launch_qprep_bwd(dqt, dtq, tq, d_q_proj, dq, R * E, 1.0f / sqrtf((float)E), stream)) return r;
        return mpo_linear_bwd_pair(mpo_args_bwd_input(dq, w_q, d_query, R, E, E, 1.0f, d_query_accumulate ? 1 : 0),
                                   mpo_args_bwd_weight(dq, query, d_in_w, d_in_b, R, E, E, 1.0f), stream);
    };
    if (!one_pass)
        if ((rc = query_side())) return rc;
    // bag side: dK = ds1^T q~ + (dg^T tq) * (1 - TK^2),  dH = A_drop^T dctx
    // (one pass: tanh' from the staged K tile)
    for (int h = 0; h < NH; ++h) {
        const float *qth = qt, *tqh = tq;
        if (NH > 1) {
            if ((rc = copy_cols(hb[4], EH, qt + h * EH, E, R, EH, stream))) return rc;
            if ((rc = copy_cols(hb[5], EH, tq + h * EH, E, R, EH, stream))) return rc;
            qth = hb[4]; tqh = hb[5];
        }
        if (one_pass) {
            if ((rc = mpo_launch_bag_key_grad(reinterpret_cast<const float*>(kbag), cu_rows, n_slides, EH, ds1_map, qth, dg_map, tqh,
                                              d_kbag, dk_dtype == MPO_F32, part_cs, part, part2, n_q, splits, stream))) return rc;
        } else if ((rc = mpo_launch_bag_outer_gated(reinterpret_cast<const float*>(static_cast<const char*>(kbag) + h * half_k), cu_rows,
                                             n_slides, EH, ds1_map, qth, dg_map, tqh, static_cast<char*>(d_kbag) + h * half_dk,
                                             dk_dtype == MPO_F32, part_cs, n_q, splits, stream))) return rc;
        // one launch: the key bag's column sums, and (first half) zeros for the key slice of the packed in-projection (it
        // belongs to the caller's K = H W_k^T + b_k; a caller may have the key-bias gradient written straight into its slice)
        BagFinish f{};
        f.part_cs = d_kbag_colsum ? part_cs : nullptr; f.colsum = d_kbag_colsum ? d_kbag_colsum + h * EH : nullptr; f.cs_cols = EH;
        if (h == 0) {
            f.zero[0] = d_in_w + (size_t)E * E; f.n_zero[0] = E * E;
            if (d_kbag_colsum != d_in_b + E) { f.zero[1] = d_in_b + E; f.n_zero[1] = E; }
        }
        if (one_pass) { f.part[0] = part; f.out[0] = dqt; f.part[1] = part2; f.out[1] = dtq; f.n_red = 2; }
        if ((rc = mpo_launch_bag_finish(f, n_slides, n_q, EH, splits, stream))) return rc;
    }
    if (one_pass)
        if ((rc = query_side())) return rc;
    if (d_ctx == nullptr)
        for (int h = 0; h < NH; ++h)
            if ((rc = mpo_launch_bag_outer(cu_rows, n_slides, EH, attn_map, NH > 1 ? hb[h] : dctx, nullptr, nullptr,
                                           static_cast<char*>(d_hbag) + h * half_h, f32, n_q, splits, stream))) return rc;
    return 0;
}

int mpo_nacagat_patch_grad(const int32_t* cu_rows, int n_slides, int total_rows, int max_rows, int n_q, int embed,
                           const float* attn_map, const float* d_ctx, const void* addend_bf16, const void* hbag_bf16,
                           void* d_bag_bf16, float relu_gate, float* d_bias, const mpo_bag_plan* plan_,
                           void* workspace, size_t workspace_bytes, mpo_stream_t stream) {
    if (int rc = check_common(MPO_BF16, n_slides, total_rows, max_rows, n_q, embed)) return rc;
    MPO_CHECK(attn_map && d_ctx && addend_bf16 && hbag_bf16 && d_bag_bf16, "nacagat patch grad: null operand");
    MPO_CHECK(((reinterpret_cast<uintptr_t>(addend_bf16) | reinterpret_cast<uintptr_t>(hbag_bf16) |
                reinterpret_cast<uintptr_t>(d_bag_bf16)) & 15) == 0, "nacagat patch grad: bag operands must be 16-byte aligned");
    const BagPlan splits = make_plan(plan_, n_slides, max_rows);
    if (int rc = check_plan(splits, n_slides)) return rc;
    Arena ws(workspace, workspace_bytes);
    float* part_cs = d_bias ? ws.floats(plan_parts(splits) * embed) : nullptr;
    MPO_CHECK(part_cs || !d_bias, "nacagat patch grad: workspace too small (%zu bytes)", workspace_bytes);
    int rc;
    if ((rc = mpo_launch_bag_outer_gate(cu_rows, n_slides, embed, attn_map, d_ctx, addend_bf16, hbag_bf16, d_bag_bf16,
                                        relu_gate, part_cs, n_q, splits, stream))) return rc;
    if (d_bias)
        if ((rc = mpo_launch_colsum(part_cs, d_bias, (int)plan_parts(splits), embed, embed, 0, stream))) return rc;
    return 0;
}

int mpo_nacagat_patch_grad_fused(const int32_t* cu_rows, int n_slides, int total_rows, int max_rows, int n_q, int embed,
                                 const float* attn_map, const float* d_ctx, const void* d_kbag_bf16, const float* w_k,
                                 const void* hbag_bf16, void* d_bag_bf16, float relu_gate, float* d_bias,
                                 const mpo_bag_plan* plan_, void* workspace, size_t workspace_bytes, mpo_stream_t stream) {
    if (int rc = check_common(MPO_BF16, n_slides, total_rows, max_rows, n_q, embed)) return rc;
    MPO_CHECK(attn_map && d_ctx && d_kbag_bf16 && w_k && hbag_bf16 && d_bag_bf16, "nacagat patch grad: null operand");
    MPO_CHECK(d_bag_bf16 != d_kbag_bf16, "nacagat patch grad (fused): d_bag must not alias d_kbag");
    MPO_CHECK(((reinterpret_cast<uintptr_t>(d_kbag_bf16) | reinterpret_cast<uintptr_t>(hbag_bf16) |
                reinterpret_cast<uintptr_t>(d_bag_bf16)) & 15) == 0, "nacagat patch grad: bag operands must be 16-byte aligned");
    const BagPlan splits = make_plan(plan_, n_slides, max_rows);
    if (int rc = check_plan(splits, n_slides)) return rc;
    Arena ws(workspace, workspace_bytes);
    float* part_cs = d_bias ? ws.floats(plan_parts(splits) * embed) : nullptr;
    MPO_CHECK(part_cs || !d_bias, "nacagat patch grad: workspace too small (%zu bytes)", workspace_bytes);
    int rc;
    if ((rc = mpo_launch_k2_patch_grad(cu_rows, d_kbag_bf16, w_k, attn_map, d_ctx, hbag_bf16, d_bag_bf16, relu_gate, part_cs, n_q,
                                       embed, splits, stream))) return rc;
    if (d_bias)
        if ((rc = mpo_launch_colsum(part_cs, d_bias, (int)plan_parts(splits), embed, embed, 0, stream))) return rc;
    return 0;
}

int mpo_coattn_fwd_bagpass(const void* bag, int bag_dtype, const int32_t* cu_rows, int n_slides, int embed,
                           const float* qk2, float* part_ml, float* part_ctx, float* raw_logits, int n_q, int max_rows,
                           const mpo_bag_plan* plan_, mpo_stream_t stream) {
    const BagPlan plan = make_plan(plan_, n_slides, max_rows);
    if (int rc = check_plan(plan, n_slides)) return rc;
    return mpo_launch_coattn_fwd_partial(bag, bag_dtype == MPO_F32, cu_rows, n_slides, embed, qk2, part_ml, part_ctx,
                                         raw_logits, n_q, plan, stream);
}
// per-(query, slide) dot products over the ragged maps, and the per-slide scaling that is the backward of a map norm
int mpo_map_block_dot(const float* a_map, const float* b_map, const int32_t* cu_rows, int n_slides, int n_q, float* out,
                      mpo_stream_t stream) {
    MPO_CHECK(a_map && b_map && cu_rows && out && n_slides >= 1 && n_q >= 1, "map block dot: bad argument");
    return mpo_launch_map_rowdot(a_map, b_map, cu_rows, out, n_slides, n_q, 0, stream);
}
int mpo_map_block_scale(const float* a_map, const float* scale, const int32_t* cu_rows, int n_slides, int n_q, float* out,
                        mpo_stream_t stream) {
    MPO_CHECK(a_map && scale && cu_rows && out && n_slides >= 1 && n_q >= 1, "map block scale: bad argument");
    return mpo_launch_map_block_scale(a_map, scale, cu_rows, out, n_slides, n_q, stream);
}
int mpo_key_projection(const void* hbag_bf16, int64_t rows, int embed, const float* w_k, const float* b_k, float* kbag,
                       mpo_stream_t stream) {
    MPO_CHECK(rows >= 1 && rows <= 0x7fffffff, "key projection: %lld rows out of range", (long long)rows);
    return mpo_launch_key_proj(hbag_bf16, w_k, b_k, kbag, (int)rows, embed, stream);
}
int mpo_nacagat_fwd_bagpass(const float* kbag, const int32_t* cu_rows, int n_slides, int embed, const float* qs2,
                            const float* tq, float* a_map, float* g_map, int n_q, int max_rows, const mpo_bag_plan* plan_,
                            mpo_stream_t stream) {
    const BagPlan plan = make_plan(plan_, n_slides, max_rows);
    if (int rc = check_plan(plan, n_slides)) return rc;
    return mpo_launch_bag_rowdot_gated(kbag, 1, cu_rows, n_slides, embed, qs2, tq, a_map, g_map, n_q, plan, stream);
}
int mpo_coattn_bwd_bagpass(const void* bag, int bag_dtype, const int32_t* cu_rows, int n_slides, int embed,
                           const float* qk2, const float* lse2, const float* dctx, const float* delta,
                           const float* d_attn_map, void* d_bag, float* part_dqk, int n_q, int max_rows,
                           const mpo_bag_plan* plan_, mpo_stream_t stream) {
    const BagPlan plan = make_plan(plan_, n_slides, max_rows);
    if (int rc = check_plan(plan, n_slides)) return rc;
    MPO_CHECK(delta, "coattn backward bag pass: delta is required here");
    return mpo_launch_coattn_bwd(bag, bag_dtype == MPO_F32, cu_rows, n_slides, embed, qk2, lse2, dctx, delta, nullptr, nullptr,
                                 d_attn_map, d_bag, part_dqk, nullptr, n_q, plan, 0.f, stream);
}

// ------------------------------------------------------------------------------------------- patch layer epilogue
int mpo_patch_epilogue_forward(void* h_bf16, const float* bias, int64_t rows, int cols, float drop_p, uint64_t seed,
                               uint64_t offset, const uint64_t* rng_epoch, mpo_stream_t stream) {
    return mpo_launch_bias_relu_dropout_bf16(h_bf16, bias, (size_t)rows, cols, drop_p, seed, offset,
                                             reinterpret_cast<const unsigned long long*>(rng_epoch), stream);
}
int mpo_colsum_bf16(const void* x_bf16, float* out, int64_t rows, int cols, mpo_stream_t stream) {
    return mpo_launch_colsum_bf16(x_bf16, out, (size_t)rows, cols, stream);
}
size_t mpo_patch_epilogue_backward_workspace_bytes(int64_t n, int cols) {
    return (size_t)mpo_relu_dropout_bwd_blocks((size_t)n, 1) * (size_t)cols * sizeof(float) + 256;
}
int mpo_patch_epilogue_backward(const void* h_bf16, const void* dy_bf16, void* g_bf16, int64_t n, int cols, float drop_p,
                                float* d_bias, void* workspace, size_t workspace_bytes, mpo_stream_t stream) {
    if (!d_bias) return mpo_launch_relu_dropout_bwd_bf16(h_bf16, dy_bf16, g_bf16, (size_t)n, drop_p, cols, nullptr, stream);
    MPO_CHECK(workspace && workspace_bytes >= mpo_patch_epilogue_backward_workspace_bytes(n, cols),
              "patch epilogue backward: workspace too small (%zu bytes)", workspace_bytes);
    float* part = static_cast<float*>(workspace);
    if (int rc = mpo_launch_relu_dropout_bwd_bf16(h_bf16, dy_bf16, g_bf16, (size_t)n, drop_p, cols, part, stream)) return rc;
    return mpo_launch_colsum(part, d_bias, mpo_relu_dropout_bwd_blocks((size_t)n, 1), cols, cols, 0, stream);
}

// dW_H = g^T X (models/mcat/mcat.py:24-29 backward): g = d(pre-activation) [rows, embed] bf16, X [rows, patch_dim] bf16
size_t mpo_patch_weight_grad_workspace_bytes(int embed, int patch_dim) {
    return mpo_patch_wgrad_partial_floats(embed, patch_dim) * sizeof(float) + 256;
}
int mpo_patch_weight_grad(const void* g_bf16, const void* patches_bf16, int64_t total_rows, int embed, int patch_dim,
                          float* d_weight, int workgroups, void* workspace, size_t workspace_bytes, mpo_stream_t stream) {
    MPO_CHECK(g_bf16 && patches_bf16 && d_weight, "patch weight gradient: null argument");
    MPO_CHECK(total_rows >= 1 && total_rows < (int64_t)1 << 31, "patch weight gradient: %lld rows", (long long)total_rows);
    Arena ws(workspace, workspace_bytes);
    float* part = ws.floats(mpo_patch_wgrad_partial_floats(embed, patch_dim));
    MPO_CHECK(part, "patch weight gradient: workspace too small (%zu bytes)", workspace_bytes);
    return mpo_launch_patch_wgrad(g_bf16, patches_bf16, total_rows, embed, patch_dim, part, d_weight, workgroups,
                                  static_cast<hipStream_t>(stream));
}

// ------------------------------------------------------------------------------------------- optimiser
int mpo_adam_step_flat(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                       float beta1, float beta2, float eps, float weight_decay, int step, const int32_t* step_dev,
                       mpo_stream_t stream) {
    MPO_CHECK(step >= 1 || step_dev, "adam: step counts from 1 (got %d)", step);
    return mpo_launch_adam_flat(params, grads, exp_avg, exp_avg_sq, (size_t)n, lr, beta1, beta2, eps, weight_decay,
                                step < 1 ? 1 : step, step_dev, stream);
}

// Verification hook: the small-row GEMMs have a branch-free body for regular products and a general body; both must
// give the same bits.  enabled = 0 routes every product through the general body.  Returns the previous setting.
int mpo_set_gemm_fast_path(int enabled) { return mpo_gemm_fast_path(enabled); }
int mpo_set_coattn_bwd_two_wave(int enabled) { return mpo_coattn_bwd8_enable(enabled); }
// verification hook: 0 = fp32 bags take the general (matrix-pipe) K1 backward
int mpo_set_coattn_bwd_f32_vector(int enabled) { return mpo_coattn_bwd_f32_enable(enabled); }
// verification hook: 0 = K2's backward reads K twice (bag_colacc_gated, then bag_outer_gated, both on the matrix pipe) as in rounds 1-2
int mpo_set_nacagat_one_pass_key_grad(int enabled) {
    const int was = g_k2_one_pass_key ? 1 : 0;
    g_k2_one_pass_key = enabled != 0;
    return was;
}

// The two device-resident per-step counters of a captured training step, bumped by one launch.
int mpo_step_counters_bump(uint64_t* rng_epoch, int32_t* adam_step, mpo_stream_t stream) {
    MPO_CHECK(rng_epoch || adam_step, "step counters: nothing to bump");
    return mpo_launch_counters_bump(reinterpret_cast<unsigned long long*>(rng_epoch), adam_step, static_cast<hipStream_t>(stream));
}

}  // extern "C"
