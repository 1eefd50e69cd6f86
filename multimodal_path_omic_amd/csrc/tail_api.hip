// C-ABI entries of the 6 x d token tail (declared in include/mpo_hip.h): K3 Contextual Attention Gate,
// K4 set-Transformer encoder, K5 gated attention-MIL pooling, K6 fusion + survival head.
// Each entry is a fixed sequence of launches (GEMMs with fused epilogues/gates + the kernels of tail.hip)
// on the caller's stream, in caller-provided buffers.
#include <type_traits>
#include <vector>

#include "../../include/mpo_hip.h"
#include "mpo_common.h"
#include "mpo_kernels.h"

namespace {

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
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
struct Sizer {                         // mirrors Arena to size a workspace
    size_t off = 0;
    void floats(size_t n) { off = align_up(off, 256) + n * 4; }
};
// carve the saved buffer sequentially (floats, 64-float aligned so float4 accesses stay aligned)
struct Carver {
    float* p;
    explicit Carver(float* base) : p(base) {}
    float* take(size_t n) { float* r = p; p += (n + 63) / 64 * 64; return r; }
};
struct CarveSizer {
    size_t n = 0;
    float* take(size_t k) { n += (k + 63) / 64 * 64; return nullptr; }
};

inline DropSpec stream_of(float p, uint64_t seed, uint64_t base, uint64_t stride, int k, const uint64_t* epoch) {
    DropSpec d;
    d.p = p; d.seed = seed; d.off = base + stride * (uint64_t)k;
    d.epoch = reinterpret_cast<const unsigned long long*>(epoch);
    return d;
}
inline GateSpec gate(const float* g, int mode, float p = 0.f) {
    GateSpec s;
    s.g = g; s.mode = mode; s.p = p;
    return s;
}
inline GateSpec gate_rng(DropSpec d) {
    GateSpec s;
    s.mode = d.p > 0.f ? MPO_GATE_RNG : MPO_GATE_NONE; s.p = d.p; s.seed = d.seed; s.off = d.off; s.epoch = d.epoch;
    return s;
}

#define RC(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)
// a layer's dx (layout 2) and dW (layout 0) products share one mixed launch
#define PAIR(dxa, dwa) RC(mpo_linear_bwd_pair((dxa), (dwa), stream))

// ------------------------------------------------------------------------------------------- K4 encoder
enum { P_INW, P_INB, P_OUTW, P_OUTB, P_L1W, P_L1B, P_L2W, P_L2B, P_N1W, P_N1B, P_N2W, P_N2B, P_PER_LAYER };

struct EncLayerSaved {
    float *qkv, *psave, *o, *s1, *st1, *x1, *f, *s2, *st2, *x2;
};
template <typename C>
void enc_carve(C& c, EncLayerSaved* out, int B, int T, int d, int ff, int H) {
    const size_t R = (size_t)B * T;
    EncLayerSaved s;
    s.qkv = c.take(R * 3 * d);
    // attention state: the two T x T probability matrices per head of the token tail, or (T = the rows of a bag) one
    // log-sum-exp per head and row (+ the operand forms of the three-term bf16 path)
    s.psave = c.take(T <= kSmallAttnMaxT ? (size_t)B * 2 * H * T * T : mpo_bag_sa_saved_floats(B, T, d, H));
    s.o = c.take(R * d);
    s.s1 = c.take(R * d); s.st1 = c.take(2 * R); s.x1 = c.take(R * d); s.f = c.take(R * ff);
    s.s2 = c.take(R * d); s.st2 = c.take(2 * R); s.x2 = c.take(R * d);
    if (out) *out = s;
}
inline uint64_t enc_stream_stride(int B, int T, int d, int ff) {
    const uint64_t R = (uint64_t)B * T;
    const uint64_t m = R * (uint64_t)(ff > 3 * d ? ff : 3 * d);
    return m / 4 + 2;
}

}  // namespace

extern "C" {

size_t mpo_encoder_saved_floats(int n_slides, int T, int d, int ff, int heads, int layers) {
    CarveSizer c;
    for (int l = 0; l < layers; ++l) enc_carve(c, (EncLayerSaved*)nullptr, n_slides, T, d, ff, heads);
    return c.n;
}
size_t mpo_encoder_workspace_bytes(int n_slides, int T, int d, int ff) {
    const size_t R = (size_t)n_slides * T;
    Sizer s;
    for (int l = 0; l < 8; ++l) {                          // one buffer set per layer (max 8 layers)
        s.floats(R * d); s.floats(R * ff); s.floats(R * d); s.floats(R * d); s.floats(R * d); s.floats(R * 3 * d); s.floats(R * d);
        // long token axes: scratch of the bag self-attention backward (heads are not known here: 8 of width 32 is the
        // one geometry with more than the per-head row sums, and d floats per row covers it)
        if (T > kSmallAttnMaxT) s.floats(mpo_bag_sa_bwd_floats(n_slides, T, d, d / 32 > 0 ? d / 32 : 1) + (size_t)R * d);
    }
    return s.off + 256;
}
uint64_t mpo_encoder_rng_span(int n_slides, int T, int d, int ff, int layers) {
    return enc_stream_stride(n_slides, T, d, ff) * 4 * (uint64_t)layers;
}

// ---- branch batching: n_branches independent modules of IDENTICAL geometry (the path and the omic set-Transformer /
// pooling head of one model) run as ONE launch sequence.  Activations are [branch][rows][width]; every GEMM becomes a
// grouped launch with one member per branch; LayerNorm picks its parameters by row.  The dependent launch chain of
// the token tail is latency-bound, so the second branch rides along for free.
namespace {
// dropout stream of branch br inside one stream slot: branches are rows_x_width elements apart
inline DropSpec drop_br(DropSpec d, int br, size_t elems_per_branch) { d.off += (uint64_t)br * ((elems_per_branch + 3) / 4); return d; }
inline GateSpec gate_rng_br(DropSpec d, int br, size_t elems_per_branch) { return gate_rng(drop_br(d, br, elems_per_branch)); }
// any number of same-layout products, eight members to a launch
inline int launch_in_groups(const std::vector<GemmArgs>& list, int a_kc, int b_kc, hipStream_t stream) {
    for (size_t i0 = 0; i0 < list.size(); i0 += 8) {
        GemmGroup grp;
        for (size_t i = i0; i < list.size() && i < i0 + 8; ++i) grp.g[grp.n++] = list[i];
        if (int rc = mpo_launch_gemm_group(grp, a_kc, b_kc, stream)) return rc;
    }
    return 0;
}
struct GroupBuilder {
    GemmGroup g;
    int add(const GemmArgs& a) {
        if (g.n >= 8) { mpo_set_error("grouped gemm: more than 8 members"); return 1; }
        g.g[g.n++] = a;
        return 0;
    }
    int launch(hipStream_t s) { return g.n ? mpo_launch_gemm_mixed(g, s) : 0; }
};
}  // namespace

// nn.TransformerEncoder (post-norm layers, ReLU FFN, no final norm): models/mcat/mcat.py:51-53,101-102;
// layer arithmetic torch/nn/modules/transformer.py:661 (norm_first=False).
int mpo_encoder_forward(const float* x, int n_branches, int n_slides, int T, int d, int ff, int heads, int layers,
                        const float* const* params, float drop_p, uint64_t seed, uint64_t offset, const uint64_t* rng_epoch,
                        float* y, float* saved, mpo_stream_t stream) {
    MPO_CHECK(n_slides >= 1 && layers >= 1 && d % heads == 0, "encoder: bad geometry (slides %d, layers %d, d %d, heads %d)",
              n_slides, layers, d, heads);
    MPO_CHECK(n_branches >= 1 && n_branches <= kMaxBranches, "encoder: 1..%d branches (got %d)", kMaxBranches, n_branches);
    const int NB = n_branches, R = n_slides * T, RT = NB * R, BT = NB * n_slides;
    const uint64_t stride = enc_stream_stride(BT, T, d, ff);
    Carver c(saved);
    const float* in = x;
    for (int l = 0; l < layers; ++l) {
        EncLayerSaved S;
        enc_carve(c, &S, BT, T, d, ff, heads);
        const uint64_t base = offset + stride * 4 * (uint64_t)l;
        const DropSpec d0 = stream_of(drop_p, seed, base, stride, 0, rng_epoch), d1 = stream_of(drop_p, seed, base, stride, 1, rng_epoch),
                       d2 = stream_of(drop_p, seed, base, stride, 2, rng_epoch), d3 = stream_of(drop_p, seed, base, stride, 3, rng_epoch);
        float* out = (l == layers - 1) ? y : S.x2;
        auto P = [&](int br, int i) { return params[((size_t)br * layers + l) * P_PER_LAYER + i]; };
        LnBranches n1, n2;
        n1.n = n2.n = NB; n1.rows_per_branch = n2.rows_per_branch = R;
        for (int br = 0; br < NB; ++br) { n1.w[br] = P(br, P_N1W); n1.b[br] = P(br, P_N1B); n2.w[br] = P(br, P_N2W); n2.b[br] = P(br, P_N2B); }
        const size_t Rd = (size_t)R * d, Rf = (size_t)R * ff, Rq = (size_t)R * 3 * d;
        GroupBuilder q, o, f1, f2;
        for (int br = 0; br < NB; ++br) {
            RC(q.add(mpo_args_fwd(in + br * Rd, P(br, P_INW), P(br, P_INB), S.qkv + br * Rq, R, d, 3 * d, 1.0f, MPO_ACT_NONE)));
            RC(o.add(mpo_args_fwd(S.o + br * Rd, P(br, P_OUTW), P(br, P_OUTB), S.s1 + br * Rd, R, d, d, 1.0f, MPO_ACT_NONE,
                                  in + br * Rd, drop_br(d1, br, Rd))));
            RC(f1.add(mpo_args_fwd(S.x1 + br * Rd, P(br, P_L1W), P(br, P_L1B), S.f + br * Rf, R, d, ff, 1.0f, MPO_ACT_RELU, nullptr,
                                   drop_br(d2, br, Rf))));
            RC(f2.add(mpo_args_fwd(S.f + br * Rf, P(br, P_L2W), P(br, P_L2B), S.s2 + br * Rd, R, ff, d, 1.0f, MPO_ACT_NONE,
                                   S.x1 + br * Rd, drop_br(d3, br, Rd))));
        }
        RC(q.launch(stream));
        if (T <= kSmallAttnMaxT) RC(mpo_launch_mha_small_fwd(S.qkv, S.o, S.psave, BT, T, d, heads, d0.p, d0.seed, d0.off, d0.epoch, stream));
        else RC(mpo_launch_bag_sa_fwd(S.qkv, BT, T, d, heads, d0.p, d0.seed, d0.off, d0.epoch, S.o, S.psave, nullptr, stream));
        RC(o.launch(stream));
        RC(mpo_launch_ln_fwd_br(S.s1, n1, S.x1, S.st1, RT, d, 1e-5f, stream));
        RC(f1.launch(stream));
        RC(f2.launch(stream));
        RC(mpo_launch_ln_fwd_br(S.s2, n2, out, S.st2, RT, d, 1e-5f, stream));
        in = out;
    }
    return 0;
}

int mpo_encoder_backward(const float* x, int n_branches, int n_slides, int T, int d, int ff, int heads, int layers,
                         const float* const* params, float drop_p, uint64_t seed, uint64_t offset, const uint64_t* rng_epoch,
                         const float* saved, const float* dy, float* dx, float* const* grads,
                         void* workspace, size_t workspace_bytes, mpo_stream_t stream) {
    MPO_CHECK(n_branches >= 1 && n_branches <= kMaxBranches, "encoder: 1..%d branches (got %d)", kMaxBranches, n_branches);
    const int NB = n_branches, R = n_slides * T, RT = NB * R, BT = NB * n_slides;
    const uint64_t stride = enc_stream_stride(BT, T, d, ff);
    MPO_CHECK(layers <= 8, "encoder: at most 8 layers (got %d)", layers);
    EncLayerSaved S[8];
    Carver c(const_cast<float*>(saved));
    for (int l = 0; l < layers; ++l) enc_carve(c, &S[l], BT, T, d, ff, heads);
    Arena ws(workspace, workspace_bytes);
    // each layer's dx and dW products share one grouped launch (every layer has its own buffer set)
    const float* dcur = dy;
    const size_t Rd = (size_t)R * d, Rf = (size_t)R * ff, Rq = (size_t)R * 3 * d;
    for (int l = layers - 1; l >= 0; --l) {
        float* ds2 = ws.floats((size_t)RT * d);
        float* df = ws.floats((size_t)RT * ff);
        float* dx1 = ws.floats((size_t)RT * d);
        float* ds1 = ws.floats((size_t)RT * d);
        float* dob = ws.floats((size_t)RT * d);
        float* dqkv = ws.floats((size_t)RT * 3 * d);
        float* din = l == 0 ? dx : ws.floats((size_t)RT * d);
        MPO_CHECK(ds2 && df && dx1 && ds1 && dob && dqkv && din, "encoder backward: workspace too small (%zu bytes)", workspace_bytes);
        auto P = [&](int br, int i) { return params[((size_t)br * layers + l) * P_PER_LAYER + i]; };
        auto G = [&](int br, int i) { return grads[((size_t)br * layers + l) * P_PER_LAYER + i]; };
        const float* in = l == 0 ? x : S[l - 1].x2;
        const uint64_t base = offset + stride * 4 * (uint64_t)l;
        const DropSpec d1 = stream_of(drop_p, seed, base, stride, 1, rng_epoch), d3 = stream_of(drop_p, seed, base, stride, 3, rng_epoch);
        LnBranches n1, n2;
        n1.n = n2.n = NB; n1.rows_per_branch = n2.rows_per_branch = R;
        for (int br = 0; br < NB; ++br) {
            n1.w[br] = P(br, P_N1W); n1.dw[br] = G(br, P_N1W); n1.db[br] = G(br, P_N1B);
            n2.w[br] = P(br, P_N2W); n2.dw[br] = G(br, P_N2W); n2.db[br] = G(br, P_N2B);
        }
        // one grouped launch per product pair: members (dx_br, dW_br) for every branch
        auto pairs = [&](auto&& mk_dx, auto&& mk_dw) -> int {
            GroupBuilder main_g;
            for (int br = 0; br < NB; ++br) {
                RC(main_g.add(mk_dx(br)));
                RC(main_g.add(mk_dw(br)));
            }
            RC(main_g.launch(stream));
            return 0;
        };
        // x2 = LN2(s2)
        RC(mpo_launch_ln_bwd_br(dcur, S[l].s2, S[l].st2, n2, ds2, RT, d, 0, 3, stream));
        // s2 = x1 + drop3(f W2^T + b2)
        RC(pairs([&](int br) { return mpo_args_bwd_input(ds2 + br * Rd, P(br, P_L2W), df + br * Rf, R, ff, d, 1.0f, 0, gate_rng_br(d3, br, Rd)); },
                 [&](int br) { return mpo_args_bwd_weight(ds2 + br * Rd, S[l].f + br * Rf, G(br, P_L2W), G(br, P_L2B), R, ff, d, 1.0f, gate_rng_br(d3, br, Rd)); }));
        // f = drop2(relu(x1 W1^T + b1));  dx1 = ds2 + (df*gate) W1
        RC(pairs([&](int br) {
                     GemmArgs g;
                     g.A = df + br * Rf; g.B = P(br, P_L1W); g.C = dx1 + br * Rd; g.residual = ds2 + br * Rd;
                     g.M = R; g.N = d; g.K = ff; g.lda = ff; g.ldb = d; g.ldc = d;
                     g.gate = S[l].f + br * Rf; g.gate_mode = MPO_GATE_RELU; g.gate_p = drop_p; g.layout = 2;
                     return g;
                 },
                 [&](int br) { return mpo_args_bwd_weight(df + br * Rf, S[l].x1 + br * Rd, G(br, P_L1W), G(br, P_L1B), R, d, ff, 1.0f,
                                                          gate(S[l].f + br * Rf, MPO_GATE_RELU, drop_p)); }));
        // x1 = LN1(s1)
        RC(mpo_launch_ln_bwd_br(dx1, S[l].s1, S[l].st1, n1, ds1, RT, d, 0, 3, stream));
        // s1 = in + drop1(o W_o^T + b_o)
        RC(pairs([&](int br) { return mpo_args_bwd_input(ds1 + br * Rd, P(br, P_OUTW), dob + br * Rd, R, d, d, 1.0f, 0, gate_rng_br(d1, br, Rd)); },
                 [&](int br) { return mpo_args_bwd_weight(ds1 + br * Rd, S[l].o + br * Rd, G(br, P_OUTW), G(br, P_OUTB), R, d, d, 1.0f, gate_rng_br(d1, br, Rd)); }));
        if (T <= kSmallAttnMaxT) {
            RC(mpo_launch_mha_small_bwd(S[l].qkv, S[l].psave, dob, dqkv, BT, T, d, heads, stream));
        } else {
            float* sa_ws = ws.floats(mpo_bag_sa_bwd_floats(BT, T, d, heads));
            MPO_CHECK(sa_ws, "encoder backward: workspace too small for the attention scratch (%zu bytes)", workspace_bytes);
            const DropSpec d0 = stream_of(drop_p, seed, base, stride, 0, rng_epoch);
            RC(mpo_launch_bag_sa_bwd(S[l].qkv, S[l].o, S[l].psave, dob, BT, T, d, heads, d0.p, d0.seed, d0.off, d0.epoch, dqkv, sa_ws, stream));
        }
        // qkv = in W_in^T + b_in;  d_in = ds1 + dqkv W_in
        RC(pairs([&](int br) {
                     GemmArgs g;
                     g.A = dqkv + br * Rq; g.B = P(br, P_INW); g.C = din + br * Rd; g.residual = ds1 + br * Rd;
                     g.M = R; g.N = d; g.K = 3 * d; g.lda = 3 * d; g.ldb = d; g.ldc = d; g.layout = 2;
                     return g;
                 },
                 [&](int br) { return mpo_args_bwd_weight(dqkv + br * Rq, in + br * Rd, G(br, P_INW), G(br, P_INB), R, d, 3 * d, 1.0f); }));
        dcur = din;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------- f3 bag self-attention
// The attention core of nn.MultiheadAttention over the M rows of a bag (models/ge_nacagat/ge_nacagat.py:27,49): the packed
// projections qkv [n_bags][M][3 d] come from the caller's in_proj product, the out_proj follows on the caller's side.
size_t mpo_bag_self_attention_saved_floats(int n_bags, int M, int d, int heads) { return mpo_bag_sa_saved_floats(n_bags, M, d, heads); }
size_t mpo_bag_self_attention_workspace_bytes(int n_bags, int M, int d, int heads) { return mpo_bag_sa_bwd_floats(n_bags, M, d, heads) * sizeof(float); }
int mpo_set_bag_self_attention_bf16x3(int enabled) { return mpo_bag_sa_set_bf16x3(enabled); }
int mpo_bag_self_attention_forward(const float* qkv, int n_bags, int M, int d, int heads, float drop_p, uint64_t seed, uint64_t offset,
                                   const uint64_t* rng_epoch, float* out, float* saved, float* attn_map, mpo_stream_t stream) {
    MPO_CHECK(qkv && out && saved, "bag self-attention: NULL buffer");
    return mpo_launch_bag_sa_fwd(qkv, n_bags, M, d, heads, drop_p, seed, offset, (const unsigned long long*)rng_epoch, out, saved,
                                 attn_map, (hipStream_t)stream);
}
int mpo_bag_self_attention_backward(const float* qkv, const float* out, const float* saved, const float* d_out, int n_bags, int M, int d,
                                    int heads, float drop_p, uint64_t seed, uint64_t offset, const uint64_t* rng_epoch, float* d_qkv,
                                    void* workspace, size_t workspace_bytes, mpo_stream_t stream) {
    MPO_CHECK(qkv && out && saved && d_out && d_qkv, "bag self-attention backward: NULL buffer");
    MPO_CHECK(workspace && workspace_bytes >= mpo_bag_self_attention_workspace_bytes(n_bags, M, d, heads),
              "bag self-attention backward: workspace too small (%zu bytes)", workspace_bytes);
    return mpo_launch_bag_sa_bwd(qkv, out, saved, d_out, n_bags, M, d, heads, drop_p, seed, offset, (const unsigned long long*)rng_epoch,
                                 d_qkv, static_cast<float*>(workspace), (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------- K5 gated pooling
// params: attention_a.0.weight, .bias, attention_b.0.weight, .bias, attention_c.weight, .bias, rho.0.weight, .bias
// saved: a [R,d] | b [R,d] | ab [R,d] | w [R] | hpool [B,d]      (R, B over all branches)
size_t mpo_gated_pool_saved_floats(int n_slides, int L, int d) {
    CarveSizer c;
    const size_t R = (size_t)n_slides * L;
    c.take(R * d); c.take(R * d); c.take(R * d); c.take(R); c.take((size_t)n_slides * d);
    return c.n;
}
size_t mpo_gated_pool_workspace_bytes(int n_slides, int L, int d) {
    const size_t R = (size_t)n_slides * L;
    Sizer s;
    s.floats((size_t)n_slides * d); s.floats(R); s.floats(R * d); s.floats(R * d); s.floats(R * d);
    return s.off + 256;
}
uint64_t mpo_gated_pool_rng_span(int n_slides, int L, int d) { return 3 * ((uint64_t)n_slides * L * d / 4 + 2 + kMaxBranches); }

// AttentionNetGated (models/blocks.py:13-48) + the pooling idiom of models/mcat/mcat.py:105-109:
// scores = W_c[drop(tanh(W_a x)) * drop(sigmoid(W_b x))] + b_c; h = drop(relu(W_rho (softmax_L(scores) x) + b_rho))
// The pooling head's scorer (models/blocks.py:42-48: attention_c on a (.) b) runs inside the pooling kernels for the token
// tail (<= 64 rows per slide, <= 2 branches); long bags (row f3: L = M rows) keep the grid-wide pooling kernels and the
// scorer as a many-row product.
static bool pool_fuses_scorer(int n_branches, int L, int d) { return n_branches <= 2 && L <= 64 && d <= 1024; }

int mpo_gated_pool_forward(const float* x, int n_branches, int n_slides, int L, int d, const float* const* params,
                           float head_drop_p, float rho_drop_p, uint64_t seed, uint64_t offset, const uint64_t* rng_epoch,
                           float* scores, float* h, int h_interleaved, float* saved, mpo_stream_t stream) {
    MPO_CHECK(n_branches >= 1 && n_branches <= kMaxBranches, "gated pool: 1..%d branches (got %d)", kMaxBranches, n_branches);
    MPO_CHECK(!h_interleaved || (d & 3) == 0, "gated pool: interleaved h needs d %% 4 == 0 (got %d)", d);
    const int NB = n_branches, R = n_slides * L, RT = NB * R, BT = NB * n_slides;
    const uint64_t stride = (uint64_t)RT * d / 4 + 2 + kMaxBranches;
    const size_t Rd = (size_t)R * d, Bd = (size_t)n_slides * d;
    Carver c(saved);
    float* a = c.take((size_t)RT * d); float* b = c.take((size_t)RT * d); float* ab = c.take((size_t)RT * d);
    float* w = c.take(RT); float* hpool = c.take((size_t)BT * d);
    auto P = [&](int br, int i) { return params[br * 8 + i]; };
    const DropSpec s0 = stream_of(head_drop_p, seed, offset, stride, 0, rng_epoch), s1 = stream_of(head_drop_p, seed, offset, stride, 1, rng_epoch),
                   s2 = stream_of(rho_drop_p, seed, offset, stride, 2, rng_epoch);
    const bool fuse_ab = pool_fuses_scorer(NB, L, d);
    GroupBuilder gab, gsc, grho;
    for (int br = 0; br < NB; ++br) {
        RC(gab.add(mpo_args_fwd(x + br * Rd, P(br, 0), P(br, 1), a + br * Rd, R, d, d, 1.0f, MPO_ACT_TANH, nullptr, drop_br(s0, br, Rd))));
        RC(gab.add(mpo_args_fwd(x + br * Rd, P(br, 2), P(br, 3), b + br * Rd, R, d, d, 1.0f, MPO_ACT_SIGMOID, nullptr, drop_br(s1, br, Rd))));
        if (!fuse_ab)
            RC(gsc.add(mpo_args_fwd(ab + br * Rd, P(br, 4), P(br, 5), scores + (size_t)br * R, R, d, 1, 1.0f, MPO_ACT_NONE)));
        if (h_interleaved) {
            // h [n_slides][branch][d] = the concatenated [h_path | h_omic] rows the fusion layer reads (no transposing copy);
            // one dropout stream over the interleaved rows: branch br starts d / 4 counters in
            DropSpec sd = s2;
            sd.off += (uint64_t)br * (d / 4);
            GemmArgs m = mpo_args_fwd(hpool + br * Bd, P(br, 6), P(br, 7), h + (size_t)br * d, n_slides, d, d, 1.0f, MPO_ACT_RELU,
                                      nullptr, sd);
            m.ldc = NB * d;
            RC(grho.add(m));
        } else {
            RC(grho.add(mpo_args_fwd(hpool + br * Bd, P(br, 6), P(br, 7), h + br * Bd, n_slides, d, d, 1.0f, MPO_ACT_RELU, nullptr,
                                     drop_br(s2, br, Bd))));
        }
    }
    RC(gab.launch(stream));
    if (fuse_ab) {
        // scores = (a (.) b) W_c^T + b_c, softmax over the slide's rows and the weighted sum in ONE launch
        PoolScorer ps;
        for (int br = 0; br < NB; ++br) { ps.wc[br] = P(br, 4); ps.bc[br] = P(br, 5); }
        ps.n_slides = n_slides;
        RC(mpo_launch_pool_score_fwd(a, b, x, ps, scores, w, hpool, BT, L, d, stream));
    } else {
        RC(mpo_launch_ew_mul(a, b, ab, RT * d, stream));
        RC(gsc.launch(stream));
        RC(mpo_launch_pool_fwd(scores, x, w, hpool, BT, L, d, stream));
    }
    RC(grho.launch(stream));
    return 0;
}

int mpo_gated_pool_backward(const float* x, int n_branches, int n_slides, int L, int d, const float* const* params,
                            float head_drop_p, float rho_drop_p, const float* saved, const float* h,
                            const float* dh, int h_interleaved, const float* d_scores_ext, float* dx, float* const* grads,
                            void* workspace, size_t workspace_bytes, mpo_stream_t stream) {
    MPO_CHECK(n_branches >= 1 && n_branches <= 2, "gated pool backward: 1..2 branches (got %d)", n_branches);
    const int NB = n_branches, R = n_slides * L, RT = NB * R, BT = NB * n_slides;
    const size_t Rd = (size_t)R * d, Bd = (size_t)n_slides * d;
    Carver c(const_cast<float*>(saved));
    const float* a = c.take((size_t)RT * d); const float* b = c.take((size_t)RT * d); const float* ab = c.take((size_t)RT * d);
    const float* w = c.take(RT); const float* hpool = c.take((size_t)BT * d);
    Arena ws(workspace, workspace_bytes);
    float* dhpool = ws.floats((size_t)BT * d);
    float* dscores = ws.floats(RT);
    float* dab = ws.floats((size_t)RT * d);
    float* da = ws.floats((size_t)RT * d);
    float* db = ws.floats((size_t)RT * d);
    MPO_CHECK(dhpool && dscores && dab && da && db, "gated pool backward: workspace too small (%zu bytes)", workspace_bytes);
    auto P = [&](int br, int i) { return params[br * 8 + i]; };
    auto G = [&](int br, int i) { return grads[br * 8 + i]; };
    auto pairs = [&](auto&& mk_dx, auto&& mk_dw) -> int {
        GroupBuilder main_g;
        for (int br = 0; br < NB; ++br) {
            RC(main_g.add(mk_dx(br)));
            RC(main_g.add(mk_dw(br)));
        }
        RC(main_g.launch(stream));
        return 0;
    };
    // h = drop(relu(hpool W_rho^T + b_rho));  h / dh either [branch][n_slides][d] or interleaved [n_slides][branch][d]
    const size_t h_off = h_interleaved ? (size_t)d : Bd;
    const int h_ld = h_interleaved ? NB * d : d;
    RC(pairs([&](int br) { GemmArgs m = mpo_args_bwd_input(dh + br * h_off, P(br, 6), dhpool + br * Bd, n_slides, d, d, 1.0f, 0, gate(h + br * h_off, MPO_GATE_RELU, rho_drop_p)); m.lda = h_ld; return m; },
             [&](int br) { GemmArgs m = mpo_args_bwd_weight(dh + br * h_off, hpool + br * Bd, G(br, 6), G(br, 7), n_slides, d, d, 1.0f, gate(h + br * h_off, MPO_GATE_RELU, rho_drop_p)); m.lda = h_ld; return m; }));
    if (pool_fuses_scorer(NB, L, d)) {
        // pooling + scorer backward in ONE launch: d_scores, dx (pooling part), da, db (dW_c, db_c: below)
        PoolScorer ps;
        for (int br = 0; br < NB; ++br) { ps.wc[br] = P(br, 4); ps.bc[br] = P(br, 5); }
        ps.n_slides = n_slides;
        RC(mpo_launch_pool_score_bwd(dhpool, x, w, d_scores_ext, a, b, ps, dscores, dx, da, db, BT, L, d, stream));
    } else {
        RC(mpo_launch_pool_bwd(dhpool, x, w, d_scores_ext, dscores, dx, BT, L, d, stream));
        // scores = ab W_c^T + b_c
        RC(pairs([&](int br) { return mpo_args_bwd_input(dscores + (size_t)br * R, P(br, 4), dab + br * Rd, R, d, 1, 1.0f, 0); },
                 [&](int br) { return mpo_args_bwd_weight(dscores + (size_t)br * R, ab + br * Rd, G(br, 4), G(br, 5), R, d, 1, 1.0f); }));
        RC(mpo_launch_ew_mul2(dab, b, a, da, db, RT * d, stream));                 // da = dab * b,  db = dab * a
    }
    // a = drop(tanh(x W_a^T + b_a)), b = drop(sigmoid(x W_b^T + b_b)): dx accumulates both products, the second follows alone
    {
        GroupBuilder first, second;
        for (int br = 0; br < NB; ++br) {
            const GateSpec ga = gate(a + br * Rd, MPO_GATE_TANH, head_drop_p), gb = gate(b + br * Rd, MPO_GATE_SIGMOID, head_drop_p);
            RC(first.add(mpo_args_bwd_input(da + br * Rd, P(br, 0), dx + br * Rd, R, d, d, 1.0f, 1, ga)));
            RC(second.add(mpo_args_bwd_input(db + br * Rd, P(br, 2), dx + br * Rd, R, d, d, 1.0f, 1, gb)));
            RC(first.add(mpo_args_bwd_weight(da + br * Rd, x + br * Rd, G(br, 0), G(br, 1), R, d, d, 1.0f, ga)));
            RC(first.add(mpo_args_bwd_weight(db + br * Rd, x + br * Rd, G(br, 2), G(br, 3), R, d, d, 1.0f, gb)));
            if (pool_fuses_scorer(NB, L, d)) {
                // the scorer's own gradients ride in the second launch:
                //   dW_c^T [d x 1] = (a (.) b)^T d_scores: a as the (row-contiguous) A operand with the value gate "multiply by b";
                //   db_c = sum(d_scores): the bias-gradient side output of a 1 x 1 weight-gradient product (its dW goes to scratch)
                const float* ds = dscores + (size_t)br * R;
                GemmArgs m3;
                m3.A = a + br * Rd; m3.lda = d;                    // A(m = column j, k = row r) = a[r][j]
                m3.gate = b + br * Rd; m3.gate_mode = MPO_GATE_MUL;
                m3.B = ds; m3.ldb = 1;                             // B(n = 0, k = r) = d_scores[r]
                m3.C = G(br, 4); m3.ldc = 1;
                m3.M = d; m3.N = 1; m3.K = R; m3.layout = 0;
                RC(second.add(m3));
                RC(second.add(mpo_args_bwd_weight(ds, ds, dab + br, G(br, 5), R, 1, 1, 1.0f)));
            }
        }
        RC(first.launch(stream));
        RC(second.launch(stream));
    }
    return 0;
}

// ------------------------------------------------------------------------------------------- K6 fusion + head
// params: fusion_layer.0.weight, .bias, fusion_layer.2.weight, .bias, classifier.weight, .bias
// saved: z1 [B,hidden] | z2 [B,dout] | logits [B,C]
size_t mpo_fusion_head_saved_floats(int n_slides, int hidden, int dout, int n_classes) {
    CarveSizer c;
    c.take((size_t)n_slides * hidden); c.take((size_t)n_slides * dout); c.take((size_t)n_slides * n_classes);
    return c.n;
}
size_t mpo_fusion_head_workspace_bytes(int n_slides, int hidden, int dout, int n_classes) {
    Sizer s;
    s.floats((size_t)n_slides * n_classes); s.floats((size_t)n_slides * dout); s.floats((size_t)n_slides * hidden);
    return s.off + 256;
}

// ConcatFusion (models/fusion.py:7-19) on the concatenated [h_path | h_omic], classifier and the
// survival head of models/mcat/mcat.py:126-138.
int mpo_fusion_head_forward(const float* hcat, int n_slides, int din, int hidden, int dout, int n_classes,
                            const float* const* P, float* hazards, float* survs, float* y, float* saved,
                            mpo_stream_t stream) {
    Carver c(saved);
    float* z1 = c.take((size_t)n_slides * hidden); float* z2 = c.take((size_t)n_slides * dout);
    float* logits = c.take((size_t)n_slides * n_classes);
    RC(mpo_linear_fwd(hcat, P[0], P[1], z1, n_slides, din, hidden, 1.0f, MPO_ACT_RELU, stream));
    RC(mpo_linear_fwd(z1, P[2], P[3], z2, n_slides, hidden, dout, 1.0f, MPO_ACT_RELU, stream));
    RC(mpo_linear_fwd(z2, P[4], P[5], logits, n_slides, dout, n_classes, 1.0f, MPO_ACT_NONE, stream));
    RC(mpo_launch_head_fwd(logits, hazards, survs, y, n_slides, n_classes, stream));
    return 0;
}

int mpo_fusion_head_backward(const float* hcat, int n_slides, int din, int hidden, int dout, int n_classes,
                             const float* const* P, const float* saved, const float* hazards, const float* survs,
                             const float* y, const float* d_hazards, const float* d_survs, const float* d_y,
                             float* d_hcat, float* const* G, void* workspace, size_t workspace_bytes,
                             mpo_stream_t stream) {
    Carver c(const_cast<float*>(saved));
    const float* z1 = c.take((size_t)n_slides * hidden); const float* z2 = c.take((size_t)n_slides * dout);
    Arena ws(workspace, workspace_bytes);
    float* dlogits = ws.floats((size_t)n_slides * n_classes);
    float* dz2 = ws.floats((size_t)n_slides * dout);
    float* dz1 = ws.floats((size_t)n_slides * hidden);
    MPO_CHECK(dlogits && dz2 && dz1, "fusion head backward: workspace too small (%zu bytes)", workspace_bytes);
    RC(mpo_launch_head_bwd(hazards, survs, y, d_hazards, d_survs, d_y, dlogits, n_slides, n_classes, stream));
    PAIR(mpo_args_bwd_input(dlogits, P[4], dz2, n_slides, dout, n_classes, 1.0f, 0),
         mpo_args_bwd_weight(dlogits, z2, G[4], G[5], n_slides, dout, n_classes, 1.0f));
    PAIR(mpo_args_bwd_input(dz2, P[2], dz1, n_slides, hidden, dout, 1.0f, 0, gate(z2, MPO_GATE_RELU)),
         mpo_args_bwd_weight(dz2, z1, G[2], G[3], n_slides, hidden, dout, 1.0f, gate(z2, MPO_GATE_RELU)));
    PAIR(mpo_args_bwd_input(dz1, P[0], d_hcat, n_slides, din, hidden, 1.0f, 0, gate(z1, MPO_GATE_RELU)),
         mpo_args_bwd_weight(dz1, hcat, G[0], G[1], n_slides, din, hidden, 1.0f, gate(z1, MPO_GATE_RELU)));
    return 0;
}

// Training-step form: the same MLP, then head + 'ces' loss + their backward in ONE launch (head_loss_kernel); the
// gradient w.r.t. the logits is kept in `saved` and the backward starts at the classifier products.
// saved: z1 [B,hidden] | z2 [B,dout] | logits [B,C] | d_logits [B,C]
size_t mpo_fusion_head_loss_saved_floats(int n_slides, int hidden, int dout, int n_classes) {
    CarveSizer c;
    c.take((size_t)n_slides * hidden); c.take((size_t)n_slides * dout); c.take((size_t)n_slides * n_classes);
    c.take((size_t)n_slides * n_classes);
    return c.n;
}
int mpo_fusion_head_loss_forward(const float* hcat, int n_slides, int din, int hidden, int dout, int n_classes,
                                 const float* const* P, const int64_t* label, const float* censorship,
                                 const float* slide_weight, float alpha, float eps, float* hazards, float* survs, float* y,
                                 float* loss, float* risk, float* saved, mpo_stream_t stream) {
    MPO_CHECK(hcat && P && label && censorship && slide_weight && hazards && survs && y && loss && saved,
              "fusion head + loss forward: null argument");
    Carver c(saved);
    float* z1 = c.take((size_t)n_slides * hidden); float* z2 = c.take((size_t)n_slides * dout);
    float* logits = c.take((size_t)n_slides * n_classes); float* dlogits = c.take((size_t)n_slides * n_classes);
    RC(mpo_linear_fwd(hcat, P[0], P[1], z1, n_slides, din, hidden, 1.0f, MPO_ACT_RELU, stream));
    RC(mpo_linear_fwd(z1, P[2], P[3], z2, n_slides, hidden, dout, 1.0f, MPO_ACT_RELU, stream));
    RC(mpo_linear_fwd(z2, P[4], P[5], logits, n_slides, dout, n_classes, 1.0f, MPO_ACT_NONE, stream));
    RC(mpo_launch_head_loss(logits, reinterpret_cast<const long long*>(label), censorship, slide_weight, hazards, survs, y,
                            loss, risk, dlogits, n_slides, n_classes, alpha, eps, static_cast<hipStream_t>(stream)));
    return 0;
}
int mpo_fusion_head_loss_backward(const float* hcat, int n_slides, int din, int hidden, int dout, int n_classes,
                                  const float* const* P, const float* saved, float* d_hcat, float* const* G,
                                  void* workspace, size_t workspace_bytes, mpo_stream_t stream) {
    Carver c(const_cast<float*>(saved));
    const float* z1 = c.take((size_t)n_slides * hidden); const float* z2 = c.take((size_t)n_slides * dout);
    c.take((size_t)n_slides * n_classes);
    const float* dlogits = c.take((size_t)n_slides * n_classes);
    Arena ws(workspace, workspace_bytes);
    ws.floats((size_t)n_slides * n_classes);                    // (layout of mpo_fusion_head_workspace_bytes)
    float* dz2 = ws.floats((size_t)n_slides * dout);
    float* dz1 = ws.floats((size_t)n_slides * hidden);
    MPO_CHECK(dz2 && dz1, "fusion head + loss backward: workspace too small (%zu bytes)", workspace_bytes);
    PAIR(mpo_args_bwd_input(dlogits, P[4], dz2, n_slides, dout, n_classes, 1.0f, 0),
         mpo_args_bwd_weight(dlogits, z2, G[4], G[5], n_slides, dout, n_classes, 1.0f));
    PAIR(mpo_args_bwd_input(dz2, P[2], dz1, n_slides, hidden, dout, 1.0f, 0, gate(z2, MPO_GATE_RELU)),
         mpo_args_bwd_weight(dz2, z1, G[2], G[3], n_slides, hidden, dout, 1.0f, gate(z2, MPO_GATE_RELU)));
    PAIR(mpo_args_bwd_input(dz1, P[0], d_hcat, n_slides, din, hidden, 1.0f, 0, gate(z1, MPO_GATE_RELU)),
         mpo_args_bwd_weight(dz1, hcat, G[0], G[1], n_slides, din, hidden, 1.0f, gate(z1, MPO_GATE_RELU)));
    return 0;
}

// ------------------------------------------------------------------------------------------- survival head alone
// hazards = sigmoid(logits), survs = cumprod(1 - hazards), Y = softmax(logits)   (models/mcat/mcat.py:130-138)
int mpo_survival_head_forward(const float* logits, int n_slides, int n_classes, float* hazards, float* survs, float* y,
                              mpo_stream_t stream) {
    MPO_CHECK(logits && hazards && survs && y, "survival head forward: null argument");
    return mpo_launch_head_fwd(logits, hazards, survs, y, n_slides, n_classes, static_cast<hipStream_t>(stream));
}
int mpo_survival_head_backward(const float* hazards, const float* survs, const float* y, const float* d_hazards,
                               const float* d_survs, const float* d_y, int n_slides, int n_classes, float* d_logits,
                               mpo_stream_t stream) {
    MPO_CHECK(hazards && survs && y && d_logits, "survival head backward: null argument");
    return mpo_launch_head_bwd(hazards, survs, y, d_hazards, d_survs, d_y, d_logits, n_slides, n_classes,
                               static_cast<hipStream_t>(stream));
}

// ------------------------------------------------------------------------------------------- 'ces' loss
int mpo_ces_loss_forward(const float* hazards, const float* survs, const int64_t* label, const float* censorship,
                         int n_slides, int n_classes, float alpha, float eps, float* loss, float* risk, mpo_stream_t stream) {
    MPO_CHECK(hazards && survs && label && censorship && loss, "ces loss forward: null argument");
    return mpo_launch_ces_loss_fwd(hazards, survs, reinterpret_cast<const long long*>(label), censorship, loss, risk,
                                   n_slides, n_classes, alpha, eps, static_cast<hipStream_t>(stream));
}
int mpo_ces_loss_backward(const float* hazards, const float* survs, const int64_t* label, const float* censorship,
                          int n_slides, int n_classes, float alpha, float eps, const float* d_loss, int d_loss_is_scalar,
                          float* d_hazards, float* d_survs, mpo_stream_t stream) {
    MPO_CHECK(hazards && survs && label && censorship && d_loss && d_hazards && d_survs, "ces loss backward: null argument");
    return mpo_launch_ces_loss_bwd(hazards, survs, reinterpret_cast<const long long*>(label), censorship, d_loss,
                                   d_loss_is_scalar, d_hazards, d_survs, n_slides, n_classes, alpha, eps,
                                   static_cast<hipStream_t>(stream));
}

// ------------------------------------------------------------------------------------------- K3 CAG
// params: fc1.0.weight,.bias, fc2.0.weight,.bias, fc3.0.weight,.bias, G.1.weight,.bias, E.1.weight,.bias, fc_c.0.weight,.bias
// saved: u1 u2 u3 t1 t3 G E m  [R,h each] | stats_g [R,2] | stats_e [R,2]
size_t mpo_cag_saved_floats(int rows, int hidden) {
    CarveSizer c;
    for (int i = 0; i < 8; ++i) c.take((size_t)rows * hidden);
    c.take(2 * (size_t)rows); c.take(2 * (size_t)rows);
    return c.n;
}
size_t mpo_cag_workspace_bytes(int rows, int hidden) {
    Sizer s;
    for (int i = 0; i < 5; ++i) s.floats((size_t)rows * hidden);
    return s.off + 256;
}

// ContextualAttentionGate.forward, models/blocks.py:247-253
int mpo_cag_forward(const float* q, const float* q_hat, int rows, int dim, int hidden, const float* const* P,
                    float* c_out, float* saved, const float* residual, float* sum_out, mpo_stream_t stream) {
    MPO_CHECK((residual == nullptr) == (sum_out == nullptr), "CAG forward: residual and sum_out go together");
    Carver c(saved);
    float* u1 = c.take((size_t)rows * hidden); float* u2 = c.take((size_t)rows * hidden); float* u3 = c.take((size_t)rows * hidden);
    float* t1 = c.take((size_t)rows * hidden); float* t3 = c.take((size_t)rows * hidden);
    float* g = c.take((size_t)rows * hidden); float* e = c.take((size_t)rows * hidden); float* m = c.take((size_t)rows * hidden);
    float* sg = c.take(2 * (size_t)rows); float* se = c.take(2 * (size_t)rows);
    {
        const GemmArgs g3 = mpo_args_fwd(q_hat, P[4], P[5], u3, rows, dim, hidden, 1.0f, MPO_ACT_ELU);
        RC(mpo_gemm_together(stream, mpo_args_fwd(q, P[0], P[1], u1, rows, dim, hidden, 1.0f, MPO_ACT_ELU),
                             mpo_args_fwd(q_hat, P[2], P[3], u2, rows, dim, hidden, 1.0f, MPO_ACT_ELU), &g3));
    }
    RC(mpo_launch_cag_mid_fwd(u1, u2, u3, P[6], P[7], P[8], P[9], t1, t3, g, e, m, sg, se, rows, hidden, 1e-5f, stream));
    if (sum_out == nullptr) {
        RC(mpo_linear_fwd(m, P[10], P[11], c_out, rows, hidden, hidden, 1.0f, MPO_ACT_ELU, stream));
    } else {
        // the caller's  residual + C  (models/blocks.py:110: attn_output + CAG(...)) rides in the same launch as a second member
        // of the same product: C alone stays in c_out (its ELU derivative is read off it in the backward)
        RC(mpo_gemm_together(stream, mpo_args_fwd(m, P[10], P[11], c_out, rows, hidden, hidden, 1.0f, MPO_ACT_ELU),
                             mpo_args_fwd(m, P[10], P[11], sum_out, rows, hidden, hidden, 1.0f, MPO_ACT_ELU, residual)));
    }
    return 0;
}

int mpo_cag_backward(const float* q, const float* q_hat, int rows, int dim, int hidden, const float* const* P,
                     const float* saved, const float* c_out, const float* d_c, float* d_q, int d_q_accumulate, float* d_q_hat,
                     float* const* G, void* workspace, size_t workspace_bytes, mpo_stream_t stream) {
    Carver c(const_cast<float*>(saved));
    const float* u1 = c.take((size_t)rows * hidden); const float* u2 = c.take((size_t)rows * hidden);
    const float* u3 = c.take((size_t)rows * hidden); const float* t1 = c.take((size_t)rows * hidden);
    const float* t3 = c.take((size_t)rows * hidden); const float* g = c.take((size_t)rows * hidden);
    const float* e = c.take((size_t)rows * hidden); const float* m = c.take((size_t)rows * hidden);
    const float* sg = c.take(2 * (size_t)rows); const float* se = c.take(2 * (size_t)rows);
    Arena ws(workspace, workspace_bytes);
    float* dm = ws.floats((size_t)rows * hidden); float* dG = ws.floats((size_t)rows * hidden);
    float* dE = ws.floats((size_t)rows * hidden); float* ds12 = ws.floats((size_t)rows * hidden);
    float* ds3 = ws.floats((size_t)rows * hidden);
    MPO_CHECK(dm && dG && dE && ds12 && ds3, "CAG backward: workspace too small (%zu bytes)", workspace_bytes);
    // C = ELU(m Wc^T + bc)
    RC(mpo_linear_bwd_pair(mpo_args_bwd_input(d_c, P[10], dm, rows, hidden, hidden, 1.0f, 0, gate(c_out, MPO_GATE_ELU)),
                           mpo_args_bwd_weight(d_c, m, G[10], G[11], rows, hidden, hidden, 1.0f, gate(c_out, MPO_GATE_ELU)), stream));
    RC(mpo_launch_cag_mid_bwd(dm, t1, t3, g, e, P[6], P[8], sg, se, dG, dE, ds12, ds3, rows, hidden, stream));
    RC(mpo_launch_ln_bwd_params_only(dG, t1, sg, G[6], G[7], rows, hidden, stream));
    RC(mpo_launch_ln_bwd_params_only(dE, t3, se, G[8], G[9], rows, hidden, stream));
    // u1 = ELU(fc1 q), u2 = ELU(fc2 qh), u3 = ELU(fc3 qh)
    {   // five independent products in one mixed launch; d_q_hat accumulates its second product afterwards
        GemmGroup grp;
        grp.g[0] = mpo_args_bwd_input(ds12, P[0], d_q, rows, dim, hidden, 1.0f, d_q_accumulate ? 1 : 0, gate(u1, MPO_GATE_ELU));
        grp.g[1] = mpo_args_bwd_weight(ds12, q, G[0], G[1], rows, dim, hidden, 1.0f, gate(u1, MPO_GATE_ELU));
        grp.g[2] = mpo_args_bwd_input(ds12, P[2], d_q_hat, rows, dim, hidden, 1.0f, 0, gate(u2, MPO_GATE_ELU));
        grp.g[3] = mpo_args_bwd_weight(ds12, q_hat, G[2], G[3], rows, dim, hidden, 1.0f, gate(u2, MPO_GATE_ELU));
        grp.g[4] = mpo_args_bwd_weight(ds3, q_hat, G[4], G[5], rows, dim, hidden, 1.0f, gate(u3, MPO_GATE_ELU));
        grp.n = 5;
        RC(mpo_launch_gemm_mixed(grp, stream));
    }
    RC(mpo_linear_bwd_input(ds3, P[4], d_q_hat, rows, dim, hidden, 1.0f, 1, stream, gate(u3, MPO_GATE_ELU)));
    return 0;
}

// ------------------------------------------------------------------------------------------- omic SNNs (self.G)
// models/mcat/mcat.py:32-45,90-92: per omic group i  G_i(x) = AD(ELU(W2 AD(ELU(W1 x + b1)) + b2)), AD = AlphaDropout(p).
// All groups advance together: ONE grouped GEMM launch per layer (blockIdx.z = group).  x_i [n_slides, width_i];
// the second layer writes straight into G_bag [n_slides, n_groups, d] (row stride n_groups * d).
// params per group: 0.0.weight [d, width_i], 0.0.bias, 1.0.weight [d, d], 1.0.bias        saved: u1 [n_groups][n_slides, d]
size_t mpo_omic_snn_saved_floats(int n_slides, int n_groups, int d) {
    CarveSizer c;
    for (int i = 0; i < n_groups; ++i) c.take((size_t)n_slides * d);
    return c.n;
}
size_t mpo_omic_snn_workspace_bytes(int n_slides, int n_groups, int d) {
    Sizer s;
    for (int i = 0; i < n_groups; ++i) s.floats((size_t)n_slides * d);
    return s.off + 256;
}
uint64_t mpo_omic_snn_rng_span(int n_slides, int n_groups, int d) { return 2ull * n_groups * ((uint64_t)n_slides * n_groups * d / 4 + 2); }

int mpo_omic_snn_forward(const float* const* x, const int* widths, int n_groups, int n_slides, int d,
                         const float* const* params, float drop_p, uint64_t seed, uint64_t offset,
                         const uint64_t* rng_epoch, float* g_bag, float* saved, mpo_stream_t stream) {
    MPO_CHECK(n_groups >= 1, "omic SNN: at least one group (got %d)", n_groups);
    const uint64_t stride = (uint64_t)n_slides * n_groups * d / 4 + 2;
    Carver c(saved);
    std::vector<GemmArgs> l1(n_groups), l2(n_groups);
    for (int i = 0; i < n_groups; ++i) {
        float* u1 = c.take((size_t)n_slides * d);
        const float* const* P = params + 4 * i;
        GemmArgs& a = l1[i];
        a.A = x[i]; a.B = P[0]; a.bias = P[1]; a.C = u1;
        a.M = n_slides; a.N = d; a.K = widths[i]; a.lda = widths[i]; a.ldb = widths[i]; a.ldc = d;
        a.act = MPO_ACT_ELU; a.alpha_dropout = 1; a.drop_p = drop_p; a.drop_seed = seed; a.drop_off = offset + stride * (2 * i);
        a.rng_epoch = reinterpret_cast<const unsigned long long*>(rng_epoch);
        GemmArgs& b = l2[i];
        b.A = u1; b.B = P[2]; b.bias = P[3]; b.C = g_bag + (size_t)i * d;
        b.M = n_slides; b.N = d; b.K = d; b.lda = d; b.ldb = d; b.ldc = n_groups * d;
        b.act = MPO_ACT_ELU; b.alpha_dropout = 1; b.drop_p = drop_p; b.drop_seed = seed; b.drop_off = offset + stride * (2 * i + 1);
        b.rng_epoch = a.rng_epoch;
    }
    RC(launch_in_groups(l1, 1, 1, stream));
    RC(launch_in_groups(l2, 1, 1, stream));
    return 0;
}

int mpo_omic_snn_backward(const float* const* x, const int* widths, int n_groups, int n_slides, int d,
                          const float* const* params, float drop_p, uint64_t seed, uint64_t offset,
                          const uint64_t* rng_epoch, const float* g_bag, const float* saved, const float* d_g_bag,
                          float* const* grads, void* workspace, size_t workspace_bytes, mpo_stream_t stream) {
    MPO_CHECK(n_groups >= 1, "omic SNN: at least one group (got %d)", n_groups);
    const uint64_t stride = (uint64_t)n_slides * n_groups * d / 4 + 2;
    Carver c(const_cast<float*>(saved));
    Arena ws(workspace, workspace_bytes);
    std::vector<GemmArgs> dx(n_groups), dw2(n_groups), dw1(n_groups);
    const unsigned long long* ep = reinterpret_cast<const unsigned long long*>(rng_epoch);
    for (int i = 0; i < n_groups; ++i) {
        const float* u1 = c.take((size_t)n_slides * d);
        float* du1 = ws.floats((size_t)n_slides * d);
        MPO_CHECK(du1, "omic SNN backward: workspace too small (%zu bytes)", workspace_bytes);
        const float* const* P = params + 4 * i;
        float* const* G = grads + 4 * i;
        const float* dy = d_g_bag + (size_t)i * d;          // [n_slides, d] with row stride n_groups * d
        const float* y = g_bag + (size_t)i * d;
        const int ldy = n_groups * d;
        // layer 2: y = AD(ELU(u1 W2^T + b2)); gate indexed like y (row stride ldy) -> same stream index as forward
        GemmArgs& a = dx[i];                                 // du1 = (dy*gate) W2
        a.A = dy; a.B = P[2]; a.C = du1; a.M = n_slides; a.N = d; a.K = d; a.lda = ldy; a.ldb = d; a.ldc = d;
        a.gate = y; a.gate_mode = MPO_GATE_ELU_ADROP; a.gate_p = drop_p; a.gate_seed = seed; a.gate_off = offset + stride * (2 * i + 1);
        a.rng_epoch = ep;
        GemmArgs& b = dw2[i];                                // dW2 = (dy*gate)^T u1, db2
        b.A = dy; b.B = u1; b.C = G[2]; b.bias_grad = G[3]; b.M = d; b.N = d; b.K = n_slides; b.lda = ldy; b.ldb = d; b.ldc = d;
        b.gate = y; b.gate_mode = MPO_GATE_ELU_ADROP; b.gate_p = drop_p; b.gate_seed = seed; b.gate_off = a.gate_off; b.rng_epoch = ep;
        GemmArgs& e = dw1[i];                                // dW1 = (du1*gate1)^T x, db1   (x needs no gradient: it is data)
        e.A = du1; e.B = x[i]; e.C = G[0]; e.bias_grad = G[1]; e.M = d; e.N = widths[i]; e.K = n_slides; e.lda = d; e.ldb = widths[i];
        e.ldc = widths[i];
        e.gate = u1; e.gate_mode = MPO_GATE_ELU_ADROP; e.gate_p = drop_p; e.gate_seed = seed; e.gate_off = offset + stride * (2 * i);
        e.rng_epoch = ep;
    }
    RC(launch_in_groups(dx, 1, 0, stream));
    RC(launch_in_groups(dw2, 0, 0, stream));
    RC(launch_in_groups(dw1, 0, 0, stream));
    return 0;
}

}  // extern "C"
