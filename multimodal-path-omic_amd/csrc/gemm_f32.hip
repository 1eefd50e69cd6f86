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
// Tile: 32 x 32 outputs per 256-thread workgroup (4 waves x 16x16), and a whole K chunk of up to 256
// staged at once: every global load of a chunk is in flight together (these GEMMs are a few hundred KB
// and launch/latency-bound, not bandwidth-bound), the next chunk is prefetched into registers while the
// MFMAs of the current one run.  Lane (i = l&15, kq = l>>4) takes k = 16kk + 4kq + j for the j-th MFMA of a
// block of four (any partition of k is valid as long as A and B agree).  An operand whose k is contiguous in
// memory is imaged [row][k] (stride 264 floats) and read with one conflict-free ds_read_b128 per four MFMAs;
// an operand whose ROW index is contiguous is imaged as it lies, [k][row] (stride 36 floats: float4 stores,
// no transposition, conflict-free ds_read_b32).
#include <stdlib.h>

#include "mpo_common.h"
#include "mpo_kernels.h"
#include "gemm_f32_gate.h"

namespace {

constexpr int BM = 32, BN = 32, KC = 256, LDK = KC + 8, LDM = 36;
constexpr int IMG_FLOATS = KC * LDM > BM * LDK ? KC * LDM : BM * LDK;


// One operand tile (32 rows x KC k) in flight in registers: 8 float4 per thread.
template <bool KCONTIG>
struct OperandStage {
    f32x4 v[8];
    // multiply the staged values by the gate of the same (mn, k) elements (same layout/ld as the operand)
    __device__ __forceinline__ void apply_gate(const GateFn& gf, int ld, int mn0, int mn_lim, int k0, int k_lim, int tid) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = i * 256 + tid;
            int mn, k;
            if (KCONTIG) { mn = f >> 6; k = (f & 63) << 2; } else { k = f >> 3; mn = (f & 7) << 2; }
            // the four elements of a float4 are consecutive in memory: one Philox draw covers them when aligned
            const size_t idx0 = KCONTIG ? (size_t)(mn0 + mn) * ld + (k0 + k) : (size_t)(k0 + k) * ld + (mn0 + mn);
            const bool live0 = KCONTIG ? (mn0 + mn < mn_lim && k0 + k < k_lim) : (k0 + k < k_lim && mn0 + mn < mn_lim);
            if (gf.draws() && (idx0 & 3) == 0) {
                if (!live0) continue;
                const uint64_t ctr = gf.off + (idx0 >> 2);
                const uint4 r = philox4x32((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)gf.seed, (uint32_t)(gf.seed >> 32));
                const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool live = KCONTIG ? k0 + k + j < k_lim : mn0 + mn + j < mn_lim;
                    if (live) v[i][j] *= gf.with_word(gf.g ? gf.g[idx0 + j] : 0.f, w[j]);
                }
                continue;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gmn = mn0 + mn + (KCONTIG ? 0 : j), gk = k0 + k + (KCONTIG ? j : 0);
                if (gmn < mn_lim && gk < k_lim) {
                    const size_t idx = KCONTIG ? (size_t)gmn * ld + gk : (size_t)gk * ld + gmn;
                    v[i][j] *= gf(gf.g ? gf.g[idx] : 0.f, idx);
                }
            }
        }
    }
    // element (mn, k) lives at p[mn*ld + k] (KCONTIG) or p[k*ld + mn]
    __device__ __forceinline__ void load(const float* __restrict__ p, int ld, int mn0, int mn_lim, int k0, int k_lim,
                                         bool vec_ok, int tid) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = i * 256 + tid;
            int mn, k;
            if (KCONTIG) { mn = f >> 6; k = (f & 63) << 2; } else { k = f >> 3; mn = (f & 7) << 2; }
            const int gmn = mn0 + mn, gk = k0 + k;
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
            if (KCONTIG) {
                if (gmn < mn_lim) {
                    const float* q = p + (size_t)gmn * ld + gk;
                    if (vec_ok && gk + 3 < k_lim) r = *reinterpret_cast<const f32x4*>(q);
                    else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (gk + j < k_lim) r[j] = q[j];
                    }
                }
            } else {
                if (gk < k_lim) {
                    const float* q = p + (size_t)gk * ld + gmn;
                    if (vec_ok && gmn + 3 < mn_lim) r = *reinterpret_cast<const f32x4*>(q);
                    else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (gmn + j < mn_lim) r[j] = q[j];
                    }
                }
            }
            v[i] = r;
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ lds, int tid) const {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = i * 256 + tid;
            if (KCONTIG) {
                *reinterpret_cast<f32x4*>(lds + (f >> 6) * LDK + ((f & 63) << 2)) = v[i];
            } else {
                *reinterpret_cast<f32x4*>(lds + (f >> 3) * LDM + ((f & 7) << 2)) = v[i];
            }
        }
    }
    // the four k-values of MFMA block kk for row `row` of this operand
    __device__ __forceinline__ static f32x4 frag(const float* __restrict__ lds, int row, int kk, int kq) {
        if (KCONTIG) return *reinterpret_cast<const f32x4*>(lds + row * LDK + 16 * kk + 4 * kq);
        const float* p = lds + (16 * kk + 4 * kq) * LDM + row;
        return f32x4{p[0], p[LDM], p[2 * LDM], p[3 * LDM]};
    }
};

struct GemmLds {
    __attribute__((aligned(16))) float As[IMG_FLOATS];
    __attribute__((aligned(16))) float Bs[IMG_FLOATS];
    float bred[4][32];
};

template <bool A_KC, bool B_KC>
__device__ __forceinline__ void gemm_f32_body(const GemmArgs& g, GemmLds& lds);

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256)
void gemm_f32_kernel(GemmArgs g) {
    __shared__ GemmLds lds;
    gemm_f32_body<A_KC, B_KC>(g, lds);
}

template <bool A_KC, bool B_KC>
__device__ __forceinline__ void gemm_f32_body(const GemmArgs& g, GemmLds& lds) {
    if ((int)blockIdx.y * BM >= g.M || (int)blockIdx.x * BN >= g.N) return;      // grouped launch: grid is the max extent
    float* As = lds.As;
    float* Bs = lds.Bs;
    float (*bred)[32] = lds.bred;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int i16 = lane & 15, kq = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const bool a_vec = (g.lda & 3) == 0 && (reinterpret_cast<uintptr_t>(g.A) & 15) == 0;
    const bool b_vec = (g.ldb & 3) == 0 && (reinterpret_cast<uintptr_t>(g.B) & 15) == 0;

    GateFn gf;
    gf.g = g.gate; gf.mode = g.gate_mode; gf.p = g.gate_p; gf.seed = g.gate_seed;
    gf.off = epoch_offset(g.gate_off, g.rng_epoch);
    gf.inv_keep = g.gate_p > 0.f ? 1.0f / (1.0f - g.gate_p) : 1.0f;
    const bool gated = g.gate_mode != MPO_GATE_NONE;

    OperandStage<A_KC> sa;
    OperandStage<B_KC> sb;
    sa.load(g.A, g.lda, m0, g.M, 0, g.K, a_vec, tid);
    if (gated) sa.apply_gate(gf, g.lda, m0, g.M, 0, g.K, tid);
    sb.load(g.B, g.ldb, n0, g.N, 0, g.K, b_vec, tid);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;                                           // bias gradient: row sums of the (gated) A tile

    for (int k0 = 0; k0 < g.K; k0 += KC) {
        sa.store(As, tid);
        sb.store(Bs, tid);
        if (k0 + KC < g.K) {                                    // next chunk flies under this chunk's MFMAs
            sa.load(g.A, g.lda, m0, g.M, k0 + KC, g.K, a_vec, tid);
            if (gated) sa.apply_gate(gf, g.lda, m0, g.M, k0 + KC, g.K, tid);
            sb.load(g.B, g.ldb, n0, g.N, k0 + KC, g.K, b_vec, tid);
        }
        __syncthreads();
        const int kblocks = (min(KC, g.K - k0) + 15) >> 4;
        if (g.bias_grad != nullptr && blockIdx.x == 0) {       // row sums of the staged A tile: 8 threads per row
            const int row = tid & 31, part = tid >> 5, klen = 2 * kblocks;
            float ps = 0.f;
            for (int k = part * klen; k < (part + 1) * klen; ++k) ps += A_KC ? As[row * LDK + k] : As[k * LDM + row];
            ps += __shfl_xor(ps, 32);                          // parts (2w, 2w+1) share a wave
            if (lane < 32) bred[wave][lane] = ps;
        }
        for (int kk = 0; kk < kblocks; ++kk) {
            const f32x4 a = OperandStage<A_KC>::frag(As, 16 * wm + i16, kk, kq);
            const f32x4 b = OperandStage<B_KC>::frag(Bs, 16 * wn + i16, kk, kq);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc1, 0, 0, 0);
        }
        __syncthreads();
        if (g.bias_grad != nullptr && blockIdx.x == 0 && tid < BM) bsum += bred[0][tid] + bred[1][tid] + bred[2][tid] + bred[3][tid];
    }
    // ---- epilogue: D col = lane&15, row = 4*(lane>>4) + r
    const int n = n0 + 16 * wn + i16;
    if (n < g.N) {
        const float bias = g.bias ? g.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * wm + 4 * kq + r;
            if (m >= g.M) continue;
            float v = (acc0[r] + acc1[r] + bias) * g.alpha;
            v = apply_act(v, g.act);
            const size_t o = (size_t)m * g.ldc + n;
            if (g.drop_p > 0.f) {
                const unsigned long long doff = epoch_offset(g.drop_off, g.rng_epoch);
                if (g.alpha_dropout) {
                    const bool keep = dropout_keep(g.drop_seed, doff, o, g.drop_p, 1.0f) != 0.f;
                    v = alpha_drop_a(g.drop_p) * (keep ? v : kAlphaPrime) + alpha_drop_b(g.drop_p);
                } else {
                    v *= dropout_keep(g.drop_seed, doff, o, g.drop_p, 1.0f / (1.0f - g.drop_p));
                }
            }
            if (g.mask) v *= g.mask[o];
            if (g.residual) v += g.residual[o];
            if (g.accumulate) v += g.C[o];
            g.C[o] = v;
        }
    }
    if (g.bias_grad != nullptr && blockIdx.x == 0 && tid < BM && m0 + tid < g.M) g.bias_grad[m0 + tid] = bsum;
}


// ------------------------------------------------------------------------------------------------ direct variant
// (templates in gemm_f32_direct.h, instantiated per NBMAX in their own translation units)
constexpr int DB = 16;
constexpr int DMAXB = 8;
}  // namespace
void mpo_direct_single_nb4(const GemmArgs& g, int layout, dim3 grid, hipStream_t stream);
void mpo_direct_single_nb8(const GemmArgs& g, int layout, dim3 grid, hipStream_t stream);
void mpo_direct_group_nb4(const GemmGroup& grp, dim3 grid, hipStream_t stream);
void mpo_direct_group_nb8(const GemmGroup& grp, dim3 grid, hipStream_t stream);
namespace {
inline int direct_nbmax(int k) { return k <= 256 ? 4 : DMAXB; }
void launch_direct_single(const GemmArgs& g, int layout, dim3 grid, hipStream_t stream) {
    if (direct_nbmax(g.K) == 4) mpo_direct_single_nb4(g, layout, grid, stream);
    else mpo_direct_single_nb8(g, layout, grid, stream);
}
void launch_direct_group(const GemmGroup& grp, dim3 grid, hipStream_t stream) {
    int kmax = 0;
    for (int i = 0; i < grp.n; ++i) kmax = grp.g[i].K > kmax ? grp.g[i].K : kmax;
    // a launch that fills the chip several times over is throughput-bound: the 4-block variant's smaller register
    // footprint (4 instead of 2 workgroups per CU) then beats having all of K in flight at once
    const size_t wgs = (size_t)grid.x * grid.y * grid.z;
    if (wgs > 1024 || direct_nbmax(kmax) == 4) mpo_direct_group_nb4(grp, grid, stream);
    else mpo_direct_group_nb8(grp, grid, stream);
}

// MPO_GEMM_STAGED=1 selects the LDS-staged kernels everywhere (A/B comparison, debugging)
bool use_direct() {
    static const bool v = [] { const char* e = getenv("MPO_GEMM_STAGED"); return !(e && e[0] == '1'); }();
    return v;
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

int mpo_launch_gemm(const GemmArgs& g, int a_kc, int b_kc, hipStream_t stream) {
    if (g.M <= 0 || g.N <= 0) return 0;
    MPO_CHECK(g.K > 0, "gemm: K must be positive (got %d)", g.K);
    if (use_direct()) {
        dim3 dgrid((g.N + DB - 1) / DB, (g.M + DB - 1) / DB);
        launch_direct_single(g, 2 * (a_kc ? 1 : 0) + (b_kc ? 1 : 0), dgrid, stream);
        MPO_LAUNCH_CHECK();
        return 0;
    }
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM);
    if (a_kc && b_kc) gemm_f32_kernel<true, true><<<grid, 256, 0, stream>>>(g);
    else if (a_kc && !b_kc) gemm_f32_kernel<true, false><<<grid, 256, 0, stream>>>(g);
    else if (!a_kc && b_kc) gemm_f32_kernel<false, true><<<grid, 256, 0, stream>>>(g);
    else gemm_f32_kernel<false, false><<<grid, 256, 0, stream>>>(g);
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
    if (use_direct()) {
        GemmGroup tagged = grp;
        for (int i = 0; i < tagged.n; ++i) tagged.g[i].layout = 2 * (a_kc ? 1 : 0) + (b_kc ? 1 : 0);
        dim3 dgrid((nx + DB - 1) / DB, (mx + DB - 1) / DB, grp.n);
        launch_direct_group(tagged, dgrid, stream);
        MPO_LAUNCH_CHECK();
        return 0;
    }
    // staged A/B path: one launch per member (the grouped staged kernels were dropped: 70 s of compile time for a debug switch)
    for (int i = 0; i < grp.n; ++i)
        if (int rc = mpo_launch_gemm(grp.g[i], a_kc, b_kc, stream)) return rc;
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
    if (use_direct()) {
        dim3 dgrid((nx + DB - 1) / DB, (mx + DB - 1) / DB, grp.n);
        launch_direct_group(grp, dgrid, stream);
        MPO_LAUNCH_CHECK();
        return 0;
    }
    for (int i = 0; i < grp.n; ++i)
        if (int rc = mpo_launch_gemm(grp.g[i], grp.g[i].layout >> 1, grp.g[i].layout & 1, stream)) return rc;
    return 0;
}

int mpo_launch_colsum(const float* x, float* out, int M, int N, int ld, int accumulate, hipStream_t stream) {
    if (N <= 0) return 0;
    colsum_kernel<<<(N + 15) / 16, 256, 0, stream>>>(x, out, M, N, ld, accumulate);
    MPO_LAUNCH_CHECK();
    return 0;
}
