// Generic single-pass bag kernels + N x M map kernels: the building blocks of K2, NaCAGaT's
// narrow-gated co-attention (models/blocks.py:114-206), in its first, modular form.
//
//   S[n][m] = (q~[n].k[m]) * (tanh(q)[n].tanh(k)[m] + 1) / 2,  A = softmax_m(S),  A_drop = dropout(A)
//   ctx[n]  = sum_m A_drop[n][m] H[m]          (value projection folded out: A v = (A H) W_v^T + b_v sum_m A)
//
// Every long tensor (K = H W_k^T + b_k, TK = tanh K, H, and their gradients) is streamed exactly once
// per kernel through the LDS tile image of coattn_tile.h; the coupling between them goes through
// ragged N x M fp32 maps (24 bytes per patch, against 512-1024 bytes per patch of bag data):
//   bag_rowdot   : map[n][m]  = sum_e X[m][e] r[n][e]                 (a = qs.K, g = tq.TK, dA = dctx.H)
//   bag_colacc   : acc[n][e]  = sum_m W[n][m] X[m][e]   (split-M)     (ctx = A.H, dq~ = ds1.K, dtq = dg.TK)
//   bag_outer    : dX[m][e]   = sum_n W1[n][m] Z1[n][e] + W2[n][m] Z2[n][e]   (dK, dTK, dH)
//   map kernels  : gated softmax statistics / apply (+ dropout) / backward.
// The same MFMA orientation rules as K1 apply (query index on the MFMA column).  A fully fused K2
// (in-kernel K projection, one pass) is the planned successor; this form is parity-complete.
#include <type_traits>

#include "coattn_tile.h"
#include "mpo_kernels.h"

namespace {

template <int E_, bool F32BAG>
struct BagCfg {
    static constexpr int NT = F32BAG ? 2 : 1;
    static constexpr int WAVES = (F32BAG && E_ == 512) ? 2 : 4;
    static constexpr int WAVE_LDS = NT * TileGeom<E_>::TILEB;
    static constexpr int LDS_BYTES = WAVES * WAVE_LDS;
};

struct SplitGeom {
    int b, row_begin, m_rows, r0, r1, n_my;
    size_t part;
};
template <int WAVES>
__device__ __forceinline__ SplitGeom split_geom(const int* cu, const BagPlan& plan, int wave) {
    const WgGeom g = wg_geom(cu, plan);
    SplitGeom s;
    s.b = g.b; s.row_begin = g.row_begin; s.m_rows = g.m_rows; s.r0 = g.r0; s.r1 = g.r1; s.part = g.part;
    s.n_my = wave < g.ntiles ? (g.ntiles - wave + WAVES - 1) / WAVES : 0;
    return s;
}

// ------------------------------------------------------------------ (i) map[n][m] = alpha * X[m] . r[n]
template <int E_, bool F32BAG>
__global__ __launch_bounds__((BagCfg<E_, F32BAG>::WAVES * 64), 1)
void bag_rowdot_kernel(const void* __restrict__ bag_, const int* __restrict__ cu, const float* __restrict__ r,
                       float* __restrict__ map, float alpha, int n_q, BagPlan plan) {
    using G = TileGeom<E_>;
    using C = BagCfg<E_, F32BAG>;
    __shared__ __attribute__((aligned(16))) char lds[C::LDS_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const SplitGeom sg = split_geom<C::WAVES>(cu, plan, wave);
    const int b = sg.b;
    char* thi = lds + wave * C::WAVE_LDS;
    char* tlo = thi + (C::NT - 1) * G::TILEB;
    const int q = lane & 15, g = lane >> 4;
    bf16x8 rh[G::KS], rl[G::KS];
    load_query_frags<E_>(r + (size_t)b * n_q * E_, n_q, lane, rh, rl);
    float* mrow = map + (size_t)n_q * sg.row_begin + (size_t)q * sg.m_rows;
    const char* slide = reinterpret_cast<const char*>(bag_) + (size_t)sg.row_begin * E_ * (F32BAG ? 4 : 2);
    Stage<E_, F32BAG> st0, st1;
    if (sg.n_my > 0) {
        st0.load(slide, sg.r0 + kTileRows * wave, sg.m_rows, 0, lane);
        if constexpr (F32BAG) st1.load(slide, sg.r0 + kTileRows * wave, sg.m_rows, 1, lane);
    }
    for (int it = 0; it < sg.n_my; ++it) {
        const int trow = sg.r0 + kTileRows * (wave + it * C::WAVES);
        const int nvalid = min(kTileRows, sg.r1 - trow);
        st0.store(thi, tlo, 0, lane);
        if constexpr (F32BAG) st1.store(thi, tlo, 1, lane);
        if (it + 1 < sg.n_my) {
            st0.load(slide, trow + kTileRows * C::WAVES, sg.m_rows, 0, lane);
            if constexpr (F32BAG) st1.load(slide, trow + kTileRows * C::WAVES, sg.m_rows, 1, lane);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
        tile_dot_rows<E_, C::NT>(thi, tlo, rh, rl, s0, s1, lane);
        if (q < n_q) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                if (4 * g + rr < nvalid) mrow[trow + 4 * g + rr] = s0[rr] * alpha;
                if (16 + 4 * g + rr < nvalid) mrow[trow + 16 + 4 * g + rr] = s1[rr] * alpha;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------ (ii) part[n][e] = sum_m W[n][m] X[m][e]
template <int E_, bool F32BAG>
__global__ __launch_bounds__((BagCfg<E_, F32BAG>::WAVES * 64), 1)
void bag_colacc_kernel(const void* __restrict__ bag_, const int* __restrict__ cu, const float* __restrict__ wmap,
                       float* __restrict__ part, int n_q, BagPlan plan) {
    using G = TileGeom<E_>;
    using C = BagCfg<E_, F32BAG>;
    __shared__ __attribute__((aligned(16))) char lds[C::LDS_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const SplitGeom sg = split_geom<C::WAVES>(cu, plan, wave);
    char* thi = lds + wave * C::WAVE_LDS;
    char* tlo = thi + (C::NT - 1) * G::TILEB;
    const int q = lane & 15, g = lane >> 4;
    const float* wrow = wmap + (size_t)n_q * sg.row_begin + (size_t)(q < n_q ? q : 0) * sg.m_rows;
    const char* slide = reinterpret_cast<const char*>(bag_) + (size_t)sg.row_begin * E_ * (F32BAG ? 4 : 2);
    f32x4 acc[G::DT];
#pragma unroll
    for (int t = 0; t < G::DT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    Stage<E_, F32BAG> st0, st1;
    if (sg.n_my > 0) {
        st0.load(slide, sg.r0 + kTileRows * wave, sg.m_rows, 0, lane);
        if constexpr (F32BAG) st1.load(slide, sg.r0 + kTileRows * wave, sg.m_rows, 1, lane);
    }
    for (int it = 0; it < sg.n_my; ++it) {
        const int trow = sg.r0 + kTileRows * (wave + it * C::WAVES);
        const int nvalid = min(kTileRows, sg.r1 - trow);
        float w[8];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            w[rr] = (q < n_q && 4 * g + rr < nvalid) ? wrow[trow + 4 * g + rr] : 0.f;
            w[4 + rr] = (q < n_q && 16 + 4 * g + rr < nvalid) ? wrow[trow + 16 + 4 * g + rr] : 0.f;
        }
        st0.store(thi, tlo, 0, lane);
        if constexpr (F32BAG) st1.store(thi, tlo, 1, lane);
        if (it + 1 < sg.n_my) {
            st0.load(slide, trow + kTileRows * C::WAVES, sg.m_rows, 0, lane);
            if constexpr (F32BAG) st1.load(slide, trow + kTileRows * C::WAVES, sg.m_rows, 1, lane);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        bf16x8 wh, wl;
        pack_hi_lo(w, wh, wl);
        tile_accum_cols<E_, C::NT>(thi, tlo, wh, wl, acc, lane);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {
        float* wq = reinterpret_cast<float*>(thi);
#pragma unroll
        for (int t = 0; t < G::DT; ++t) *reinterpret_cast<f32x4*>(wq + q * E_ + 16 * t + 4 * g) = acc[t];
    }
    __syncthreads();
    const size_t pbase = sg.part;
    for (int idx = threadIdx.x; idx < n_q * E_; idx += C::WAVES * 64) {
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < C::WAVES; ++w) a += reinterpret_cast<const float*>(lds + w * C::WAVE_LDS)[idx];
        part[pbase * n_q * E_ + idx] = a;
    }
}

// ------------------------------------------------------------------ (iii) dX[m][e] = sum_n W1[n][m] Z1[n][e] + W2[n][m] Z2[n][e]
// Output only (no bag input).  OUT_F32: dtype of dX.  w2/z2 may be null.
template <int E_, bool OUT_F32>
__global__ __launch_bounds__(256, 1)
void bag_outer_kernel(const int* __restrict__ cu, const float* __restrict__ w1, const float* __restrict__ z1,
                      const float* __restrict__ w2, const float* __restrict__ z2, void* __restrict__ dx_,
                      int n_q, BagPlan plan) {
    using G = TileGeom<E_>;
    constexpr int EB = OUT_F32 ? 4 : 2;
    constexpr int IMG = kTileRows * E_ * EB;
    __shared__ __attribute__((aligned(16))) char lds[4 * IMG];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const SplitGeom sg = split_geom<4>(cu, plan, wave);
    const int b = sg.b;
    char* img = lds + wave * IMG;
    const int c16 = lane & 15, g = lane >> 4;
    const float* z1b = z1 + (size_t)b * n_q * E_;
    const float* z2b = z2 ? z2 + (size_t)b * n_q * E_ : nullptr;
    bf16x8 zh[G::DT], zl[G::DT];
#pragma unroll
    for (int t = 0; t < G::DT; ++t) {
        float z[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int qq = 4 * g + j;
            const int qc = qq < n_q ? qq : n_q - 1;
            const float live = qq < n_q ? 1.0f : 0.0f;
            z[j] = z1b[qc * E_ + 16 * t + c16] * live;
            z[4 + j] = z2b ? z2b[qc * E_ + 16 * t + c16] * live : 0.f;
        }
        pack_hi_lo(z, zh[t], zl[t]);
    }
    const float* w1b = w1 + (size_t)n_q * sg.row_begin;
    const float* w2b = w2 ? w2 + (size_t)n_q * sg.row_begin : nullptr;
    char* dslide = reinterpret_cast<char*>(dx_) + (size_t)sg.row_begin * E_ * EB;
    for (int it = 0; it < sg.n_my; ++it) {
        const int trow = sg.r0 + kTileRows * (wave + it * 4);
        const int nvalid = min(kTileRows, sg.r1 - trow);
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const int row = 16 * pt + c16;
            const bool ok = row < nvalid;
            float w[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qq = 4 * g + r;
                const bool live = ok && qq < n_q;
                w[r] = live ? w1b[(size_t)qq * sg.m_rows + trow + row] : 0.f;
                w[4 + r] = (live && w2b) ? w2b[(size_t)qq * sg.m_rows + trow + row] : 0.f;
            }
            bf16x8 wh, wl;
            pack_hi_lo(w, wh, wl);
#pragma unroll
            for (int t = 0; t < G::DT; ++t) {
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
                o = mfma_bf16(zh[t], wh, o);
                o = mfma_bf16(zh[t], wl, o);
                o = mfma_bf16(zl[t], wh, o);
                if constexpr (!OUT_F32) {
                    bf16x4 ob = {f2bf(o[0]), f2bf(o[1]), f2bf(o[2]), f2bf(o[3])};
                    const int c = (2 * t + (g >> 1)) ^ ((row & 7) << 1);
                    *reinterpret_cast<bf16x4*>(img + row * G::ROWB + (c << 4) + 8 * (g & 1)) = ob;
                } else {
                    const int c = (4 * t + g) ^ ((row & 7) << 1);
                    *reinterpret_cast<f32x4*>(img + row * (E_ * 4) + (c << 4)) = o;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        constexpr int CH_PER_ROW = E_ * EB / 16;
        constexpr int NCH = kTileRows * CH_PER_ROW / 64;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ci = i * 64 + lane;
            const int r = ci / CH_PER_ROW, cc = ci % CH_PER_ROW;
            const f32x4 v = *reinterpret_cast<const f32x4*>(img + r * (E_ * EB) + ((cc ^ ((r & 7) << 1)) << 4));
            if (r < nvalid) *reinterpret_cast<f32x4*>(dslide + ((size_t)(trow + r) * CH_PER_ROW + cc) * 16) = v;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------ (iii'') patch-side gradient of K2 in one pass
//   G[m][e] = ( sum_n W1[n][m] Z1[n][e]  +  ADD[m][e] ) * (H[m][e] > 0 ? gate : 0)          bf16 bag
// with W1 = A_drop (ragged map), Z1 = dL/dctx, ADD = dK W_k from the library GEMM and H the bag itself
// (H = dropout(relu(pre)): its sign is the ReLU/dropout derivative of the layer that produced the bag, gate = 1/(1-p)).
// Replaces bag_outer + addmm's read-modify-write of dH + the element-wise derivative pass: one read of ADD and H,
// one write of G (which may alias ADD), column sums of G (= that layer's bias gradient) on the way out.
// The outer product is kept in fp32 in the LDS image until the sum is rounded once.
template <int E_>
__global__ __launch_bounds__(256, 1)
void bag_outer_gate_kernel(const int* __restrict__ cu, const float* __restrict__ w1, const float* __restrict__ z1,
                           const uint16_t* addend /* may alias out */, const uint16_t* __restrict__ hbag, uint16_t* out,
                           float gate /* 0: no gating */, float* __restrict__ part_colsum /* nullable [parts][E] */,
                           int n_q, BagPlan plan) {
    using G = TileGeom<E_>;
    constexpr int HR = 16;                                   // rows per step: one MFMA row block (half a 32-row tile)
    constexpr int IMG = HR * E_ * 4;
    constexpr int CH_PER_ROW = E_ / 8;                       // 16-byte chunks (8 bf16) per row
    constexpr int NCH = HR * CH_PER_ROW / 64;
    static_assert(CH_PER_ROW <= 64 && 64 % CH_PER_ROW == 0, "copy-out: a lane keeps one 8-column chunk");
    __shared__ __attribute__((aligned(16))) char lds[4 * IMG];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const SplitGeom sg = split_geom<4>(cu, plan, wave);
    const int b = sg.b;
    char* img = lds + wave * IMG;
    const int c16 = lane & 15, g = lane >> 4;
    const float* z1b = z1 + (size_t)b * n_q * E_;
    bf16x8 zh[G::DT], zl[G::DT];
#pragma unroll
    for (int t = 0; t < G::DT; ++t) {
        float z[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int qq = 4 * g + j;
            const int qc = qq < n_q ? qq : n_q - 1;
            z[j] = z1b[qc * E_ + 16 * t + c16] * (qq < n_q ? 1.0f : 0.0f);
            z[4 + j] = 0.f;
        }
        pack_hi_lo(z, zh[t], zl[t]);
    }
    const float* w1b = w1 + (size_t)n_q * sg.row_begin;
    const size_t slide_off = (size_t)sg.row_begin * E_;
    const f32x4* add4 = reinterpret_cast<const f32x4*>(addend + slide_off);
    const f32x4* h4 = reinterpret_cast<const f32x4*>(hbag + slide_off);
    f32x4* out4 = reinterpret_cast<f32x4*>(out + slide_off);
    f32x4 csum0 = {0.f, 0.f, 0.f, 0.f}, csum1 = csum0;       // column sums of the chunk this lane copies out
    const int n_steps = 2 * sg.n_my;
    auto step_row = [&](int st) { return sg.r0 + kTileRows * (wave + (st >> 1) * 4) + HR * (st & 1); };
    // map values one step ahead (clamped addresses; masked when used), ADD / H chunks of the current step at its start:
    // they are consumed after the MFMAs, and four waves per CU keep 64 KB in flight meanwhile
    f32x4 wn;
    auto fetch_w = [&](int st) {
        const int mrow = min(step_row(st) + c16, sg.m_rows - 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) wn[r] = w1b[(size_t)min(4 * g + r, n_q - 1) * sg.m_rows + mrow];
    };
    if (n_steps > 0) fetch_w(0);
    for (int st = 0; st < n_steps; ++st) {
        const int row0 = step_row(st);
        const int nvalid = max(0, min(HR, sg.r1 - row0));
        float w[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            w[r] = (c16 < nvalid && 4 * g + r < n_q) ? wn[r] : 0.f;
            w[4 + r] = 0.f;
        }
        fetch_w(st + 1 < n_steps ? st + 1 : st);
        f32x4 av[NCH], hv[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ci = i * 64 + lane;
            const int r = ci / CH_PER_ROW, cc = ci % CH_PER_ROW;
            const int grow = min(row0 + r, sg.m_rows - 1);
            av[i] = add4[(size_t)grow * CH_PER_ROW + cc];
            hv[i] = h4[(size_t)grow * CH_PER_ROW + cc];
        }
        {
            const int row = c16;
            bf16x8 wh, wl;
            pack_hi_lo(w, wh, wl);
#pragma unroll
            for (int t = 0; t < G::DT; ++t) {
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
                o = mfma_bf16(zh[t], wh, o);
                o = mfma_bf16(zh[t], wl, o);
                o = mfma_bf16(zl[t], wh, o);
                const int c = (4 * t + g) ^ ((row & 7) << 1);
                *reinterpret_cast<f32x4*>(img + row * (E_ * 4) + (c << 4)) = o;      // lane holds [row][16t + 4g .. +3]
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ci = i * 64 + lane;
            const int r = ci / CH_PER_ROW, cc = ci % CH_PER_ROW;
            // columns 8cc .. 8cc+7 = fp32 chunks 2cc and 2cc+1 (the swizzle leaves bit 0 alone: the pair stays adjacent)
            const char* rowp = img + r * (E_ * 4);
            const f32x4 o0 = *reinterpret_cast<const f32x4*>(rowp + (((2 * cc) ^ ((r & 7) << 1)) << 4));
            const f32x4 o1 = *reinterpret_cast<const f32x4*>(rowp + (((2 * cc + 1) ^ ((r & 7) << 1)) << 4));
            const bf16x8 ab = __builtin_bit_cast(bf16x8, av[i]);
            const bf16x8 hb = __builtin_bit_cast(bf16x8, hv[i]);
            bf16x8 ob;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = (j < 4 ? o0[j] : o1[j - 4]) + (float)ab[j];
                if (gate != 0.f) v *= (float)hb[j] > 0.f ? gate : 0.f;
                ob[j] = f2bf(v);
            }
            if (r < nvalid) {
                out4[(size_t)(row0 + r) * CH_PER_ROW + cc] = __builtin_bit_cast(f32x4, ob);
                csum0 += f32x4{(float)ob[0], (float)ob[1], (float)ob[2], (float)ob[3]};   // what the caller would sum
                csum1 += f32x4{(float)ob[4], (float)ob[5], (float)ob[6], (float)ob[7]};
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (part_colsum != nullptr) {
        __syncthreads();
#pragma unroll
        for (int o = CH_PER_ROW; o < 64; o <<= 1) {                      // lanes l, l + CH_PER_ROW, ... share a chunk
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                csum0[j] += __shfl_xor(csum0[j], o);
                csum1[j] += __shfl_xor(csum1[j], o);
            }
        }
        if (lane < CH_PER_ROW) {                                         // lane == column chunk
            *reinterpret_cast<f32x4*>(img + lane * 32) = csum0;
            *reinterpret_cast<f32x4*>(img + lane * 32 + 16) = csum1;
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < E_; idx += 256) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) a += reinterpret_cast<const float*>(lds + w * IMG)[idx];
            part_colsum[(size_t)sg.part * E_ + idx] = a;
        }
    }
}

// ------------------------------------------------------------------ gated (tanh on the fly) variants for K2
// tanh(x) = 1 - 2 / (2^(2x log2 e) + 1): two transcendentals, saturates cleanly, abs error ~1e-7.
__device__ __forceinline__ float fast_tanh(float v) {
    const float e = __builtin_amdgcn_exp2f(v * (2.0f * kLog2e));
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
// fragment of tanh(X) from the hi (+lo) images, split again into bf16 hi/lo
template <int NT>
__device__ __forceinline__ void tanh_frag(bf16x8 xh, bf16x8 xl, bf16x8& th, bf16x8& tl) {
    float t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = fast_tanh(NT == 2 ? (float)xh[j] + (float)xl[j] : (float)xh[j]);
    pack_hi_lo(t, th, tl);
}

// (i') ONE pass over an fp32 K:  a_map[n][m] = K[m] . r1[n]   and   g_map[n][m] = tanh(K[m]) . r2[n],
// exact: both products on the fp32-input MFMA, no operand splitting, so
// a_map / g_map carry plain fp32 rounding and the peaky-softmax parity bar of 1e-3 holds with margin.
// With N = 6 queries the 16x16x4 shape would idle 10 of its 16 columns (measured: 158.9 us, MFMA-bound);
// v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4x1 blocks, 8 cycles) keeps 6 of 8:
//   block b = lane/4 takes A from lanes 4b..4b+3 and B from its own lanes: D[lane][r] = A[4(lane/4)+r] * B[lane]
//   (lane mapping checked on the card by tools/probe_mfma4x4.hip).
// A wave owns a 16-row fp32 tile (raw rows, stride E*4 + 16 bytes: conflict-free ds_read_b128); lane
// (row = l%16, phase = l/16) contracts k = 16u + 4*phase + j at step (u, j), so one b128 read of the row feeds four
// steps, and the four phases are summed by two lane exchanges at the end of the tile.  B operands come from two
// per-slide LDS tables [k/4][column 0..7][4] (16-byte entries, broadcast reads): qs and -2*tq, with
// tanh(x) = 1 - 2r, r = 1/(2^(2x log2 e) + 1) folded as g = sum(tq) - 2 sum(r tq).  Eight waves per CU, tiles staged
// through registers (NOTES.md r02-K2: the direct-to-LDS form with 8-row tiles has half the bytes in flight and is slower).
template <int E_, int NG>                                    // NG column groups of four queries: 2 (N <= 8) or 4 (N <= 16)
struct GateCfg {
    static constexpr int WAVES = NG <= 2 ? 8 : 4;            // the wider operand tables leave room for four tiles
    static constexpr int PAIRS = WAVES / 2;                  // a wave pair shares one of the plan's 32-row tiles
    static constexpr int ROWS = 16;
    static constexpr int ROWB = E_ * 4 + 16;
    static constexpr int TILEB = ROWS * ROWB;
    static constexpr int NCOL = 4 * NG;
    static constexpr int TAB = (E_ / 4) * NCOL * 16;
    static constexpr int OFF_TAB = WAVES * TILEB;
    static constexpr int OFF_SUM = OFF_TAB + 2 * TAB;
    static constexpr int LDS_BYTES = OFF_SUM + 64;           // E = 256: 149 568 (NG = 2) / 132 160 (NG = 4) bytes
    static constexpr int CH = E_ / 4;                        // 16-byte chunks per row
    static constexpr int NLD = ROWS * CH / 64;               // global loads per lane per tile
    static constexpr int KU = E_ / 16;
};

template <int E_>
struct GateStage {
    static constexpr int ROWS = 16, ROWB = E_ * 4 + 16, CH = E_ / 4, NLD = ROWS * CH / 64;
    f32x4 v[NLD];
    __device__ __forceinline__ void load(const char* slide, int row0, int m_rows, int lane) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int ci = i * 64 + lane;
            int grow = row0 + ci / CH;
            grow = grow < m_rows ? grow : m_rows - 1;
            v[i] = *reinterpret_cast<const f32x4*>(slide + ((size_t)grow * CH + ci % CH) * 16);
        }
    }
    __device__ __forceinline__ void store(char* tile, int lane) const {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int ci = i * 64 + lane;
            *reinterpret_cast<f32x4*>(tile + (ci / CH) * ROWB + (ci % CH) * 16) = v[i];
        }
    }
};

__device__ __forceinline__ f32x4 mfma_4x4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 sum_phases(f32x4 v) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        v[r] += __shfl_xor(v[r], 16, 64);
        v[r] += __shfl_xor(v[r], 32, 64);
    }
    return v;
}
__device__ __forceinline__ float pick(f32x4 v, int p) {
    return p == 0 ? v[0] : p == 1 ? v[1] : p == 2 ? v[2] : v[3];
}

template <int E_, int NG>
__global__ __launch_bounds__((GateCfg<E_, NG>::WAVES * 64), 1)
void bag_rowdot_gated_exact_kernel(const float* __restrict__ bag, const int* __restrict__ cu,
                                   const float* __restrict__ r1, const float* __restrict__ r2,
                                   float* __restrict__ a_map, float* __restrict__ g_map, int n_q, BagPlan plan) {
    using C = GateCfg<E_, NG>;
    __shared__ __attribute__((aligned(16))) char lds[C::LDS_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const SplitGeom sg = split_geom<C::PAIRS>(cu, plan, wave >> 1);
    char* tile = lds + wave * C::TILEB;
    // operand tables of this slide: entry (k4, col) = r[col][4 k4 .. +3], zero for col >= n_q
    for (int idx = threadIdx.x; idx < (E_ / 4) * C::NCOL; idx += C::WAVES * 64) {
        const int k4 = idx / C::NCOL, col = idx % C::NCOL;
        f32x4 va = {0.f, 0.f, 0.f, 0.f}, vg = va;
        if (col < n_q) {
            va = *reinterpret_cast<const f32x4*>(r1 + ((size_t)sg.b * n_q + col) * E_ + 4 * k4);
            vg = *reinterpret_cast<const f32x4*>(r2 + ((size_t)sg.b * n_q + col) * E_ + 4 * k4) * -2.0f;
        }
        *reinterpret_cast<f32x4*>(lds + C::OFF_TAB + idx * 16) = va;
        *reinterpret_cast<f32x4*>(lds + C::OFF_TAB + C::TAB + idx * 16) = vg;
    }
    for (int col = wave * 8 + (lane >> 3); col < C::NCOL; col += C::WAVES * 8) {   // sum_e tq[col][e]: eight lanes per column
        const int part = lane & 7;
        float acc = 0.f;
        if (col < n_q)
            for (int e = part; e < E_; e += 8) acc += r2[((size_t)sg.b * n_q + col) * E_ + e];
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        if (part == 0) reinterpret_cast<float*>(lds + C::OFF_SUM)[col] = acc;
    }
    __syncthreads();
    const int row = lane & 15, ph = lane >> 4;
    const char* arow = tile + row * C::ROWB + 16 * ph;
    const char* btab = lds + C::OFF_TAB + (ph * C::NCOL + (lane & 3)) * 16;
    const char* slide = reinterpret_cast<const char*>(bag) + (size_t)sg.row_begin * E_ * 4;
    const int half = C::ROWS * (wave & 1);
    const int orow = 4 * ((lane >> 2) & 3) + ph;                       // the output row this lane stores
    float osum[NG];
#pragma unroll
    for (int c = 0; c < NG; ++c) osum[c] = reinterpret_cast<const float*>(lds + C::OFF_SUM)[4 * c + (lane & 3)];
    float* am = a_map + (size_t)n_q * sg.row_begin;
    float* gm = g_map + (size_t)n_q * sg.row_begin;
    GateStage<E_> st;
    if (sg.n_my > 0) st.load(slide, sg.r0 + kTileRows * (wave >> 1) + half, sg.m_rows, lane);
    for (int it = 0; it < sg.n_my; ++it) {
        const int trow = sg.r0 + kTileRows * ((wave >> 1) + it * C::PAIRS) + half;
        const int nvalid = min(C::ROWS, sg.r1 - trow);                 // may be <= 0 for the upper half of a last tile
        st.store(tile, lane);
        if (it + 1 < sg.n_my) st.load(slide, trow + kTileRows * C::PAIRS, sg.m_rows, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        f32x4 a[NG], g[NG];
#pragma unroll
        for (int c = 0; c < NG; ++c) a[c] = g[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < C::KU; ++u) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(arow + 64 * u);
            f32x4 qa[NG], qg[NG];
#pragma unroll
            for (int c = 0; c < NG; ++c) {
                qa[c] = *reinterpret_cast<const f32x4*>(btab + (4 * C::NCOL * 16) * u + 64 * c);
                qg[c] = *reinterpret_cast<const f32x4*>(btab + C::TAB + (4 * C::NCOL * 16) * u + 64 * c);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float r = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x[j] * (2.0f * kLog2e)) + 1.0f);
#pragma unroll
                for (int c = 0; c < NG; ++c) {
                    a[c] = mfma_4x4(x[j], qa[c][j], a[c]);
                    g[c] = mfma_4x4(r, qg[c][j], g[c]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < NG; ++c) {
            a[c] = sum_phases(a[c]);
            g[c] = sum_phases(g[c]);
            const int col = 4 * c + (lane & 3);
            if (orow < nvalid && col < n_q) {
                am[(size_t)col * sg.m_rows + trow + orow] = pick(a[c], ph);
                gm[(size_t)col * sg.m_rows + trow + orow] = pick(g[c], ph) + osum[c];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// (ii') ONE pass over K:  part1[n][e] = sum_m W1[n][m] K[m][e]   and   part2[n][e] = sum_m W2[n][m] tanh(K[m][e])
template <int E_, bool F32BAG>
__global__ __launch_bounds__((BagCfg<E_, F32BAG>::WAVES * 64), 1)
void bag_colacc_gated_kernel(const void* __restrict__ bag_, const int* __restrict__ cu, const float* __restrict__ w1map,
                             const float* __restrict__ w2map, float* __restrict__ part1, float* __restrict__ part2,
                             int n_q, BagPlan plan) {
    using G = TileGeom<E_>;
    using C = BagCfg<E_, F32BAG>;
    constexpr int NT = C::NT;
    __shared__ __attribute__((aligned(16))) char lds[C::LDS_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const SplitGeom sg = split_geom<C::WAVES>(cu, plan, wave);
    char* thi = lds + wave * C::WAVE_LDS;
    char* tlo = thi + (NT - 1) * G::TILEB;
    const int q = lane & 15, g = lane >> 4;
    const size_t mbase = (size_t)n_q * sg.row_begin + (size_t)(q < n_q ? q : 0) * sg.m_rows;
    const char* slide = reinterpret_cast<const char*>(bag_) + (size_t)sg.row_begin * E_ * (F32BAG ? 4 : 2);
    f32x4 acc1[G::DT], acc2[G::DT];
#pragma unroll
    for (int t = 0; t < G::DT; ++t) { acc1[t] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[t] = acc1[t]; }
    Stage<E_, F32BAG> st0, st1;
    if (sg.n_my > 0) {
        st0.load(slide, sg.r0 + kTileRows * wave, sg.m_rows, 0, lane);
        if constexpr (F32BAG) st1.load(slide, sg.r0 + kTileRows * wave, sg.m_rows, 1, lane);
    }
    for (int it = 0; it < sg.n_my; ++it) {
        const int trow = sg.r0 + kTileRows * (wave + it * C::WAVES);
        const int nvalid = min(kTileRows, sg.r1 - trow);
        float w1[8], w2[8];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const bool ok0 = q < n_q && 4 * g + rr < nvalid, ok1 = q < n_q && 16 + 4 * g + rr < nvalid;
            w1[rr] = ok0 ? w1map[mbase + trow + 4 * g + rr] : 0.f;
            w2[rr] = ok0 ? w2map[mbase + trow + 4 * g + rr] : 0.f;
            w1[4 + rr] = ok1 ? w1map[mbase + trow + 16 + 4 * g + rr] : 0.f;
            w2[4 + rr] = ok1 ? w2map[mbase + trow + 16 + 4 * g + rr] : 0.f;
        }
        st0.store(thi, tlo, 0, lane);
        if constexpr (F32BAG) st1.store(thi, tlo, 1, lane);
        if (it + 1 < sg.n_my) {
            st0.load(slide, trow + kTileRows * C::WAVES, sg.m_rows, 0, lane);
            if constexpr (F32BAG) st1.load(slide, trow + kTileRows * C::WAVES, sg.m_rows, 1, lane);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        bf16x8 w1h, w1l, w2h, w2l;
        pack_hi_lo(w1, w1h, w1l);
        pack_hi_lo(w2, w2h, w2l);
#pragma unroll
        for (int t = 0; t < G::DT; ++t) {
            const bf16x8 xh = col_frag<E_>(thi, t, lane);
            const bf16x8 xl = NT == 2 ? col_frag<E_>(tlo, t, lane) : xh;
            acc1[t] = mfma_bf16(xh, w1h, acc1[t]);
            acc1[t] = mfma_bf16(xh, w1l, acc1[t]);
            if (NT == 2) acc1[t] = mfma_bf16(xl, w1h, acc1[t]);
            bf16x8 th, tl;
            tanh_frag<NT>(xh, xl, th, tl);
            acc2[t] = mfma_bf16(th, w2h, acc2[t]);
            acc2[t] = mfma_bf16(th, w2l, acc2[t]);
            acc2[t] = mfma_bf16(tl, w2h, acc2[t]);
        }
        __builtin_amdgcn_wave_barrier();
    }
    const size_t pbase = sg.part;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
        {
            float* wq = reinterpret_cast<float*>(thi);
#pragma unroll
            for (int t = 0; t < G::DT; ++t)
                *reinterpret_cast<f32x4*>(wq + q * E_ + 16 * t + 4 * g) = pass == 0 ? acc1[t] : acc2[t];
        }
        __syncthreads();
        float* part = pass == 0 ? part1 : part2;
        for (int idx = threadIdx.x; idx < n_q * E_; idx += C::WAVES * 64) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < C::WAVES; ++w) a += reinterpret_cast<const float*>(lds + w * C::WAVE_LDS)[idx];
            part[pbase * n_q * E_ + idx] = a;
        }
    }
}

// (iii') dK[m][e] = sum_n W1[n][m] Z1[n][e] + (sum_n W2[n][m] Z2[n][e]) * (1 - tanh(K[m][e])^2),  K and dK fp32.
// The K tile is staged (coalesced) into the fp32 image that then receives dK: each slot is read (K) right before it
// is overwritten (dK), so tanh' costs no extra pass over the bag.
template <int E_, bool OUT_BF16>
__global__ __launch_bounds__(256, 1)
void bag_outer_gated_kernel(const float* __restrict__ kbag, const int* __restrict__ cu, const float* __restrict__ w1,
                            const float* __restrict__ z1, const float* __restrict__ w2, const float* __restrict__ z2,
                            void* __restrict__ dk, float* __restrict__ part_colsum /* nullable [parts][E] */, int n_q,
                            BagPlan plan) {
    using G = TileGeom<E_>;
    constexpr int HR = 16;                                   // rows per step: half of a 32-row tile (one MFMA row block)
    constexpr int IMG = HR * E_ * 4;
    constexpr int CH_PER_ROW = E_ / 4;
    constexpr int NCH = HR * CH_PER_ROW / 64;                // float4 per lane per step (16 for E = 256)
    // LDS: one fp32 half-tile image per wave + the per-slide Z fragments (hi/lo of both products), shared by the
    // waves.  As registers (256 per lane) the Z fragments left no room to prefetch the next K rows.
    constexpr int ZB = G::DT * 64 * 16;                      // one fragment set: [t][lane] x 16 bytes
    __shared__ __attribute__((aligned(16))) char lds[4 * IMG + 4 * ZB];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const SplitGeom sg = split_geom<4>(cu, plan, wave);
    const int b = sg.b;
    char* img = lds + wave * IMG;
    bf16x8* zb = reinterpret_cast<bf16x8*>(lds + 4 * IMG);   // [z1h | z1l | z2h | z2l][t][lane]
    const int c16 = lane & 15, g = lane >> 4;
    const float* z1b = z1 + (size_t)b * n_q * E_;
    const float* z2b = z2 + (size_t)b * n_q * E_;
    for (int t = wave; t < G::DT; t += 4) {
        float a[8], c[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int qq = 4 * g + (j & 3);
            const int qc = qq < n_q ? qq : n_q - 1;
            const float live = (qq < n_q && j < 4) ? 1.0f : 0.0f;       // k-slots 4..7 unused (zero)
            a[j] = z1b[qc * E_ + 16 * t + c16] * live;
            c[j] = z2b[qc * E_ + 16 * t + c16] * live;
        }
        bf16x8 ah, al, ch, cl;
        pack_hi_lo(a, ah, al);
        pack_hi_lo(c, ch, cl);
        zb[(0 * G::DT + t) * 64 + lane] = ah;
        zb[(1 * G::DT + t) * 64 + lane] = al;
        zb[(2 * G::DT + t) * 64 + lane] = ch;
        zb[(3 * G::DT + t) * 64 + lane] = cl;
    }
    __syncthreads();
    const float* w1b = w1 + (size_t)n_q * sg.row_begin;
    const float* w2b = w2 + (size_t)n_q * sg.row_begin;
    const float* kslide = kbag + (size_t)sg.row_begin * E_;
    char* dslide = reinterpret_cast<char*>(dk) + (size_t)sg.row_begin * E_ * (OUT_BF16 ? 2 : 4);
    // step st covers rows  tile(st >> 1) + 16 (st & 1);  the K rows of the NEXT step travel in registers meanwhile
    auto step_row = [&](int st) { return sg.r0 + kTileRows * (wave + (st >> 1) * 4) + HR * (st & 1); };
    const int n_steps = 2 * sg.n_my;
    // Everything a step needs from memory is loaded ONE STEP AHEAD, the two map columns first and the K rows after them,
    // with clamped addresses instead of predicates.  gfx950 counts loads and stores in one in-order counter: loading the
    // map values after the K prefetch (as the first version did) made their wait a wait for the whole prefetch, i.e.
    // load -> compute -> store with nothing overlapped.
    f32x4 kv[NCH];
    float wan[4], wbn[4];
    auto fetch = [&](int st) {
        st = st < n_steps ? st : n_steps - 1;
        const int row0 = step_row(st);
        int mrow = row0 + c16;
        mrow = mrow < sg.m_rows ? mrow : sg.m_rows - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int qq = 4 * g + j;
            const int qc = qq < n_q ? qq : n_q - 1;
            wan[j] = w1b[(size_t)qc * sg.m_rows + mrow];
            wbn[j] = w2b[(size_t)qc * sg.m_rows + mrow];
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ci = i * 64 + lane;
            const int r = ci / CH_PER_ROW, cc = ci % CH_PER_ROW;
            int grow = row0 + r;
            grow = grow < sg.m_rows ? grow : sg.m_rows - 1;
            kv[i] = *reinterpret_cast<const f32x4*>(kslide + (size_t)grow * E_ + cc * 4);
        }
    };
    static_assert(CH_PER_ROW <= 64 && 64 % CH_PER_ROW == 0, "copy-out: a lane keeps one 4-column chunk");
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};                       // column sums of the rows this lane copies out
    auto step = [&](int st, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;        // all HR rows exist: nothing in the step is predicated
        const int row0 = step_row(st);
        const int nvalid = FULL ? HR : max(0, min(HR, sg.r1 - row0));
        const int row = c16;
        float wa[8], wb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool live = row < nvalid && 4 * g + (j & 3) < n_q && j < 4;   // k-slots 4..7 unused (zero)
            wa[j] = live ? wan[j & 3] : 0.f;
            wb[j] = live ? wbn[j & 3] : 0.f;
        }
        // stage K (fp32) into the image, same chunk swizzle as the output image
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ci = i * 64 + lane;
            const int r = ci / CH_PER_ROW, cc = ci % CH_PER_ROW;
            *reinterpret_cast<f32x4*>(img + r * (E_ * 4) + ((cc ^ ((r & 7) << 1)) << 4)) = kv[i];
        }
        fetch(st + 1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (FULL || nvalid > 0) {
            bf16x8 wah, wal, wbh, wbl;
            pack_hi_lo(wa, wah, wal);
            pack_hi_lo(wb, wbh, wbl);
#pragma unroll
            for (int t = 0; t < G::DT; ++t) {
                const bf16x8 z1h = zb[(0 * G::DT + t) * 64 + lane], z1l = zb[(1 * G::DT + t) * 64 + lane];
                const bf16x8 z2h = zb[(2 * G::DT + t) * 64 + lane], z2l = zb[(3 * G::DT + t) * 64 + lane];
                f32x4 o1 = {0.f, 0.f, 0.f, 0.f}, o2 = o1;
                o1 = mfma_bf16(z1h, wah, o1);
                o1 = mfma_bf16(z1h, wal, o1);
                o1 = mfma_bf16(z1l, wah, o1);
                o2 = mfma_bf16(z2h, wbh, o2);
                o2 = mfma_bf16(z2h, wbl, o2);
                o2 = mfma_bf16(z2l, wbh, o2);
                const int c = (4 * t + g) ^ ((row & 7) << 1);
                f32x4* slot = reinterpret_cast<f32x4*>(img + row * (E_ * 4) + (c << 4));
                const f32x4 kvs = *slot;
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float tk = fast_tanh(kvs[j]);
                    o[j] = o1[j] + o2[j] * (1.0f - tk * tk);
                }
                *slot = o;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ci = i * 64 + lane;
            const int r = ci / CH_PER_ROW, cc = ci % CH_PER_ROW;
            const f32x4 v = *reinterpret_cast<const f32x4*>(img + r * (E_ * 4) + ((cc ^ ((r & 7) << 1)) << 4));
            if (FULL || r < nvalid) {
                if constexpr (OUT_BF16) {
                    bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    *reinterpret_cast<bf16x4*>(dslide + ((size_t)(row0 + r) * E_ + cc * 4) * 2) = o;
                    csum += f32x4{(float)o[0], (float)o[1], (float)o[2], (float)o[3]};      // what the caller will sum
                } else {
                    *reinterpret_cast<f32x4*>(dslide + ((size_t)(row0 + r) * E_ + cc * 4) * 4) = v;
                    csum += v;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    // hipcc's wait counts must hold on every path into a wait, so (a) full steps -- no predicated store, the count of
    // stores between a prefetch and its use is a constant -- run in their own loop, the ragged last steps after it, and
    // (b) the first step is peeled: the loop's entry edge (prefetch loads youngest) and its back edge (a step's stores
    // younger than the prefetch) would otherwise be joined conservatively, i.e. every step would wait for the previous
    // step's STORES to be acknowledged before staging the next rows.
    if (n_steps > 0) {
        using Full = std::integral_constant<bool, true>;
        using Guarded = std::integral_constant<bool, false>;
        int n_full = 0;
        while (n_full < n_steps && step_row(n_full) + HR <= sg.r1) ++n_full;
        fetch(0);
        int st = 0;
        if (n_full > 0) {
            step(0, Full());
            for (st = 1; st < n_full; ++st) step(st, Full());
        }
        for (; st < n_steps; ++st) step(st, Guarded());
    }
    if (part_colsum != nullptr) {
        __syncthreads();
#pragma unroll
        for (int o = CH_PER_ROW; o < 64; o <<= 1) {                      // lanes l, l + CH_PER_ROW, ... share a chunk
#pragma unroll
            for (int j = 0; j < 4; ++j) csum[j] += __shfl_xor(csum[j], o);
        }
        if (lane < CH_PER_ROW) *reinterpret_cast<f32x4*>(img + lane * 16) = csum;   // lane == column chunk
        __syncthreads();
        for (int idx = threadIdx.x; idx < E_; idx += 256) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) a += reinterpret_cast<const float*>(lds + w * IMG)[idx];
            part_colsum[(size_t)sg.part * E_ + idx] = a;
        }
    }
}

// (ii' + iii') for N <= 6 queries, ONE pass over K on the vector ALUs:
//   part1[n][e] = sum_m W1[n][m] K[m][e],   part2[n][e] = sum_m W2[n][m] tanh(K[m][e])          (the query-side gradient)
//   dK[m][e]    = sum_n W1[n][m] Z1[n][e] + (sum_n W2[n][m] Z2[n][e]) (1 - tanh(K[m][e])^2)     (+ its column sums)
// Both need K and the two maps and nothing of each other.  With six queries every product is SKINNY: per element of K
// 12 FMAs for dK and 12 for the accumulations -- as rank-6 MFMA work (k = 32 slots, six used, three split terms each) the
// same dK costs 96 matrix instructions per 16 rows, twice the cycles of the plain FMAs, and needs K staged through an LDS
// image to change layout.  Here a lane keeps its 4 columns: K arrives as one float4 per row (a wave reads whole rows),
// tanh is evaluated once, Z1 / Z2 / the accumulators of the lane's columns sit in registers, the step's map columns are
// published to a 1 KB per-wave LDS table and read back as broadcasts, dK leaves as one 8-byte (bf16) store per row.
// No image, no fragments, no matrix pipe: 220 registers, so 8 waves per workgroup (two per SIMD) cover each other's
// waits; the K rows of the next 16-row step are requested into the register a row has just left (ring of 16, distance
// 16 rows).  737 MB per 32 x 15 000 window in 146 us (5.06 TB/s) against 127 + 152 us for the two matrix-pipe passes; the
// same fusion on the image / MFMA kernel measured 213 us (the vector work of both passes behind one wave per SIMD).
// The ragged last step of a row range is a plain loop (predicated stores inside the unrolled ring made the compiler spill
// several hundred registers across its sixteen exec-mask branches).
template <int E_, bool OUT_BF16, int NQA>
__global__ __launch_bounds__(512, 1)
void bag_key_grad_kernel(const float* __restrict__ kbag, const int* __restrict__ cu, const float* __restrict__ w1,
                         const float* __restrict__ z1, const float* __restrict__ w2, const float* __restrict__ z2,
                         void* __restrict__ dk, float* __restrict__ part_colsum /* nullable [parts][E] */,
                         float* __restrict__ part1, float* __restrict__ part2, int n_q, BagPlan plan) {
    constexpr int WAVES = 8;
    constexpr int HR = 16;                                   // rows per step
    constexpr int CH_PER_ROW = E_ / 4;                       // 16-byte chunks per fp32 row
    constexpr int RPP = 64 / CH_PER_ROW;                     // rows a wave covers with one float4 per lane (1 at E = 256)
    constexpr int NCH = HR / RPP;                            // float4 per lane per step
    constexpr int WTAB = 2 * HR * 8;                         // floats per wave: [map][row][8 queries]
    static_assert(NQA >= 1 && NQA <= 8 && 64 % CH_PER_ROW == 0 && CH_PER_ROW <= 64, "geometry");
    constexpr int RED = 2 * 8 * E_ + E_;                     // per wave: [product][8][E] + [E] column sums
    __shared__ __attribute__((aligned(16))) float lds[WAVES * WTAB + WAVES * RED];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const SplitGeom sg = split_geom<WAVES>(cu, plan, wave);
    float* wt = lds + wave * WTAB;
    float* red = lds + WAVES * WTAB + wave * RED;
    const int cc = lane % CH_PER_ROW, rs = lane / CH_PER_ROW;
    const int b = sg.b;
    // the lane's columns of Z1 / Z2 (rows n >= n_q: any finite value, their map columns are zero)
    f32x4 zz1[NQA], zz2[NQA];
#pragma unroll
    for (int n = 0; n < NQA; ++n) {
        const int qc = n < n_q ? n : n_q - 1;
        zz1[n] = *reinterpret_cast<const f32x4*>(z1 + ((size_t)b * n_q + qc) * E_ + 4 * cc);
        zz2[n] = *reinterpret_cast<const f32x4*>(z2 + ((size_t)b * n_q + qc) * E_ + 4 * cc);
    }
    f32x4 acc1[NQA], acc2[NQA];
#pragma unroll
    for (int n = 0; n < NQA; ++n) { acc1[n] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[n] = acc1[n]; }
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};
    const float* w1b = w1 + (size_t)n_q * sg.row_begin;
    const float* w2b = w2 + (size_t)n_q * sg.row_begin;
    const float* kslide = kbag + (size_t)sg.row_begin * E_;
    char* dslide = reinterpret_cast<char*>(dk) + (size_t)sg.row_begin * E_ * (OUT_BF16 ? 2 : 4);
    // 16-row steps dealt to the waves one by one (not 32-row tiles: 59 tiles over 8 waves leave five waves idle for a whole
    // tile in the last round, 118 steps for half of one)
    const int n_units = sg.r1 > sg.r0 ? (sg.r1 - sg.r0 + HR - 1) / HR : 0;
    auto step_row = [&](int st) { return sg.r0 + HR * (wave + st * WAVES); };
    const int n_steps = wave < n_units ? (n_units - wave + WAVES - 1) / WAVES : 0;
    // map columns of a step: lane (m = lane & 15, quarter = lane >> 4) holds queries quarter and quarter + 4
    const int tm = lane & 15, tq = lane >> 4;
    float wn[4];                                             // W1[tq], W1[tq + 4], W2[tq], W2[tq + 4] at row tm of the NEXT step
    auto fetch_maps = [&](int st) {
        st = st < n_steps ? st : n_steps - 1;
        const int row0 = step_row(st);
        int mrow = row0 + tm;
        const bool live = mrow < sg.r1;
        mrow = mrow < sg.m_rows ? mrow : sg.m_rows - 1;
        const int qa = tq < n_q ? tq : n_q - 1, qb = tq + 4 < n_q ? tq + 4 : n_q - 1;
        const float a0 = w1b[(size_t)qa * sg.m_rows + mrow], a1 = w1b[(size_t)qb * sg.m_rows + mrow];
        const float b0 = w2b[(size_t)qa * sg.m_rows + mrow], b1 = w2b[(size_t)qb * sg.m_rows + mrow];
        wn[0] = live && tq < n_q ? a0 : 0.f;
        wn[1] = live && tq + 4 < n_q ? a1 : 0.f;
        wn[2] = live && tq < n_q ? b0 : 0.f;
        wn[3] = live && tq + 4 < n_q ? b1 : 0.f;
    };
    f32x4 kv[NCH];
    auto fetch_row = [&](int st, int i) {
        st = st < n_steps ? st : n_steps - 1;
        int grow = step_row(st) + i * RPP + rs;
        grow = grow < sg.m_rows ? grow : sg.m_rows - 1;
        kv[i] = *reinterpret_cast<const f32x4*>(kslide + (size_t)grow * E_ + 4 * cc);
    };
    // one row (r of the step, K values k) of the lane's 4 columns
    auto row_math = [&](int r, const f32x4& k, f32x4& o) {
        const f32x4 t = {fast_tanh(k[0]), fast_tanh(k[1]), fast_tanh(k[2]), fast_tanh(k[3])};
        const f32x4 wa0 = *reinterpret_cast<const f32x4*>(wt + (0 * HR + r) * 8);
        const f32x4 wb0 = *reinterpret_cast<const f32x4*>(wt + (1 * HR + r) * 8);
        f32x4 wa1 = wa0, wb1 = wb0;
        if constexpr (NQA > 4) {
            wa1 = *reinterpret_cast<const f32x4*>(wt + (0 * HR + r) * 8 + 4);
            wb1 = *reinterpret_cast<const f32x4*>(wt + (1 * HR + r) * 8 + 4);
        }
        f32x4 u = {0.f, 0.f, 0.f, 0.f}, v = u;
#pragma unroll
        for (int n = 0; n < NQA; ++n) {
            const float a = n < 4 ? wa0[n & 3] : wa1[n & 3], bb = n < 4 ? wb0[n & 3] : wb1[n & 3];
            acc1[n] += k * a;
            acc2[n] += t * bb;
            u += zz1[n] * a;
            v += zz2[n] * bb;
        }
        o = u + v * (1.0f - t * t);
    };
    auto row_store = [&](int grow, const f32x4& o) {
        if constexpr (OUT_BF16) {
            const bf16x4 ob = {(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3]};
            *reinterpret_cast<bf16x4*>(dslide + ((size_t)grow * E_ + 4 * cc) * 2) = ob;
            csum += f32x4{(float)ob[0], (float)ob[1], (float)ob[2], (float)ob[3]};      // what the caller will sum
        } else {
            *reinterpret_cast<f32x4*>(dslide + ((size_t)grow * E_ + 4 * cc) * 4) = o;
            csum += o;
        }
    };
    auto publish_maps = [&](int st) {
        wt[(0 * HR + tm) * 8 + tq] = wn[0];
        wt[(0 * HR + tm) * 8 + tq + 4] = wn[1];
        wt[(1 * HR + tm) * 8 + tq] = wn[2];
        wt[(1 * HR + tm) * 8 + tq + 4] = wn[3];
        fetch_maps(st + 1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // a step whose 16 rows all exist: nothing is predicated, the register ring runs
    auto step = [&](int st) {
        const int row0 = step_row(st);
        publish_maps(st);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int r = i * RPP + rs;
            const f32x4 k = kv[i];
            fetch_row(st + 1, i);                               // the register is free again: next step's row i
            f32x4 o;
            row_math(r, k, o);
            row_store(row0 + r, o);
            __builtin_amdgcn_sched_barrier(0);                  // one row's working set at a time (the scheduler otherwise hoists all 16)
        }
        __builtin_amdgcn_wave_barrier();                        // the table is rewritten by the next step
    };
    // the ragged last step(s) of a range: a plain loop over the rows that exist, each loaded where it is used (the ring's
    // requests for these steps were clamped into the slide and are dropped)
    auto ragged_step = [&](int st) {
        const int row0 = step_row(st);
        const int nvalid = max(0, min(HR, sg.r1 - row0));
        publish_maps(st);
        for (int r = rs; r < nvalid; r += RPP) {
            const f32x4 k = *reinterpret_cast<const f32x4*>(kslide + (size_t)(row0 + r) * E_ + 4 * cc);
            f32x4 o;
            row_math(r, k, o);
            row_store(row0 + r, o);
        }
        __builtin_amdgcn_wave_barrier();
    };
    if (n_steps > 0) {
        int n_full = 0;
        while (n_full < n_steps && step_row(n_full) + HR <= sg.r1) ++n_full;
        fetch_maps(0);
#pragma unroll
        for (int i = 0; i < NCH; ++i) fetch_row(0, i);
        int st = 0;
        for (; st < n_full; ++st) step(st);
        for (; st < n_steps; ++st) ragged_step(st);
    }
    // lanes l, l + CH_PER_ROW, ... hold the same column chunk (other rows); then the four waves through LDS in a fixed order
#pragma unroll
    for (int o = CH_PER_ROW; o < 64; o <<= 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            csum[j] += __shfl_xor(csum[j], o);
#pragma unroll
            for (int n = 0; n < NQA; ++n) { acc1[n][j] += __shfl_xor(acc1[n][j], o); acc2[n][j] += __shfl_xor(acc2[n][j], o); }
        }
    }
    if (lane < CH_PER_ROW) {
#pragma unroll
        for (int n = 0; n < NQA; ++n) {
            *reinterpret_cast<f32x4*>(red + (0 * 8 + n) * E_ + 4 * lane) = acc1[n];
            *reinterpret_cast<f32x4*>(red + (1 * 8 + n) * E_ + 4 * lane) = acc2[n];
        }
        *reinterpret_cast<f32x4*>(red + 2 * 8 * E_ + 4 * lane) = csum;
    }
    __syncthreads();
    const float* all = lds + WAVES * WTAB;
    for (int idx = threadIdx.x; idx < n_q * E_; idx += WAVES * 64) {
        const int n = idx / E_, e = idx % E_;
        float a = 0.f, bsum = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            a += all[w * RED + (0 * 8 + n) * E_ + e];
            bsum += all[w * RED + (1 * 8 + n) * E_ + e];
        }
        part1[(size_t)sg.part * n_q * E_ + idx] = a;
        part2[(size_t)sg.part * n_q * E_ + idx] = bsum;
    }
    if (part_colsum != nullptr)
        for (int idx = threadIdx.x; idx < E_; idx += WAVES * 64) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) a += all[w * RED + 2 * 8 * E_ + idx];
            part_colsum[(size_t)sg.part * E_ + idx] = a;
        }
}

// ------------------------------------------------------------------ map kernels (one workgroup per (query, slide))
constexpr int kMapThreads = 512;                               // 8 waves per (query, slide) row
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int w = 1; w < kMapThreads / 64; ++w) r = fmaxf(r, red[w]);
    return r;
}
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int w = 1; w < kMapThreads / 64; ++w) r += red[w];
    return r;
}

// The map kernels walk a (query, slide) row of the ragged map in ALIGNED GROUPS of four elements (absolute index
// 4G .. 4G+3): one float4 access per array and ONE draw per group (the dropout counter is index >> 2), with
// the group's first/last elements masked at the row ends.  (Element-wise they drew once per element: 53 / 106 us.)
struct MapRow {
    size_t base;       // absolute index of the row's first element
    int m_rows;
    size_t g0;         // first group
    int n_groups;
    __device__ __forceinline__ MapRow(const int* cu, int n_q, int q, int b) {
        const int row_begin = cu[b];
        m_rows = cu[b + 1] - row_begin;
        base = (size_t)n_q * row_begin + (size_t)q * m_rows;
        g0 = base >> 2;
        n_groups = (int)(((base + m_rows + 3) >> 2) - g0);
    }
    // lanes of group G that belong to the row: bit j set <=> element 4G + j is inside
    __device__ __forceinline__ unsigned live(size_t G) const {
        unsigned m = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t i = 4 * G + j;
            if (i >= base && i < base + m_rows) m |= 1u << j;
        }
        return m;
    }
};
__device__ __forceinline__ f32x4 map_load4(const float* __restrict__ p, size_t G, unsigned live) {
    if (live == 0xFu) return *reinterpret_cast<const f32x4*>(p + 4 * G);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) if (live >> j & 1) v[j] = p[4 * G + j];
    return v;
}
__device__ __forceinline__ void map_store4(float* __restrict__ p, size_t G, unsigned live, const f32x4& v) {
    if (live == 0xFu) { *reinterpret_cast<f32x4*>(p + 4 * G) = v; return; }
#pragma unroll
    for (int j = 0; j < 4; ++j) if (live >> j & 1) p[4 * G + j] = v[j];
}
// keep-scales of the four elements of group G (same draw as dropout_keep(seed, offset, 4G + j, ...))
__device__ __forceinline__ f32x4 map_keep4(unsigned long long seed, unsigned long long offset, size_t G, float p, float inv_keep) {
    const unsigned long long ctr = offset + G;
    const uint4 r = draw4x32((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
    f32x4 k;
#pragma unroll
    for (int j = 0; j < 4; ++j) k[j] = (float)(w[j] >> 8) * (1.0f / 16777216.0f) >= p ? inv_keep : 0.0f;
    return k;
}

// a: log2-unit half-logits (qs2.k), g: gate dot (tq.tk).  S2 = a (g + 1).
// Writes lse2[b][q], the (post-dropout) map A_drop in place of `amap`, and asum[b][q] = sum_m A_drop.
__global__ __launch_bounds__(kMapThreads)
void gated_softmax_fwd_kernel(const float* __restrict__ amap_a, const float* __restrict__ gmap, const int* __restrict__ cu,
                              float* __restrict__ out_map, float* __restrict__ lse2, float* __restrict__ asum,
                              int n_q, float drop_p, unsigned long long seed, unsigned long long offset_,
                              const unsigned long long* epoch) {
    __shared__ float red[kMapThreads / 64];
    const unsigned long long offset = epoch_offset(offset_, epoch);
    const int q = blockIdx.x, b = blockIdx.y;
    const MapRow row(cu, n_q, q, b);
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < row.n_groups; i += kMapThreads) {
        const size_t G = row.g0 + i;
        const unsigned lv = row.live(G);
        const f32x4 a = map_load4(amap_a, G, lv), g = map_load4(gmap, G, lv);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (lv >> j & 1) mx = fmaxf(mx, a[j] * (g[j] + 1.0f));
    }
    mx = block_max(mx, red);
    float l = 0.f;
    for (int i = threadIdx.x; i < row.n_groups; i += kMapThreads) {
        const size_t G = row.g0 + i;
        const unsigned lv = row.live(G);
        const f32x4 a = map_load4(amap_a, G, lv), g = map_load4(gmap, G, lv);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (lv >> j & 1) l += __builtin_amdgcn_exp2f(a[j] * (g[j] + 1.0f) - mx);
    }
    l = block_sum(l, red);
    const float lse = mx + __builtin_amdgcn_logf(l);
    const float inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    float s = 0.f;
    for (int i = threadIdx.x; i < row.n_groups; i += kMapThreads) {
        const size_t G = row.g0 + i;
        const unsigned lv = row.live(G);
        const f32x4 a = map_load4(amap_a, G, lv), g = map_load4(gmap, G, lv);
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = __builtin_amdgcn_exp2f(a[j] * (g[j] + 1.0f) - lse);
        if (drop_p > 0.f) {
            const f32x4 k = map_keep4(seed, offset, G, drop_p, inv_keep);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= k[j];
        }
        map_store4(out_map, G, lv, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (lv >> j & 1) s += v[j];
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        lse2[(size_t)b * n_q + q] = lse;
        asum[(size_t)b * n_q + q] = s;
    }
}

// Backward of the gated softmax.  da_map holds dctx.H (from bag_rowdot) on entry; d_ext (nullable) is the
// gradient arriving on the returned (post-dropout) map; dasum[b][q] the gradient of the row sums.
// On exit: ds1_map[n][m] = dS (g+1)/2 (natural units, for q~.k) and dg_map[n][m] = dS * s1/2.
__global__ __launch_bounds__(kMapThreads)
void gated_softmax_bwd_kernel(const float* __restrict__ amap_a, const float* __restrict__ gmap, const int* __restrict__ cu,
                              const float* __restrict__ lse2, const float* __restrict__ dasum,
                              const float* __restrict__ d_ext, float* __restrict__ da_map /* in: dctx.H, out: ds1 */,
                              float* __restrict__ dg_map, int n_q, float drop_p, unsigned long long seed,
                              unsigned long long offset_, const unsigned long long* epoch) {
    __shared__ float red[kMapThreads / 64];
    const unsigned long long offset = epoch_offset(offset_, epoch);
    const int q = blockIdx.x, b = blockIdx.y;
    const MapRow row(cu, n_q, q, b);
    const float lse = lse2[(size_t)b * n_q + q];
    const float das = dasum[(size_t)b * n_q + q];
    const float inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    float delta = 0.f;
    for (int i = threadIdx.x; i < row.n_groups; i += kMapThreads) {
        const size_t G = row.g0 + i;
        const unsigned lv = row.live(G);
        const f32x4 ah = map_load4(amap_a, G, lv), gg = map_load4(gmap, G, lv), dd = map_load4(da_map, G, lv);
        const f32x4 de = d_ext ? map_load4(d_ext, G, lv) : f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 ks = drop_p > 0.f ? map_keep4(seed, offset, G, drop_p, inv_keep) : f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (lv >> j & 1) {
                const float a = __builtin_amdgcn_exp2f(ah[j] * (gg[j] + 1.0f) - lse);
                delta += a * ks[j] * (dd[j] + das + de[j]);
            }
        }
    }
    delta = block_sum(delta, red);
    for (int i = threadIdx.x; i < row.n_groups; i += kMapThreads) {
        const size_t G = row.g0 + i;
        const unsigned lv = row.live(G);
        const f32x4 ah = map_load4(amap_a, G, lv), gg = map_load4(gmap, G, lv), dd = map_load4(da_map, G, lv);
        const f32x4 de = d_ext ? map_load4(d_ext, G, lv) : f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 ks = drop_p > 0.f ? map_keep4(seed, offset, G, drop_p, inv_keep) : f32x4{1.f, 1.f, 1.f, 1.f};
        f32x4 o1, o2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = __builtin_amdgcn_exp2f(ah[j] * (gg[j] + 1.0f) - lse);
            const float ds = a * (ks[j] * (dd[j] + das + de[j]) - delta);
            o1[j] = ds * (gg[j] + 1.0f) * 0.5f;      // d/d(q~.k)
            o2[j] = ds * ah[j] * kLn2;               // dS * s1/2 with s1/2 = ah / log2(e)
        }
        map_store4(da_map, G, lv, o1);
        map_store4(dg_map, G, lv, o2);
    }
}

// elementwise over a bag-shaped tensor: y = tanh(x)  /  dx += dy * (1 - y^2)
template <typename T>
__global__ void bag_tanh_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = (float)x[i];
        const float e = __builtin_amdgcn_exp2f(v * (2.0f * kLog2e));
        y[i] = (T)(1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f));
    }
}
template <typename T>
__global__ void bag_tanh_bwd_kernel(const T* __restrict__ y, const T* __restrict__ dy, T* __restrict__ dx, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float t = (float)y[i];
        dx[i] = (T)((float)dx[i] + (float)dy[i] * (1.0f - t * t));
    }
}

// q-side preparation: q [R][E] -> qt = q / sqrt(E), qs2 = qt * log2e / 2, tq = tanh(q)
__global__ void qprep_kernel(const float* __restrict__ q, float* __restrict__ qt, float* __restrict__ qs2,
                             float* __restrict__ tq, int n, float c_nat) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = q[i];
    qt[i] = v * c_nat;
    qs2[i] = v * c_nat * (0.5f * kLog2e);
    tq[i] = tanhf(v);
}
// dq = dqt * c_nat + dtq * (1 - tq^2) [+ d_ext]
__global__ void qprep_bwd_kernel(const float* __restrict__ dqt, const float* __restrict__ dtq, const float* __restrict__ tq,
                                 const float* __restrict__ d_ext, float* __restrict__ dq, int n, float c_nat) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float t = tq[i];
    dq[i] = dqt[i] * c_nat + dtq[i] * (1.0f - t * t) + (d_ext ? d_ext[i] : 0.f);
}

// y[r][:] += s[r] * b[:]      (value-bias term  b_v * sum_m A_drop)
__global__ void row_scaled_bias_kernel(float* __restrict__ y, const float* __restrict__ s, const float* __restrict__ bias,
                                       int rows, int cols) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    y[i] += s[i / cols] * bias[i % cols];
}

// patch-layer epilogue: h = drop(relu(h + bias)), bf16 in place, 8 elements (16 bytes) per lane.
// The launch guarantees (total threads) % (cols / 8) == 0, so a thread meets the same 8 columns on every
// grid-stride iteration and keeps their biases in registers (8 scalar, poorly coalesced bias loads per
// iteration made the first version 3x slower than a plain element-wise pass).
__global__ void bias_relu_dropout_bf16_kernel(bf16x8* __restrict__ h, const float* __restrict__ bias, size_t n8, int cols,
                                              float drop_p, unsigned long long seed, unsigned long long offset_,
                                              const unsigned long long* epoch) {
    const float inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    const unsigned long long offset = epoch_offset(offset_, epoch);
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c0 = (int)((tid * 8) % (size_t)cols);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + c0), b1 = *reinterpret_cast<const f32x4*>(bias + c0 + 4);
    const float bv[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    const uint32_t thr = (uint32_t)(drop_p * 65536.0f);
    for (size_t i = tid; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        bf16x8 v = h[i];
        // one Philox call per 8 elements: 16 random bits each (keep iff u16 >= p * 65536)
        uint4 r0 = {0, 0, 0, 0};
        if (drop_p > 0.f)
            r0 = philox4x32((uint32_t)(offset + i), (uint32_t)((offset + i) >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
        const uint32_t rw[4] = {r0.x, r0.y, r0.z, r0.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float x = fmaxf((float)v[j] + bv[j], 0.f);
            if (drop_p > 0.f) x = (((rw[j >> 1] >> (16 * (j & 1))) & 0xFFFFu) >= thr) ? x * inv_keep : 0.f;
            v[j] = (__bf16)x;
        }
        h[i] = v;
    }
}
// g = dy * (h > 0 ? 1/(1-p) : 0).  part_colsum (nullable, [gridDim.x][cols]): per-workgroup column sums of g -- the bias
// gradient of the layer -- from the same pass (a thread keeps one 8-column group: the grid stride is a multiple of a row).
__global__ __launch_bounds__(256)
void relu_dropout_bwd_bf16_kernel(const bf16x8* __restrict__ h, const bf16x8* __restrict__ dy, bf16x8* __restrict__ g,
                                  size_t n8, float inv_keep, int cols, float* __restrict__ part_colsum) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const bf16x8 hv = h[i], d = dy[i];
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (float)hv[j] > 0.f ? (__bf16)((float)d[j] * inv_keep) : (__bf16)0.f;
        g[i] = o;
        if (part_colsum != nullptr) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (float)o[j];
        }
    }
    if (part_colsum != nullptr) {
        __shared__ float red[256][9];
        const int tpr = cols / 8, c8 = threadIdx.x % tpr, rl = threadIdx.x / tpr;
#pragma unroll
        for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = acc[j];
        __syncthreads();
        if (rl == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = 0.f;
                for (int k = 0; k < 256 / tpr; ++k) t += red[k * tpr + c8][j];
                part_colsum[(size_t)blockIdx.x * cols + 8 * c8 + j] = t;
            }
        }
    }
}

// out[c] = sum_r x[r][c] over a bf16 [rows][cols] tensor (cols = 8 * a divisor of 256): the patch layer's bias gradient.
// Each thread owns 8 fixed columns (16-byte loads), workgroups take row chunks, fp32 atomics merge them.
__global__ __launch_bounds__(256)
void colsum_bf16_kernel(const bf16x8* __restrict__ x, float* __restrict__ out, size_t rows, int cols) {
    const int tpr = cols / 8;                       // threads per row
    const int rpb = 256 / tpr;                      // rows per block-iteration
    const int c8 = threadIdx.x % tpr, rl = threadIdx.x / tpr;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (size_t r = (size_t)blockIdx.x * rpb + rl; r < rows; r += (size_t)gridDim.x * rpb) {
        const bf16x8 v = x[r * tpr + c8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
    }
    __shared__ float red[256][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = acc[j];
    __syncthreads();
    if (rl == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = 0.f;
            for (int k = 0; k < rpb; ++k) t += red[k * tpr + c8][j];
            atomicAdd(out + 8 * c8 + j, t);
        }
    }
}

// torch.optim.Adam semantics (L2 weight decay folded into the gradient), one pass over the flat buffers
__global__ void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                 size_t n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                 const int* __restrict__ step_dev) {
    if (step_dev) {                                     // graph replay: the step count lives on the device
        const float t = (float)(*step_dev);
        bc1 = 1.0f - powf(b1, t);
        bc2_sqrt = sqrtf(1.0f - powf(b2, t));
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float pi = p[i];
        const float gi = g[i] + wd * pi;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
    }
}

}  // namespace

int mpo_launch_adam_flat(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                         float wd, int step, const int* step_dev, hipStream_t stream) {
    const float bc1 = 1.0f - powf(b1, (float)step), bc2s = sqrtf(1.0f - powf(b2, (float)step));
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    adam_flat_kernel<<<blocks, 256, 0, stream>>>(p, g, m, v, n, lr, b1, b2, eps, wd, bc1, bc2s, step_dev);
    MPO_LAUNCH_CHECK();
    return 0;
}

int mpo_launch_colsum_bf16(const void* x, float* out, size_t rows, int cols, hipStream_t stream) {
    MPO_CHECK(cols % 8 == 0 && 256 % (cols / 8) == 0, "bf16 column sum: width %d must be 8 * a divisor of 256", cols);
    MPO_HIP(hipMemsetAsync(out, 0, (size_t)cols * sizeof(float), stream));
    const size_t rpb = 256 / (cols / 8);
    size_t blocks = (rows + rpb - 1) / rpb;
    if (blocks > 2048) blocks = 2048;
    colsum_bf16_kernel<<<(int)blocks, 256, 0, stream>>>((const bf16x8*)x, out, rows, cols);
    MPO_LAUNCH_CHECK();
    return 0;
}

int mpo_launch_bias_relu_dropout_bf16(void* h, const float* bias, size_t rows, int cols, float drop_p,
                                      unsigned long long seed, unsigned long long offset, const unsigned long long* epoch,
                                      hipStream_t stream) {
    MPO_CHECK(cols % 8 == 0 && 256 % (cols / 8) == 0, "patch epilogue: width %d must be 8 * a divisor of 256", cols);
    const size_t n8 = rows * (size_t)cols / 8;
    const int blocks = (int)((n8 + 255) / 256 < 8192 ? (n8 + 255) / 256 : 8192);   // blocks * 256 is a multiple of cols / 8
    bias_relu_dropout_bf16_kernel<<<blocks, 256, 0, stream>>>((bf16x8*)h, bias, n8, cols, drop_p, seed, offset, epoch);
    MPO_LAUNCH_CHECK();
    return 0;
}
int mpo_relu_dropout_bwd_blocks(size_t n, int with_colsum) {
    const size_t n8 = n / 8;
    const size_t cap = with_colsum ? 512 : 8192;           // column sums: fewer, longer workgroups (one partial row each)
    return (int)((n8 + 255) / 256 < cap ? (n8 + 255) / 256 : cap);
}
int mpo_launch_relu_dropout_bwd_bf16(const void* h, const void* dy, void* g, size_t n, float drop_p, int cols,
                                     float* part_colsum /* nullable [blocks][cols] */, hipStream_t stream) {
    MPO_CHECK(n % 8 == 0, "patch epilogue backward: %zu elements not a multiple of 8", n);
    MPO_CHECK(!part_colsum || (cols >= 8 && cols % 8 == 0 && 256 % (cols / 8) == 0 && n % (size_t)cols == 0),
              "patch epilogue backward: column sums need cols in {8,..,2048} dividing 2048 (got %d)", cols);
    const size_t n8 = n / 8;
    const int blocks = mpo_relu_dropout_bwd_blocks(n, part_colsum != nullptr);
    relu_dropout_bwd_bf16_kernel<<<blocks, 256, 0, stream>>>((const bf16x8*)h, (const bf16x8*)dy, (bf16x8*)g, n8,
                                                             drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f, cols, part_colsum);
    MPO_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------- host launchers
#define MPO_E_SWITCH(embed, CALL)                                                     \
    switch (embed) {                                                                  \
        case 128: { constexpr int EV = 128; CALL; } break;                            \
        case 256: { constexpr int EV = 256; CALL; } break;                            \
        default: mpo_set_error("bag kernels: embed_dim %d not in {128,256}", embed); return 1; \
    }

int mpo_launch_bag_rowdot(const void* bag, int bag_f32, const int* cu, int n_slides, int embed, const float* r,
                          float* map, float alpha, int n_q, const BagPlan& plan, hipStream_t stream) {
    (void)n_slides;
    dim3 grid = plan_grid(plan);
    if (bag_f32) {
        MPO_E_SWITCH(embed, (bag_rowdot_kernel<EV, true><<<grid, BagCfg<EV, true>::WAVES * 64, 0, stream>>>(bag, cu, r, map, alpha, n_q, plan)))
    } else {
        MPO_E_SWITCH(embed, (bag_rowdot_kernel<EV, false><<<grid, BagCfg<EV, false>::WAVES * 64, 0, stream>>>(bag, cu, r, map, alpha, n_q, plan)))
    }
    MPO_LAUNCH_CHECK();
    return 0;
}

int mpo_launch_bag_colacc(const void* bag, int bag_f32, const int* cu, int n_slides, int embed, const float* wmap,
                          float* part, int n_q, const BagPlan& plan, hipStream_t stream) {
    (void)n_slides;
    dim3 grid = plan_grid(plan);
    if (bag_f32) {
        MPO_E_SWITCH(embed, (bag_colacc_kernel<EV, true><<<grid, BagCfg<EV, true>::WAVES * 64, 0, stream>>>(bag, cu, wmap, part, n_q, plan)))
    } else {
        MPO_E_SWITCH(embed, (bag_colacc_kernel<EV, false><<<grid, BagCfg<EV, false>::WAVES * 64, 0, stream>>>(bag, cu, wmap, part, n_q, plan)))
    }
    MPO_LAUNCH_CHECK();
    return 0;
}

int mpo_launch_bag_outer(const int* cu, int n_slides, int embed, const float* w1, const float* z1, const float* w2,
                         const float* z2, void* dx, int out_f32, int n_q, const BagPlan& plan, hipStream_t stream) {
    (void)n_slides;
    dim3 grid = plan_grid(plan);
    if (out_f32) {
        MPO_E_SWITCH(embed, (bag_outer_kernel<EV, true><<<grid, 256, 0, stream>>>(cu, w1, z1, w2, z2, dx, n_q, plan)))
    } else {
        MPO_E_SWITCH(embed, (bag_outer_kernel<EV, false><<<grid, 256, 0, stream>>>(cu, w1, z1, w2, z2, dx, n_q, plan)))
    }
    MPO_LAUNCH_CHECK();
    return 0;
}

int mpo_launch_bag_outer_gate(const int* cu, int n_slides, int embed, const float* w1, const float* z1, const void* addend,
                              const void* hbag, void* out, float gate, float* part_colsum, int n_q, const BagPlan& plan,
                              hipStream_t stream) {
    (void)n_slides;
    dim3 grid = plan_grid(plan);
    MPO_E_SWITCH(embed, (bag_outer_gate_kernel<EV><<<grid, 256, 0, stream>>>(cu, w1, z1, static_cast<const uint16_t*>(addend),
                                                                            static_cast<const uint16_t*>(hbag),
                                                                            static_cast<uint16_t*>(out), gate, part_colsum, n_q, plan)))
    MPO_LAUNCH_CHECK();
    return 0;
}

int mpo_launch_bag_rowdot_gated(const void* bag, int bag_f32, const int* cu, int n_slides, int embed, const float* r1,
                                const float* r2, float* a_map, float* g_map, int n_q, const BagPlan& plan, hipStream_t stream) {
    (void)n_slides;
    dim3 grid = plan_grid(plan);
    MPO_CHECK(bag_f32, "bag_rowdot_gated: the key bag is fp32 (K2 projects bf16 bags into fp32 keys first)");
    MPO_CHECK(n_q >= 1 && n_q <= 16, "bag_rowdot_gated: 1..16 queries (got %d)", n_q);
    if (n_q <= 8) {
        MPO_E_SWITCH(embed, (bag_rowdot_gated_exact_kernel<EV, 2><<<grid, GateCfg<EV, 2>::WAVES * 64, 0, stream>>>(static_cast<const float*>(bag), cu, r1, r2, a_map, g_map, n_q, plan)))
    } else {
        MPO_E_SWITCH(embed, (bag_rowdot_gated_exact_kernel<EV, 4><<<grid, GateCfg<EV, 4>::WAVES * 64, 0, stream>>>(static_cast<const float*>(bag), cu, r1, r2, a_map, g_map, n_q, plan)))
    }
    MPO_LAUNCH_CHECK();
    return 0;
}

int mpo_launch_bag_colacc_gated(const void* bag, int bag_f32, const int* cu, int n_slides, int embed, const float* w1map,
                                const float* w2map, float* part1, float* part2, int n_q, const BagPlan& plan,
                                hipStream_t stream) {
    (void)n_slides;
    dim3 grid = plan_grid(plan);
    if (bag_f32) {
        MPO_E_SWITCH(embed, (bag_colacc_gated_kernel<EV, true><<<grid, BagCfg<EV, true>::WAVES * 64, 0, stream>>>(bag, cu, w1map, w2map, part1, part2, n_q, plan)))
    } else {
        MPO_E_SWITCH(embed, (bag_colacc_gated_kernel<EV, false><<<grid, BagCfg<EV, false>::WAVES * 64, 0, stream>>>(bag, cu, w1map, w2map, part1, part2, n_q, plan)))
    }
    MPO_LAUNCH_CHECK();
    return 0;
}

int mpo_launch_bag_key_grad(const float* kbag, const int* cu, int n_slides, int embed, const float* w1, const float* z1,
                            const float* w2, const float* z2, void* dk, int dk_f32, float* part_colsum, float* part1,
                            float* part2, int n_q, const BagPlan& plan, hipStream_t stream) {
    (void)n_slides;
    dim3 grid = plan_grid(plan);
    if (part1 == nullptr || part2 == nullptr || n_q < 1 || n_q > 6) {
        mpo_set_error("bag key-gradient pass: needs both partial buffers and 1 <= n_q <= 6 (got %d)", n_q);
        return 1;
    }
#define MPO_KG(F32OUT, NQA_) MPO_E_SWITCH(embed, (bag_key_grad_kernel<EV, !(F32OUT), NQA_><<<grid, 512, 0, stream>>>(kbag, cu, w1, z1, w2, z2, dk, part_colsum, part1, part2, n_q, plan)))
    if (n_q <= 4) {
        if (dk_f32) { MPO_KG(true, 4) } else { MPO_KG(false, 4) }
    } else {
        if (dk_f32) { MPO_KG(true, 6) } else { MPO_KG(false, 6) }
    }
#undef MPO_KG
    MPO_LAUNCH_CHECK();
    return 0;
}

int mpo_launch_bag_outer_gated(const float* kbag, const int* cu, int n_slides, int embed, const float* w1, const float* z1,
                               const float* w2, const float* z2, void* dk, int dk_f32, float* part_colsum, int n_q,
                               const BagPlan& plan, hipStream_t stream) {
    (void)n_slides;
    dim3 grid = plan_grid(plan);
    if (dk_f32) {
        MPO_E_SWITCH(embed, (bag_outer_gated_kernel<EV, false><<<grid, 256, 0, stream>>>(kbag, cu, w1, z1, w2, z2, dk, part_colsum, n_q, plan)))
    } else {
        MPO_E_SWITCH(embed, (bag_outer_gated_kernel<EV, true><<<grid, 256, 0, stream>>>(kbag, cu, w1, z1, w2, z2, dk, part_colsum, n_q, plan)))
    }
    MPO_LAUNCH_CHECK();
    return 0;
}

int mpo_launch_gated_softmax_fwd(const float* amap_a, const float* gmap, const int* cu, float* out_map, float* lse2,
                                 float* asum, int n_slides, int n_q, float drop_p, unsigned long long seed,
                                 unsigned long long offset, const unsigned long long* epoch, hipStream_t stream) {
    gated_softmax_fwd_kernel<<<dim3(n_q, n_slides), kMapThreads, 0, stream>>>(amap_a, gmap, cu, out_map, lse2, asum, n_q, drop_p, seed, offset, epoch);
    MPO_LAUNCH_CHECK();
    return 0;
}

int mpo_launch_gated_softmax_bwd(const float* amap_a, const float* gmap, const int* cu, const float* lse2,
                                 const float* dasum, const float* d_ext, float* da_map, float* dg_map, int n_slides,
                                 int n_q, float drop_p, unsigned long long seed, unsigned long long offset,
                                 const unsigned long long* epoch, hipStream_t stream) {
    gated_softmax_bwd_kernel<<<dim3(n_q, n_slides), kMapThreads, 0, stream>>>(amap_a, gmap, cu, lse2, dasum, d_ext, da_map, dg_map, n_q,
                                                                      drop_p, seed, offset, epoch);
    MPO_LAUNCH_CHECK();
    return 0;
}

int mpo_launch_bag_tanh_fwd(const void* x, void* y, size_t n, int f32, hipStream_t stream) {
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (f32) bag_tanh_fwd_kernel<float><<<blocks, 256, 0, stream>>>((const float*)x, (float*)y, n);
    else bag_tanh_fwd_kernel<__bf16><<<blocks, 256, 0, stream>>>((const __bf16*)x, (__bf16*)y, n);
    MPO_LAUNCH_CHECK();
    return 0;
}
int mpo_launch_bag_tanh_bwd(const void* y, const void* dy, void* dx, size_t n, int f32, hipStream_t stream) {
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (f32) bag_tanh_bwd_kernel<float><<<blocks, 256, 0, stream>>>((const float*)y, (const float*)dy, (float*)dx, n);
    else bag_tanh_bwd_kernel<__bf16><<<blocks, 256, 0, stream>>>((const __bf16*)y, (const __bf16*)dy, (__bf16*)dx, n);
    MPO_LAUNCH_CHECK();
    return 0;
}
int mpo_launch_qprep(const float* q, float* qt, float* qs2, float* tq, int n, float c_nat, hipStream_t stream) {
    qprep_kernel<<<(n + 255) / 256, 256, 0, stream>>>(q, qt, qs2, tq, n, c_nat);
    MPO_LAUNCH_CHECK();
    return 0;
}
int mpo_launch_qprep_bwd(const float* dqt, const float* dtq, const float* tq, const float* d_ext, float* dq, int n,
                         float c_nat, hipStream_t stream) {
    qprep_bwd_kernel<<<(n + 255) / 256, 256, 0, stream>>>(dqt, dtq, tq, d_ext, dq, n, c_nat);
    MPO_LAUNCH_CHECK();
    return 0;
}
int mpo_launch_row_scaled_bias(float* y, const float* s, const float* bias, int rows, int cols, hipStream_t stream) {
    row_scaled_bias_kernel<<<(rows * cols + 255) / 256, 256, 0, stream>>>(y, s, bias, rows, cols);
    MPO_LAUNCH_CHECK();
    return 0;
}
