// K1 forward: genomic-guided co-attention of MCAT over a long patch bag, folded form.
//
// Replaces the M-proportional part of nn.MultiheadAttention(E, heads=1) as the reference
// calls it (models/mcat/mcat.py:48,97; arithmetic torch/nn/functional.py:6206-6660):
//   S = (q/sqrt(E)) k^T,  A = softmax(S),  ctx = A v      with k = H W_k^T + b_k, v = H W_v^T + b_v.
// One head and key == value == H let K and V disappear (hard part 3):
//   S[n][m] = qk[n] . H[m] + const(n)   with qk = (q/sqrt(E)) W_k   (the constant cancels in softmax)
//   A v     = (A H) W_v^T + b_v                                      (rows of A sum to 1)
// so the bag is streamed ONCE: per 32-row tile  S^T = H qk^T  and  ctx^T += H^T P  on the
// MFMA (bf16 operands, hi/lo split of qk and P, fp32 accumulate), online softmax in registers.
// Logits are kept in log2 units (qk is pre-multiplied by log2 e) so exp is a bare v_exp_f32.
//
// Launch: grid (splits, n_slides); a workgroup takes a contiguous range of one slide's rows,
// its waves take 32-row tiles round-robin and never wait on each other inside the loop.
// Tiles are staged through registers one tile ahead (coattn_tile.h); an fp32 bag becomes a
// bf16 hi + lo image.  Every workgroup writes one partial (max, sum, ctx[n_q][E]);
// coattn_combine_kernel merges the partials of a slide (split-M).
#include "coattn_tile.h"

namespace {

template <int E_, bool F32BAG>
struct FwdCfg {
    static constexpr int NT = F32BAG ? 2 : 1;                       // image tiles per wave (hi [+ lo])
    static constexpr int WAVES = (F32BAG && E_ == 512) ? 2 : 4;     // LDS budget 160 KiB
    // bf16 bag, E <= 256: tiles travel global -> LDS directly (global_load_lds_dwordx4), two images per wave
    static constexpr bool DMA = !F32BAG && E_ <= 256;
    static constexpr int WAVE_LDS = (DMA ? 2 : NT) * TileGeom<E_>::TILEB;
    static constexpr int ML_OFF = WAVES * WAVE_LDS;                 // [WAVES][16][2] floats after the images
    static constexpr int LDS_BYTES = ML_OFF + WAVES * 128;
};

// One 32-row tile: scores, online-softmax update, context accumulate.
// nvalid = rows of this tile that exist (1..32).  s_out (nullable) -> raw log2 logits of query q
// at this tile's first row (row stride 1), written only for lanes with q < n_q.
template <int E_, int NT>
__device__ __forceinline__ void fwd_tile(const char* thi, const char* tlo, int nvalid,
                                         const bf16x8 (&qh)[TileGeom<E_>::KS],
                                         const bf16x8 (&qm)[E_ <= 256 ? TileGeom<E_>::KS : 1],
                                         const bf16x8 (&ql)[TileGeom<E_>::KS], float& m_run, float& l_run, f32x4 (&acc)[TileGeom<E_>::DT],
                                         float* s_out, bool q_live, int lane) {
    const int g = lane >> 4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (E_ <= 256) tile_dot_rows3<E_, NT>(thi, tlo, qh, qm, ql, s0, s1, lane);
    else tile_dot_rows<E_, NT>(thi, tlo, qh, ql, s0, s1, lane);       // ('big': the third term does not fit the register file)
    float sv[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sv[r] = (4 * g + r < nvalid) ? s0[r] : -INFINITY;
        sv[4 + r] = (16 + 4 * g + r < nvalid) ? s1[r] : -INFINITY;
    }
    if (s_out != nullptr && q_live) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (4 * g + r < nvalid) s_out[4 * g + r] = sv[r];
            if (16 + 4 * g + r < nvalid) s_out[16 + 4 * g + r] = sv[4 + r];
        }
    }
    float mx = sv[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) mx = fmaxf(mx, sv[j]);
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);                    // finite: row 0 of a processed tile is valid
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // 0 on the first tile (m_run = -inf)
    float pv[8], ps = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        pv[j] = __builtin_amdgcn_exp2f(sv[j] - m_new);
        ps += pv[j];
    }
    l_run = l_run * alpha + ps;
    m_run = m_new;
    if (!__all(alpha == 1.0f)) {
#pragma unroll
        for (int t = 0; t < TileGeom<E_>::DT; ++t) acc[t] *= alpha;
    }
    bf16x8 ph, pl;
    pack_hi_lo(pv, ph, pl);
    tile_accum_cols<E_, NT>(thi, tlo, ph, pl, acc, lane);
}

template <int E_, bool F32BAG>
__global__ __launch_bounds__((FwdCfg<E_, F32BAG>::WAVES * 64), 1)
void coattn_fwd_partial_kernel(const void* __restrict__ bag_, const int* __restrict__ cu,
                               const float* __restrict__ qk2,     // [n_slides][n_q][E], log2 units
                               float* __restrict__ part_ml,       // [n_slides][splits][16][2]
                               float* __restrict__ part_ctx,      // [n_slides][splits][n_q][E]
                               float* __restrict__ s_out,         // nullable; slide b at n_q*cu[b], [n_q][M_b]
                               int n_q, BagPlan plan) {
    using G = TileGeom<E_>;
    using C = FwdCfg<E_, F32BAG>;
    constexpr int WAVES = C::WAVES;
    __shared__ __attribute__((aligned(16))) char lds[C::LDS_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // scalar: uniform control flow
    const WgGeom wg = wg_geom(cu, plan);
    const int b = wg.b, row_begin = wg.row_begin, m_rows = wg.m_rows, r0 = wg.r0, r1 = wg.r1, ntiles = wg.ntiles;
    const int n_my = wave < ntiles ? (ntiles - wave + WAVES - 1) / WAVES : 0;

    char* thi = lds + wave * C::WAVE_LDS;
    char* tlo = thi + (C::NT - 1) * G::TILEB;                     // == thi for a bf16 bag (unused)
    const int q = lane & 15;
    const bool q_live = q < n_q;

    bf16x8 qh[G::KS], qm[E_ <= 256 ? G::KS : 1], ql[G::KS];          // the score operand in three bf16 terms (coattn_tile.h)
    if constexpr (E_ <= 256) load_query_frags3<E_>(qk2 + (size_t)b * n_q * E_, n_q, lane, qh, qm, ql);
    else load_query_frags<E_>(qk2 + (size_t)b * n_q * E_, n_q, lane, qh, ql);

    float m_run = -INFINITY, l_run = 0.f;
    f32x4 acc[G::DT];
#pragma unroll
    for (int t = 0; t < G::DT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    float* s_row = (s_out != nullptr) ? s_out + (size_t)n_q * row_begin + (size_t)q * m_rows : nullptr;
    const char* slide = reinterpret_cast<const char*>(bag_) + (size_t)row_begin * E_ * (F32BAG ? 4 : 2);

    const int tstride = kTileRows * WAVES;
    const int t0row = r0 + kTileRows * wave;
    if constexpr (C::DMA) {
        // The tile goes global -> LDS without passing through registers: no staging registers, no ds_write, and the
        // wait for a tile is counted BY HAND (hipcc does not track these loads; with register staging it waited for both
        // tiles in flight before staging one, so the stream ran in bursts: 4.8 TB/s where this schedule -- one tile
        // ahead, tools/gpu_probe_readbw.py -- reads 6.1).  A wave-instruction fills 1 KiB of the image linearly
        // (lane l -> byte 16 l), so the image's chunk swizzle is applied to the GLOBAL chunk each lane fetches.
        constexpr int CH = E_ / 8;                                // 16-byte chunks per bag row
        constexpr int RPI = 64 / CH;                              // rows per wave-instruction
        constexpr int NI = kTileRows / RPI;                       // wave-instructions per tile
        static_assert(64 % CH == 0 && NI * 1024 == G::TILEB, "one wave-instruction = 1 KiB of whole rows");
        auto issue = [&](int trow, char* image) {
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const int r = RPI * k + lane / CH, cs = lane % CH;
                const int c = cs ^ ((r & 7) << 1);
                const int grow = min(trow + r, m_rows - 1);       // rows past the slide: clamped (finite; masked in the tile)
                // (inline asm, not __builtin_amdgcn_global_load_lds: hipcc tracks the builtin's LDS write and puts a
                //  vmcnt(0) in front of the tile's transposing LDS reads, i.e. waits for the NEXT tile in mid-tile.
                //  Untracked loads only make its own waits for other memory operations stricter, never weaker.)
                const char* src = slide + ((size_t)grow * CH + c) * 16;
                const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(image + k * 1024);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"                   // m0 is "reserved": nothing else in this kernel uses it
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                             :: "v"(src), "s"(dst) : "memory", "m0");
#pragma clang diagnostic pop
            }
        };
        // s_waitcnt vmcnt(N): N = loads that may still be outstanding (the tile just requested)
        constexpr int kWaitNext = (NI & 15) | ((NI >> 4) << 14) | 0x0F70;
        constexpr int kWaitAll = 0x0F70;
        char* img1 = thi + G::TILEB;
        // The tiles of a range are walked from its END (MPO_K1_FWD_FORWARD_WALK: from its start): in the window step this pass
        // follows the patch-layer kernel, whose workgroups wrote H_bag front to back -- the rows written last are the ones
        // the caches still hold, and a front-to-back read would evict them with the rows it misses on.  (The online softmax
        // does not care about the order.)
#ifdef MPO_K1_FWD_FORWARD_WALK
        const int tfirst = t0row, tstep = tstride;
#else
        const int tfirst = t0row + (n_my - 1) * tstride, tstep = -tstride;
#endif
        if (n_my > 0) issue(tfirst, thi);
        for (int it = 0; it < n_my; ++it) {
            const int trow = tfirst + it * tstep;
            char* cur = (it & 1) ? img1 : thi;
            if (it + 1 < n_my) {
                issue(trow + tstep, (it & 1) ? thi : img1);       // its last readers finished with the previous tile
                __builtin_amdgcn_s_waitcnt(kWaitNext);
            } else {
                __builtin_amdgcn_s_waitcnt(kWaitAll);
            }
            asm volatile("" ::: "memory");                        // the tile's LDS reads stay below the wait
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            fwd_tile<E_, 1>(cur, cur, min(kTileRows, r1 - trow), qh, qm, ql, m_run, l_run, acc,
                            s_row ? s_row + trow : nullptr, q_live, lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // (the merge below reuses image 0 of every wave)
    } else if constexpr (!F32BAG) {
        // bf16 bag: TWO tiles in flight per wave (32 KiB), register sets alternate (static names: unrolled by 2)
        Stage<E_, false> sa, sb;
        if (n_my > 0) sa.load(slide, t0row, m_rows, 0, lane);
        if (n_my > 1) sb.load(slide, t0row + tstride, m_rows, 0, lane);
        for (int it = 0; it < n_my; it += 2) {
            int trow = t0row + it * tstride;
            sa.store(thi, tlo, 0, lane);
            if (it + 2 < n_my) sa.load(slide, trow + 2 * tstride, m_rows, 0, lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            fwd_tile<E_, 1>(thi, tlo, min(kTileRows, r1 - trow), qh, qm, ql, m_run, l_run, acc,
                            s_row ? s_row + trow : nullptr, q_live, lane);
            __builtin_amdgcn_wave_barrier();
            if (it + 1 < n_my) {
                trow += tstride;
                sb.store(thi, tlo, 0, lane);
                if (it + 3 < n_my) sb.load(slide, trow + 2 * tstride, m_rows, 0, lane);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                fwd_tile<E_, 1>(thi, tlo, min(kTileRows, r1 - trow), qh, qm, ql, m_run, l_run, acc,
                                s_row ? s_row + trow : nullptr, q_live, lane);
                __builtin_amdgcn_wave_barrier();
            }
        }
    } else {
        Stage<E_, true> st0, st1;                                 // the two 16-row halves of an fp32 tile
        if (n_my > 0) {
            st0.load(slide, t0row, m_rows, 0, lane);
            st1.load(slide, t0row, m_rows, 1, lane);
        }
        for (int it = 0; it < n_my; ++it) {
            const int trow = t0row + it * tstride;
            st0.store(thi, tlo, 0, lane);
            st1.store(thi, tlo, 1, lane);
            if (it + 1 < n_my) {                                  // next tile's loads fly under this tile's MFMAs
                st0.load(slide, trow + tstride, m_rows, 0, lane);
                st1.load(slide, trow + tstride, m_rows, 1, lane);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            fwd_tile<E_, 2>(thi, tlo, min(kTileRows, r1 - trow), qh, qm, ql, m_run, l_run, acc,
                            s_row ? s_row + trow : nullptr, q_live, lane);
            __builtin_amdgcn_wave_barrier();
        }
    }

    // ---- merge the waves of this workgroup through LDS (each wave reuses its own image)
    float l_tot = l_run + __shfl_xor(l_run, 16);
    l_tot += __shfl_xor(l_tot, 32);
    {
        float* wctx = reinterpret_cast<float*>(thi);              // [16][E] floats == TILEB bytes
        float* wml = reinterpret_cast<float*>(lds + C::ML_OFF) + wave * 32;
        const int g = lane >> 4;
#pragma unroll
        for (int t = 0; t < G::DT; ++t)
            *reinterpret_cast<f32x4*>(wctx + q * E_ + 16 * t + 4 * g) = acc[t];
        if (g == 0) {
            wml[2 * q] = m_run;
            wml[2 * q + 1] = l_tot;
        }
    }
    __syncthreads();
    const float* ml = reinterpret_cast<const float*>(lds + C::ML_OFF);
    const size_t pbase = wg.part;
    for (int idx = threadIdx.x; idx < n_q * E_; idx += WAVES * 64) {
        const int qq = idx / E_;
        float mt = -INFINITY;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) mt = fmaxf(mt, ml[w * 32 + 2 * qq]);
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const float mw = ml[w * 32 + 2 * qq];
            const float wgt = mw == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mw - mt);
            a += wgt * reinterpret_cast<const float*>(lds + w * C::WAVE_LDS)[idx];
        }
        part_ctx[pbase * n_q * E_ + idx] = a;
    }
    if (threadIdx.x < 16) {
        const int qq = threadIdx.x;
        float mt = -INFINITY, lt = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) mt = fmaxf(mt, ml[w * 32 + 2 * qq]);
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const float mw = ml[w * 32 + 2 * qq];
            lt += (mw == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mw - mt)) * ml[w * 32 + 2 * qq + 1];
        }
        part_ml[pbase * 32 + 2 * qq] = mt;
        part_ml[pbase * 32 + 2 * qq + 1] = lt;
    }
}

// Merge the split partials of one (slide, query): ctx = sum_s 2^(m_s - m) ctx_s / l,  lse2 = m + log2 l.
// One workgroup per (query, slide); the splits (<= 1024) are spread over the threads: weights go
// through LDS, the context is summed as float4 columns x split groups and reduced across the groups.
template <int E_>
__global__ __launch_bounds__(256)
void coattn_combine_kernel(const float* __restrict__ part_ml, const float* __restrict__ part_ctx,
                           float* __restrict__ ctx, float* __restrict__ lse2, int n_q, BagPlan plan) {
    constexpr int DG = E_ / 4;                 // float4 columns
    constexpr int NSG = 256 / DG;              // split groups (E=256: 4)
    __shared__ float wts[1024];
    __shared__ float red[8];
    __shared__ __attribute__((aligned(16))) float accs[NSG][E_];
    const int q = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    int s0, s1;
    slide_parts(plan, b, s0, s1);
    const size_t p0 = (size_t)s0;
    const int splits = s1 - s0;
    float mt = -INFINITY;
    for (int s = tid; s < splits; s += 256) mt = fmaxf(mt, part_ml[(p0 + s) * 32 + 2 * q]);
    mt = wave_max(mt);
    if ((tid & 63) == 0) red[tid >> 6] = mt;
    __syncthreads();
    mt = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float lt = 0.f;
    for (int s = tid; s < splits; s += 256) {
        const float ms = part_ml[(p0 + s) * 32 + 2 * q];
        const float w = ms == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(ms - mt);
        wts[s] = w;
        lt += w * part_ml[(p0 + s) * 32 + 2 * q + 1];
    }
    lt = wave_sum(lt);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = lt;
    __syncthreads();
    lt = red[4] + red[5] + red[6] + red[7];
    const int dg = tid % DG, sg = tid / DG;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int s = sg; s < splits; s += NSG) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(part_ctx + ((p0 + s) * n_q + q) * E_ + 4 * dg);
        a += v * wts[s];
    }
    *reinterpret_cast<f32x4*>(&accs[sg][4 * dg]) = a;
    __syncthreads();
    for (int d = tid; d < E_; d += 256) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < NSG; ++g) t += accs[g][d];
        ctx[((size_t)b * n_q + q) * E_ + d] = t / lt;
    }
    if (tid == 0) lse2[(size_t)b * n_q + q] = mt + __builtin_amdgcn_logf(lt);
}

// A[n][m] = 2^(S2[n][m] - lse2[n]) in place over the ragged [n_q][M_b] blocks; optional
// attention-weight dropout (NaCAGaT, models/blocks.py:189-190) writes the post-dropout map.
__global__ void coattn_normalize_kernel(float* __restrict__ a, const float* __restrict__ lse2, const int* __restrict__ cu,
                                        int n_q, float drop_p, unsigned long long seed, unsigned long long offset) {
    const int b = blockIdx.z, q = blockIdx.y;
    const int row_begin = cu[b], m_rows = cu[b + 1] - row_begin;
    const float l = lse2[(size_t)b * n_q + q];
    const size_t base = (size_t)n_q * row_begin + (size_t)q * m_rows;
    const float inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < m_rows; m += gridDim.x * blockDim.x) {
        float v = __builtin_amdgcn_exp2f(a[base + m] - l);
        if (drop_p > 0.f) v *= dropout_keep(seed, offset, base + m, drop_p, inv_keep);
        a[base + m] = v;
    }
}

}  // namespace

// ---------------------------------------------------------------------------- host launchers
int mpo_launch_coattn_fwd_partial(const void* bag, int bag_f32, const int* cu, int n_slides, int embed,
                                  const float* qk2, float* part_ml, float* part_ctx, float* s_out,
                                  int n_q, const BagPlan& plan, hipStream_t stream) {
    (void)n_slides;
    dim3 grid = plan_grid(plan);
#define MPO_FWD_CASE(EV)                                                                                   \
    case EV:                                                                                               \
        if (bag_f32)                                                                                       \
            coattn_fwd_partial_kernel<EV, true><<<grid, FwdCfg<EV, true>::WAVES * 64, 0, stream>>>(        \
                bag, cu, qk2, part_ml, part_ctx, s_out, n_q, plan);                                      \
        else                                                                                               \
            coattn_fwd_partial_kernel<EV, false><<<grid, FwdCfg<EV, false>::WAVES * 64, 0, stream>>>(      \
                bag, cu, qk2, part_ml, part_ctx, s_out, n_q, plan);                                      \
        break;
    switch (embed) {
        MPO_FWD_CASE(128)
        MPO_FWD_CASE(256)
        MPO_FWD_CASE(512)
        default:
            mpo_set_error("coattn: embed_dim %d not in {128,256,512}", embed);
            return 1;
    }
#undef MPO_FWD_CASE
    MPO_LAUNCH_CHECK();
    return 0;
}

int mpo_launch_coattn_combine(const float* part_ml, const float* part_ctx, float* ctx, float* lse2,
                              int n_slides, int n_q, int embed, const BagPlan& plan, hipStream_t stream) {
    MPO_CHECK((plan.wg_start ? plan.n_wg : plan.splits) <= 1024 + n_slides, "coattn combine: too many partials per slide");
    dim3 grid(n_q, n_slides);
    switch (embed) {
        case 128: coattn_combine_kernel<128><<<grid, 256, 0, stream>>>(part_ml, part_ctx, ctx, lse2, n_q, plan); break;
        case 256: coattn_combine_kernel<256><<<grid, 256, 0, stream>>>(part_ml, part_ctx, ctx, lse2, n_q, plan); break;
        case 512: coattn_combine_kernel<512><<<grid, 256, 0, stream>>>(part_ml, part_ctx, ctx, lse2, n_q, plan); break;
        default: mpo_set_error("coattn: embed_dim %d not in {128,256,512}", embed); return 1;
    }
    MPO_LAUNCH_CHECK();
    return 0;
}

int mpo_launch_coattn_normalize(float* a, const float* lse2, const int* cu, int n_slides, int n_q, int max_rows,
                                float drop_p, unsigned long long seed, unsigned long long offset, hipStream_t stream) {
    int bx = (max_rows + 255) / 256;
    if (bx > 1024) bx = 1024;
    if (bx < 1) bx = 1;
    dim3 grid(bx, n_q, n_slides);
    coattn_normalize_kernel<<<grid, 256, 0, stream>>>(a, lse2, cu, n_q, drop_p, seed, offset);
    MPO_LAUNCH_CHECK();
    return 0;
}
