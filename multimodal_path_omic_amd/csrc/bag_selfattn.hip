// Self-attention over the M rows of ONE bag: the long-bag shapes of
// /root/reference/models/ge_nacagat/ge_nacagat.py -- `nn.MultiheadAttention(embed, num_heads=1)(H_bag, H_bag, H_bag)` with its
// M x M map returned (:27, :49) and the two `nn.TransformerEncoderLayer(nhead=8)` blocks over the same M rows (:30-33, :53).
// The token-tail kernel of tail.hip keeps a T x T probability matrix per head in LDS (T <= 16); at T = M = 15 000 that matrix
// is 0.9 GB per head, so these kernels never form it: scores are recomputed tile by tile (forward: online softmax, backward:
// from the saved log-sum-exp), and only the one map the reference RETURNS is ever written.
//
// fp32 throughout on v_mfma_f32_16x16x4_f32 (the 1e-3 bar on maps is an fp32 bar; a three-term bf16 split is the faster
// follow-up).  All three kernels work on TRANSPOSED tiles so that the MFMA result layout is already the next product's
// operand layout and no probability tile ever travels through LDS:
//   forward / dQ : a wave owns 16 queries;  S^T[key][q] = K_tile . Q^T  puts query q in lane column q, keys 4*(lane/16)+r
//                  in the four result registers -- exactly the B operand of  O^T[c][q] += V^T[c][key] . P^T[key][q];  the
//                  soft-max statistics of a query are lane-local (+ two cross-lane steps) and the rescale is a per-lane scalar.
//   dK / dV      : a wave owns 16 keys;  S[q][key] = Q_tile . K^T  puts key in the lane column, which is the B operand of
//                  dV^T[c][key] += dO^T[c][q] . P[q][key]  and  dK^T[c][key] += Q^T[c][q] . dS[q][key].
// The row-major K / V (or Q / dO) tile in LDS serves both as the A operand of the score product (one ds_read_b128 per four
// MFMAs, row stride HD + 4 floats: conflict-free) and, read one float per MFMA, as the transposed A operand of the second.
//
// Dropout on the probabilities (the encoder layers' attention dropout, p = 0.25 in training): one Philox call per 4 x 4
// block of (query, key), 8 bits per element (realised p = round(256 p) / 256, as in the fused patch layer); forward, dQ and
// dK/dV regenerate the same block from (seed, stream, head, q / 4, key / 4) -- both orientations hold four elements of one
// block per lane, so each pays one call per four elements.
#include "mpo_common.h"
#include "mpo_kernels.h"

namespace {

constexpr int kSaWaves = 4;                      // 4 waves x 16 queries (or keys) per workgroup

template <int HD> struct SaCfg {
    static constexpr int BN = HD >= 128 ? 32 : 64;          // rows of the streamed tile per step
    static constexpr int LDR = HD + 4;                       // LDS row stride in floats
    static constexpr int C16 = HD / 16;                      // 16-column groups of the head dimension
    static constexpr int NT = BN / 16;                       // 16-row tiles of the streamed tile
    static constexpr size_t TILE_FLOATS = (size_t)BN * LDR;
};

struct SaDrop {
    unsigned thr;                                            // keep when byte >= thr (0: no dropout)
    float inv_keep;
    uint32_t k0, k1, c2, c3;
};
__device__ __forceinline__ SaDrop sa_drop(float p, unsigned long long seed, unsigned long long offset,
                                          const unsigned long long* epoch, int head) {
    SaDrop d;
    const unsigned t = p > 0.f ? (unsigned)(p * 256.0f + 0.5f) : 0u;
    d.thr = t > 255u ? 255u : t;
    d.inv_keep = 256.0f / (256.0f - (float)d.thr);
    const unsigned long long ctr = epoch_offset(offset, epoch);
    d.k0 = (uint32_t)seed; d.k1 = (uint32_t)(seed >> 32);
    d.c2 = (uint32_t)ctr ^ ((uint32_t)head * 0x9E3779B1u);
    d.c3 = (uint32_t)(ctr >> 32) | 0x80000000u;              // the element-counter streams of the other kernels have c3 = 0
    return d;
}
// the 16 bytes of block (q / 4, key / 4); element (q % 4, key % 4) is byte 4 * (q % 4) + key % 4
__device__ __forceinline__ uint4 sa_block(const SaDrop& d, int qb, int kb) { return philox4x32((uint32_t)kb, (uint32_t)qb, d.c2, d.c3, d.k0, d.k1); }
__device__ __forceinline__ float sa_keep(const SaDrop& d, uint32_t word, int byte) {
    return ((word >> (8 * byte)) & 255u) >= d.thr ? d.inv_keep : 0.f;
}
__device__ __forceinline__ uint32_t sa_word(const uint4& b, int i) { return i == 0 ? b.x : i == 1 ? b.y : i == 2 ? b.z : b.w; }

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// rows [r0, r0 + BN) x columns [col0, col0 + HD) of a [M][ld] matrix -> LDS tile (row stride LDR); rows >= M as zeros
template <int HD>
__device__ __forceinline__ void sa_load_tile(float* tile, const float* __restrict__ src, int ld, int col0, int r0, int M) {
    using C = SaCfg<HD>;
    constexpr int V4 = HD / 4;
    for (int idx = threadIdx.x; idx < C::BN * V4; idx += 64 * kSaWaves) {
        const int r = idx / V4, c4 = idx % V4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r0 + r < M) v = *reinterpret_cast<const float4*>(src + (size_t)(r0 + r) * ld + col0 + 4 * c4);
        *reinterpret_cast<float4*>(tile + r * C::LDR + 4 * c4) = v;
    }
}
// the B operand of a wave: row `row` (clamped) of the matrix, 16 floats apart per 16-column group: f[c] = x[row][16 c + 4 kk ..]
template <int HD>
__device__ __forceinline__ void sa_load_frag(float4 (&f)[HD / 16], const float* __restrict__ src, int ld, int col0, int row, int M,
                                             int kk, float mul) {
    const int r = row < M ? row : M - 1;
#pragma unroll
    for (int c = 0; c < HD / 16; ++c) {
        float4 v = *reinterpret_cast<const float4*>(src + (size_t)r * ld + col0 + 16 * c + 4 * kk);
        if (row >= M) v = make_float4(0.f, 0.f, 0.f, 0.f);
        f[c] = make_float4(v.x * mul, v.y * mul, v.z * mul, v.w * mul);
    }
}
// T[16 tile rows][lane column] = tile rows (A operand, b128 reads) . frag (B operand): result register r = tile row 4 kk + r
template <int HD>
__device__ __forceinline__ f32x4 sa_tile_dot(const float* tile, int t, const float4 (&f)[HD / 16], int j, int kk) {
    using C = SaCfg<HD>;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const float* rowp = tile + (16 * t + j) * C::LDR + 4 * kk;
#pragma unroll
    for (int c = 0; c < HD / 16; ++c) {
        const float4 a = *reinterpret_cast<const float4*>(rowp + 16 * c);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, f[c].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, f[c].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, f[c].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, f[c].w, acc1, 0, 0, 0);
    }
    return acc0 + acc1;
}
// acc[ct][.] (rows = columns 16 ct + 4 kk + r of the tile, lane column unchanged) += tile^T . w, w[r] the lane's weight for
// tile row 16 t + 4 kk + r
template <int HD>
__device__ __forceinline__ void sa_tile_tacc(f32x4 (&acc)[HD / 16], const float* tile, int t, const f32x4& w, int j, int kk) {
    using C = SaCfg<HD>;
    const float* base = tile + (16 * t + 4 * kk) * C::LDR + j;
#pragma unroll
    for (int ct = 0; ct < HD / 16; ++ct) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(base[r * C::LDR + 16 * ct], w[r], acc[ct], 0, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------------ forward
// grid (query blocks of 64, heads, sequences).  o [M][d] (head h: columns h HD ..), lse2 [heads][M] = log2 of the row's
// exp-sum in the scaled-by-log2(e) score domain.
template <int HD>
__global__ __launch_bounds__(64 * kSaWaves)
void bag_sa_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ o, float* __restrict__ lse2, int M, int d, float scale,
                       float drop_p, unsigned long long seed, unsigned long long offset, const unsigned long long* epoch) {
    using C = SaCfg<HD>;
    __shared__ __attribute__((aligned(16))) float sm[2 * C::TILE_FLOATS];
    float* kt = sm;
    float* vt = sm + C::TILE_FLOATS;
    const int h = blockIdx.y, seq = blockIdx.z, H = gridDim.y;
    qkv += (size_t)seq * M * 3 * d;
    o += (size_t)seq * M * d;
    lse2 += ((size_t)seq * H + h) * M;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 15, kk = lane >> 4;
    const int q = blockIdx.x * 64 + 16 * wv + j;
    const SaDrop dr = sa_drop(drop_p, seed, offset, epoch, seq * H + h);
    float4 qf[C::C16];
    sa_load_frag<HD>(qf, qkv, 3 * d, h * HD, q, M, kk, scale * kLog2e);
    f32x4 acc[C::C16];
#pragma unroll
    for (int c = 0; c < C::C16; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;
    for (int n0 = 0; n0 < M; n0 += C::BN) {
        __syncthreads();
        sa_load_tile<HD>(kt, qkv, 3 * d, d + h * HD, n0, M);
        sa_load_tile<HD>(vt, qkv, 3 * d, 2 * d + h * HD, n0, M);
        __syncthreads();
        f32x4 s[C::NT];
        float mx = m;
#pragma unroll
        for (int t = 0; t < C::NT; ++t) {
            s[t] = sa_tile_dot<HD>(kt, t, qf, j, kk);
            if (n0 + C::BN > M) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n0 + 16 * t + 4 * kk + r >= M) s[t][r] = -INFINITY;
            }
            mx = fmaxf(mx, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float alpha = fast_exp2(m - mx);               // first block: exp2(-inf) = 0
        m = mx;
        l *= alpha;
#pragma unroll
        for (int c = 0; c < C::C16; ++c) acc[c] *= alpha;
#pragma unroll
        for (int t = 0; t < C::NT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[t][r] = fast_exp2(s[t][r] - mx); l += s[t][r]; }
            if (dr.thr) {
                const uint4 blk = sa_block(dr, q >> 2, (n0 + 16 * t + 4 * kk) >> 2);
                const uint32_t w = sa_word(blk, q & 3);
#pragma unroll
                for (int r = 0; r < 4; ++r) s[t][r] *= sa_keep(dr, w, r);
            }
            sa_tile_tacc<HD>(acc, vt, t, s[t], j, kk);
        }
    }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    if (q < M) {
        const float inv = 1.0f / l;
#pragma unroll
        for (int c = 0; c < C::C16; ++c)
            *reinterpret_cast<float4*>(o + (size_t)q * d + h * HD + 16 * c + 4 * kk) =
                make_float4(acc[c][0] * inv, acc[c][1] * inv, acc[c][2] * inv, acc[c][3] * inv);
        if (kk == 0) lse2[q] = m + __builtin_amdgcn_logf(l);   // v_log_f32 = log2
    }
}

// the returned map of the one-head layer: map[q][key] = exp2(s - lse2[q])   (no dropout: nn.MultiheadAttention default)
template <int HD>
__global__ __launch_bounds__(64 * kSaWaves)
void bag_sa_map_kernel(const float* __restrict__ qkv, const float* __restrict__ lse2, float* __restrict__ map, int M, int d, float scale) {
    using C = SaCfg<HD>;
    __shared__ __attribute__((aligned(16))) float sm[C::TILE_FLOATS];
    float* kt = sm;
    const int seq = blockIdx.z;
    qkv += (size_t)seq * M * 3 * d;
    lse2 += (size_t)seq * M;
    map += (size_t)seq * M * M;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 15, kk = lane >> 4;
    const int q = blockIdx.x * 64 + 16 * wv + j;
    float4 qf[C::C16];
    sa_load_frag<HD>(qf, qkv, 3 * d, 0, q, M, kk, scale * kLog2e);
    const float ls = q < M ? lse2[q] : 0.f;
    // key blocks are split over grid.y so that the 0.9 GB write is spread over more than M / 64 workgroups
    const int nblk = (M + C::BN - 1) / C::BN, per = (nblk + gridDim.y - 1) / gridDim.y;
    const int b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
    const bool vec = (M & 3) == 0;
    for (int b = b0; b < b1; ++b) {
        const int n0 = b * C::BN;
        __syncthreads();
        sa_load_tile<HD>(kt, qkv, 3 * d, d, n0, M);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < C::NT; ++t) {
            const f32x4 s = sa_tile_dot<HD>(kt, t, qf, j, kk);
            const int key = n0 + 16 * t + 4 * kk;
            if (q < M) {
                float* dst = map + (size_t)q * M + key;
                if (vec && key + 3 < M) {
                    *reinterpret_cast<float4*>(dst) = make_float4(fast_exp2(s[0] - ls), fast_exp2(s[1] - ls), fast_exp2(s[2] - ls), fast_exp2(s[3] - ls));
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (key + r < M) dst[r] = fast_exp2(s[r] - ls);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward, dQ (+ delta)
// delta[h][q] = sum_c dO[q][c] O[q][c] is computed here from the wave's own rows and written for the dK/dV kernel.
template <int HD>
__global__ __launch_bounds__(64 * kSaWaves)
void bag_sa_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ o, const float* __restrict__ lse2,
                          const float* __restrict__ d_o, float* __restrict__ dqkv, float* __restrict__ delta, int M, int d, float scale,
                          float drop_p, unsigned long long seed, unsigned long long offset, const unsigned long long* epoch) {
    using C = SaCfg<HD>;
    __shared__ __attribute__((aligned(16))) float sm[2 * C::TILE_FLOATS];
    float* kt = sm;
    float* vt = sm + C::TILE_FLOATS;
    const int h = blockIdx.y, seq = blockIdx.z, H = gridDim.y;
    qkv += (size_t)seq * M * 3 * d;
    dqkv += (size_t)seq * M * 3 * d;
    o += (size_t)seq * M * d;
    d_o += (size_t)seq * M * d;
    lse2 += ((size_t)seq * H + h) * M;
    delta += ((size_t)seq * H + h) * M;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 15, kk = lane >> 4;
    const int q = blockIdx.x * 64 + 16 * wv + j;
    const SaDrop dr = sa_drop(drop_p, seed, offset, epoch, seq * H + h);
    float4 qf[C::C16], dof[C::C16];
    sa_load_frag<HD>(qf, qkv, 3 * d, h * HD, q, M, kk, scale * kLog2e);
    sa_load_frag<HD>(dof, d_o, d, h * HD, q, M, kk, 1.0f);
    float dl = 0.f;
    {
        float4 of[C::C16];
        sa_load_frag<HD>(of, o, d, h * HD, q, M, kk, 1.0f);
#pragma unroll
        for (int c = 0; c < C::C16; ++c) dl += (of[c].x * dof[c].x + of[c].y * dof[c].y) + (of[c].z * dof[c].z + of[c].w * dof[c].w);
    }
    dl += __shfl_xor(dl, 16);
    dl += __shfl_xor(dl, 32);
    const float ls = q < M ? lse2[q] : INFINITY;              // rows past the end: p = exp2(-inf) = 0
    if (q < M && kk == 0) delta[q] = dl;
    f32x4 acc[C::C16];
#pragma unroll
    for (int c = 0; c < C::C16; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int n0 = 0; n0 < M; n0 += C::BN) {
        __syncthreads();
        sa_load_tile<HD>(kt, qkv, 3 * d, d + h * HD, n0, M);
        sa_load_tile<HD>(vt, qkv, 3 * d, 2 * d + h * HD, n0, M);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < C::NT; ++t) {
            f32x4 s = sa_tile_dot<HD>(kt, t, qf, j, kk);
            f32x4 dp = sa_tile_dot<HD>(vt, t, dof, j, kk);
            if (dr.thr) {
                const uint4 blk = sa_block(dr, q >> 2, (n0 + 16 * t + 4 * kk) >> 2);
                const uint32_t w = sa_word(blk, q & 3);
#pragma unroll
                for (int r = 0; r < 4; ++r) dp[r] *= sa_keep(dr, w, r);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = (n0 + 16 * t + 4 * kk + r < M) ? fast_exp2(s[r] - ls) : 0.f;
                s[r] = p * (dp[r] - dl);
            }
            sa_tile_tacc<HD>(acc, kt, t, s, j, kk);
        }
    }
    if (q < M) {
#pragma unroll
        for (int c = 0; c < C::C16; ++c)
            *reinterpret_cast<float4*>(dqkv + (size_t)q * 3 * d + h * HD + 16 * c + 4 * kk) =
                make_float4(acc[c][0] * scale, acc[c][1] * scale, acc[c][2] * scale, acc[c][3] * scale);
    }
}

// ------------------------------------------------------------------------------------------------ backward, dK and dV
template <int HD>
__global__ __launch_bounds__(64 * kSaWaves)
void bag_sa_bwd_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ lse2, const float* __restrict__ delta,
                           const float* __restrict__ d_o, float* __restrict__ dqkv, int M, int d, float scale,
                           float drop_p, unsigned long long seed, unsigned long long offset, const unsigned long long* epoch) {
    using C = SaCfg<HD>;
    __shared__ __attribute__((aligned(16))) float sm[2 * C::TILE_FLOATS + 2 * C::BN];
    float* qt = sm;
    float* dot = sm + C::TILE_FLOATS;
    float* ls_t = sm + 2 * C::TILE_FLOATS;                   // [BN] lse2, then [BN] delta
    float* dl_t = ls_t + C::BN;
    const int h = blockIdx.y, seq = blockIdx.z, H = gridDim.y;
    qkv += (size_t)seq * M * 3 * d;
    dqkv += (size_t)seq * M * 3 * d;
    d_o += (size_t)seq * M * d;
    lse2 += ((size_t)seq * H + h) * M;
    delta += ((size_t)seq * H + h) * M;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 15, kk = lane >> 4;
    const int key = blockIdx.x * 64 + 16 * wv + j;
    const SaDrop dr = sa_drop(drop_p, seed, offset, epoch, seq * H + h);
    float4 kf[C::C16], vf[C::C16];
    sa_load_frag<HD>(kf, qkv, 3 * d, d + h * HD, key, M, kk, scale * kLog2e);
    sa_load_frag<HD>(vf, qkv, 3 * d, 2 * d + h * HD, key, M, kk, 1.0f);
    f32x4 dk[C::C16], dv[C::C16];
#pragma unroll
    for (int c = 0; c < C::C16; ++c) { dk[c] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int q0 = 0; q0 < M; q0 += C::BN) {
        __syncthreads();
        sa_load_tile<HD>(qt, qkv, 3 * d, h * HD, q0, M);
        sa_load_tile<HD>(dot, d_o, d, h * HD, q0, M);
        if (threadIdx.x < C::BN) {
            const int qq = q0 + threadIdx.x;
            ls_t[threadIdx.x] = qq < M ? lse2[qq] : INFINITY;
            dl_t[threadIdx.x] = qq < M ? delta[qq] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < C::NT; ++t) {
            f32x4 s = sa_tile_dot<HD>(qt, t, kf, j, kk);      // s[r]: query q0 + 16 t + 4 kk + r, key = lane column
            f32x4 dp = sa_tile_dot<HD>(dot, t, vf, j, kk);
            const float4 ls = *reinterpret_cast<const float4*>(ls_t + 16 * t + 4 * kk);
            const float4 dl = *reinterpret_cast<const float4*>(dl_t + 16 * t + 4 * kk);
            f32x4 p = {fast_exp2(s[0] - ls.x), fast_exp2(s[1] - ls.y), fast_exp2(s[2] - ls.z), fast_exp2(s[3] - ls.w)};
            f32x4 pd = p;
            if (dr.thr) {
                const uint4 blk = sa_block(dr, (q0 + 16 * t + 4 * kk) >> 2, key >> 2);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float keep = sa_keep(dr, sa_word(blk, r), key & 3);
                    pd[r] *= keep;
                    dp[r] *= keep;
                }
            }
            f32x4 ds = {p[0] * (dp[0] - dl.x), p[1] * (dp[1] - dl.y), p[2] * (dp[2] - dl.z), p[3] * (dp[3] - dl.w)};
            sa_tile_tacc<HD>(dv, dot, t, pd, j, kk);
            sa_tile_tacc<HD>(dk, qt, t, ds, j, kk);
        }
    }
    if (key < M) {
#pragma unroll
        for (int c = 0; c < C::C16; ++c) {
            float* at = dqkv + (size_t)key * 3 * d + h * HD + 16 * c + 4 * kk;
            *reinterpret_cast<float4*>(at + d) = make_float4(dk[c][0] * scale, dk[c][1] * scale, dk[c][2] * scale, dk[c][3] * scale);
            *reinterpret_cast<float4*>(at + 2 * d) = make_float4(dv[c][0], dv[c][1], dv[c][2], dv[c][3]);
        }
    }
}

template <int HD>
int sa_forward(const float* qkv, int n_seq, int M, int d, int H, float drop_p, unsigned long long seed, unsigned long long offset,
               const unsigned long long* epoch, float* o, float* lse2, float* map, hipStream_t s) {
    const float scale = 1.0f / sqrtf((float)HD);
    const dim3 grid((M + 63) / 64, H, n_seq);
    bag_sa_fwd_kernel<HD><<<grid, 64 * kSaWaves, 0, s>>>(qkv, o, lse2, M, d, scale, drop_p, seed, offset, epoch);
    MPO_LAUNCH_CHECK();
    if (map) {
        const int qb = (M + 63) / 64;
        int split = (2048 + qb - 1) / qb;                    // ~2048 workgroups
        const int nblk = (M + SaCfg<HD>::BN - 1) / SaCfg<HD>::BN;
        if (split > nblk) split = nblk;
        bag_sa_map_kernel<HD><<<dim3(qb, split, n_seq), 64 * kSaWaves, 0, s>>>(qkv, lse2, map, M, d, scale);
        MPO_LAUNCH_CHECK();
    }
    return 0;
}
template <int HD>
int sa_backward(const float* qkv, const float* o, const float* lse2, const float* d_o, int n_seq, int M, int d, int H, float drop_p,
                unsigned long long seed, unsigned long long offset, const unsigned long long* epoch, float* dqkv, float* delta,
                hipStream_t s) {
    const float scale = 1.0f / sqrtf((float)HD);
    const dim3 grid((M + 63) / 64, H, n_seq);
    bag_sa_bwd_dq_kernel<HD><<<grid, 64 * kSaWaves, 0, s>>>(qkv, o, lse2, d_o, dqkv, delta, M, d, scale, drop_p, seed, offset, epoch);
    MPO_LAUNCH_CHECK();
    bag_sa_bwd_dkv_kernel<HD><<<grid, 64 * kSaWaves, 0, s>>>(qkv, lse2, delta, d_o, dqkv, M, d, scale, drop_p, seed,
                                                                                         offset, epoch);
    MPO_LAUNCH_CHECK();
    return 0;
}
inline bool sa_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

int mpo_bag_sa_supported_head_dim(int hd) { return hd == 16 || hd == 32 || hd == 64 || hd == 128 || hd == 256; }

int mpo_launch_bag_sa_fwd(const float* qkv, int n_seq, int M, int d, int H, float drop_p, unsigned long long seed,
                          unsigned long long offset, const unsigned long long* epoch, float* o, float* lse2, float* map, hipStream_t s) {
    MPO_CHECK(n_seq >= 1 && M >= 1 && H >= 1 && d % H == 0, "bag self-attention: %d sequences of %d rows, d=%d, heads=%d", n_seq, M, d, H);
    MPO_CHECK(n_seq <= 65535 && H <= 65535, "bag self-attention: %d sequences x %d heads exceed the grid", n_seq, H);
    const int hd = d / H;
    MPO_CHECK(mpo_bag_sa_supported_head_dim(hd), "bag self-attention: head dimension %d (16, 32, 64, 128 or 256)", hd);
    MPO_CHECK(map == nullptr || H == 1, "bag self-attention: the M x M map is returned for one head only (heads=%d)", H);
    MPO_CHECK(sa_al16(qkv) && sa_al16(o) && (map == nullptr || sa_al16(map)), "bag self-attention: buffers must be 16-byte aligned");
    switch (hd) {
        case 16: return sa_forward<16>(qkv, n_seq, M, d, H, drop_p, seed, offset, epoch, o, lse2, map, s);
        case 32: return sa_forward<32>(qkv, n_seq, M, d, H, drop_p, seed, offset, epoch, o, lse2, map, s);
        case 64: return sa_forward<64>(qkv, n_seq, M, d, H, drop_p, seed, offset, epoch, o, lse2, map, s);
        case 128: return sa_forward<128>(qkv, n_seq, M, d, H, drop_p, seed, offset, epoch, o, lse2, map, s);
        default: return sa_forward<256>(qkv, n_seq, M, d, H, drop_p, seed, offset, epoch, o, lse2, map, s);
    }
}
int mpo_launch_bag_sa_bwd(const float* qkv, const float* o, const float* lse2, const float* d_o, int n_seq, int M, int d, int H,
                          float drop_p, unsigned long long seed, unsigned long long offset, const unsigned long long* epoch,
                          float* dqkv, float* delta, hipStream_t s) {
    MPO_CHECK(n_seq >= 1 && M >= 1 && H >= 1 && d % H == 0, "bag self-attention: %d sequences of %d rows, d=%d, heads=%d", n_seq, M, d, H);
    MPO_CHECK(n_seq <= 65535 && H <= 65535, "bag self-attention: %d sequences x %d heads exceed the grid", n_seq, H);
    const int hd = d / H;
    MPO_CHECK(mpo_bag_sa_supported_head_dim(hd), "bag self-attention: head dimension %d (16, 32, 64, 128 or 256)", hd);
    MPO_CHECK(sa_al16(qkv) && sa_al16(o) && sa_al16(d_o) && sa_al16(dqkv), "bag self-attention: buffers must be 16-byte aligned");
    switch (hd) {
        case 16: return sa_backward<16>(qkv, o, lse2, d_o, n_seq, M, d, H, drop_p, seed, offset, epoch, dqkv, delta, s);
        case 32: return sa_backward<32>(qkv, o, lse2, d_o, n_seq, M, d, H, drop_p, seed, offset, epoch, dqkv, delta, s);
        case 64: return sa_backward<64>(qkv, o, lse2, d_o, n_seq, M, d, H, drop_p, seed, offset, epoch, dqkv, delta, s);
        case 128: return sa_backward<128>(qkv, o, lse2, d_o, n_seq, M, d, H, drop_p, seed, offset, epoch, dqkv, delta, s);
        default: return sa_backward<256>(qkv, o, lse2, d_o, n_seq, M, d, H, drop_p, seed, offset, epoch, dqkv, delta, s);
    }
}
