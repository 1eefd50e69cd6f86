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
// Dropout on the probabilities (the encoder layers' attention dropout, p = 0.25 in training): one counter-hash call per
// 4 x 4 block of (query, key), 8 bits per element (realised p = round(256 p) / 256, as in the fused patch layer); forward, dQ
// and dK/dV regenerate the same block from (seed, stream, head, q / 4, key / 4) -- both orientations hold four elements of
// one block per lane.
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

// Counter-based hash, not Philox: 32-bit integer multiplies run at a quarter of the vector rate on this chip, and the ten
// Philox rounds (40 multiplies) of one block cost a third of the bf16 kernels' step.  One block = 16 bytes = four words, each
// the murmur3 finaliser (two multiplies, full avalanche) of a key mixed with the block's coordinates; the block's 32-bit
// x is a bottleneck (at M = 15 000, ~2e4 of a head's 1.4e7 blocks repeat another block's 16 bytes) -- irrelevant to parity
// (all three kernels regenerate the same mask), stated so that nobody takes the mask for 128 independent bits per block.
__device__ __forceinline__ uint32_t sa_fmix(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
struct SaDrop {
    unsigned thr;                                            // keep when byte >= thr (0: no dropout)
    float inv_keep;
    uint32_t key;                                            // seed, stream offset (+ epoch) and head, hashed
    uint32_t inc;                                            // odd word stride, also from the key: two keys' blocks are not
                                                             // the same draws at permuted coordinates (mpo_common.h hash4x32)
};
__device__ __forceinline__ SaDrop sa_drop(float p, unsigned long long seed, unsigned long long offset,
                                          const unsigned long long* epoch, int head) {
    SaDrop d;
    const unsigned t = p > 0.f ? (unsigned)(p * 256.0f + 0.5f) : 0u;
    d.thr = t > 255u ? 255u : t;
    d.inv_keep = 256.0f / (256.0f - (float)d.thr);
    const unsigned long long ctr = epoch_offset(offset, epoch);
    uint32_t k = sa_fmix((uint32_t)seed ^ 0x5A17u);
    k = sa_fmix(k ^ (uint32_t)(seed >> 32));
    k = sa_fmix(k ^ (uint32_t)ctr);
    k = sa_fmix(k ^ (uint32_t)(ctr >> 32));
    d.key = sa_fmix(k ^ (uint32_t)head);
    d.inc = sa_fmix(d.key ^ 0x9E3779B9u) | 1u;
    return d;
}
// the 16 bytes of block (q / 4, key / 4); element (q % 4, key % 4) is byte 4 * (q % 4) + key % 4
__device__ __forceinline__ uint32_t sa_block_x(const SaDrop& d, int qb, int kb) {
    return (d.key ^ ((uint32_t)qb * 0x9E3779B1u)) + (uint32_t)kb * 0x85EBCA77u;
}
// word i (= q % 4) of the block alone: what the forward / dQ orientation needs (one query per lane)
__device__ __forceinline__ uint32_t sa_block_word(const SaDrop& d, int qb, int kb, int i) {
    return sa_fmix(sa_block_x(d, qb, kb) + (uint32_t)i * d.inc);
}
__device__ __forceinline__ uint4 sa_block(const SaDrop& d, int qb, int kb) {
    const uint32_t x = sa_block_x(d, qb, kb);
    return make_uint4(sa_fmix(x), sa_fmix(x + d.inc), sa_fmix(x + 2u * d.inc), sa_fmix(x + 3u * d.inc));
}
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

// the same for NCT 16-column groups starting at group ct0 (one column pass of a head wider than the register file)
template <int HD, int NCT>
__device__ __forceinline__ void sa_tile_tacc_cols(f32x4 (&acc)[NCT], const float* tile, int t, const f32x4& w, int j, int kk, int ct0) {
    using C = SaCfg<HD>;
    const float* base = tile + (16 * t + 4 * kk) * C::LDR + j + 16 * ct0;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
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
                const uint32_t w = sa_block_word(dr, q >> 2, (n0 + 16 * t + 4 * kk) >> 2, q & 3);
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
                const uint32_t w = sa_block_word(dr, q >> 2, (n0 + 16 * t + 4 * kk) >> 2, q & 3);
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
// NP column passes (grid.y = heads x NP): at HD = 512 the two key-side operands and both accumulators would take the whole
// register file, so a pass recomputes the scores from full rows and accumulates HD / NP columns of dK and dV.
template <int HD, int NP>
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
    constexpr int NCT = C::C16 / NP;                         // 16-column groups of this pass
    const int h = blockIdx.y / NP, ct0 = (blockIdx.y % NP) * NCT, seq = blockIdx.z, H = gridDim.y / NP;
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
    f32x4 dk[NCT], dv[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) { dk[c] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
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
            sa_tile_tacc_cols<HD, NCT>(dv, dot, t, pd, j, kk, ct0);
            sa_tile_tacc_cols<HD, NCT>(dk, qt, t, ds, j, kk, ct0);
        }
    }
    if (key < M) {
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            float* at = dqkv + (size_t)key * 3 * d + h * HD + 16 * (ct0 + c) + 4 * kk;
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
    constexpr int NP = HD > 256 ? 2 : 1;
    bag_sa_bwd_dkv_kernel<HD, NP><<<dim3(grid.x, H * NP, n_seq), 64 * kSaWaves, 0, s>>>(qkv, lse2, delta, d_o, dqkv, M, d, scale, drop_p, seed,
                                                                                         offset, epoch);
    MPO_LAUNCH_CHECK();
    return 0;
}
// ================================================================================================ three-term bf16 path
// Every operand is split x = hi + lo into two bf16 and each product runs as hi*hi + lo*hi + hi*lo on
// v_mfma_f32_16x16x32_bf16: three instructions (48 cycles) cover 32 of the inner dimension, against eight fp32 instructions
// (256 cycles).  ~16 mantissa bits per operand -- the arithmetic the fused patch layer's co-attention uses for its query
// operand, inside the 1e-3 bar on maps (checked on peaky rows).  Used for head dimensions 32 (the encoder layers) and 256
// (the one-head layer whose map is returned); the fp32 kernels above serve the other widths and remain selectable
// (mpo_set_bag_self_attention_bf16x3) as the check of this path.
// A split pass writes each operand once in the two forms the kernels read, so every tile load is a contiguous copy:
//   row form  R[head][Mp][HD]               (A operand of a score product, or a wave's own rows as B operand)
//   T form    T[head][Mp / 32][HD c][32 p]  (A operand of a second product: for column c the 32 rows of a group in MFMA
//             k-slot order p = 8 kk + 4 u + e  <->  row 16 u + 4 kk + e, i.e. the order in which a lane holds two
//             consecutive score tiles' results)
// Mp = M rounded up to 64, padding rows zero.  The same transposed-tile scheme as the fp32 kernels above otherwise.
template <int HD> struct B3Cfg {
    static constexpr int BN = HD == 32 ? 64 : 32;           // streamed rows per step
    static constexpr int NT = BN / 16;                       // 16-row tiles per step
    static constexpr int NG = BN / 32;                       // 32-row groups per step
    static constexpr int KS = HD / 32;                       // MFMA k-steps of a score product
    static constexpr int CT = HD / 16;                       // 16-column tiles of the head dimension
    // head dimension 256: tiles go global -> LDS directly (global_load_lds_dwordx4), so the images are unpadded; row-form
    // rows (512 B) keep their 16-byte chunks XOR-swizzled by the row, T-form rows (64 B) are conflict-free as they stand
    static constexpr bool kDma = HD == 256;
    // LDS images are unpadded.  ds_read_b128 is serviced in four NON-contiguous 16-lane groups ({0-3, 12-15, 20-27}, {4-11, 16-19,
    // 28-31} and the same + 32: MI355X_MICROARCH.md, LDS); a fragment read has lane (j = row, kk = 16-byte chunk), so on 64-byte
    // rows a group holds rows j, j + 12 at chunk kk and rows j + 4, j + 8 at chunk kk ^ 1 of every residue j mod 4, all four in
    // the same 64 bytes of the 256-byte bank line: the chunk is stored at position chunk ^ sw64(row), sw64 = 2 for rows 8-15 of
    // a 16-row tile, which spreads them over the four positions (r03's 80-byte padded rows were 2-way on 3 of 16 slots per
    // group: SQ_LDS_BANK_CONFLICT = 48 % of the LDS cycles of the HD-32 kernels).  512-byte rows: chunk ^ (row & 31), as before.
    static constexpr int RLD = HD;                           // LDS row stride of a row-form tile, bf16
    static constexpr int TLD = 32;                           // LDS row stride of a T-form tile, bf16 (64 bytes)
    static constexpr int RTILE = BN * RLD;                   // bf16 elements of one row-form tile
    static constexpr int TTILE = NG * HD * TLD;              // ... of one T-form tile
};
__device__ __forceinline__ int sw64(int row) { return (row >> 2) & 2; }      // (row & 8) ? 2 : 0
bool g_sa_b3 = true;

struct B3Form { const __bf16 *rh, *rl, *th, *tl; };

__device__ __forceinline__ f32x4 mma3(const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
    return acc;
}
// Tiles travel global -> registers -> LDS in two steps so that the loads of step n + 1 are in flight while step n computes
// (one workgroup per CU at head dimension 256: nothing else would hide them).
// BN rows of a row form (contiguous in memory) <-> LDS rows of HD + 8 bf16
// (At head dimension 256 the staged pieces -- 16 to 36 uint4 per thread -- no longer fit next to the operands and
// accumulators: measured with the prefetch on, dK/dV 5.9 -> 8.6 ms at 15 000 rows through spills.  There the tiles are loaded
// and committed in one go, kPipe = false.)
template <int HD> struct B3Stage {
    static constexpr bool kPipe = HD <= 64;
    static constexpr int RV = kPipe ? B3Cfg<HD>::BN * (HD / 8) / (64 * kSaWaves) : 1;     // 16-byte pieces per thread of a row-form tile
    static constexpr int TV = kPipe ? B3Cfg<HD>::NG * HD * 4 / (64 * kSaWaves) : 1;       // ... of a T-form tile (1: unused)
};
template <int HD>
__device__ __forceinline__ void b3_stage_rows(uint4 (&v)[B3Stage<HD>::RV], const __bf16* __restrict__ src) {
#pragma unroll
    for (int i = 0; i < B3Stage<HD>::RV; ++i) v[i] = reinterpret_cast<const uint4*>(src)[threadIdx.x + 64 * kSaWaves * i];
}
template <int HD>
__device__ __forceinline__ void b3_commit_rows(__bf16* tile, const uint4 (&v)[B3Stage<HD>::RV]) {
    using C = B3Cfg<HD>;
    constexpr int V8 = HD / 8;
#pragma unroll
    for (int i = 0; i < B3Stage<HD>::RV; ++i) {
        const int idx = threadIdx.x + 64 * kSaWaves * i;
        *reinterpret_cast<uint4*>(tile + (idx / V8) * C::RLD + (((idx % V8) ^ (V8 == 4 ? sw64(idx / V8) : 0)) << 3)) = v[i];
    }
}
// NG groups of a T form (contiguous: [group][HD c][32 slots]) <-> LDS rows of 40 bf16
template <int HD>
__device__ __forceinline__ void b3_stage_t(uint4 (&v)[B3Stage<HD>::TV], const __bf16* __restrict__ src) {
#pragma unroll
    for (int i = 0; i < B3Stage<HD>::TV; ++i) v[i] = reinterpret_cast<const uint4*>(src)[threadIdx.x + 64 * kSaWaves * i];
}
template <int HD>
__device__ __forceinline__ void b3_commit_t(__bf16* tile, const uint4 (&v)[B3Stage<HD>::TV]) {
    using C = B3Cfg<HD>;
#pragma unroll
    for (int i = 0; i < B3Stage<HD>::TV; ++i) {
        const int idx = threadIdx.x + 64 * kSaWaves * i;
        *reinterpret_cast<uint4*>(tile + (idx >> 2) * C::TLD + (((idx & 3) ^ sw64(idx >> 2)) << 3)) = v[i];
    }
}
// the two steps in one (no prefetch); at head dimension 256 as LDS-DMA: no staging registers, every piece of every tile of a
// step in flight at once (through registers the compiler, out of registers, waited for each 16-byte piece in turn: 24 round
// trips per step).  The caller waits (b3_dma_wait) before its barrier; the compiler does not count these loads.
__device__ __forceinline__ void b3_dma(const void* src_base /* uniform */, unsigned byte_off, void* lds_dst /* uniform */) {
    const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds_dst;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(byte_off), "s"(src_base), "s"(dst) : "memory", "m0");
#pragma clang diagnostic pop
}
__device__ __forceinline__ void b3_dma_wait() {
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0)
}
template <int HD>
__device__ __forceinline__ void b3_load_rows(__bf16* tile, const __bf16* __restrict__ src) {
    using C = B3Cfg<HD>;
    constexpr int V8 = HD / 8;
    if constexpr (C::kDma) {
        // BN rows of 512 B = BN / 2 KiB; wave w issues KiB q = w, w + 4, ..: rows 2q, 2q + 1, chunk = slot ^ (row & 31)
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
        for (int q = wave; q < C::BN / 2; q += kSaWaves) {
            const int r = 2 * q + (lane >> 5), c = (lane & 31) ^ (r & 31);
            b3_dma(src, (unsigned)(r * (HD * 2) + c * 16), reinterpret_cast<char*>(tile) + q * 1024);
        }
    } else {
#pragma unroll
        for (int idx = threadIdx.x; idx < C::BN * V8; idx += 64 * kSaWaves) {
            const uint4 v = reinterpret_cast<const uint4*>(src)[idx];
            *reinterpret_cast<uint4*>(tile + (idx / V8) * C::RLD + (((idx % V8) ^ (V8 == 4 ? sw64(idx / V8) : 0)) << 3)) = v;
        }
    }
}
template <int HD>
__device__ __forceinline__ void b3_load_t(__bf16* tile, const __bf16* __restrict__ src) {
    using C = B3Cfg<HD>;
    if constexpr (C::kDma) {
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
        for (int q = wave; q < C::NG * HD * 64 / 1024; q += kSaWaves) {        // KiB q = 16 rows of 64 B; lane -> (row, position)
            const int row = lane >> 2;
            b3_dma(src, (unsigned)(q * 1024 + row * 64 + (((lane & 3) ^ sw64(row)) << 4)), reinterpret_cast<char*>(tile) + q * 1024);
        }
    } else {
#pragma unroll
        for (int idx = threadIdx.x; idx < C::NG * HD * 4; idx += 64 * kSaWaves) {
            const uint4 v = reinterpret_cast<const uint4*>(src)[idx];
            *reinterpret_cast<uint4*>(tile + (idx >> 2) * C::TLD + (((idx & 3) ^ sw64(idx >> 2)) << 3)) = v;
        }
    }
}
// score product of tile t of a row-form tile with a wave's own rows (B operand in registers)
template <int HD>
__device__ __forceinline__ f32x4 b3_dot(const __bf16* th, const __bf16* tl, int t, const bf16x8 (&bh)[HD / 32], const bf16x8 (&bl)[HD / 32],
                                        int j, int kk) {
    using C = B3Cfg<HD>;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (C::kDma) {
        const int row = 16 * t + j;
#pragma unroll
        for (int k = 0; k < C::KS; ++k) {
            const int at = row * C::RLD + (((4 * k + kk) ^ (row & 31)) << 3);          // chunk 4 k + kk of the row, swizzled
            acc = mma3(*reinterpret_cast<const bf16x8*>(th + at), *reinterpret_cast<const bf16x8*>(tl + at), bh[k], bl[k], acc);
        }
    } else {
        static_assert(C::kDma || HD == 32, "row-form swizzle: 64-byte rows");
        const int at = (16 * t + j) * C::RLD + ((kk ^ sw64(j)) << 3);
#pragma unroll
        for (int k = 0; k < C::KS; ++k)
            acc = mma3(*reinterpret_cast<const bf16x8*>(th + at + 32 * k), *reinterpret_cast<const bf16x8*>(tl + at + 32 * k), bh[k], bl[k], acc);
    }
    return acc;
}
// second product: acc[ct] += T-form tile (group g, column 16 ct + j) . (wh, wl); two column tiles at a time with the three
// terms interleaved, so that consecutive MFMAs do not wait for each other's accumulator
template <int HD>
__device__ __forceinline__ void b3_tacc(f32x4 (&acc)[HD / 16], const __bf16* th, const __bf16* tl, int g, const bf16x8& wh, const bf16x8& wl,
                                        int j, int kk) {
    using C = B3Cfg<HD>;
#pragma unroll
    for (int ct = 0; ct < C::CT; ct += 2) {
        const int at0 = (g * HD + 16 * ct + j) * C::TLD + ((kk ^ sw64(j)) << 3), at1 = at0 + 16 * C::TLD;
        const bf16x8 ah0 = *reinterpret_cast<const bf16x8*>(th + at0), ah1 = *reinterpret_cast<const bf16x8*>(th + at1);
        const bf16x8 al0 = *reinterpret_cast<const bf16x8*>(tl + at0), al1 = *reinterpret_cast<const bf16x8*>(tl + at1);
        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah0, wh, acc[ct], 0, 0, 0);
        acc[ct + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah1, wh, acc[ct + 1], 0, 0, 0);
        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al0, wh, acc[ct], 0, 0, 0);
        acc[ct + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al1, wh, acc[ct + 1], 0, 0, 0);
        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah0, wl, acc[ct], 0, 0, 0);
        acc[ct + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah1, wl, acc[ct + 1], 0, 0, 0);
    }
}
template <int HD>
__device__ __forceinline__ void b3_own_rows(bf16x8 (&h)[HD / 32], bf16x8 (&l)[HD / 32], const B3Form& f, size_t hoff, int row, int kk) {
#pragma unroll
    for (int k = 0; k < HD / 32; ++k) {
        h[k] = *reinterpret_cast<const bf16x8*>(f.rh + hoff + (size_t)row * HD + 32 * k + 8 * kk);
        l[k] = *reinterpret_cast<const bf16x8*>(f.rl + hoff + (size_t)row * HD + 32 * k + 8 * kk);
    }
}
// two consecutive result tiles of a lane (rows 4 kk + e of tile 2g, then of tile 2g + 1) -> the B operand of a second product
__device__ __forceinline__ void b3_split8(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        __bf16 h, l;
        split_bf16(a[e], h, l); hi[e] = h; lo[e] = l;
        split_bf16(b[e], h, l); hi[4 + e] = h; lo[4 + e] = l;
    }
}
// The same split, pair by pair: one packed conversion for the two high parts, their fp32 values back by a shift and a mask, one
// packed subtraction, one packed conversion for the two low parts (5 vector instructions per two elements; element by element the
// compiler spends 6-7).  These kernels are bound by the vector ALU: every instruction per score element counts.
typedef __attribute__((ext_vector_type(2))) float f32x2_sa;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_sa;
__device__ __forceinline__ void b3_split_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_sa{a, b}, bf16x2_sa));
    const f32x2_sa hf = {__builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xFFFF0000u)};
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_sa{a, b} - hf, bf16x2_sa));
}
__device__ __forceinline__ void b3_split8p(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_sa;
    uint32_t h[4], l[4];
    b3_split_pair(a[0], a[1], h[0], l[0]);
    b3_split_pair(a[2], a[3], h[1], l[1]);
    b3_split_pair(b[0], b[1], h[2], l[2]);
    b3_split_pair(b[2], b[3], h[3], l[3]);
    hi = __builtin_bit_cast(bf16x8, u32x4_sa{h[0], h[1], h[2], h[3]});
    lo = __builtin_bit_cast(bf16x8, u32x4_sa{l[0], l[1], l[2], l[3]});
}
// keep ? v : 0 for the four elements of a lane's row group (bytes of the lane's mask word): a byte compare and a select each
__device__ __forceinline__ f32x4 b3_keep4(const SaDrop& d, uint32_t word, const f32x4& v) {
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = ((word >> (8 * r)) & 255u) >= d.thr ? v[r] : 0.f;
    return o;
}
// forward / dQ orientation (lane column = query q, NT key tiles): w[t] = word (q % 4) of the block of (q / 4, tile t's keys) --
// the one word of each block this lane's query uses, hashed directly (two multiplies per tile; no exchange between lanes)
template <int NT>
__device__ __forceinline__ void b3_words_q(const SaDrop& dr, int q, int n0, int kk, int lane, uint32_t (&w)[NT]) {
    (void)lane;
#pragma unroll
    for (int t = 0; t < NT; ++t) w[t] = sa_block_word(dr, q >> 2, (n0 + 16 * t + 4 * kk) >> 2, q & 3);
}
// dK/dV orientation (lane column = key, NT query tiles, all four words of a block used by every lane of the quad)
template <int NT>
__device__ __forceinline__ void b3_blocks_key(const SaDrop& dr, int key, int q0, int kk, int lane, uint32_t (&w)[NT][4]) {
    if constexpr (NT == 4) {
        const int lq = lane & 3;
        const uint4 b = sa_block(dr, (q0 + 16 * lq + 4 * kk) >> 2, key >> 2);
        // quad broadcasts as DPP moves (quad_perm [t, t, t, t]): one vector instruction each, no LDS round trip
#define MPO_QUAD_BCAST(v, t) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (t) * 0x55, 0xF, 0xF, false)
        w[0][0] = MPO_QUAD_BCAST(b.x, 0); w[0][1] = MPO_QUAD_BCAST(b.y, 0); w[0][2] = MPO_QUAD_BCAST(b.z, 0); w[0][3] = MPO_QUAD_BCAST(b.w, 0);
        w[1][0] = MPO_QUAD_BCAST(b.x, 1); w[1][1] = MPO_QUAD_BCAST(b.y, 1); w[1][2] = MPO_QUAD_BCAST(b.z, 1); w[1][3] = MPO_QUAD_BCAST(b.w, 1);
        w[2][0] = MPO_QUAD_BCAST(b.x, 2); w[2][1] = MPO_QUAD_BCAST(b.y, 2); w[2][2] = MPO_QUAD_BCAST(b.z, 2); w[2][3] = MPO_QUAD_BCAST(b.w, 2);
        w[3][0] = MPO_QUAD_BCAST(b.x, 3); w[3][1] = MPO_QUAD_BCAST(b.y, 3); w[3][2] = MPO_QUAD_BCAST(b.z, 3); w[3][3] = MPO_QUAD_BCAST(b.w, 3);
#undef MPO_QUAD_BCAST
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const uint4 b = sa_block(dr, (q0 + 16 * t + 4 * kk) >> 2, key >> 2);
            w[t][0] = b.x; w[t][1] = b.y; w[t][2] = b.z; w[t][3] = b.w;
        }
    }
}

// src columns [col0 + HD h, +HD) of rows [32 bx, +32) -> row and / or T form (either pointer pair may be null)
template <int HD>
__global__ __launch_bounds__(256)
void sa_b3_split_kernel(const float* __restrict__ src, int ld, int col0, int M, int Mp, __bf16* __restrict__ rh, __bf16* __restrict__ rl,
                        __bf16* __restrict__ th, __bf16* __restrict__ tl) {
    __shared__ __bf16 sh[2][32][36];
    const int h = blockIdx.y, H = gridDim.y, seq = blockIdx.z;
    const int rho = threadIdx.x >> 3, c4 = threadIdx.x & 7, row = blockIdx.x * 32 + rho;
    const size_t head = (size_t)seq * H + h;
    for (int cb = 0; cb < HD; cb += 32) {                    // 32 columns per round
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < M) v = *reinterpret_cast<const float4*>(src + ((size_t)seq * M + row) * ld + col0 + HD * h + cb + 4 * c4);
        const float x[4] = {v.x, v.y, v.z, v.w};
        bf16x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) { __bf16 a, b; split_bf16(x[e], a, b); hi[e] = a; lo[e] = b; }
        if (rh) {
            const size_t at = (head * Mp + row) * HD + cb + 4 * c4;
            *reinterpret_cast<bf16x4*>(rh + at) = hi;
            *reinterpret_cast<bf16x4*>(rl + at) = lo;
        }
        if (th) {
            const int slot = ((rho >> 2) & 3) * 8 + (rho >> 4) * 4 + (rho & 3);      // rho = 16 u + 4 kk + e
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; ++e) { sh[0][4 * c4 + e][slot] = hi[e]; sh[1][4 * c4 + e][slot] = lo[e]; }
            __syncthreads();
            const int c = threadIdx.x >> 3, p4 = (threadIdx.x & 7) * 4;
            const size_t at = ((head * (Mp / 32) + blockIdx.x) * HD + cb + c) * 32 + p4;
            bf16x4 oh, ol;
#pragma unroll
            for (int e = 0; e < 4; ++e) { oh[e] = sh[0][c][p4 + e]; ol[e] = sh[1][c][p4 + e]; }
            *reinterpret_cast<bf16x4*>(th + at) = oh;
            *reinterpret_cast<bf16x4*>(tl + at) = ol;
        }
    }
}

// grid (Mp / 64, heads, sequences); Q, K in row form, V in T form.  DROP: attention dropout on (a template parameter, not a
// uniform branch: a branch around each tile's select makes the compiler copy the whole score block at every join)
template <int HD, bool DROP>
__global__ __launch_bounds__(64 * kSaWaves)
void bag_sa_b3_fwd_kernel(B3Form Q, B3Form K, B3Form V, float* __restrict__ o, float* __restrict__ lse2, int M, int Mp, int d, float scale,
                          float drop_p, unsigned long long seed, unsigned long long offset, const unsigned long long* epoch) {
    using C = B3Cfg<HD>;
    __shared__ __attribute__((aligned(1024))) __bf16 sm[2 * C::RTILE + 2 * C::TTILE];
    __bf16 *kh = sm, *kl = sm + C::RTILE, *vh = sm + 2 * C::RTILE, *vl = vh + C::TTILE;
    const int h = blockIdx.y, seq = blockIdx.z, H = gridDim.y;
    const size_t head = (size_t)seq * H + h, hoff = head * Mp * HD;
    o += (size_t)seq * M * d;
    lse2 += head * M;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 15, kk = lane >> 4;
    const int q = blockIdx.x * 64 + 16 * wv + j;
    const SaDrop dr = sa_drop(drop_p, seed, offset, epoch, (int)head);
    bf16x8 qh[C::KS], ql[C::KS];
    b3_own_rows<HD>(qh, ql, Q, hoff, q, kk);
    const float c2 = scale * kLog2e;
    f32x4 acc[C::CT];
#pragma unroll
    for (int c = 0; c < C::CT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY;
    f32x2_sa l2 = {0.f, 0.f};                                // row sum (before dropout) in two halves: packed adds
    uint4 s_kh[B3Stage<HD>::RV], s_kl[B3Stage<HD>::RV], s_vh[B3Stage<HD>::TV], s_vl[B3Stage<HD>::TV];
    auto stage = [&](int n0) __attribute__((always_inline)) {
        b3_stage_rows<HD>(s_kh, K.rh + hoff + (size_t)n0 * HD);
        b3_stage_rows<HD>(s_kl, K.rl + hoff + (size_t)n0 * HD);
        b3_stage_t<HD>(s_vh, V.th + hoff + (size_t)n0 * HD);
        b3_stage_t<HD>(s_vl, V.tl + hoff + (size_t)n0 * HD);
    };
    if (B3Stage<HD>::kPipe) stage(0);
    for (int n0 = 0; n0 < M; n0 += C::BN) {
        __syncthreads();
        if constexpr (B3Stage<HD>::kPipe) {
            b3_commit_rows<HD>(kh, s_kh);
            b3_commit_rows<HD>(kl, s_kl);
            b3_commit_t<HD>(vh, s_vh);
            b3_commit_t<HD>(vl, s_vl);
            if (n0 + C::BN < M) stage(n0 + C::BN);
        } else {
            b3_load_rows<HD>(kh, K.rh + hoff + (size_t)n0 * HD);
            b3_load_rows<HD>(kl, K.rl + hoff + (size_t)n0 * HD);
            b3_load_t<HD>(vh, V.th + hoff + (size_t)n0 * HD);
            b3_load_t<HD>(vl, V.tl + hoff + (size_t)n0 * HD);
            if (C::kDma) b3_dma_wait();
        }
        __syncthreads();
        // Per score element: its share of the row maximum (max3 over raw products), ONE packed fma + exp2 (the scale and the
        // maximum go in together: p = exp2(c2 dot - mx)), a packed add into the row sum, the dropout select (its keep-scale is
        // applied once, to the output), the hi / lo split of the second product's operand.
        f32x4 s[C::NT];
        const bool edge = n0 + C::BN > M;                    // uniform: only the last step has keys past the end
        float rmx = -INFINITY;                               // maximum of the RAW products of the step (c2 > 0)
#pragma unroll
        for (int t = 0; t < C::NT; ++t) {
            s[t] = b3_dot<HD>(kh, kl, t, qh, ql, j, kk);
            if (edge) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s[t][r] = (n0 + 16 * t + 4 * kk + r < M) ? s[t][r] : -INFINITY;
            }
            rmx = fmaxf(fmaxf(rmx, s[t][0]), s[t][1]);
            rmx = fmaxf(fmaxf(rmx, s[t][2]), s[t][3]);
        }
        float mx = fmaxf(m, rmx * c2);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float alpha = fast_exp2(m - mx);
        m = mx;
        l2 *= alpha;
#pragma unroll
        for (int c = 0; c < C::CT; ++c) acc[c] *= alpha;
        uint32_t w[C::NT];
        if constexpr (DROP) b3_words_q<C::NT>(dr, q, n0, kk, lane, w);
        const f32x2_sa c2v = {c2, c2}, nmx = {-mx, -mx};
#pragma unroll
        for (int t = 0; t < C::NT; ++t) {
            const f32x2_sa a0 = f32x2_sa{s[t][0], s[t][1]} * c2v + nmx, a1 = f32x2_sa{s[t][2], s[t][3]} * c2v + nmx;
            s[t] = f32x4{fast_exp2(a0[0]), fast_exp2(a0[1]), fast_exp2(a1[0]), fast_exp2(a1[1])};
            l2 += f32x2_sa{s[t][0], s[t][1]};
            l2 += f32x2_sa{s[t][2], s[t][3]};
            if constexpr (DROP) s[t] = b3_keep4(dr, w[t], s[t]);
        }
#pragma unroll
        for (int g = 0; g < C::NG; ++g) {
            bf16x8 ph, pl;
            b3_split8p(s[2 * g], s[2 * g + 1], ph, pl);
            b3_tacc<HD>(acc, vh, vl, g, ph, pl, j, kk);
        }
    }
    float l = l2[0] + l2[1];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    if (q < M) {
        const float inv = dr.inv_keep / l;                   // (the kept probabilities were accumulated unscaled)
#pragma unroll
        for (int c = 0; c < C::CT; ++c)
            *reinterpret_cast<float4*>(o + (size_t)q * d + h * HD + 16 * c + 4 * kk) =
                make_float4(acc[c][0] * inv, acc[c][1] * inv, acc[c][2] * inv, acc[c][3] * inv);
        if (kk == 0) lse2[q] = m + __builtin_amdgcn_logf(l);
    }
}

// the returned map of the one-head layer on the same arithmetic: map[q][key] = exp2(s - lse2[q]); key steps split over grid.y
template <int HD>
__global__ __launch_bounds__(64 * kSaWaves)
void bag_sa_b3_map_kernel(B3Form Q, B3Form K, const float* __restrict__ lse2, float* __restrict__ map, int M, int Mp, float scale) {
    using C = B3Cfg<HD>;
    __shared__ __attribute__((aligned(1024))) __bf16 sm[2 * C::RTILE];
    __bf16 *kh = sm, *kl = sm + C::RTILE;
    const int seq = blockIdx.z;
    const size_t hoff = (size_t)seq * Mp * HD;
    lse2 += (size_t)seq * M;
    map += (size_t)seq * M * M;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 15, kk = lane >> 4;
    const int q = blockIdx.x * 64 + 16 * wv + j;
    bf16x8 qh[C::KS], ql[C::KS];
    b3_own_rows<HD>(qh, ql, Q, hoff, q, kk);
    const float ls = q < M ? lse2[q] : 0.f, c2 = scale * kLog2e;
    const int nblk = (M + C::BN - 1) / C::BN, per = (nblk + gridDim.y - 1) / gridDim.y;
    const int b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
    const bool vec = (M & 3) == 0;
    uint4 s_kh[B3Stage<HD>::RV], s_kl[B3Stage<HD>::RV];
    if (B3Stage<HD>::kPipe && b0 < b1) {
        b3_stage_rows<HD>(s_kh, K.rh + hoff + (size_t)b0 * C::BN * HD);
        b3_stage_rows<HD>(s_kl, K.rl + hoff + (size_t)b0 * C::BN * HD);
    }
    for (int b = b0; b < b1; ++b) {
        const int n0 = b * C::BN;
        __syncthreads();
        if constexpr (B3Stage<HD>::kPipe) {
            b3_commit_rows<HD>(kh, s_kh);
            b3_commit_rows<HD>(kl, s_kl);
            if (b + 1 < b1) {
                b3_stage_rows<HD>(s_kh, K.rh + hoff + (size_t)(n0 + C::BN) * HD);
                b3_stage_rows<HD>(s_kl, K.rl + hoff + (size_t)(n0 + C::BN) * HD);
            }
        } else {
            b3_load_rows<HD>(kh, K.rh + hoff + (size_t)n0 * HD);
            b3_load_rows<HD>(kl, K.rl + hoff + (size_t)n0 * HD);
            if (C::kDma) b3_dma_wait();
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < C::NT; ++t) {
            const f32x4 s = b3_dot<HD>(kh, kl, t, qh, ql, j, kk) * c2;
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

// dQ (+ delta): Q, dO row form (the wave's own rows); K, V row form and K T form streamed
template <int HD, bool DROP>
__global__ __launch_bounds__(64 * kSaWaves)
void bag_sa_b3_dq_kernel(B3Form Q, B3Form K, B3Form V, B3Form DO, const float* __restrict__ o, const float* __restrict__ d_o,
                         const float* __restrict__ lse2, float* __restrict__ dqkv, float* __restrict__ delta, int M, int Mp, int d,
                         float scale, float drop_p, unsigned long long seed, unsigned long long offset, const unsigned long long* epoch) {
    using C = B3Cfg<HD>;
    __shared__ __attribute__((aligned(1024))) __bf16 sm[4 * C::RTILE + 2 * C::TTILE];
    __bf16 *kh = sm, *kl = sm + C::RTILE, *vh = sm + 2 * C::RTILE, *vl = sm + 3 * C::RTILE, *kth = sm + 4 * C::RTILE, *ktl = kth + C::TTILE;
    const int h = blockIdx.y, seq = blockIdx.z, H = gridDim.y;
    const size_t head = (size_t)seq * H + h, hoff = head * Mp * HD;
    dqkv += (size_t)seq * M * 3 * d;
    o += (size_t)seq * M * d;
    d_o += (size_t)seq * M * d;
    lse2 += head * M;
    delta += head * M;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 15, kk = lane >> 4;
    const int q = blockIdx.x * 64 + 16 * wv + j;
    const SaDrop dr = sa_drop(drop_p, seed, offset, epoch, (int)head);
    bf16x8 qh[C::KS], ql[C::KS], doh[C::KS], dol[C::KS];
    b3_own_rows<HD>(qh, ql, Q, hoff, q, kk);
    b3_own_rows<HD>(doh, dol, DO, hoff, q, kk);
    float dl = 0.f;
    if (q < M) {
#pragma unroll
        for (int k = 0; k < C::KS; ++k) {
            const float* op = o + (size_t)q * d + h * HD + 32 * k + 8 * kk;
            const float* gp = d_o + (size_t)q * d + h * HD + 32 * k + 8 * kk;
#pragma unroll
            for (int e = 0; e < 8; ++e) dl += op[e] * gp[e];
        }
    }
    dl += __shfl_xor(dl, 16);
    dl += __shfl_xor(dl, 32);
    const float ls = q < M ? lse2[q] : INFINITY;
    if (q < M && kk == 0) delta[q] = dl;
    const float c2 = scale * kLog2e;
    f32x4 acc[C::CT];
#pragma unroll
    for (int c = 0; c < C::CT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint4 s_kh[B3Stage<HD>::RV], s_kl[B3Stage<HD>::RV], s_vh[B3Stage<HD>::RV], s_vl[B3Stage<HD>::RV], s_th[B3Stage<HD>::TV], s_tl[B3Stage<HD>::TV];
    auto stage = [&](int n0) __attribute__((always_inline)) {
        b3_stage_rows<HD>(s_kh, K.rh + hoff + (size_t)n0 * HD);
        b3_stage_rows<HD>(s_kl, K.rl + hoff + (size_t)n0 * HD);
        b3_stage_rows<HD>(s_vh, V.rh + hoff + (size_t)n0 * HD);
        b3_stage_rows<HD>(s_vl, V.rl + hoff + (size_t)n0 * HD);
        b3_stage_t<HD>(s_th, K.th + hoff + (size_t)n0 * HD);
        b3_stage_t<HD>(s_tl, K.tl + hoff + (size_t)n0 * HD);
    };
    if (B3Stage<HD>::kPipe) stage(0);
    for (int n0 = 0; n0 < M; n0 += C::BN) {
        __syncthreads();
        if constexpr (B3Stage<HD>::kPipe) {
            b3_commit_rows<HD>(kh, s_kh);
            b3_commit_rows<HD>(kl, s_kl);
            b3_commit_rows<HD>(vh, s_vh);
            b3_commit_rows<HD>(vl, s_vl);
            b3_commit_t<HD>(kth, s_th);
            b3_commit_t<HD>(ktl, s_tl);
            if (n0 + C::BN < M) stage(n0 + C::BN);
        } else {
            b3_load_rows<HD>(kh, K.rh + hoff + (size_t)n0 * HD);
            b3_load_rows<HD>(kl, K.rl + hoff + (size_t)n0 * HD);
            b3_load_rows<HD>(vh, V.rh + hoff + (size_t)n0 * HD);
            b3_load_rows<HD>(vl, V.rl + hoff + (size_t)n0 * HD);
            b3_load_t<HD>(kth, K.th + hoff + (size_t)n0 * HD);
            b3_load_t<HD>(ktl, K.tl + hoff + (size_t)n0 * HD);
            if (C::kDma) b3_dma_wait();
        }
        __syncthreads();
        uint32_t w[C::NT];
        if constexpr (DROP) b3_words_q<C::NT>(dr, q, n0, kk, lane, w);
        const bool edge = n0 + C::BN > M;
        f32x4 ds[C::NT];
        // per element: p = exp2(c2 s - lse) (packed fma), dS = p (keep ? dP / (1 - p_drop) - delta : -delta) (packed fma, byte
        // compare + select, packed multiply), the pairwise hi / lo split
        const f32x2_sa c2v = {c2, c2}, nls = {-ls, -ls}, ikv = {dr.inv_keep, dr.inv_keep}, ndl = {-dl, -dl};
#pragma unroll
        for (int t = 0; t < C::NT; ++t) {
            const f32x4 s = b3_dot<HD>(kh, kl, t, qh, ql, j, kk);
            const f32x4 dp = b3_dot<HD>(vh, vl, t, doh, dol, j, kk);
            const f32x2_sa a0 = f32x2_sa{s[0], s[1]} * c2v + nls, a1 = f32x2_sa{s[2], s[3]} * c2v + nls;
            f32x4 p = {fast_exp2(a0[0]), fast_exp2(a0[1]), fast_exp2(a1[0]), fast_exp2(a1[1])};
            if (edge) {
#pragma unroll
                for (int r = 0; r < 4; ++r) p[r] = (n0 + 16 * t + 4 * kk + r < M) ? p[r] : 0.f;
            }
            const f32x2_sa u0 = f32x2_sa{dp[0], dp[1]} * ikv + ndl, u1 = f32x2_sa{dp[2], dp[3]} * ikv + ndl;
            f32x4 u = {u0[0], u0[1], u1[0], u1[1]};
            if constexpr (DROP) {
#pragma unroll
                for (int r = 0; r < 4; ++r) u[r] = ((w[t] >> (8 * r)) & 255u) >= dr.thr ? u[r] : -dl;
            }
            const f32x2_sa d0 = f32x2_sa{p[0], p[1]} * f32x2_sa{u[0], u[1]}, d1 = f32x2_sa{p[2], p[3]} * f32x2_sa{u[2], u[3]};
            ds[t] = f32x4{d0[0], d0[1], d1[0], d1[1]};
        }
#pragma unroll
        for (int g = 0; g < C::NG; ++g) {
            bf16x8 dh, dlo;
            b3_split8p(ds[2 * g], ds[2 * g + 1], dh, dlo);
            b3_tacc<HD>(acc, kth, ktl, g, dh, dlo, j, kk);
        }
    }
    if (q < M) {
#pragma unroll
        for (int c = 0; c < C::CT; ++c)
            *reinterpret_cast<float4*>(dqkv + (size_t)q * 3 * d + h * HD + 16 * c + 4 * kk) =
                make_float4(acc[c][0] * scale, acc[c][1] * scale, acc[c][2] * scale, acc[c][3] * scale);
    }
}

// dK, dV: K, V row form (the wave's own rows); Q, dO row and T forms streamed
template <int HD, bool DROP>
__global__ __launch_bounds__(64 * kSaWaves)
void bag_sa_b3_dkv_kernel(B3Form Q, B3Form K, B3Form V, B3Form DO, const float* __restrict__ lse2, const float* __restrict__ delta,
                          float* __restrict__ dqkv, int M, int Mp, int d, float scale, float drop_p, unsigned long long seed,
                          unsigned long long offset, const unsigned long long* epoch) {
    using C = B3Cfg<HD>;
    __shared__ __attribute__((aligned(1024))) __bf16 sm[4 * C::RTILE + 4 * C::TTILE];
    __shared__ __attribute__((aligned(16))) float st[2 * C::BN];
    __bf16 *qh = sm, *ql = sm + C::RTILE, *gh = sm + 2 * C::RTILE, *gl = sm + 3 * C::RTILE;
    __bf16 *qth = sm + 4 * C::RTILE, *qtl = qth + C::TTILE, *gth = qth + 2 * C::TTILE, *gtl = qth + 3 * C::TTILE;
    float *ls_t = st, *dl_t = st + C::BN;
    const int h = blockIdx.y, seq = blockIdx.z, H = gridDim.y;
    const size_t head = (size_t)seq * H + h, hoff = head * Mp * HD;
    dqkv += (size_t)seq * M * 3 * d;
    lse2 += head * M;
    delta += head * M;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 15, kk = lane >> 4;
    const int key = blockIdx.x * 64 + 16 * wv + j;
    const SaDrop dr = sa_drop(drop_p, seed, offset, epoch, (int)head);
    bf16x8 kh[C::KS], kl[C::KS], vh[C::KS], vl[C::KS];
    b3_own_rows<HD>(kh, kl, K, hoff, key, kk);
    b3_own_rows<HD>(vh, vl, V, hoff, key, kk);
    const float c2 = scale * kLog2e;
    f32x4 dk[C::CT], dv[C::CT];
#pragma unroll
    for (int c = 0; c < C::CT; ++c) { dk[c] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    uint4 s_qh[B3Stage<HD>::RV], s_ql[B3Stage<HD>::RV], s_gh[B3Stage<HD>::RV], s_gl[B3Stage<HD>::RV];
    uint4 s_qth[B3Stage<HD>::TV], s_qtl[B3Stage<HD>::TV], s_gth[B3Stage<HD>::TV], s_gtl[B3Stage<HD>::TV];
    auto stage = [&](int q0) __attribute__((always_inline)) {
        b3_stage_rows<HD>(s_qh, Q.rh + hoff + (size_t)q0 * HD);
        b3_stage_rows<HD>(s_ql, Q.rl + hoff + (size_t)q0 * HD);
        b3_stage_rows<HD>(s_gh, DO.rh + hoff + (size_t)q0 * HD);
        b3_stage_rows<HD>(s_gl, DO.rl + hoff + (size_t)q0 * HD);
        b3_stage_t<HD>(s_qth, Q.th + hoff + (size_t)q0 * HD);
        b3_stage_t<HD>(s_qtl, Q.tl + hoff + (size_t)q0 * HD);
        b3_stage_t<HD>(s_gth, DO.th + hoff + (size_t)q0 * HD);
        b3_stage_t<HD>(s_gtl, DO.tl + hoff + (size_t)q0 * HD);
    };
    if (B3Stage<HD>::kPipe) stage(0);
    for (int q0 = 0; q0 < M; q0 += C::BN) {
        __syncthreads();
        if constexpr (B3Stage<HD>::kPipe) {
            b3_commit_rows<HD>(qh, s_qh);
            b3_commit_rows<HD>(ql, s_ql);
            b3_commit_rows<HD>(gh, s_gh);
            b3_commit_rows<HD>(gl, s_gl);
            b3_commit_t<HD>(qth, s_qth);
            b3_commit_t<HD>(qtl, s_qtl);
            b3_commit_t<HD>(gth, s_gth);
            b3_commit_t<HD>(gtl, s_gtl);
            if (q0 + C::BN < M) stage(q0 + C::BN);
        } else {
            b3_load_rows<HD>(qh, Q.rh + hoff + (size_t)q0 * HD);
            b3_load_rows<HD>(ql, Q.rl + hoff + (size_t)q0 * HD);
            b3_load_rows<HD>(gh, DO.rh + hoff + (size_t)q0 * HD);
            b3_load_rows<HD>(gl, DO.rl + hoff + (size_t)q0 * HD);
            b3_load_t<HD>(qth, Q.th + hoff + (size_t)q0 * HD);
            b3_load_t<HD>(qtl, Q.tl + hoff + (size_t)q0 * HD);
            b3_load_t<HD>(gth, DO.th + hoff + (size_t)q0 * HD);
            b3_load_t<HD>(gtl, DO.tl + hoff + (size_t)q0 * HD);
            if (C::kDma) b3_dma_wait();
        }
        if (threadIdx.x < C::BN) {
            const int qq = q0 + threadIdx.x;
            ls_t[threadIdx.x] = qq < M ? lse2[qq] : INFINITY;
            dl_t[threadIdx.x] = qq < M ? delta[qq] : 0.f;
        }
        __syncthreads();
        uint32_t w[C::NT][4];
        if constexpr (DROP) b3_blocks_key<C::NT>(dr, key, q0, kk, lane, w);
        f32x4 pd[C::NT], ds[C::NT];
        // per element: p = exp2(c2 s - lse_q) (packed fma; rows past the end carry lse = +inf: p = 0), the kept probability for
        // dV (its keep-scale is applied once, to the result), dS = p (keep ? dP / (1 - p_drop) - delta_q : -delta_q), the splits
        const f32x2_sa c2v = {c2, c2}, ikv = {dr.inv_keep, dr.inv_keep};
        const int ksh = 8 * (key & 3);                       // this lane's byte of a block word
#pragma unroll
        for (int t = 0; t < C::NT; ++t) {
            const f32x4 s = b3_dot<HD>(qh, ql, t, kh, kl, j, kk);
            const f32x4 dp = b3_dot<HD>(gh, gl, t, vh, vl, j, kk);
            const f32x4 ls = *reinterpret_cast<const f32x4*>(ls_t + 16 * t + 4 * kk);
            const f32x4 dl = *reinterpret_cast<const f32x4*>(dl_t + 16 * t + 4 * kk);
            const f32x2_sa a0 = f32x2_sa{s[0], s[1]} * c2v - f32x2_sa{ls[0], ls[1]}, a1 = f32x2_sa{s[2], s[3]} * c2v - f32x2_sa{ls[2], ls[3]};
            const f32x4 p = {fast_exp2(a0[0]), fast_exp2(a0[1]), fast_exp2(a1[0]), fast_exp2(a1[1])};
            const f32x2_sa u0 = f32x2_sa{dp[0], dp[1]} * ikv - f32x2_sa{dl[0], dl[1]}, u1 = f32x2_sa{dp[2], dp[3]} * ikv - f32x2_sa{dl[2], dl[3]};
            f32x4 u = {u0[0], u0[1], u1[0], u1[1]};
            pd[t] = p;
            if constexpr (DROP) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool kept = ((w[t][r] >> ksh) & 255u) >= dr.thr;
                    pd[t][r] = kept ? p[r] : 0.f;
                    u[r] = kept ? u[r] : -dl[r];
                }
            }
            const f32x2_sa d0 = f32x2_sa{p[0], p[1]} * f32x2_sa{u[0], u[1]}, d1 = f32x2_sa{p[2], p[3]} * f32x2_sa{u[2], u[3]};
            ds[t] = f32x4{d0[0], d0[1], d1[0], d1[1]};
        }
#pragma unroll
        for (int g = 0; g < C::NG; ++g) {
            bf16x8 ph, pl, sh_, sl_;
            b3_split8p(pd[2 * g], pd[2 * g + 1], ph, pl);
            b3_split8p(ds[2 * g], ds[2 * g + 1], sh_, sl_);
            b3_tacc<HD>(dv, gth, gtl, g, ph, pl, j, kk);
            b3_tacc<HD>(dk, qth, qtl, g, sh_, sl_, j, kk);
        }
    }
    if (key < M) {
#pragma unroll
        for (int c = 0; c < C::CT; ++c) {
            float* at = dqkv + (size_t)key * 3 * d + h * HD + 16 * c + 4 * kk;
            *reinterpret_cast<float4*>(at + d) = make_float4(dk[c][0] * scale, dk[c][1] * scale, dk[c][2] * scale, dk[c][3] * scale);
            const float ik = dr.inv_keep;                    // (the kept probabilities were accumulated unscaled)
            *reinterpret_cast<float4*>(at + 2 * d) = make_float4(dv[c][0] * ik, dv[c][1] * ik, dv[c][2] * ik, dv[c][3] * ik);
        }
    }
}

inline int b3_mp(int M) { return (M + 63) / 64 * 64; }
// head widths of the three-term path: 32 with several heads (encoder layers), 256 (the one-head layer of the medium model)
inline bool b3_geometry(int d, int H) { return (H > 1 && d == 32 * H) || (H == 1 && d == 256); }
inline bool b3_applies(int d, int H) { return g_sa_b3 && b3_geometry(d, H); }
// one operand's forms inside a workspace of bf16: [rh | rl | th | tl], each n_seq * Mp * d elements
inline B3Form b3_form(__bf16* base, int idx, size_t each) {
    __bf16* p = base + (size_t)idx * 4 * each;
    return B3Form{p, p + each, p + 2 * each, p + 3 * each};
}
template <int HD>
int b3_split(const float* src, int ld, int col0, int n_seq, int M, int H, const B3Form& f, bool rows, bool tform, hipStream_t s) {
    const int Mp = b3_mp(M);
    sa_b3_split_kernel<HD><<<dim3(Mp / 32, H, n_seq), 256, 0, s>>>(src, ld, col0, M, Mp, rows ? const_cast<__bf16*>(f.rh) : nullptr,
                                                                 rows ? const_cast<__bf16*>(f.rl) : nullptr,
                                                                 tform ? const_cast<__bf16*>(f.th) : nullptr,
                                                                 tform ? const_cast<__bf16*>(f.tl) : nullptr);
    MPO_LAUNCH_CHECK();
    return 0;
}
template <int HD>
int b3_forward(const float* qkv, int n_seq, int M, int d, int H, float drop_p, unsigned long long seed, unsigned long long offset,
               const unsigned long long* epoch, float* o, float* saved, float* map, hipStream_t s) {
    const int Mp = b3_mp(M);
    const size_t each = (size_t)n_seq * Mp * d;
    __bf16* forms = reinterpret_cast<__bf16*>(saved + ((size_t)n_seq * H * M + 3) / 4 * 4);
    const B3Form Q = b3_form(forms, 0, each), K = b3_form(forms, 1, each), V = b3_form(forms, 2, each);
    if (int rc = b3_split<HD>(qkv, 3 * d, 0, n_seq, M, H, Q, true, true, s)) return rc;
    if (int rc = b3_split<HD>(qkv, 3 * d, d, n_seq, M, H, K, true, true, s)) return rc;
    if (int rc = b3_split<HD>(qkv, 3 * d, 2 * d, n_seq, M, H, V, true, true, s)) return rc;
    const float scale = 1.0f / sqrtf((float)HD);
    if (drop_p > 0.f && (unsigned)(drop_p * 256.0f + 0.5f) > 0u)
        bag_sa_b3_fwd_kernel<HD, true><<<dim3(Mp / 64, H, n_seq), 64 * kSaWaves, 0, s>>>(Q, K, V, o, saved, M, Mp, d, scale, drop_p, seed, offset, epoch);
    else
        bag_sa_b3_fwd_kernel<HD, false><<<dim3(Mp / 64, H, n_seq), 64 * kSaWaves, 0, s>>>(Q, K, V, o, saved, M, Mp, d, scale, 0.f, seed, offset, epoch);
    MPO_LAUNCH_CHECK();
    if (map) {
        const int qb = Mp / 64, nblk = (M + B3Cfg<HD>::BN - 1) / B3Cfg<HD>::BN;
        int split = (2048 + qb - 1) / qb;
        if (split > nblk) split = nblk;
        bag_sa_b3_map_kernel<HD><<<dim3(qb, split, n_seq), 64 * kSaWaves, 0, s>>>(Q, K, saved, map, M, Mp, scale);
        MPO_LAUNCH_CHECK();
    }
    return 0;
}
template <int HD>
int b3_backward(const float* o, const float* saved, const float* d_o, int n_seq, int M, int d, int H, float drop_p, unsigned long long seed,
                unsigned long long offset, const unsigned long long* epoch, float* dqkv, float* scratch, hipStream_t s) {
    const int Mp = b3_mp(M);
    const size_t each = (size_t)n_seq * Mp * d, lse_floats = ((size_t)n_seq * H * M + 3) / 4 * 4;
    __bf16* forms = reinterpret_cast<__bf16*>(const_cast<float*>(saved) + lse_floats);
    const B3Form Q = b3_form(forms, 0, each), K = b3_form(forms, 1, each), V = b3_form(forms, 2, each);
    const B3Form DO = b3_form(reinterpret_cast<__bf16*>(scratch + lse_floats), 0, each);
    if (int rc = b3_split<HD>(d_o, d, 0, n_seq, M, H, DO, true, true, s)) return rc;
    const float scale = 1.0f / sqrtf((float)HD);
    const dim3 grid(Mp / 64, H, n_seq);
    const bool drop_on = drop_p > 0.f && (unsigned)(drop_p * 256.0f + 0.5f) > 0u;
    if (drop_on)
        bag_sa_b3_dq_kernel<HD, true><<<grid, 64 * kSaWaves, 0, s>>>(Q, K, V, DO, o, d_o, saved, dqkv, scratch, M, Mp, d, scale, drop_p, seed, offset, epoch);
    else
        bag_sa_b3_dq_kernel<HD, false><<<grid, 64 * kSaWaves, 0, s>>>(Q, K, V, DO, o, d_o, saved, dqkv, scratch, M, Mp, d, scale, 0.f, seed, offset, epoch);
    MPO_LAUNCH_CHECK();
    if (drop_on)
        bag_sa_b3_dkv_kernel<HD, true><<<grid, 64 * kSaWaves, 0, s>>>(Q, K, V, DO, saved, scratch, dqkv, M, Mp, d, scale, drop_p, seed, offset, epoch);
    else
        bag_sa_b3_dkv_kernel<HD, false><<<grid, 64 * kSaWaves, 0, s>>>(Q, K, V, DO, saved, scratch, dqkv, M, Mp, d, scale, 0.f, seed, offset, epoch);
    MPO_LAUNCH_CHECK();
    return 0;
}

inline bool sa_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

int mpo_bag_sa_supported_head_dim(int hd) { return hd == 16 || hd == 32 || hd == 64 || hd == 128 || hd == 256 || hd == 512; }
int mpo_bag_sa_set_bf16x3(int enabled) {
    const int was = g_sa_b3 ? 1 : 0;
    g_sa_b3 = enabled != 0;
    return was;
}
// floats of saved state: the log-sum-exps (rounded up to 16 bytes) + on the three-term bf16 path the forward's operand forms
// (Q rows, K rows + T, V rows + T: 3 operands x 4 arrays of bf16), which the backward reads again
size_t mpo_bag_sa_saved_floats(int n_seq, int M, int d, int H) {
    size_t n = ((size_t)n_seq * H * M + 3) / 4 * 4;
    if (b3_geometry(d, H)) n += 3 * 4 * ((size_t)n_seq * b3_mp(M) * d) / 2;
    return n;
}
// floats of backward scratch: delta + (three-term path) the four forms of dO
size_t mpo_bag_sa_bwd_floats(int n_seq, int M, int d, int H) {
    size_t n = ((size_t)n_seq * H * M + 3) / 4 * 4;
    if (b3_geometry(d, H)) n += 4 * ((size_t)n_seq * b3_mp(M) * d) / 2;
    return n;
}

int mpo_launch_bag_sa_fwd(const float* qkv, int n_seq, int M, int d, int H, float drop_p, unsigned long long seed,
                          unsigned long long offset, const unsigned long long* epoch, float* o, float* saved, float* map, hipStream_t s) {
    MPO_CHECK(n_seq >= 1 && M >= 1 && H >= 1 && d % H == 0, "bag self-attention: %d sequences of %d rows, d=%d, heads=%d", n_seq, M, d, H);
    MPO_CHECK(n_seq <= 65535 && H <= 65535, "bag self-attention: %d sequences x %d heads exceed the grid", n_seq, H);
    const int hd = d / H;
    MPO_CHECK(mpo_bag_sa_supported_head_dim(hd), "bag self-attention: head dimension %d (16, 32, 64, 128, 256 or 512)", hd);
    MPO_CHECK(map == nullptr || H == 1, "bag self-attention: the M x M map is returned for one head only (heads=%d)", H);
    MPO_CHECK(sa_al16(qkv) && sa_al16(o) && sa_al16(saved) && (map == nullptr || sa_al16(map)), "bag self-attention: buffers must be 16-byte aligned");
    float* lse2 = saved;
    if (b3_applies(d, H)) {
        if (hd == 32) return b3_forward<32>(qkv, n_seq, M, d, H, drop_p, seed, offset, epoch, o, saved, map, s);
        return b3_forward<256>(qkv, n_seq, M, d, H, drop_p, seed, offset, epoch, o, saved, map, s);
    }
    switch (hd) {
        case 16: return sa_forward<16>(qkv, n_seq, M, d, H, drop_p, seed, offset, epoch, o, lse2, map, s);
        case 32: return sa_forward<32>(qkv, n_seq, M, d, H, drop_p, seed, offset, epoch, o, lse2, map, s);
        case 64: return sa_forward<64>(qkv, n_seq, M, d, H, drop_p, seed, offset, epoch, o, lse2, map, s);
        case 128: return sa_forward<128>(qkv, n_seq, M, d, H, drop_p, seed, offset, epoch, o, lse2, map, s);
        case 512: return sa_forward<512>(qkv, n_seq, M, d, H, drop_p, seed, offset, epoch, o, lse2, map, s);
        default: return sa_forward<256>(qkv, n_seq, M, d, H, drop_p, seed, offset, epoch, o, lse2, map, s);
    }
}
// scratch: mpo_bag_sa_bwd_floats() floats
int mpo_launch_bag_sa_bwd(const float* qkv, const float* o, const float* saved, const float* d_o, int n_seq, int M, int d, int H,
                          float drop_p, unsigned long long seed, unsigned long long offset, const unsigned long long* epoch,
                          float* dqkv, float* scratch, hipStream_t s) {
    MPO_CHECK(n_seq >= 1 && M >= 1 && H >= 1 && d % H == 0, "bag self-attention: %d sequences of %d rows, d=%d, heads=%d", n_seq, M, d, H);
    MPO_CHECK(n_seq <= 65535 && H <= 65535, "bag self-attention: %d sequences x %d heads exceed the grid", n_seq, H);
    const int hd = d / H;
    MPO_CHECK(mpo_bag_sa_supported_head_dim(hd), "bag self-attention: head dimension %d (16, 32, 64, 128, 256 or 512)", hd);
    MPO_CHECK(sa_al16(qkv) && sa_al16(o) && sa_al16(d_o) && sa_al16(dqkv) && sa_al16(saved) && sa_al16(scratch),
              "bag self-attention: buffers must be 16-byte aligned");
    const float* lse2 = saved;
    float* delta = scratch;
    if (b3_applies(d, H)) {
        if (hd == 32) return b3_backward<32>(o, saved, d_o, n_seq, M, d, H, drop_p, seed, offset, epoch, dqkv, scratch, s);
        return b3_backward<256>(o, saved, d_o, n_seq, M, d, H, drop_p, seed, offset, epoch, dqkv, scratch, s);
    }
    switch (hd) {
        case 16: return sa_backward<16>(qkv, o, lse2, d_o, n_seq, M, d, H, drop_p, seed, offset, epoch, dqkv, delta, s);
        case 32: return sa_backward<32>(qkv, o, lse2, d_o, n_seq, M, d, H, drop_p, seed, offset, epoch, dqkv, delta, s);
        case 64: return sa_backward<64>(qkv, o, lse2, d_o, n_seq, M, d, H, drop_p, seed, offset, epoch, dqkv, delta, s);
        case 128: return sa_backward<128>(qkv, o, lse2, d_o, n_seq, M, d, H, drop_p, seed, offset, epoch, dqkv, delta, s);
        case 512: return sa_backward<512>(qkv, o, lse2, d_o, n_seq, M, d, H, drop_p, seed, offset, epoch, dqkv, delta, s);
        default: return sa_backward<256>(qkv, o, lse2, d_o, n_seq, M, d, H, drop_p, seed, offset, epoch, dqkv, delta, s);
    }
}
