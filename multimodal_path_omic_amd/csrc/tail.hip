// Kernels of the 6 x d token tail that are not GEMMs (K3 CAG, K4 set-Transformer,
// K5 gated attention-MIL pooling, K6 fusion head).  Every linear of the tail runs on gemm_f32.hip with
// fused bias / activation / dropout / residual epilogues and fused activation-derivative gates; what is
// left is LayerNorm, the 8-head self-attention over T = N tokens of one slide, the softmax pooling and
// the survival head.  All of it is launch-latency-bound small-row work: one wave per row or one
// workgroup per slide, fp32 throughout.
#include "mpo_common.h"
#include "mpo_kernels.h"

namespace {

__device__ __forceinline__ float elu_f(float v) { return v > 0.f ? v : expm1f(v); }
__device__ __forceinline__ float elu_grad_from_out(float y) { return y > 0.f ? 1.0f : y + 1.0f; }

// ------------------------------------------------------------------ LayerNorm (one wave per row)
// Rows may belong to several BRANCHES (independent modules of identical shape batched into one launch, e.g. the
// path and the omic set-Transformer): branch = row / rows_per_branch picks the affine parameters.
// D4: float4 columns per lane fixed at compile time (1: d = 256, the model's width; 2: d = 512; 0: any d, strided loops).
// The fixed-width versions read the row ONCE into registers; the strided one re-reads it for the variance and the output
// and waits for every load of its run-time loops in turn (4.8 us -> the launch floor for 384 rows).
template <int D4>
__global__ void ln_fwd_kernel(const float* __restrict__ x, LnBranches p, float* __restrict__ y, float* __restrict__ stats,
                              int rows, int d, float eps) {
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const int br = r / p.rows_per_branch;
    const float* __restrict__ w = p.w[br];
    const float* __restrict__ b = p.b[br];
    const float* xr = x + (size_t)r * d;
    if constexpr (D4 > 0) {
        float4 v[D4], wv[D4], bv[D4];
#pragma unroll
        for (int k = 0; k < D4; ++k) {
            v[k] = *reinterpret_cast<const float4*>(xr + (k * 64 + lane) * 4);
            wv[k] = *reinterpret_cast<const float4*>(w + (k * 64 + lane) * 4);
            bv[k] = *reinterpret_cast<const float4*>(b + (k * 64 + lane) * 4);
        }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < D4; ++k) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
        const float mean = wave_sum(s) / d;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < D4; ++k) {
            v[k].x -= mean; v[k].y -= mean; v[k].z -= mean; v[k].w -= mean;
            q += (v[k].x * v[k].x + v[k].y * v[k].y) + (v[k].z * v[k].z + v[k].w * v[k].w);
        }
        const float rstd = rsqrtf(wave_sum(q) / d + eps);
#pragma unroll
        for (int k = 0; k < D4; ++k) {
            float4 o;
            o.x = v[k].x * rstd * wv[k].x + bv[k].x;
            o.y = v[k].y * rstd * wv[k].y + bv[k].y;
            o.z = v[k].z * rstd * wv[k].z + bv[k].z;
            o.w = v[k].w * rstd * wv[k].w + bv[k].w;
            *reinterpret_cast<float4*>(y + (size_t)r * d + (k * 64 + lane) * 4) = o;
        }
        if (lane == 0) { stats[2 * r] = mean; stats[2 * r + 1] = rstd; }
    } else {
        float s = 0.f;
        for (int c = lane; c < d; c += 64) s += xr[c];
        const float mean = wave_sum(s) / d;
        float v = 0.f;
        for (int c = lane; c < d; c += 64) { const float t = xr[c] - mean; v += t * t; }
        const float rstd = rsqrtf(wave_sum(v) / d + eps);
        for (int c = lane; c < d; c += 64) y[(size_t)r * d + c] = (xr[c] - mean) * rstd * w[c] + b[c];
        if (lane == 0) { stats[2 * r] = mean; stats[2 * r + 1] = rstd; }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * w    [dx may alias dy]
template <int D4>
__device__ __forceinline__ void ln_bwd_rows(int rb, const float* __restrict__ dy, const float* __restrict__ x,
                                            const float* __restrict__ stats, const LnBranches& p, float* __restrict__ dx,
                                            int rows, int d, int accumulate) {
    const int r = rb * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* __restrict__ w = p.w[r / p.rows_per_branch];
    const float mean = stats[2 * r], rstd = stats[2 * r + 1];
    if constexpr (D4 > 0) {
        float g[D4][4], xh[D4][4], acc[D4][4];
#pragma unroll
        for (int k = 0; k < D4; ++k) {
            const size_t at = (size_t)r * d + (k * 64 + lane) * 4;
            const float4 gv = *reinterpret_cast<const float4*>(dy + at);
            const float4 xv = *reinterpret_cast<const float4*>(x + at);
            const float4 wv = *reinterpret_cast<const float4*>(w + (k * 64 + lane) * 4);
            float4 av = make_float4(0.f, 0.f, 0.f, 0.f);
            if (accumulate) av = *reinterpret_cast<const float4*>(dx + at);
            g[k][0] = gv.x * wv.x; g[k][1] = gv.y * wv.y; g[k][2] = gv.z * wv.z; g[k][3] = gv.w * wv.w;
            xh[k][0] = (xv.x - mean) * rstd; xh[k][1] = (xv.y - mean) * rstd;
            xh[k][2] = (xv.z - mean) * rstd; xh[k][3] = (xv.w - mean) * rstd;
            acc[k][0] = av.x; acc[k][1] = av.y; acc[k][2] = av.z; acc[k][3] = av.w;
        }
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int k = 0; k < D4; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) { c1 += g[k][j]; c2 += g[k][j] * xh[k][j]; }
        c1 = wave_sum(c1) / d;
        c2 = wave_sum(c2) / d;
#pragma unroll
        for (int k = 0; k < D4; ++k) {
            float4 o;
            o.x = acc[k][0] + rstd * (g[k][0] - c1 - xh[k][0] * c2);
            o.y = acc[k][1] + rstd * (g[k][1] - c1 - xh[k][1] * c2);
            o.z = acc[k][2] + rstd * (g[k][2] - c1 - xh[k][2] * c2);
            o.w = acc[k][3] + rstd * (g[k][3] - c1 - xh[k][3] * c2);
            *reinterpret_cast<float4*>(dx + (size_t)r * d + (k * 64 + lane) * 4) = o;
        }
    } else {
        float c1 = 0.f, c2 = 0.f;
        for (int c = lane; c < d; c += 64) {
            const float g = dy[(size_t)r * d + c] * w[c];
            const float xh = (x[(size_t)r * d + c] - mean) * rstd;
            c1 += g;
            c2 += g * xh;
        }
        c1 = wave_sum(c1) / d;
        c2 = wave_sum(c2) / d;
        for (int c = lane; c < d; c += 64) {
            const float g = dy[(size_t)r * d + c] * w[c];
            const float xh = (x[(size_t)r * d + c] - mean) * rstd;
            const float v = rstd * (g - c1 - xh * c2);
            dx[(size_t)r * d + c] = accumulate ? dx[(size_t)r * d + c] + v : v;
        }
    }
}

// dw[c] = sum_r dy * xhat,  db[c] = sum_r dy over the rows of one branch.  16 columns per workgroup, 16 row groups of
// 16 lanes each (the first version used 64 columns x 4 row groups = 4 workgroups for d = 256 and took 19 us on 192 rows).
// chunk / n_chunks > 1 (long row axes: the rows of a bag): the branch's rows are cut into n_chunks pieces, each workgroup adds
// its piece into dw / db (zeroed by the launcher) -- 16 workgroups walking 15 000 rows took 0.31 ms per call
__device__ __forceinline__ void ln_bwd_params_cols(int cb, int br, const float* __restrict__ dy, const float* __restrict__ x,
                                                   const float* __restrict__ stats, const LnBranches& p, int d, int chunk, int n_chunks) {
    __shared__ float red[2][16][17];
    const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int col = cb * 16 + c;
    const int per = (p.rows_per_branch + n_chunks - 1) / n_chunks;
    const int r0 = br * p.rows_per_branch + chunk * per, r1 = min(br * p.rows_per_branch + p.rows_per_branch, r0 + per);
    float a = 0.f, bsum = 0.f;
    if (col < d)
        for (int r = r0 + rg; r < r1; r += 16) {
            const float g = dy[(size_t)r * d + col];
            a += g * (x[(size_t)r * d + col] - stats[2 * r]) * stats[2 * r + 1];
            bsum += g;
        }
    red[0][rg][c] = a;
    red[1][rg][c] = bsum;
    __syncthreads();
    if (rg == 0 && col < d) {
        float ta = 0.f, tb = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { ta += red[0][k][c]; tb += red[1][k][c]; }
        if (n_chunks > 1) {
            atomicAdd(&p.dw[br][col], ta);
            atomicAdd(&p.db[br][col], tb);
        } else {
            p.dw[br][col] = ta;
            p.db[br][col] = tb;
        }
    }
}
// what: 1 = dx rows, 2 = parameter gradients, 3 = both in ONE launch (row blocks first, then column blocks)
template <int D4>
__global__ __launch_bounds__(256)
void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ stats, LnBranches p,
                   float* __restrict__ dx, int rows, int d, int accumulate, int what, int n_chunks) {
    const int row_blocks = (what & 1) ? (rows + 3) / 4 : 0;
    if ((int)blockIdx.x < row_blocks) {
        ln_bwd_rows<D4>(blockIdx.x, dy, x, stats, p, dx, rows, d, accumulate);
    } else {
        const int cblocks = (d + 15) / 16, i = blockIdx.x - row_blocks;
        ln_bwd_params_cols(i % cblocks, (i / cblocks) % p.n, dy, x, stats, p, d, i / (cblocks * p.n), n_chunks);
    }
}

// ------------------------------------------------------------------ self-attention over the T tokens of one slide
// qkv [B*T][3d] (q | k | v), H heads of hd = d / H.  One workgroup per slide, everything through LDS.
constexpr int kMaxT = 16;
constexpr int kPoolLongL = 64;          // pooling axes longer than this are spread over the grid

// One WAVE per (slide, head): the four waves of a workgroup take four consecutive (slide, head) pairs and never wait for
// each other beyond the workgroup barriers between phases.  (First version: one workgroup per slide, everything for its 8
// heads through LDS in five phases, the softmax + dropout phase on 48 threads with six Philox draws each: 11 us forward,
// 14 us backward for 64 slides -- the latency of the phases, not the work.)
// LDS of a wave: q | k | v rows of its head [3][T][hd], then scratch.
__device__ __forceinline__ void mha_load_head(float* sq, const float* __restrict__ qkv, int b, int h, int T, int d, int hd, int lane) {
    // row (part, i) of the head: hd contiguous floats at qkv[(b T + i) 3d + part d + h hd]
    const int n = 3 * T * hd;
    for (int base = 0; base < n; base += 12 * 64) {         // (576 values at T = 6, hd = 32: one round of nine loads)
        float v[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int idx = base + k * 64 + lane;
            const int ii = idx < n ? idx : 0;
            const int part = ii / (T * hd), i = (ii / hd) % T, c = ii % hd;
            v[k] = qkv[((size_t)b * T + i) * 3 * d + part * d + h * hd + c];
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int idx = base + k * 64 + lane;
            if (idx < n) sq[idx] = v[k];
        }
    }
}

// TC / HDC: compile-time T and head dimension (0: run-time values).  With run-time loop bounds every LDS read of the dot
// and context loops is waited for before the next is issued; the model's shape (T = 6 tokens, head dimension 32) gets
// fully unrolled loops.
template <int TC, int HDC>
__global__ __launch_bounds__(256)
void mha_small_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ o, float* __restrict__ p_save /* [B][H][T][T] x2: p, p_post */,
                          int B, int T_, int d, int H, float drop_p, unsigned long long seed, unsigned long long offset_,
                          const unsigned long long* epoch) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const unsigned long long offset = epoch_offset(offset_, epoch);
    const int T = TC ? TC : T_;
    const int hd = HDC ? HDC : d / H, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per_wave = 3 * T * hd + 2 * T * T + 2 * T;
    float* sq = sm + wave * per_wave;     // [3][T][hd]
    float* sp = sq + 3 * T * hd;          // [T][T] scores, then p_post
    float* sst = sp + 2 * T * T;          // [T] row max, [T] 1 / row sum
    const int gid = blockIdx.x * 4 + wave;
    const bool live = gid < B * H;
    const int b = live ? gid / H : 0, h = live ? gid % H : 0;
    const float scale = rsqrtf((float)hd);
    if (live) mha_load_head(sq, qkv, b, h, T, d, hd, lane);
    __syncthreads();
    if (live) {
        for (int it = lane; it < T * T; it += 64) {
            const int i = it / T, j = it % T;
            const float* qi = sq + i * hd;
            const float* kj = sq + (T + j) * hd;
            // the rows of all (i, j) start on the same LDS banks: walk the head dimension from a per-lane rotation
            const int rot = lane % hd;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < hd; ++c) {
                const int cc = c + rot < hd ? c + rot : c + rot - hd;
                s += qi[cc] * kj[cc];
            }
            sp[it] = s * scale;
        }
    }
    __syncthreads();
    if (live && lane < T) {
        const float* row = sp + lane * T;
        float mx = row[0];
        for (int j = 1; j < T; ++j) mx = fmaxf(mx, row[j]);
        float l = 0.f;
        for (int j = 0; j < T; ++j) l += __expf(row[j] - mx);
        sst[lane] = mx;
        sst[T + lane] = 1.0f / l;
    }
    __syncthreads();
    if (live) {
        const float inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
        float* pb = p_save + (size_t)b * 2 * H * T * T;
        for (int it = lane; it < T * T; it += 64) {
            const int i = it / T;
            const int row_g = h * T + i;                                         // row index of the [H T][T] layout
            const float p = __expf(sp[it] - sst[i]) * sst[T + i];
            float pp = p;
            if (drop_p > 0.f) pp *= dropout_keep(seed, offset, ((size_t)b * H * T + row_g) * T + (it % T), drop_p, inv_keep);
            pb[row_g * T + (it % T)] = p;
            pb[H * T * T + row_g * T + (it % T)] = pp;
            sp[T * T + it] = pp;
        }
    }
    __syncthreads();
    if (live) {
        for (int idx = lane; idx < T * hd; idx += 64) {
            const int i = idx / hd, c = idx % hd;
            const float* pr = sp + T * T + i * T;
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < T; ++j) a += pr[j] * sq[(2 * T + j) * hd + c];
            o[((size_t)b * T + i) * d + h * hd + c] = a;
        }
    }
}

template <int TC, int HDC>
__global__ __launch_bounds__(256)
void mha_small_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ p_save, const float* __restrict__ d_o,
                          float* __restrict__ dqkv, int B, int T_, int d, int H) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int T = TC ? TC : T_;
    const int hd = HDC ? HDC : d / H, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per_wave = 4 * T * hd + 4 * T * T;
    float* sq = sm + wave * per_wave;     // [3][T][hd]
    float* sdo = sq + 3 * T * hd;         // [T][hd]
    float* spp = sdo + T * hd;            // [2][T][T]  P, P after dropout
    float* sdp = spp + 2 * T * T;         // [T][T]  dP * keep-scale
    float* sds = sdp + T * T;             // [T][T]  dS
    const int gid = blockIdx.x * 4 + wave;
    const bool live = gid < B * H;
    const int b = live ? gid / H : 0, h = live ? gid % H : 0;
    const float scale = rsqrtf((float)hd);
    if (live) {
        mha_load_head(sq, qkv, b, h, T, d, hd, lane);
        for (int idx = lane; idx < T * hd; idx += 64) sdo[idx] = d_o[((size_t)b * T + idx / hd) * d + h * hd + idx % hd];
        const float* pb = p_save + (size_t)b * 2 * H * T * T;
        for (int it = lane; it < T * T; it += 64) {
            spp[it] = pb[(h * T) * T + it];
            spp[T * T + it] = pb[H * T * T + (h * T) * T + it];
        }
    }
    __syncthreads();
    if (live) {
        // dP[i][j] = do[i] . v[j], times the dropout keep-scale of (i, j)
        for (int it = lane; it < T * T; it += 64) {
            const int i = it / T, j = it % T;
            const float* vj = sq + (2 * T + j) * hd;
            const float* doi = sdo + i * hd;
            const int rot = lane % hd;
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < hd; ++c) {
                const int cc = c + rot < hd ? c + rot : c + rot - hd;
                a += doi[cc] * vj[cc];
            }
            const float p = spp[it];
            sdp[it] = a * (p > 0.f ? spp[T * T + it] / p : 0.f);
        }
    }
    __syncthreads();
    if (live && lane < T) {
        float delta = 0.f;
        for (int j = 0; j < T; ++j) delta += spp[lane * T + j] * sdp[lane * T + j];
        for (int j = 0; j < T; ++j) sds[lane * T + j] = spp[lane * T + j] * (sdp[lane * T + j] - delta) * scale;
    }
    __syncthreads();
    if (live) {
        for (int idx = lane; idx < T * hd; idx += 64) {
            const int t = idx / hd, c = idx % hd;
            float dq = 0.f, dk = 0.f, dv = 0.f;
#pragma unroll
            for (int u = 0; u < T; ++u) {
                dq += sds[t * T + u] * sq[(T + u) * hd + c];                   // dS[t][u] k[u]
                dk += sds[u * T + t] * sq[u * hd + c];                         // dS[u][t] q[u]
                dv += spp[T * T + u * T + t] * sdo[u * hd + c];                // Ppost[u][t] do[u]
            }
            float* out = dqkv + ((size_t)b * T + t) * 3 * d + h * hd + c;
            out[0] = dq;
            out[d] = dk;
            out[2 * d] = dv;
        }
    }
}

// ------------------------------------------------------------------ softmax pooling over the L rows of a slide
// scores [B*L] raw, x [B*L][d] -> w [B*L] = softmax_l(scores), h [B][d] = sum_l w_l x_l.  One workgroup per slide.
__device__ __forceinline__ float blk_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float blk_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256)
void pool_fwd_kernel(const float* __restrict__ scores, const float* __restrict__ x, float* __restrict__ w,
                     float* __restrict__ h, int L, int d) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* sc = scores + (size_t)b * L;
    float mx = -INFINITY;
    for (int l = tid; l < L; l += 256) mx = fmaxf(mx, sc[l]);
    mx = blk_max(mx, red);
    float s = 0.f;
    for (int l = tid; l < L; l += 256) s += __expf(sc[l] - mx);
    s = blk_sum(s, red);
    const float inv = 1.0f / s;
    for (int l = tid; l < L; l += 256) w[(size_t)b * L + l] = __expf(sc[l] - mx) * inv;
    __syncthreads();
    for (int c = tid; c < d; c += 256) {
        float a = 0.f;
        for (int l = 0; l < L; ++l) a += w[(size_t)b * L + l] * x[((size_t)b * L + l) * d + c];
        h[(size_t)b * d + c] = a;
    }
}

// dh [B][d], optional d_scores_ext [B*L] -> d_scores [B*L], dx [B*L][d] = w_l dh
__global__ __launch_bounds__(256)
void pool_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ x, const float* __restrict__ w,
                     const float* __restrict__ d_ext, float* __restrict__ d_scores, float* __restrict__ dx, int L, int d) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* dhb = dh + (size_t)b * d;
    // dw_l = dh . x_l  (one wave per row, strided), stashed in d_scores
    for (int l = wv; l < L; l += 4) {
        float a = 0.f;
        for (int c = lane; c < d; c += 64) a += dhb[c] * x[((size_t)b * L + l) * d + c];
        a = wave_sum(a);
        if (lane == 0) d_scores[(size_t)b * L + l] = a;
    }
    __syncthreads();
    float dl = 0.f;
    for (int l = tid; l < L; l += 256) dl += w[(size_t)b * L + l] * d_scores[(size_t)b * L + l];
    dl = blk_sum(dl, red);
    for (int l = tid; l < L; l += 256) {
        const size_t i = (size_t)b * L + l;
        d_scores[i] = w[i] * (d_scores[i] - dl) + (d_ext ? d_ext[i] : 0.f);
    }
    for (int it = tid; it < L * d; it += 256) {
        const int l = it / d, c = it % d;
        dx[((size_t)b * L + l) * d + c] = w[(size_t)b * L + l] * dhb[c];
    }
}

// ---- the pooling head's scorer folded into the pooling kernels (token tail, L <= 64): scores_l = (a_l (.) b_l) . w_c + b_c is six
// dot products per slide and branch -- as its own 256 -> 1 product it was a launch on the general (guarded) GEMM body plus an
// a * b pass before it, and in the backward a second grouped launch plus a dab * b / dab * a pass (models/blocks.py:42-48,
// models/mcat/mcat.py:105-107).  One workgroup per (branch, slide).
__global__ __launch_bounds__(256)
void pool_score_fwd_kernel(const float* __restrict__ a, const float* __restrict__ bg, const float* __restrict__ x, PoolScorer ps,
                           float* __restrict__ scores, float* __restrict__ w, float* __restrict__ h, int L, int d) {
    __shared__ float sc[64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int br = b / ps.n_slides;
    const float* wc = ps.wc[br];
    for (int l = wv; l < L; l += 4) {                             // one wave per token row
        const size_t row = ((size_t)b * L + l) * d;
        float v = 0.f;
        for (int c = lane; c < d; c += 64) v += a[row + c] * bg[row + c] * wc[c];
        v = wave_sum(v);
        if (lane == 0) {
            v += ps.bc[br][0];
            sc[l] = v;
            scores[(size_t)b * L + l] = v;
        }
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int l = 0; l < L; ++l) mx = fmaxf(mx, sc[l]);
    float sum = 0.f;
    for (int l = 0; l < L; ++l) sum += __expf(sc[l] - mx);
    const float inv = 1.0f / sum;
    if (tid < L) w[(size_t)b * L + tid] = __expf(sc[tid] - mx) * inv;
    for (int c = tid; c < d; c += 256) {
        float acc = 0.f;
        for (int l = 0; l < L; ++l) acc += __expf(sc[l] - mx) * inv * x[((size_t)b * L + l) * d + c];
        h[(size_t)b * d + c] = acc;
    }
}

// d_scores of one slide into ds[0..L) (LDS): ds_l = w_l (dh . x_l - sum_k w_k dh . x_k) + d_ext_l
__device__ __forceinline__ void pool_dscores(const float* __restrict__ dhb, const float* __restrict__ x, const float* __restrict__ w,
                                             const float* __restrict__ d_ext, size_t b, int L, int d, float* ds, int tid) {
    const int lane = tid & 63, wv = tid >> 6;
    for (int l = wv; l < L; l += 4) {
        float v = 0.f;
        for (int c = lane; c < d; c += 64) v += dhb[c] * x[(b * L + l) * d + c];
        v = wave_sum(v);
        if (lane == 0) ds[l] = v;
    }
    __syncthreads();
    float dl = 0.f;
    for (int l = 0; l < L; ++l) dl += w[b * L + l] * ds[l];
    __syncthreads();
    if (tid < L) ds[tid] = w[b * L + tid] * (ds[tid] - dl) + (d_ext ? d_ext[b * L + tid] : 0.f);
    __syncthreads();
}
// per slide: d_scores, dx = w_l dh, da = d_scores w_c (.) b, db = d_scores w_c (.) a.  (The scorer's own gradients,
// dW_c = sum_rows d_scores (a (.) b) and db_c = sum_rows d_scores, need every slide's d_scores: they ride in the launch that
// follows, tail_api.hip -- computed here by one workgroup per branch walking its 32 slides they took 130 us.)
__global__ __launch_bounds__(256)
void pool_score_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ x, const float* __restrict__ w,
                           const float* __restrict__ d_ext, const float* __restrict__ a, const float* __restrict__ bg, PoolScorer ps,
                           float* __restrict__ d_scores, float* __restrict__ dx, float* __restrict__ da, float* __restrict__ db,
                           int L, int d) {
    __shared__ float ds[64];
    const int tid = threadIdx.x;
    const size_t b = blockIdx.x;
    const int br = (int)b / ps.n_slides;
    const float* dhb = dh + b * d;
    const float* wc = ps.wc[br];
    pool_dscores(dhb, x, w, d_ext, b, L, d, ds, tid);
    if (tid < L) d_scores[b * L + tid] = ds[tid];
    for (int it = tid; it < L * d; it += 256) {
        const int l = it / d, c = it % d;
        const size_t i = (b * L + l) * d + c;
        const float g = ds[l] * wc[c];
        dx[i] = w[b * L + l] * dhb[c];
        da[i] = g * bg[i];
        db[i] = g * a[i];
    }
}

// ---- the same pooling over a LONG axis (L = the M rows of a bag: models/ge_nacagat/ge_nacagat.py:56-58).  The kernels above
// give one workgroup a whole slide and walk L serially; here L is spread over the grid.
// w = softmax_L(scores): one workgroup of 1024 threads per slide
__global__ __launch_bounds__(1024)
void pool_long_softmax_kernel(const float* __restrict__ scores, float* __restrict__ w, int L) {
    __shared__ float red[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* sc = scores + (size_t)b * L;
    float mx = -INFINITY;
    for (int l = tid; l < L; l += 1024) mx = fmaxf(mx, sc[l]);
    mx = wave_max(mx);
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) mx = fmaxf(mx, red[k]);
    __syncthreads();
    float sum = 0.f;
    for (int l = tid; l < L; l += 1024) sum += __expf(sc[l] - mx);
    sum = wave_sum(sum);
    if (lane == 0) red[wv] = sum;
    __syncthreads();
    sum = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) sum += red[k];
    const float inv = 1.0f / sum;
    for (int l = tid; l < L; l += 1024) w[(size_t)b * L + l] = __expf(sc[l] - mx) * inv;
}
// h[b][c] = sum_l w[l] x[l][c]: 16 columns per workgroup, 16 row groups of 16 lanes, fixed summation order
__global__ __launch_bounds__(256)
void pool_long_wsum_kernel(const float* __restrict__ w, const float* __restrict__ x, float* __restrict__ h, int L, int d) {
    __shared__ float red[16][17];
    const int b = blockIdx.y, c = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + c;
    const float* wb = w + (size_t)b * L;
    const float* xb = x + (size_t)b * L * d;
    float a = 0.f;
    if (col < d) {
        int l = rg;
        for (; l + 7 * 16 < L; l += 8 * 16) {
            float wv[8], xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { wv[u] = wb[l + 16 * u]; xv[u] = xb[(size_t)(l + 16 * u) * d + col]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) a += wv[u] * xv[u];
        }
        for (; l < L; l += 16) a += wb[l] * xb[(size_t)l * d + col];
    }
    red[rg][c] = a;
    __syncthreads();
    if (rg == 0 && col < d) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][c];
        h[(size_t)b * d + col] = t;
    }
}
// dw[l] = dh . x[l] (one wave per row) -> d_scores (stash), and dx[l] = w[l] dh
__global__ __launch_bounds__(256)
void pool_long_rows_kernel(const float* __restrict__ dh, const float* __restrict__ x, const float* __restrict__ w,
                           float* __restrict__ d_scores, float* __restrict__ dx, int L, int d) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int l = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (l >= L) return;
    const size_t i = (size_t)b * L + l;
    const float* dhb = dh + (size_t)b * d;
    const float wl = w[i];
    float a = 0.f;
    for (int c = lane; c < d; c += 64) {
        const float g = dhb[c];
        a += g * x[i * d + c];
        dx[i * d + c] = wl * g;
    }
    a = wave_sum(a);
    if (lane == 0) d_scores[i] = a;
}
// d_scores[l] = w[l] (dw[l] - sum_l' w[l'] dw[l']) + d_ext[l]
__global__ __launch_bounds__(1024)
void pool_long_dscore_kernel(const float* __restrict__ w, const float* __restrict__ d_ext, float* __restrict__ d_scores, int L) {
    __shared__ float red[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float dl = 0.f;
    for (int l = tid; l < L; l += 1024) dl += w[(size_t)b * L + l] * d_scores[(size_t)b * L + l];
    dl = wave_sum(dl);
    if (lane == 0) red[wv] = dl;
    __syncthreads();
    dl = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) dl += red[k];
    for (int l = tid; l < L; l += 1024) {
        const size_t i = (size_t)b * L + l;
        d_scores[i] = w[i] * (d_scores[i] - dl) + (d_ext ? d_ext[i] : 0.f);
    }
}

// ------------------------------------------------------------------ survival head (models/mcat/mcat.py:126-138)
constexpr int kMaxC = 16;
__global__ void head_fwd_kernel(const float* __restrict__ logits, float* __restrict__ hazards, float* __restrict__ survs,
                                float* __restrict__ y, int B, int C) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* lg = logits + (size_t)b * C;
    float mx = lg[0];
    for (int j = 1; j < C; ++j) mx = fmaxf(mx, lg[j]);
    float s = 0.f, run = 1.0f;
    for (int j = 0; j < C; ++j) s += __expf(lg[j] - mx);
    for (int j = 0; j < C; ++j) {
        const float hz = 1.0f / (1.0f + __expf(-lg[j]));
        run *= 1.0f - hz;
        hazards[(size_t)b * C + j] = hz;
        survs[(size_t)b * C + j] = run;
        y[(size_t)b * C + j] = __expf(lg[j] - mx) / s;
    }
}
__global__ void head_bwd_kernel(const float* __restrict__ hazards, const float* __restrict__ survs, const float* __restrict__ y,
                                const float* __restrict__ dhz, const float* __restrict__ dsv, const float* __restrict__ dy,
                                float* __restrict__ dlogits, int B, int C) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const size_t o = (size_t)b * C;
    float ydot = 0.f;
    if (dy) for (int j = 0; j < C; ++j) ydot += y[o + j] * dy[o + j];
    float tail = 0.f;                                   // sum_{j >= i} dS_j S_j, built from the back
    float dl[kMaxC];
    for (int i = C - 1; i >= 0; --i) {
        if (dsv) tail += dsv[o + i] * survs[o + i];
        const float hz = hazards[o + i];
        float dh = dhz ? dhz[o + i] : 0.f;
        dh -= tail / fmaxf(1.0f - hz, 1e-30f);
        dl[i] = dh * hz * (1.0f - hz) + (dy ? y[o + i] * (dy[o + i] - ydot) : 0.f);
    }
    for (int i = 0; i < C; ++i) dlogits[o + i] = dl[i];
}

// ------------------------------------------------------------------ 'ces' survival loss (models/loss.py:5-28), one thread per slide
//   reg = -(1-c) (log max(S_pad[y], eps) + log max(h[y], eps)),  S_pad = [1, S]
//   ce  = -(c log s + (1-c) log(1 - s)),  s = max(S[y], eps)
//   loss = (1 - alpha) ce + alpha reg;   risk = -sum_j S_j  (models/mcat/main.py:56)
__global__ void ces_loss_fwd_kernel(const float* __restrict__ hazards, const float* __restrict__ survs,
                                    const long long* __restrict__ label, const float* __restrict__ cens,
                                    float* __restrict__ loss, float* __restrict__ risk, int B, int C, float alpha, float eps) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const size_t o = (size_t)b * C;
    const int y = (int)label[b];
    const float c = cens[b];
    const float s_prev = y == 0 ? 1.0f : survs[o + y - 1];
    const float reg = -(1.0f - c) * (logf(fmaxf(s_prev, eps)) + logf(fmaxf(hazards[o + y], eps)));
    const float sy = fmaxf(survs[o + y], eps);
    const float ce = -(c * logf(sy) + (1.0f - c) * logf(1.0f - sy));
    loss[b] = (1.0f - alpha) * ce + alpha * reg;
    if (risk) {
        float r = 0.f;
        for (int j = 0; j < C; ++j) r -= survs[o + j];
        risk[b] = r;
    }
}
// d_loss: per-slide upstream gradient (B) or, when d_loss_scalar, one value broadcast to every slide
__global__ void ces_loss_bwd_kernel(const float* __restrict__ hazards, const float* __restrict__ survs,
                                    const long long* __restrict__ label, const float* __restrict__ cens,
                                    const float* __restrict__ d_loss, int d_loss_scalar, float* __restrict__ d_hazards,
                                    float* __restrict__ d_survs, int B, int C, float alpha, float eps) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const size_t o = (size_t)b * C;
    const int y = (int)label[b];
    const float c = cens[b];
    const float g = d_loss_scalar ? d_loss[0] : d_loss[b];
    for (int j = 0; j < C; ++j) { d_hazards[o + j] = 0.f; d_survs[o + j] = 0.f; }
    const float hy = hazards[o + y], sy = survs[o + y];
    if (hy >= eps) d_hazards[o + y] = -alpha * (1.0f - c) * g / hy;             // clamp(min=eps) passes gradient where x >= eps
    if (y > 0) {
        const float sp = survs[o + y - 1];
        if (sp >= eps) d_survs[o + y - 1] = -alpha * (1.0f - c) * g / sp;
    }
    if (sy >= eps) d_survs[o + y] += -(1.0f - alpha) * g * (c / sy - (1.0f - c) / (1.0f - sy));
}

// Training step: survival head, 'ces' loss and BOTH their backward passes for a slide, one thread, one launch: the
// loss's upstream gradient w[b] is known before the forward in a training step (1 / grad_acc_step per slide), and four
// dependent launches (head, loss, d loss, d head) are one.  Same arithmetic as the four kernels above.
__global__ void head_loss_kernel(const float* __restrict__ logits, const long long* __restrict__ label,
                                 const float* __restrict__ cens, const float* __restrict__ w,
                                 float* __restrict__ hazards, float* __restrict__ survs, float* __restrict__ y,
                                 float* __restrict__ loss, float* __restrict__ risk, float* __restrict__ dlogits,
                                 int B, int C, float alpha, float eps) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const size_t o = (size_t)b * C;
    float hz[kMaxC], sv[kMaxC];
    float mx = logits[o];
    for (int j = 1; j < C; ++j) mx = fmaxf(mx, logits[o + j]);
    float s = 0.f, run = 1.0f, r = 0.f;
    for (int j = 0; j < C; ++j) s += __expf(logits[o + j] - mx);
    for (int j = 0; j < C; ++j) {
        hz[j] = 1.0f / (1.0f + __expf(-logits[o + j]));
        run *= 1.0f - hz[j];
        sv[j] = run;
        r -= run;
        hazards[o + j] = hz[j];
        survs[o + j] = run;
        y[o + j] = __expf(logits[o + j] - mx) / s;
    }
    const int yb = (int)label[b];
    const float c = cens[b];
    const float s_prev = yb == 0 ? 1.0f : sv[yb - 1];
    const float reg = -(1.0f - c) * (logf(fmaxf(s_prev, eps)) + logf(fmaxf(hz[yb], eps)));
    const float sy = fmaxf(sv[yb], eps);
    const float ce = -(c * logf(sy) + (1.0f - c) * logf(1.0f - sy));
    loss[b] = (1.0f - alpha) * ce + alpha * reg;
    if (risk) risk[b] = r;
    // d loss / d (hazards, survs), then through cumprod and sigmoid (head_bwd_kernel with dy = 0)
    const float g = w[b];
    float tail = 0.f;
    float dl[kMaxC];
    for (int i = C - 1; i >= 0; --i) {
        float dsv = 0.f, dh = 0.f;
        if (i == yb) {
            if (hz[i] >= eps) dh = -alpha * (1.0f - c) * g / hz[i];
            if (sv[i] >= eps) dsv += -(1.0f - alpha) * g * (c / sv[i] - (1.0f - c) / (1.0f - sv[i]));
        }
        if (i == yb - 1 && sv[i] >= eps) dsv += -alpha * (1.0f - c) * g / sv[i];
        tail += dsv * sv[i];
        dh -= tail / fmaxf(1.0f - hz[i], 1e-30f);
        dl[i] = dh * hz[i] * (1.0f - hz[i]);
    }
    for (int i = 0; i < C; ++i) dlogits[o + i] = dl[i];
}
__global__ void counters_bump_kernel(unsigned long long* __restrict__ epoch, int* __restrict__ step) {
    if (threadIdx.x == 0) {
        if (epoch) epoch[0] += 1ull;
        if (step) step[0] += 1;
    }
}

// ------------------------------------------------------------------ element-wise helpers
__global__ void ew_mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] * b[i];
}
__global__ void ew_add_kernel(float* __restrict__ acc, const float* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc[i] += b[i];
}
__global__ void ew_mul2_kernel(const float* __restrict__ x, const float* __restrict__ p, const float* __restrict__ q,
                               float* __restrict__ xp, float* __restrict__ xq, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float v = x[i]; xp[i] = v * p[i]; xq[i] = v * q[i]; }
}
__global__ void ew_add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

// CAG middle (models/blocks.py:248-250), one wave per row:
//   t1 = ELU(u1 + u2), G = LN_G(t1);  t3 = ELU(u3), E = LN_E(t3);  m = G * E
__global__ void cag_mid_fwd_kernel(const float* __restrict__ u1, const float* __restrict__ u2, const float* __restrict__ u3,
                                   const float* __restrict__ gw, const float* __restrict__ gb,
                                   const float* __restrict__ ew, const float* __restrict__ eb,
                                   float* __restrict__ t1, float* __restrict__ t3, float* __restrict__ gout,
                                   float* __restrict__ eout, float* __restrict__ m, float* __restrict__ stats_g /* [R][2] */,
                                   float* __restrict__ stats_e /* [R][2] */, int rows, int d, float eps) {
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const size_t o = (size_t)r * d;
    float s1 = 0.f, s3 = 0.f;
    for (int c = lane; c < d; c += 64) {
        const float a = elu_f(u1[o + c] + u2[o + c]), b = elu_f(u3[o + c]);
        t1[o + c] = a;
        t3[o + c] = b;
        s1 += a;
        s3 += b;
    }
    const float m1 = wave_sum(s1) / d, m3 = wave_sum(s3) / d;
    float v1 = 0.f, v3 = 0.f;
    for (int c = lane; c < d; c += 64) {
        const float a = t1[o + c] - m1, b = t3[o + c] - m3;
        v1 += a * a;
        v3 += b * b;
    }
    const float r1 = rsqrtf(wave_sum(v1) / d + eps), r3 = rsqrtf(wave_sum(v3) / d + eps);
    for (int c = lane; c < d; c += 64) {
        const float g = (t1[o + c] - m1) * r1 * gw[c] + gb[c];
        const float e = (t3[o + c] - m3) * r3 * ew[c] + eb[c];
        gout[o + c] = g;
        eout[o + c] = e;
        m[o + c] = g * e;
    }
    if (lane == 0) { stats_g[2 * r] = m1; stats_g[2 * r + 1] = r1; stats_e[2 * r] = m3; stats_e[2 * r + 1] = r3; }
}

// dm -> dG = dm*E, dE = dm*G (written for the LN parameter gradients) and, through both LayerNorms and the
// outer ELUs, ds12 = d(u1 + u2) and ds3 = d(ELU(u3)).
__global__ void cag_mid_bwd_kernel(const float* __restrict__ dm, const float* __restrict__ t1, const float* __restrict__ t3,
                                   const float* __restrict__ gout, const float* __restrict__ eout,
                                   const float* __restrict__ gw, const float* __restrict__ ew, const float* __restrict__ stats_g,
                                   const float* __restrict__ stats_e, float* __restrict__ dG, float* __restrict__ dE, float* __restrict__ ds12, float* __restrict__ ds3,
                                   int rows, int d) {
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const size_t o = (size_t)r * d;
    const float m1 = stats_g[2 * r], r1 = stats_g[2 * r + 1], m3 = stats_e[2 * r], r3 = stats_e[2 * r + 1];
    float a1 = 0.f, a2 = 0.f, b1 = 0.f, b2 = 0.f;
    for (int c = lane; c < d; c += 64) {
        const float dg = dm[o + c] * eout[o + c], de = dm[o + c] * gout[o + c];
        dG[o + c] = dg;
        dE[o + c] = de;
        const float g1 = dg * gw[c], x1 = (t1[o + c] - m1) * r1;
        const float g3 = de * ew[c], x3 = (t3[o + c] - m3) * r3;
        a1 += g1; a2 += g1 * x1; b1 += g3; b2 += g3 * x3;
    }
    a1 = wave_sum(a1) / d; a2 = wave_sum(a2) / d; b1 = wave_sum(b1) / d; b2 = wave_sum(b2) / d;
    for (int c = lane; c < d; c += 64) {
        const float g1 = dG[o + c] * gw[c], x1 = (t1[o + c] - m1) * r1;
        const float g3 = dE[o + c] * ew[c], x3 = (t3[o + c] - m3) * r3;
        ds12[o + c] = r1 * (g1 - a1 - x1 * a2) * elu_grad_from_out(t1[o + c]);
        ds3[o + c] = r3 * (g3 - b1 - x3 * b2) * elu_grad_from_out(t3[o + c]);
    }
}

}  // namespace

// ---------------------------------------------------------------------------- launchers
static bool ln_al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
// the fixed-width LayerNorm kernels read the affine parameters as float4
static bool ln_vec_ok(const LnBranches& p, bool) {
    for (int i = 0; i < p.n; ++i)
        if (!ln_al16(p.w[i]) || !ln_al16(p.b[i])) return false;
    return true;
}
int mpo_launch_ln_fwd_br(const float* x, const LnBranches& p, float* y, float* stats, int rows, int d, float eps, hipStream_t s) {
    if (rows <= 0) return 0;
    MPO_CHECK(p.n >= 1 && p.n <= kMaxBranches && p.rows_per_branch * p.n == rows, "layer norm: %d rows over %d branches of %d",
              rows, p.n, p.rows_per_branch);
    const bool vec = ln_vec_ok(p, false) && ln_al16(x) && ln_al16(y);
    if (vec && d == 256) ln_fwd_kernel<1><<<(rows + 3) / 4, 256, 0, s>>>(x, p, y, stats, rows, d, eps);
    else if (vec && d == 512) ln_fwd_kernel<2><<<(rows + 3) / 4, 256, 0, s>>>(x, p, y, stats, rows, d, eps);
    else ln_fwd_kernel<0><<<(rows + 3) / 4, 256, 0, s>>>(x, p, y, stats, rows, d, eps);
    MPO_LAUNCH_CHECK();
    return 0;
}
int mpo_launch_ln_bwd_br(const float* dy, const float* x, const float* stats, const LnBranches& p, float* dx, int rows, int d,
                         int accumulate, int what, hipStream_t s) {
    if (rows <= 0 || !(what & 3)) return 0;
    MPO_CHECK(p.n >= 1 && p.n <= kMaxBranches && p.rows_per_branch * p.n == rows, "layer norm: %d rows over %d branches of %d",
              rows, p.n, p.rows_per_branch);
    // long row axes: the parameter sums are cut into row chunks of ~512 and added atomically into zeroed dw / db
    const int n_chunks = (what & 2) && p.rows_per_branch >= 2048 ? (p.rows_per_branch + 511) / 512 : 1;
    if (n_chunks > 1) {
        for (int br = 0; br < p.n; ++br) {
            MPO_HIP(hipMemsetAsync(p.dw[br], 0, (size_t)d * sizeof(float), s));
            MPO_HIP(hipMemsetAsync(p.db[br], 0, (size_t)d * sizeof(float), s));
        }
    }
    const int blocks = ((what & 1) ? (rows + 3) / 4 : 0) + ((what & 2) ? n_chunks * p.n * ((d + 15) / 16) : 0);
    const bool vec = ln_vec_ok(p, false) && ln_al16(x) && ln_al16(dy) && ln_al16(dx);
    if (vec && d == 256) ln_bwd_kernel<1><<<blocks, 256, 0, s>>>(dy, x, stats, p, dx, rows, d, accumulate, what, n_chunks);
    else if (vec && d == 512) ln_bwd_kernel<2><<<blocks, 256, 0, s>>>(dy, x, stats, p, dx, rows, d, accumulate, what, n_chunks);
    else ln_bwd_kernel<0><<<blocks, 256, 0, s>>>(dy, x, stats, p, dx, rows, d, accumulate, what, n_chunks);
    MPO_LAUNCH_CHECK();
    return 0;
}
static LnBranches one_branch(const float* w, const float* b, float* dw, float* db, int rows) {
    LnBranches p;
    p.n = 1; p.rows_per_branch = rows; p.w[0] = w; p.b[0] = b; p.dw[0] = dw; p.db[0] = db;
    return p;
}
int mpo_launch_ln_fwd(const float* x, const float* w, const float* b, float* y, float* stats, int rows, int d, float eps,
                      hipStream_t s) {
    return mpo_launch_ln_fwd_br(x, one_branch(w, b, nullptr, nullptr, rows), y, stats, rows, d, eps, s);
}
int mpo_launch_ln_bwd(const float* dy, const float* x, const float* stats, const float* w, float* dx, float* dw, float* db,
                      int rows, int d, int accumulate, hipStream_t s) {
    return mpo_launch_ln_bwd_br(dy, x, stats, one_branch(w, nullptr, dw, db, rows), dx, rows, d, accumulate, dw ? 3 : 1, s);
}
int mpo_launch_ln_bwd_params_only(const float* dy, const float* x, const float* stats, float* dw, float* db, int rows, int d,
                                  hipStream_t s) {
    return mpo_launch_ln_bwd_br(dy, x, stats, one_branch(nullptr, nullptr, dw, db, rows), nullptr, rows, d, 0, 2, s);
}
int mpo_launch_mha_small_fwd(const float* qkv, float* o, float* p_save, int B, int T, int d, int H, float drop_p,
                             unsigned long long seed, unsigned long long offset, const unsigned long long* epoch,
                             hipStream_t s) {
    MPO_CHECK(T >= 1 && T <= kMaxT && d % H == 0, "set-transformer attention: T=%d (max %d), d=%d, heads=%d", T, kMaxT, d, H);
    const int hd = d / H;
    const size_t lds = 4 * ((size_t)3 * T * hd + 2 * T * T + 2 * T) * sizeof(float);
    MPO_CHECK(lds <= 160 * 1024, "set-transformer attention: T=%d, head dim %d needs %zu bytes of LDS", T, hd, lds);
    if (T == 6 && hd == 32) mha_small_fwd_kernel<6, 32><<<(B * H + 3) / 4, 256, lds, s>>>(qkv, o, p_save, B, T, d, H, drop_p, seed, offset, epoch);
    else mha_small_fwd_kernel<0, 0><<<(B * H + 3) / 4, 256, lds, s>>>(qkv, o, p_save, B, T, d, H, drop_p, seed, offset, epoch);
    MPO_LAUNCH_CHECK();
    return 0;
}
int mpo_launch_mha_small_bwd(const float* qkv, const float* p_save, const float* d_o, float* dqkv, int B, int T, int d, int H,
                             hipStream_t s) {
    MPO_CHECK(T >= 1 && T <= kMaxT && d % H == 0, "set-transformer attention: T=%d (max %d), d=%d, heads=%d", T, kMaxT, d, H);
    const int hd = d / H;
    const size_t lds = 4 * ((size_t)4 * T * hd + 4 * T * T) * sizeof(float);
    MPO_CHECK(lds <= 160 * 1024, "set-transformer attention backward: T=%d, head dim %d needs %zu bytes of LDS", T, hd, lds);
    if (T == 6 && hd == 32) mha_small_bwd_kernel<6, 32><<<(B * H + 3) / 4, 256, lds, s>>>(qkv, p_save, d_o, dqkv, B, T, d, H);
    else mha_small_bwd_kernel<0, 0><<<(B * H + 3) / 4, 256, lds, s>>>(qkv, p_save, d_o, dqkv, B, T, d, H);
    MPO_LAUNCH_CHECK();
    return 0;
}
int mpo_launch_pool_fwd(const float* scores, const float* x, float* w, float* h, int B, int L, int d, hipStream_t s) {
    if (L > kPoolLongL) {
        pool_long_softmax_kernel<<<B, 1024, 0, s>>>(scores, w, L);
        MPO_LAUNCH_CHECK();
        pool_long_wsum_kernel<<<dim3((d + 15) / 16, B), 256, 0, s>>>(w, x, h, L, d);
    } else {
        pool_fwd_kernel<<<B, 256, 0, s>>>(scores, x, w, h, L, d);
    }
    MPO_LAUNCH_CHECK();
    return 0;
}
int mpo_launch_pool_score_fwd(const float* a, const float* b, const float* x, const PoolScorer& ps, float* scores, float* w,
                              float* h, int B, int L, int d, hipStream_t s) {
    MPO_CHECK(L >= 1 && L <= 64 && B % ps.n_slides == 0, "fused pooling scorer: 1..64 rows per slide (got %d)", L);
    pool_score_fwd_kernel<<<B, 256, 0, s>>>(a, b, x, ps, scores, w, h, L, d);
    MPO_LAUNCH_CHECK();
    return 0;
}
int mpo_launch_pool_score_bwd(const float* dh, const float* x, const float* w, const float* d_ext, const float* a, const float* b,
                              const PoolScorer& ps, float* d_scores, float* dx, float* da, float* db, int B, int L, int d,
                              hipStream_t s) {
    MPO_CHECK(L >= 1 && L <= 64 && B % ps.n_slides == 0, "fused pooling scorer backward: 1..64 rows per slide (got %d)", L);
    pool_score_bwd_kernel<<<B, 256, 0, s>>>(dh, x, w, d_ext, a, b, ps, d_scores, dx, da, db, L, d);
    MPO_LAUNCH_CHECK();
    return 0;
}
int mpo_launch_pool_bwd(const float* dh, const float* x, const float* w, const float* d_ext, float* d_scores, float* dx,
                        int B, int L, int d, hipStream_t s) {
    if (L > kPoolLongL) {
        MPO_CHECK(B <= 65535, "pooling over a long axis: %d slides exceed the grid", B);
        pool_long_rows_kernel<<<dim3((L + 3) / 4, B), 256, 0, s>>>(dh, x, w, d_scores, dx, L, d);
        MPO_LAUNCH_CHECK();
        pool_long_dscore_kernel<<<B, 1024, 0, s>>>(w, d_ext, d_scores, L);
    } else {
        pool_bwd_kernel<<<B, 256, 0, s>>>(dh, x, w, d_ext, d_scores, dx, L, d);
    }
    MPO_LAUNCH_CHECK();
    return 0;
}
int mpo_launch_head_fwd(const float* logits, float* hazards, float* survs, float* y, int B, int C, hipStream_t s) {
    MPO_CHECK(C >= 1 && C <= kMaxC, "survival head: n_classes %d not in 1..%d", C, kMaxC);
    head_fwd_kernel<<<(B + 63) / 64, 64, 0, s>>>(logits, hazards, survs, y, B, C);
    MPO_LAUNCH_CHECK();
    return 0;
}
int mpo_launch_head_bwd(const float* hazards, const float* survs, const float* y, const float* dhz, const float* dsv,
                        const float* dy, float* dlogits, int B, int C, hipStream_t s) {
    MPO_CHECK(C >= 1 && C <= kMaxC, "survival head: n_classes %d not in 1..%d", C, kMaxC);
    head_bwd_kernel<<<(B + 63) / 64, 64, 0, s>>>(hazards, survs, y, dhz, dsv, dy, dlogits, B, C);
    MPO_LAUNCH_CHECK();
    return 0;
}
int mpo_launch_ces_loss_fwd(const float* hazards, const float* survs, const long long* label, const float* cens, float* loss,
                            float* risk, int B, int C, float alpha, float eps, hipStream_t s) {
    MPO_CHECK(B >= 1 && C >= 1, "ces loss: empty batch (%d x %d)", B, C);
    ces_loss_fwd_kernel<<<(B + 63) / 64, 64, 0, s>>>(hazards, survs, label, cens, loss, risk, B, C, alpha, eps);
    MPO_LAUNCH_CHECK();
    return 0;
}
int mpo_launch_ces_loss_bwd(const float* hazards, const float* survs, const long long* label, const float* cens,
                            const float* d_loss, int d_loss_scalar, float* d_hazards, float* d_survs, int B, int C,
                            float alpha, float eps, hipStream_t s) {
    MPO_CHECK(B >= 1 && C >= 1, "ces loss: empty batch (%d x %d)", B, C);
    ces_loss_bwd_kernel<<<(B + 63) / 64, 64, 0, s>>>(hazards, survs, label, cens, d_loss, d_loss_scalar, d_hazards, d_survs,
                                                     B, C, alpha, eps);
    MPO_LAUNCH_CHECK();
    return 0;
}
int mpo_launch_head_loss(const float* logits, const long long* label, const float* cens, const float* w, float* hazards,
                         float* survs, float* y, float* loss, float* risk, float* dlogits, int B, int C, float alpha,
                         float eps, hipStream_t s) {
    MPO_CHECK(B >= 1 && C >= 1 && C <= kMaxC, "head + ces loss: %d slides x %d classes (classes in 1..%d)", B, C, kMaxC);
    head_loss_kernel<<<(B + 63) / 64, 64, 0, s>>>(logits, label, cens, w, hazards, survs, y, loss, risk, dlogits, B, C, alpha, eps);
    MPO_LAUNCH_CHECK();
    return 0;
}
int mpo_launch_counters_bump(unsigned long long* epoch, int* step, hipStream_t s) {
    counters_bump_kernel<<<1, 64, 0, s>>>(epoch, step);
    MPO_LAUNCH_CHECK();
    return 0;
}
int mpo_launch_ew_add(float* acc, const float* b, size_t n, hipStream_t s) {
    if (n == 0) return 0;
    const size_t blocks = (n + 255) / 256;
    ew_add_kernel<<<(unsigned)(blocks < 65536 ? blocks : 65536), 256, 0, s>>>(acc, b, n);
    MPO_LAUNCH_CHECK();
    return 0;
}
int mpo_launch_ew_mul(const float* a, const float* b, float* out, int n, hipStream_t s) {
    ew_mul_kernel<<<(n + 255) / 256, 256, 0, s>>>(a, b, out, n);
    MPO_LAUNCH_CHECK();
    return 0;
}
int mpo_launch_ew_mul2(const float* x, const float* p, const float* q, float* xp, float* xq, int n, hipStream_t s) {
    ew_mul2_kernel<<<(n + 255) / 256, 256, 0, s>>>(x, p, q, xp, xq, n);
    MPO_LAUNCH_CHECK();
    return 0;
}
int mpo_launch_ew_add(const float* a, const float* b, float* out, int n, hipStream_t s) {
    ew_add_kernel<<<(n + 255) / 256, 256, 0, s>>>(a, b, out, n);
    MPO_LAUNCH_CHECK();
    return 0;
}
int mpo_launch_cag_mid_fwd(const float* u1, const float* u2, const float* u3, const float* gw, const float* gb, const float* ew,
                           const float* eb, float* t1, float* t3, float* gout, float* eout, float* m, float* stats_g, float* stats_e,
                           int rows, int d, float eps, hipStream_t s) {
    cag_mid_fwd_kernel<<<(rows + 3) / 4, 256, 0, s>>>(u1, u2, u3, gw, gb, ew, eb, t1, t3, gout, eout, m, stats_g, stats_e, rows, d, eps);
    MPO_LAUNCH_CHECK();
    return 0;
}
int mpo_launch_cag_mid_bwd(const float* dm, const float* t1, const float* t3, const float* gout, const float* eout,
                           const float* gw, const float* ew, const float* stats_g, const float* stats_e, float* dG, float* dE,
                           float* ds12, float* ds3, int rows, int d, hipStream_t s) {
    cag_mid_bwd_kernel<<<(rows + 3) / 4, 256, 0, s>>>(dm, t1, t3, gout, eout, gw, ew, stats_g, stats_e, dG, dE, ds12, ds3, rows, d);
    MPO_LAUNCH_CHECK();
    return 0;
}
