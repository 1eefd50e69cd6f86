// The patch layer for an fp32-stored window (BASELINE cfg 5: 100 000-patch bags in fp32), hand-written forward and weight
// gradient -- replaces the two fp32 library GEMMs that were 8.3 of cfg 5's 10.9 ms (r02).
//
//     H_bag = Dropout(ReLU(X W_H^T + b_H))          X [rows, 1024] fp32, W_H [256, 1024], H_bag [rows, 256] fp32
//     dW_H  = g^T X,  db_H = colsum(g),  g = dH (.) [H_bag > 0] / (1 - p)          (models/mcat/mcat.py:24-29,87 and its backward)
//
// Arithmetic: THREE-TERM half-precision products with fp32 accumulation.  Both operands are split x = hi + lo and each
// product runs as hi*hi + lo*hi + hi*lo on the 16x16x32 MFMA (48 matrix-pipe cycles per 32 of K where eight
// v_mfma_f32_16x16x4_f32 take 256; the fp32-input MFMA peaks at 157 TF/s: 2.7 ms for this layer's 419 GF at 8 x 100 000
// rows, forward and again for dW_H).
//   * FORWARD: fp16 splits (11 + 11 significand bits: ~2^-22 per element product, i.e. the fp32 GEMM's own rounding).  The
//     forward decides the ReLU mask: with bf16 splits (2^-17 per product, ~4e-6 absolute on H_bag) a pre-activation within
//     4e-6 of zero flips its mask against the fp32 reference 40 times as often as an fp32 GEMM's does, and ONE flipped
//     element moves a row of dW_H by |dH||x| (measured r03: 1.6-4.6 % of max|dW_H| on 164-row windows, 5.9e-3 at 3 000 rows).
//     fp16's narrow exponent is handled by a power-of-two scale on the weight (W x 2^10 before the split, 2^-10 on the
//     accumulator: exact), and operands clamped to the fp16 range before conversion, so nothing overflows to infinity;
//     features must stay below 1.3e5 in magnitude (hi + lo saturate there), tiny features / weights degrade gracefully to an
//     absolute 6e-8 (fp16 subnormals).
//   * WEIGHT GRADIENT: bf16 splits (2^-17 per product, fp32's exponent range: gradients span many orders of magnitude and
//     no mask depends on them): ~1e-5 relative on dW_H.
// The split of X (and of g) happens on the fly between the global load and the LDS image; W_H is split once per call by a pack
// kernel into MFMA-fragment order (as the bf16 kernel's weight, patch_coattn_fwd.hip).
//
// Forward, one workgroup of 4 waves per CU walking blocks of 128 rows: per K-step of 32 the X tile (128 x 32 fp32 = 16 KiB)
// comes global -> registers one step ahead, is split and written to a double-buffered pair of bf16 images (hi, lo; 64-byte
// rows, chunk c of row r at c ^ ((r >> 2) & 3): conflict-free ds_read_b128 fragments), one barrier per step; wave w owns
// embed columns 64 w .. + 63 of all 128 rows (4 x 8 tiles, 128 accumulators, product taken TRANSPOSED so that a lane ends
// with four consecutive embed columns of one patch row = one 16-byte store) and reads its W fragments straight from L2 one
// step ahead.  Epilogue: bias, ReLU, dropout from the counter hash (8 bits per element, realised p = round(256 p) / 256; the
// mask lives in H_bag as zeros, the backward reads it back from there), fp32 store.
//
// Weight gradient, 256 workgroups of 8 waves = 64 row ranges x 4 column blocks of X (256 columns each; the four blocks of a
// range sit on one XCD -- workgroup ids 8 apart --: dH and H_bag are fetched from HBM once and shared through L2, as in
// patch_wgrad.hip): a 256 x 256 fp32 block of dW in registers (wave: 64 x 128, 128 accumulators), 32-row chunks: dH, H_bag
// and X global -> registers one chunk ahead, gate + split, written TRANSPOSED-readable (row-major bf16 images read with
// ds_read_b64_tr_b16), three MFMA terms; fp32 partials per row range + the reduction launch of patch_wgrad.hip.  The column
// sums of g (= db_H) fall out of the column-block-0 workgroups.
//
// Roofline: forward reads 4096 B and writes 1024 B per row: HBM-bound at 5 120 B / row (8 x 100 000 rows: 4.1 GB per launch);
// its 3 x 524 288 flop per row (1.26 PF per launch) are the co-limit at the chip's MFMA clock under load.
#include "coattn_tile.h"
#include "mpo_kernels.h"

namespace {

// waves of the forward's workgroup: 4 (one per SIMD) or 8 (two per SIMD, -DMPO_F32_FWD_WAVES=8 through tools/build_variant.py).
// Measured the same on the same box (1.27 against 1.24-1.30 ms per 8 x 100 000 window, r03): the kernel is not limited by a
// wave waiting on itself but by the matrix pipe at the clock the chip holds under this load.
#ifndef MPO_F32_FWD_WAVES
#define MPO_F32_FWD_WAVES 4
#endif
constexpr int FE = 256;                         // embed_dim
constexpr int FK = 1024;                        // patch feature width
constexpr int FBM = 128;                        // rows per block
constexpr int FBK = 32;                         // k per step
constexpr int FSTEPS = FK / FBK;                // 32
constexpr int FIMG = FBM * FBK * 2;             // one bf16 image of a stage: 8 KiB
constexpr int FROWB = FBK * 2;                  // 64 bytes per image row

__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    pack_hi_lo(v, hi, lo);
}
// fp16 hi / lo split of eight floats, round-to-nearest both (hi + lo carries 22 significand bits, the dropped lo * lo term
// is 2^-22 of the product).  Range: both operands arrive scaled by a power of two that puts their largest magnitude at
// 2^14 .. 2^15 (W_H: from its own maximum, found on the device per call; X: x_scale, which the caller derives from the window's
// maximum -- ops.patch_fc_f32 caches it per tensor), so finite data never reach the fp16 limit and small features keep
// their bits (unscaled, features of ~1e-4 sat in fp16 subnormals: ~3e-4 relative).  The clamp only matters for a caller
// that passes no scale (x_scale = 1: |x| up to 1.3e5 is carried by hi + lo, beyond that clipped).  NaN and infinities are
// NOT clamped away: v - v is 0 for finite v and NaN otherwise, so a non-finite feature makes its row of H_bag non-finite,
// as it does in the reference's fp32 GEMM (v_med3 alone returns the bound for a NaN).
__device__ __forceinline__ int fswz(int row) { return ((row >> 1) & 1) ^ ((row >> 2) & 2); }
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split8h(const f32x4& a, const f32x4& b, float scale, f16x8& hi, f16x8& lo) {
    const float v[8] = {a[0] * scale, a[1] * scale, a[2] * scale, a[3] * scale, b[0] * scale, b[1] * scale, b[2] * scale, b[3] * scale};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const _Float16 h = (_Float16)(__builtin_amdgcn_fmed3f(v[j], -65504.0f, 65504.0f) + (v[j] - v[j]));
        hi[j] = h;
        lo[j] = (_Float16)__builtin_amdgcn_fmed3f(v[j] - (float)h, -65504.0f, 65504.0f);
    }
}
__device__ __forceinline__ f32x4 mfma_f16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
// The power of two that puts max |v| into [2^14, 2^15) (1 for an all-zero or non-finite maximum).
__device__ __forceinline__ float range_scale(float vmax) {
    if (!(vmax > 0.f) || !(vmax < INFINITY)) return 1.0f;
    int e;
    (void)frexpf(vmax, &e);                     // vmax = m 2^e, m in [0.5, 1)
    return ldexpf(1.0f, min(max(15 - e, -100), 100));
}
// scale of W_H for this call -> *scale_out (one workgroup of 1024 threads over the 256 x 1024 weight)
__global__ void weight_range_kernel(const float* __restrict__ w, float* __restrict__ scale_out) {
    __shared__ float red[16];
    float m = 0.f;
    for (int i = threadIdx.x; i < 256 * 1024 / 4; i += 1024) {
        const f32x4 v = reinterpret_cast<const f32x4*>(w)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
        *scale_out = range_scale(m);
    }
}

// W_H [256][1024] fp32 (x *w_scale, weight_range_kernel) -> hi / lo fp16 in fragment order: block (((term * 4 + wave) * 32 + step) * 4 + ct) of 1 KiB holds, for lane
// (i = lane & 15, g = lane >> 4), W_H[64 wave + 16 ct + i][32 step + 8 g .. + 7]
__global__ void pack_patch_weight_f32_kernel(const float* __restrict__ w, f16x8* __restrict__ out, const float* __restrict__ w_scale) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;          // one fragment per thread: 256 * 1024 / 8
    if (t >= FE * FK / 8) return;
    const int lane = t & 63, blk = t >> 6;
    const int ct = blk & 3, step = (blk >> 2) & (FSTEPS - 1), wave = blk >> 7;
    const int row = 64 * wave + 16 * ct + (lane & 15), k0 = FBK * step + 8 * (lane >> 4);
    const f32x4 a = *reinterpret_cast<const f32x4*>(w + (size_t)row * FK + k0);
    const f32x4 b = *reinterpret_cast<const f32x4*>(w + (size_t)row * FK + k0 + 4);
    f16x8 hi, lo;
    split8h(a, b, *w_scale, hi, lo);
    out[t] = hi;
    out[FE * FK / 8 + t] = lo;
}

// NW waves per workgroup (4: one per SIMD, 64 embed columns each; 8: two per SIMD, 32 each -- the same packed weight, the same
// dropout mask, the same arithmetic per element)
template <int NW>
__global__ __launch_bounds__(64 * NW, 1)
void patch_fc_f32_kernel(const float* __restrict__ x,             // [total_rows][1024]
                         const f16x8* __restrict__ wpk,           // packed hi | lo (pack_patch_weight_f32_kernel)
                         const float* __restrict__ bias,          // [256]
                         float* __restrict__ h,                   // [total_rows][256]
                         long long total_rows, int rows_per_wg, float drop_p, unsigned long long seed,
                         unsigned long long offset_, const unsigned long long* __restrict__ epoch,
                         float x_scale, const float* __restrict__ w_scale) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * 2 * FIMG];         // [stage][hi | lo]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    const long long wg_r0 = (long long)blockIdx.x * rows_per_wg;
    const long long wg_r1 = min(total_rows, wg_r0 + rows_per_wg);
    if (wg_r0 >= wg_r1) return;
    const int nblocks = (int)((wg_r1 - wg_r0 + FBM - 1) / FBM);

    constexpr int NCT = 16 / NW;                                         // 16-column tiles a wave owns
    constexpr int TPR = NW / 2;                                          // threads per row of the X tile
    constexpr int NF4 = 8 / TPR;                                         // float4 per thread and step (k = 4 NF4 spart .. + 4 NF4 - 1)
    constexpr int NCHK = NF4 / 2;                                        // 16-byte image chunks per thread
    // staging: thread (row = tid / TPR, part = tid % TPR) carries 4 NF4 consecutive k of its row
    const int srow = tid / TPR, spart = tid % TPR;
    // Image rows are 64 bytes; chunk c of row r sits at position c ^ fswz(r).  ds_read_b128 is serviced in four NON-contiguous
    // 16-lane groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32: MI355X_MICROARCH.md, LDS): with lane = (row j, chunk g) a
    // group holds rows j, j + 12 at chunk g and rows j + 4, j + 8 at chunk g ^ 1 of every j mod 4 -- four reads inside one
    // 64-byte quarter of the bank line, which bit 3 of the row (x 2) spreads over the four positions; bit 1 of the row keeps the
    // 8-lane groups of the ds_write_b128 that fills the image (4 rows x 2 threads) off each other's banks.  (r03's
    // (row >> 2) & 3 assumed contiguous read groups: every fragment read was 2-way, SQ_LDS_BANK_CONFLICT = 50 % of LDS cycles.)
    const int sswz = fswz(srow);
    int soff[NCHK];
#pragma unroll
    for (int c = 0; c < NCHK; ++c) soff[c] = srow * FROWB + (((NCHK * spart + c) ^ sswz) << 4);
    const int wave4 = wave / (NW / 4), ct0 = NCT * (wave % (NW / 4));   // position in the packed weight's [wave 4][..][ct 4] order
    // fragment reads: row 16 rt + c16, chunk g
    const int foff = c16 * FROWB + ((g ^ fswz(c16)) << 4);
    const f16x8* whi = wpk + (size_t)wave4 * (FSTEPS * 4 * 64) + ct0 * 64 + lane;
    const f16x8* wlo = whi + FE * FK / 8;
    const uint32_t thr8 = (uint32_t)(drop_p * 256.0f + 0.5f);
    const float inv_keep = drop_p > 0.f ? 256.0f / (256.0f - (float)thr8) : 1.0f;
    const uint32_t drop_key = hash_stream_key(seed, epoch_offset(offset_, epoch));
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const float unscale = 1.0f / (x_scale * *w_scale);            // (powers of two: exact)

    for (int blk = 0; blk < nblocks; ++blk) {
        const long long rb = wg_r0 + (long long)blk * FBM;
        const long long grow = min(rb + srow, total_rows - 1);           // rows past the end: clamped (finite; never stored)
        const float* xrow = x + (size_t)grow * FK + 4 * NF4 * spart;
        f32x4 acc[NCT][8];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int rt = 0; rt < 8; ++rt) acc[ct][rt] = zero4;
        // Operand pipeline, three register sets each in static rotation (three steps per trip, no copies): step k multiplies the
        // images of x(k) (LDS stage k & 1) by weight set k % 3, requests x(k + 2) and the weight fragments of step k + 2, and
        // splits x(k + 1) -- requested in the previous step -- into the other stage AFTER the step's MFMAs are issued (four sets
        // each, three steps ahead, spilled 98 registers).  One step of matrix work (~0.8 us) does not
        // cover an HBM or a loaded-L2 round trip: with one step of lookahead every step began by waiting for its operands.
        f32x4 xs[3][NF4];
        f16x8 wh[3][NCT], wl[3][NCT];
        auto load_x = [&](f32x4 (&dst)[NF4], int k) {
            k = k < FSTEPS ? k : FSTEPS - 1;                              // (past the end: re-reads the last tile, never used)
#pragma unroll
            for (int j = 0; j < NF4; ++j) dst[j] = *reinterpret_cast<const f32x4*>(xrow + FBK * k + 4 * j);
        };
        auto load_w = [&](f16x8 (&h_)[NCT], f16x8 (&l_)[NCT], int k) {
            k = k < FSTEPS ? k : FSTEPS - 1;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                h_[ct] = whi[(k * 4 + ct) * 64];
                l_[ct] = wlo[(k * 4 + ct) * 64];
            }
        };
        auto split_tile = [&](const f32x4 (&src)[NF4], char* st) {
#pragma unroll
            for (int c = 0; c < NCHK; ++c) {
                f16x8 h0, l0;
                split8h(src[2 * c], src[2 * c + 1], x_scale, h0, l0);
                *reinterpret_cast<f16x8*>(st + soff[c]) = h0;
                *reinterpret_cast<f16x8*>(st + FIMG + soff[c]) = l0;
            }
        };
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            load_x(xs[j], j);
            load_w(wh[j], wl[j], j);
        }
        split_tile(xs[0], lds);
        // one K-step with the register sets named statically: j = k % 3
        auto step = [&](int k, const f16x8 (&wh_)[NCT], const f16x8 (&wl_)[NCT], f16x8 (&whn)[NCT], f16x8 (&wln)[NCT], f32x4 (&xn)[NF4],
                        const f32x4 (&xsplit)[NF4]) {
            const char* st = lds + (k & 1) * (2 * FIMG);
            __syncthreads();                                              // the images of x(k) are complete; the other stage is free
            load_x(xn, k + 2);
            load_w(whn, wln, k + 2);
            f16x8 xh[8], xl[8];
#pragma unroll
            for (int rt = 0; rt < 8; ++rt) {
                xh[rt] = *reinterpret_cast<const f16x8*>(st + rt * 16 * FROWB + foff);
                xl[rt] = *reinterpret_cast<const f16x8*>(st + FIMG + rt * 16 * FROWB + foff);
            }
#pragma unroll
            for (int rt = 0; rt < 8; ++rt) {
                // term-major: the three MFMAs of one accumulator are DEPENDENT (each waits for the previous result); with the
                // four column tiles in between every MFMA finds its accumulator ready
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) acc[ct][rt] = mfma_f16(wh_[ct], xh[rt], acc[ct][rt]);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) acc[ct][rt] = mfma_f16(wh_[ct], xl[rt], acc[ct][rt]);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) acc[ct][rt] = mfma_f16(wl_[ct], xh[rt], acc[ct][rt]);
            }
            // (the scheduler otherwise sinks every read to its first use: reads first, four fragments ahead of the MFMAs)
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * NCT, 0);  // one row tile's MFMAs
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);        // the fragments of row tile j + 2
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 6 * NCT, 0);
            if (k + 1 < FSTEPS) split_tile(xsplit, lds + ((k + 1) & 1) * (2 * FIMG));       // (vector work behind the queued matrix work)
        };
#pragma unroll 1
        for (int k = 0; k < FSTEPS - 2; k += 3) {                        // 30 steps, then the last two
            step(k, wh[0], wl[0], wh[2], wl[2], xs[2], xs[1]);
            step(k + 1, wh[1], wl[1], wh[0], wl[0], xs[0], xs[2]);
            step(k + 2, wh[2], wl[2], wh[1], wl[1], xs[1], xs[0]);
        }
        step(FSTEPS - 2, wh[0], wl[0], wh[2], wl[2], xs[2], xs[1]);
        step(FSTEPS - 1, wh[1], wl[1], wh[0], wl[0], xs[0], xs[2]);
        // epilogue: acc[ct][rt][r] = H^T: embed column 64 wave + 16 ct + 4 g + r of patch row 16 rt + c16
        f32x4 bv[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) bv[ct] = *reinterpret_cast<const f32x4*>(bias + 64 * wave4 + 16 * (ct0 + ct) + 4 * g);
#pragma unroll
        for (int rt = 0; rt < 8; ++rt) {
            const long long row = rb + 16 * rt + c16;
            uint4 rnd = {0u, 0u, 0u, 0u};
            if (drop_p > 0.f) rnd = hash4x32(drop_key, (unsigned long long)row * 16ull + (unsigned)(4 * wave4 + g));
            const uint32_t rw[4] = {rnd.x, rnd.y, rnd.z, rnd.w};
            uint32_t rwc[NCT];                                           // words ct0 .. ct0 + NCT - 1 (ct0 is wave-uniform: selects, no indexing)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) rwc[ct] = NCT == 4 ? rw[ct] : (ct0 == 0 ? rw[ct] : rw[(2 + ct) & 3]);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = __builtin_elementwise_maximum(acc[ct][rt][r] * unscale + bv[ct][r], 0.f);      // (v_maximum3_f32: a NaN stays a NaN, as in torch.relu)
                    if (drop_p > 0.f) v = (((rwc[ct] >> (8 * r)) & 0xFFu) >= thr8) ? v * inv_keep : 0.f;
                    o[r] = v;
                }
                if (row < wg_r1) *reinterpret_cast<f32x4*>(h + (size_t)row * FE + 64 * wave4 + 16 * (ct0 + ct) + 4 * g) = o;
            }
        }
        __syncthreads();                                                  // (the next block's first split writes stage 0)
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient
constexpr int GW = 8;                           // waves
constexpr int GCH = 32;                         // rows per chunk
constexpr int GRANGES = 64, GCOLB = 4;          // row ranges x column blocks of X = 256 workgroups
constexpr int GIMG = GCH * 256 * 2;             // one bf16 image [32 rows][256 columns]: 16 KiB
constexpr int GROWB = 512;

// [32][256] bf16 image, 16-byte chunk c of row r at c ^ (2 (r & 7)) (coattn_tile.h): col_frag<256>() reads it transposed
__device__ __forceinline__ int gimg_off(int r, int c) { return r * GROWB + ((c ^ ((r & 7) << 1)) << 4); }

__global__ __launch_bounds__(GW * 64, 1)
void patch_wgrad_f32_kernel(const float* __restrict__ dh,         // [rows][256] gradient arriving at H_bag
                            const float* __restrict__ hbag,       // [rows][256] H_bag itself (zeros = ReLU / dropout mask); NULL: no gate
                            const float* __restrict__ x,          // [rows][1024]
                            long long total_rows, float gate,     // 1 / (1 - realised p) (1: no dropout)
                            float* __restrict__ part,             // [64 ranges][256][1024] partial dW
                            float* __restrict__ part_cs) {        // [64 ranges][256] partial column sums of g
    __shared__ __attribute__((aligned(1024))) char lds[2 * 4 * GIMG];        // [stage][g hi | g lo | x hi | x lo]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroup id -> (row range, column block): the four column blocks of a row range read the same dH / H_bag rows and must
    // share an L2, i.e. an XCD (blockIdx -> XCD is round-robin, id mod 8).  r03 PMC: with the blocks of a range on FOUR
    // NEIGHBOURING ids (= four XCDs) the kernel fetched 9.90 GB per 8 x 100 000 window against 4.92 GB algorithmic -- every
    // block its own copy of dH and H_bag (profiles/r03_pmc_mcat_f32_100k.json)
    static_assert(GRANGES % 8 == 0, "row ranges are dealt to the 8 XCDs");
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int range = xcd + 8 * (slot >> 2), cb = slot & 3;
    const long long per = ((total_rows + GRANGES - 1) / GRANGES + GCH - 1) / GCH * GCH;
    const long long r0 = (long long)range * per, r1 = min(total_rows, r0 + per);
    const int nch = r1 > r0 ? (int)((r1 - r0 + GCH - 1) / GCH) : 0;
    // wave (gw = wave >> 1: 64 rows of dW = embed columns 64 gw .. + 63 of g; xw = wave & 1: 128 of this block's X columns)
    const int gw = wave >> 1, xw = wave & 1;
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // staging: thread t carries row t >> 4 (32 rows), 16-float column group t & 15 (16 x 16 = 256 columns): four float4 each
    // of dH, H_bag and X
    const int srow = tid >> 4, scg = tid & 15;
    float cs[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) cs[j] = 0.f;
    f32x4 sg[4], sh[4], sx[4];
    auto load_chunk = [&](int ch) {
        const long long row = min(r0 + (long long)ch * GCH + srow, total_rows - 1);
        const float* pg = dh + (size_t)row * FE + 16 * scg;
        const float* px = x + (size_t)row * FK + 256 * cb + 16 * scg;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sg[j] = *reinterpret_cast<const f32x4*>(pg + 4 * j);
            sx[j] = *reinterpret_cast<const f32x4*>(px + 4 * j);
        }
        if (hbag != nullptr) {
            const float* ph = hbag + (size_t)row * FE + 16 * scg;
#pragma unroll
            for (int j = 0; j < 4; ++j) sh[j] = *reinterpret_cast<const f32x4*>(ph + 4 * j);
        }
    };
    if (nch > 0) load_chunk(0);
    for (int ch = 0; ch < nch; ++ch) {
        char* st = lds + (ch & 1) * (4 * GIMG);
        {   // gate, split, write the four images of this chunk
            const bool live = r0 + (long long)ch * GCH + srow < r1;      // rows past the range contribute zeros
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float gv = sg[j][e];
                    if (hbag != nullptr) gv = sh[j][e] > 0.f ? gv * gate : 0.f;
                    sg[j][e] = live ? gv : 0.f;
                    if (!live) sx[j][e] = 0.f;
                    cs[4 * j + e] += sg[j][e];
                }
            bf16x8 h0, l0, h1, l1;
            split8(sg[0], sg[1], h0, l0);
            split8(sg[2], sg[3], h1, l1);
            *reinterpret_cast<bf16x8*>(st + gimg_off(srow, 2 * scg)) = h0;
            *reinterpret_cast<bf16x8*>(st + gimg_off(srow, 2 * scg + 1)) = h1;
            *reinterpret_cast<bf16x8*>(st + GIMG + gimg_off(srow, 2 * scg)) = l0;
            *reinterpret_cast<bf16x8*>(st + GIMG + gimg_off(srow, 2 * scg + 1)) = l1;
            split8(sx[0], sx[1], h0, l0);
            split8(sx[2], sx[3], h1, l1);
            *reinterpret_cast<bf16x8*>(st + 2 * GIMG + gimg_off(srow, 2 * scg)) = h0;
            *reinterpret_cast<bf16x8*>(st + 2 * GIMG + gimg_off(srow, 2 * scg + 1)) = h1;
            *reinterpret_cast<bf16x8*>(st + 3 * GIMG + gimg_off(srow, 2 * scg)) = l0;
            *reinterpret_cast<bf16x8*>(st + 3 * GIMG + gimg_off(srow, 2 * scg + 1)) = l1;
        }
        if (ch + 1 < nch) load_chunk(ch + 1);
        __syncthreads();
        // dW[e][k] += sum_p g[p][e] x[p][k]: A = g^T (rows = embed column, k = patch), B = x^T ... both read transposed out of
        // the row-major images: col_frag(t) gives, for column 16 t + (lane & 15), the 8 patches of k-order(g, j) -- the same
        // order on both sides
        bf16x8 ah[4], al[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ah[i] = col_frag<256>(st, 4 * gw + i, lane);
            al[i] = col_frag<256>(st + GIMG, 4 * gw + i, lane);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bf16x8 bh = col_frag<256>(st + 2 * GIMG, 8 * xw + j, lane);
            const bf16x8 bl = col_frag<256>(st + 3 * GIMG, 8 * xw + j, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = mfma_bf16(ah[i], bh, acc[i][j]);      // (term-major: dependent MFMAs four apart)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = mfma_bf16(ah[i], bl, acc[i][j]);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = mfma_bf16(al[i], bh, acc[i][j]);
        }
    }
    // partial block: acc[i][j][r] = dW[64 gw + 16 i + 4 g + r][256 cb + 128 xw + 16 j + (lane & 15)]
    const int c16 = lane & 15, g = lane >> 4;
    float* pout = part + (size_t)range * FE * FK;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                pout[(size_t)(64 * gw + 16 * i + 4 * g + r) * FK + 256 * cb + 128 * xw + 16 * j + c16] = acc[i][j][r];
    if (cb == 0 && part_cs != nullptr) {                                  // column sums of g: every thread holds 16 columns of its row
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds);                       // [32 rows][256]
#pragma unroll
        for (int j = 0; j < 16; ++j) red[srow * 256 + 16 * scg + j] = cs[j];
        __syncthreads();
        if (tid < 256) {
            float a = 0.f;
#pragma unroll 8
            for (int r = 0; r < 32; ++r) a += red[r * 256 + tid];
            part_cs[(size_t)range * FE + tid] = a;
        }
    }
}

// d_weight[i] = sum over the 64 range partials; d_bias likewise
__global__ void patch_wgrad_f32_reduce_kernel(const float* __restrict__ part, const float* __restrict__ part_cs,
                                              float* __restrict__ dw, float* __restrict__ db) {
    const int i4 = blockIdx.x * blockDim.x + threadIdx.x;                 // float4 index
    if (i4 < FE * FK / 4) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int s = 0; s < GRANGES; ++s) a += *reinterpret_cast<const f32x4*>(part + (size_t)s * FE * FK + 4 * (size_t)i4);
        *reinterpret_cast<f32x4*>(dw + 4 * (size_t)i4) = a;
    }
    if (db != nullptr && i4 < FE) {
        float a = 0.f;
        for (int s = 0; s < GRANGES; ++s) a += part_cs[s * FE + i4];
        db[i4] = a;
    }
}

}  // namespace

size_t mpo_patch_fc_f32_workspace_floats() { return (size_t)FE * FK + 64; }       // packed hi | lo fp16 weight = 1 MiB, then its scale
size_t mpo_patch_wgrad_f32_workspace_floats() { return (size_t)GRANGES * FE * FK + (size_t)GRANGES * FE; }

int mpo_launch_patch_fc_f32(const float* x, const float* w, const float* bias, float* h, long long total_rows, int embed,
                            int patch_dim, float drop_p, unsigned long long seed, unsigned long long offset,
                            const unsigned long long* epoch, float x_scale, float* ws, hipStream_t stream) {
    MPO_CHECK(embed == FE && patch_dim == FK, "fp32 patch layer kernel: built for %d -> %d (got %d -> %d)", FK, FE, patch_dim, embed);
    MPO_CHECK(drop_p >= 0.f && drop_p < 1.f, "patch-layer dropout p must be in [0,1) (got %f)", (double)drop_p);
    {
        int e = 0;
        MPO_CHECK(x_scale > 0.f && x_scale < INFINITY && frexpf(x_scale, &e) == 0.5f, "fp32 patch layer: x_scale must be a power of two (got %g)", (double)x_scale);
    }
    if (total_rows <= 0) return 0;
    float* w_scale = ws + (size_t)FE * FK;
    weight_range_kernel<<<1, 1024, 0, stream>>>(w, w_scale);
    MPO_LAUNCH_CHECK();
    pack_patch_weight_f32_kernel<<<FE * FK / 8 / 256, 256, 0, stream>>>(w, reinterpret_cast<f16x8*>(ws), w_scale);
    MPO_LAUNCH_CHECK();
    const int target = 256;
    long long per = (total_rows + target - 1) / target;
    per = (per + FBM - 1) / FBM * FBM;
    const int grid = (int)((total_rows + per - 1) / per);
    patch_fc_f32_kernel<MPO_F32_FWD_WAVES><<<grid, 64 * MPO_F32_FWD_WAVES, 0, stream>>>(x, reinterpret_cast<const f16x8*>(ws), bias, h, total_rows, (int)per, drop_p, seed,
                                                   offset, epoch, x_scale, w_scale);
    MPO_LAUNCH_CHECK();
    return 0;
}

int mpo_launch_patch_wgrad_f32(const float* dh, const float* hbag, const float* x, long long total_rows, int embed, int patch_dim,
                               float gate, float* d_weight, float* d_bias, float* ws, hipStream_t stream) {
    MPO_CHECK(embed == FE && patch_dim == FK, "fp32 patch weight gradient: built for %d x %d (got %d x %d)", FE, FK, embed, patch_dim);
    MPO_CHECK(total_rows > 0, "fp32 patch weight gradient: no rows");
    float* part = ws;
    float* part_cs = ws + (size_t)GRANGES * FE * FK;
    patch_wgrad_f32_kernel<<<GRANGES * GCOLB, GW * 64, 0, stream>>>(dh, hbag, x, total_rows, gate, part, part_cs);
    MPO_LAUNCH_CHECK();
    patch_wgrad_f32_reduce_kernel<<<FE * FK / 4 / 256, 256, 0, stream>>>(part, part_cs, d_weight, d_bias);
    MPO_LAUNCH_CHECK();
    return 0;
}
