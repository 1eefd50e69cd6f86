// Key projection of K2:  K[m][n] = sum_e H[m][e] * W_k[n][e] + b_k[n]   (models/blocks.py:151-166, the k = in-projection
// of the key), for a bf16-stored bag H [rows][256] and fp32 W_k [256][256]; K leaves in fp32.
//
// NaCAGaT needs k explicitly and to ~fp32 accuracy (its gate multiplies k's rounding error).  H is EXACT in
// bf16 (it is stored that way), so splitting only the weights, W = W_hi + W_mid + W_lo (three bf16 terms, 24 mantissa
// bits: all of fp32), gives
//     K = H W_hi^T + H W_mid^T + H W_lo^T      with fp32 accumulation, i.e. the fp32 GEMM's result
// from three bf16 MFMAs per product -- no fp32 copy of the bag (0.12 ms) and no fp32 library GEMM (0.55 ms).  (Two terms
// leave a 2^-17 weight residual, which NaCAGaT's gate amplified to 2.9e-3 on the peaky fixture's map.)
//
// Layout: a workgroup of 8 waves streams 32-row tiles of H; wave w owns output columns [32w, 32w+32) and keeps ITS slice
// of the three weight terms in registers for the whole kernel (2 column tiles x 8 k-steps x 3 terms = 192 VGPRs; with
// 16 waves x 16 columns the 128-register cap of a 1024-thread workgroup spilled), so W costs no
// memory traffic at all after the prologue.  The H tile is staged once per workgroup into a double-buffered LDS image
// (K1's swizzle: conflict-free ds_read_b128 row fragments) with the next tile's global loads in flight during the MFMAs.
// HBM-bound: reads rows*512 B, writes rows*1024 B.
#include <type_traits>

#include "coattn_tile.h"
#include "mpo_common.h"
#include "mpo_kernels.h"

namespace {

// (a native vector, not HIP's uint4 struct: arrays of the struct stayed in scratch memory -- every staged chunk went
//  global -> wait -> scratch -> LDS, which serialised the whole pipeline)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Diagnostic builds only (tools/gpu_probe_keyproj_variants.py): bit 0 drops the stores, bit 1 the LDS reads + MFMAs,
// bit 2 the global loads.  The product is built with 0.
#ifndef KP_VARIANT
#define KP_VARIANT 0
#endif
constexpr int KP_E = 256;
constexpr int KP_WAVES = 8;
constexpr int KP_THREADS = KP_WAVES * 64;
constexpr int KP_NCT = KP_E / (16 * KP_WAVES);         // 16-column tiles a wave owns
constexpr int KP_CHUNKS = kTileRows * 32 / KP_THREADS;  // 16-byte chunks of a tile each thread stages

__global__ __launch_bounds__(KP_THREADS, 1)
void key_proj_kernel(const u32x4* __restrict__ hbag /* bf16 [rows][256] as 16-byte chunks */, const float* __restrict__ w,
                     const float* __restrict__ bias, float* __restrict__ kout, int rows) {
    using G = TileGeom<KP_E>;
    __shared__ __attribute__((aligned(16))) char img[2][G::TILEB];
    __shared__ __attribute__((aligned(16))) float sbias[KP_E];   // read at store time: LDS traffic does not touch vmcnt
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    const int n0 = 16 * KP_NCT * wave;

    // this wave's weight slice as MFMA B fragments: lane supplies W[n0 + 16 ct + c16][32 s + 8 g .. + 7], split in three
    bf16x8 whi[KP_NCT][G::KS], wmid[KP_NCT][G::KS], wlo[KP_NCT][G::KS];
    // (the weight fragment is the MFMA's A operand and the H rows its B operand: D[n = 4g + r][patch = c16], so a lane ends
    //  up with FOUR CONSECUTIVE OUTPUT COLUMNS of one patch row -- one 16-byte store instead of four 4-byte ones)
#pragma unroll
    for (int ct = 0; ct < KP_NCT; ++ct) {
        const float* wr = w + (size_t)(n0 + 16 * ct + c16) * KP_E;
#pragma unroll
        for (int s = 0; s < G::KS; ++s) {
            const f32x4 lo4 = *reinterpret_cast<const f32x4*>(wr + 32 * s + 8 * g);
            const f32x4 hi4 = *reinterpret_cast<const f32x4*>(wr + 32 * s + 8 * g + 4);
            const float v[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const __bf16 h = (__bf16)v[j];
                const float r1 = v[j] - (float)h;
                const __bf16 m = (__bf16)r1;
                whi[ct][s][j] = h;
                wmid[ct][s][j] = m;
                wlo[ct][s][j] = (__bf16)(r1 - (float)m);
            }
        }
    }

    if (tid < KP_E) sbias[tid] = bias ? bias[tid] : 0.f;
    __syncthreads();

    // cooperative staging: the tile is 32 rows x 32 chunks of 16 bytes; thread t moves chunks t and t + 512
    const int ntiles = (rows + kTileRows - 1) / kTileRows;
    // TWO tiles are in flight in registers (sa: the next tile, sb: the one after): one tile of MFMA work (~0.7 us) does
    // not cover an HBM round trip, and the 8 waves of the workgroup advance in lockstep behind one barrier per tile
    u32x4 sa[KP_CHUNKS], sb[KP_CHUNKS];
    auto fetch = [&](u32x4 (&st)[KP_CHUNKS], int tile) {
#pragma unroll
        for (int i = 0; i < KP_CHUNKS; ++i) {
            const int ci = tid + i * KP_THREADS;
            const int r = ci >> 5, cc = ci & 31;
            int row = tile * kTileRows + r;
            row = row < rows ? row : rows - 1;                               // clamp: finite data, stores are guarded
            if ((KP_VARIANT & 4) == 0 || rows < 0) st[i] = hbag[(size_t)row * 32 + cc];
        }
    };
    auto stage = [&](const u32x4 (&st)[KP_CHUNKS], char* image) {
#pragma unroll
        for (int i = 0; i < KP_CHUNKS; ++i) {
            const int ci = tid + i * KP_THREADS;
            const int r = ci >> 5, cc = ci & 31;
            *reinterpret_cast<u32x4*>(image + r * G::ROWB + ((cc ^ ((r & 7) << 1)) << 4)) = st[i];
        }
    };
    auto compute = [&](const char* image, int tile, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;       // FULL: all 32 rows exist -> unconditional stores
        f32x4 acc[2][KP_NCT];
#pragma unroll
        for (int ct = 0; ct < KP_NCT; ++ct) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + n0 + 16 * ct + 4 * g);   // columns n0 + 16 ct + 4g .. + 3
            acc[0][ct] = b4;
            acc[1][ct] = b4;
        }
#pragma unroll
        for (int s = 0; s < ((KP_VARIANT & 2) ? 0 : G::KS); ++s) {
            const bf16x8 a0 = row_frag<KP_E>(image, 0, s, lane);
            const bf16x8 a1 = row_frag<KP_E>(image, 1, s, lane);
#pragma unroll
            for (int ct = 0; ct < KP_NCT; ++ct) {
                acc[0][ct] = mfma_bf16(whi[ct][s], a0, acc[0][ct]);
                acc[1][ct] = mfma_bf16(whi[ct][s], a1, acc[1][ct]);
                acc[0][ct] = mfma_bf16(wmid[ct][s], a0, acc[0][ct]);
                acc[1][ct] = mfma_bf16(wmid[ct][s], a1, acc[1][ct]);
                acc[0][ct] = mfma_bf16(wlo[ct][s], a0, acc[0][ct]);
                acc[1][ct] = mfma_bf16(wlo[ct][s], a1, acc[1][ct]);
            }
        }
        // D[n = 4g + r][patch = c16] of each 16 x 16 block: lane -> K[row0 + 16 pt + c16][n0 + 16 ct + 4g .. + 3]
        const int row0 = tile * kTileRows;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const int row = row0 + 16 * pt + c16;
            if (((KP_VARIANT & 1) == 0 || rows < 0) && (FULL || row < rows)) {
                float* o = kout + (size_t)row * KP_E + n0 + 4 * g;
#pragma unroll
                for (int ct = 0; ct < KP_NCT; ++ct) *reinterpret_cast<f32x4*>(o + 16 * ct) = acc[pt][ct];
            }
        }
    };
    // Control flow is kept uniform -- fetches are unconditional with the tile index clamped to the last tile (a few
    // redundant loads at the tail) -- so that the compiler's waitcnt bookkeeping sees ONE steady-state pattern and waits
    // with vmcnt(n > 0): conditional fetches made it fall back to vmcnt(0) before every LDS write, i.e. to waiting for
    // the previous tile's stores and the other prefetch set as well, which serialised loads, MFMAs and stores.
    const int stride = gridDim.x;
    const int last = ntiles - 1;
    auto clampt = [&](int t) { return t < last ? t : last; };
    int tile = blockIdx.x;
    fetch(sa, clampt(tile));
    fetch(sb, clampt(tile + stride));
    // two tiles per trip so that the register sets keep their roles; one barrier per tile (the buffer written now
    // was last read two barriers ago)
    const int full_tiles = rows / kTileRows;                   // tiles below this index have all 32 rows
    using Full = std::integral_constant<bool, true>;
    using Guarded = std::integral_constant<bool, false>;
    // steady state: both tiles of the trip are full -> no predicated memory operation anywhere in the loop body
    for (; tile + stride < full_tiles; tile += 2 * stride) {
        stage(sa, img[0]);
        fetch(sa, clampt(tile + 2 * stride));
        __syncthreads();
        compute(img[0], tile, Full());
        stage(sb, img[1]);
        fetch(sb, clampt(tile + 3 * stride));
        __syncthreads();
        compute(img[1], tile + stride, Full());
    }
    // tail: at most one trip per workgroup with guarded stores (the register sets already hold these tiles)
    if (tile < ntiles) {
        stage(sa, img[0]);
        __syncthreads();
        compute(img[0], tile, Guarded());
        stage(sb, img[1]);
        __syncthreads();
        if (tile + stride < ntiles) compute(img[1], tile + stride, Guarded());
    }
}

}  // namespace

int mpo_launch_key_proj(const void* hbag_bf16, const float* w, const float* bias, float* kout, int rows, int embed,
                        hipStream_t stream) {
    MPO_CHECK(embed == KP_E, "key projection kernel: embed_dim %d not built (256 only)", embed);
    MPO_CHECK(rows >= 1, "key projection: no rows");
    MPO_CHECK((reinterpret_cast<uintptr_t>(hbag_bf16) & 15) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(bias) & 15) == 0 && (reinterpret_cast<uintptr_t>(kout) & 15) == 0,
              "key projection: bag, weight, bias and output must be 16-byte aligned");
    const int ntiles = (rows + kTileRows - 1) / kTileRows;
    const int grid = ntiles < 256 ? ntiles : 256;                            // one persistent workgroup per CU
    key_proj_kernel<<<grid, KP_THREADS, 0, stream>>>(static_cast<const u32x4*>(hbag_bf16), w, bias, kout, rows);
    MPO_LAUNCH_CHECK();
    return 0;
}
