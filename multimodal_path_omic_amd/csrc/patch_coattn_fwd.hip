// Row f1 of SURVEY.md section 8(f): the patch layer fused with the forward of MCAT's co-attention (K1), ONE pass
// over the raw patch matrix.
//
// Replaces, for a bf16-stored window at embed_dim 256 (models/mcat/mcat.py:24-29,87,97):
//     H_bag = Dropout(ReLU(X W_H^T + b_H))                  X [rows, 1024], W_H [256, 1024]
//     S = qk H_bag^T, online softmax, ctx = A H_bag        (the folded co-attention of coattn_fwd.hip)
// as a persistent kernel, one workgroup of 12 waves per CU (three per SIMD); a workgroup owns a contiguous row range of
// ONE slide (the window's work plan, coattn_tile.h) and walks it in blocks of 128 patch rows.  Three roles, three code
// paths with the same sequence of workgroup barriers, so that no role's accumulators are live in another's:
//
//   waves 0-7, GEMM: 16 steps of K = 64 per block, H^T[256 x 128] += W_H[256 x 64] X^T[64 x 128] on the bf16 MFMA
//     (v_mfma_f32_16x16x32_bf16, fp32 accumulate); wave w owns embed columns 32 w .. 32 w + 31 of all 128 patch rows
//     (2 x 8 tiles, 64 accumulator registers).  The product is taken TRANSPOSED (A operand = W_H, B operand = X) so that
//     a lane ends up with four consecutive embed columns of one patch row: the epilogue packs them into one 8-byte LDS
//     store of a row-major H image.  X fragments are read from an LDS ring; the W_H fragments of a lane are 16 contiguous
//     bytes of global memory and come straight from L2 into registers one step ahead (512 KiB shared by every workgroup
//     and re-read per block: sending it through LDS as well tripled the LDS-DMA traffic of a CU and made the loaders the
//     bottleneck -- measured).  Epilogue per block: + bias, ReLU, dropout (counter hash), bf16, into the H image.
//     During the even steps a GEMM wave also copies two rows of the PREVIOUS block's image out to H_bag (the backward pass
//     reads it): the store tail of a block is issue-bound (~5 k cycles per 64 KiB), so it is spread under the next
//     block's main loop instead of standing between two blocks.
//   waves 8-11, X loaders + co-attention (one per SIMD): the X ring (5 stages x 16 KiB, global -> LDS directly by
//     global_load_lds_dwordx4, 1 KiB per wave-instruction, hand-counted s_waitcnt) is kept FOUR stages ahead of the step
//     being consumed -- about 48 KiB must be in flight per CU to cover the HBM latency at full bandwidth -- and nothing
//     else sits in these waves' memory queues: gfx950 retires a wave's memory operations in issue order, so a wave that
//     also waited for a short-latency load would wait for every stage requested before it.  Between two step barriers
//     each of them also runs one SLICE of K1's tile step (scores with the query operand split in three bf16 terms, online
//     softmax in log2 units, context accumulation; coattn_fwd.hip) on the image of the PREVIOUS block.  Waves 8/9 share
//     image rows 0-63, waves 10/11 rows 64-127 (two 32-row tiles, four slices each); within a pair both compute the
//     (bit-identical) scores and each accumulates HALF of the 256 context columns: 32 accumulator registers instead of
//     64, which is what lets the role fit beside the GEMM role at three waves per SIMD.
//   One s_barrier per step (passing barrier k means X stage k has landed and stage k - 1 has been read) + one after the
//   image is written.  The image of block t is consumed during steps 0..14 of block t + 1 (after the last block: at once).
//   Stage images are [row][64 k] with 128-byte rows, 16-byte chunk c of row r stored at c ^ ((r >> 1) & 7) (applied to the
//   GLOBAL chunk a lane fetches; the LDS side of an LDS-DMA is linear): conflict-free ds_read_b128 fragments.  The H image
//   uses coattn_tile.h's swizzle so that row_frag / col_frag read it.
//   End of range: the two (max, sum, ctx) states (one per pair) are merged through LDS into one partial per workgroup, combined per
//   slide by coattn_combine_kernel as for K1.
//
// Roofline: HBM.  Algorithmic bytes per patch row: 2048 read + 512 written; 524 288 flop per row put the MFMA floor at
// about half the HBM floor.
#include "coattn_tile.h"
#include "mpo_kernels.h"

namespace {

constexpr int PE = 256;                         // embed_dim
constexpr int PK = 1024;                        // patch feature width (models/mcat/mcat.py:25)
constexpr int BM = 128;                         // patch rows per block
constexpr int BK = 64;                          // k per step
constexpr int KSTEPS = PK / BK;                 // 16
constexpr int SROWB = BK * 2;                   // 128: bytes per row of a stage image
constexpr int A_STAGE = BM * SROWB;             // 16 KiB: X stage
constexpr int A_SLOTS = 5;
constexpr int A_AHEAD = A_SLOTS - 1;            // X stages requested ahead of the one being consumed
constexpr int OFF_A = 0;
constexpr int OFF_IMG = A_SLOTS * A_STAGE;      // 81 920: the H image of a block, 128 x 512 B = 64 KiB
constexpr int IMG_ROWB = PE * 2;                // 512
constexpr int OFF_Q = OFF_IMG + BM * IMG_ROWB;  // 147 456
constexpr int QCAP = 9;                         // query slots per fragment row: n_q <= 8 live + one zero slot
constexpr int Q_BYTES = 3 * 8 * 4 * QCAP * 16;  // [term][k-step][lane group][slot] x 16 B = 13 824
constexpr int OFF_BIAS = OFF_Q + Q_BYTES;       // 161 280
constexpr int OFF_ML = OFF_BIAS + PE * 4;       // 162 304
constexpr int LDS_TOTAL = OFF_ML + 2 * 128;     // 162 560 <= 163 840
constexpr int NTHREADS = 768;
constexpr int TAIL_STEPS = 4;                   // slices of one 32-row tile; the two tiles of a helper fit steps 0..7 of a block
static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");
static_assert(2 * TAIL_STEPS <= KSTEPS - 1 &&  2 * TAIL_STEPS >= 0, "the image must be consumed before the barrier that precedes its rewrite");

__device__ __forceinline__ unsigned lds_addr(const char* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
// one LDS-DMA wave-instruction: lane l's 16 bytes at `src` land at lds_dst + 16 l (lds_dst wave-uniform)
__device__ __forceinline__ void glds16(const char* src, unsigned lds_dst) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"       // m0 is "reserved": nothing else in this kernel uses it
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(lds_dst) : "memory", "m0");
#pragma clang diagnostic pop
}
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// wait until at most `n` of this wave's vector-memory operations are outstanding (n wave-uniform)
__device__ __forceinline__ void wait_vm(int n) {
    switch (n) {
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;       // (waiting for everything is always safe)
    }
}

// Co-attention state of one helper wave over its row range, and the tile step of coattn_fwd.hip cut into TAIL_STEPS slices
// (query fragments in LDS, three bf16 terms: qk = hi + mid + lo carries all 24 mantissa bits).
struct TailState {
    float m_run, l_run;
    f32x4 cacc[8];                      // ctx^T accumulators: embed tile 8 dhalf + t, rows 4 g + r, column = query
    f32x4 s0, s1;                       // scores of the tile in flight
    bf16x8 ph, pl;                      // its exponentiated scores, hi / lo
};
template <int SUB>
__device__ __forceinline__ void tail_slice(TailState& st, const char* tile, int nvalid, const char* qf, int qslot, int dhalf,
                                           float* s_out, bool q_live, int lane) {
    const int g = lane >> 4;
    if constexpr (SUB < 2) {                                      // scores: k-steps 4 SUB .. 4 SUB + 3 of 8
        if constexpr (SUB == 0) {
            st.s0 = f32x4{0.f, 0.f, 0.f, 0.f};
            st.s1 = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int s = 4 * SUB; s < 4 * SUB + 4; ++s) {
            const bf16x8 a0 = row_frag<PE>(tile, 0, s, lane);
            const bf16x8 a1 = row_frag<PE>(tile, 1, s, lane);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const bf16x8 q = *reinterpret_cast<const bf16x8*>(qf + ((((t * 8 + s) * 4 + g) * QCAP + qslot) << 4));
                st.s0 = mfma_bf16(a0, q, st.s0);
                st.s1 = mfma_bf16(a1, q, st.s1);
            }
        }
    } else if constexpr (SUB == 2) {                              // mask, online softmax, rescale
        float sv[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sv[r] = (4 * g + r < nvalid) ? st.s0[r] : -INFINITY;
            sv[4 + r] = (16 + 4 * g + r < nvalid) ? st.s1[r] : -INFINITY;
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
        const float m_new = fmaxf(st.m_run, mx);                  // finite: row 0 of a processed tile is valid
        const float alpha = __builtin_amdgcn_exp2f(st.m_run - m_new);     // 0 on the first tile (m_run = -inf)
        float pv[8], ps = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            pv[j] = __builtin_amdgcn_exp2f(sv[j] - m_new);
            ps += pv[j];
        }
        st.l_run = st.l_run * alpha + ps;
        st.m_run = m_new;
        if (!__all(alpha == 1.0f)) {
#pragma unroll
            for (int t = 0; t < 8; ++t) st.cacc[t] *= alpha;
        }
        pack_hi_lo(pv, st.ph, st.pl);
    } else {                                                      // context: this wave's eight embed tiles
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const bf16x8 hfr = col_frag<PE>(tile, 8 * dhalf + t, lane);
            st.cacc[t] = mfma_bf16(hfr, st.ph, st.cacc[t]);
            st.cacc[t] = mfma_bf16(hfr, st.pl, st.cacc[t]);
        }
    }
}
// slice SL (0 .. 2 TAIL_STEPS - 1) of a helper's two tiles
template <int SL>
__device__ __forceinline__ void tail_step(TailState& st, const char* img64 /* the pair's 64 image rows */, int rows64,
                                          const char* qf, int qslot, int dhalf, float* s_row64, bool q_live, int lane) {
    constexpr int TILE = SL / TAIL_STEPS, SUB = SL % TAIL_STEPS;
    const int nvalid = min(kTileRows, rows64 - kTileRows * TILE);
    if (nvalid > 0)
        tail_slice<SUB>(st, img64 + TILE * kTileRows * IMG_ROWB, nvalid, qf, qslot, dhalf,
                        s_row64 ? s_row64 + TILE * kTileRows : nullptr, q_live, lane);
}
template <int SL>
__device__ __forceinline__ void tail_steps_from(TailState& st, const char* img64, int rows64, const char* qf, int qslot, int dhalf,
                                                float* s_row64, bool q_live, int lane) {
    if constexpr (SL < 2 * TAIL_STEPS) {
        tail_step<SL>(st, img64, rows64, qf, qslot, dhalf, s_row64, q_live, lane);
        tail_steps_from<SL + 1>(st, img64, rows64, qf, qslot, dhalf, s_row64, q_live, lane);
    }
}

// one wave-store: image rows `row0`, `row0 + 1` (1 KiB) -> H_bag
__device__ __forceinline__ void copy_out_rows(const char* img, char* hrow, int rows_valid, int row0, int lane) {
    const int row = row0 + (lane >> 5), ch = lane & 31;
    const f32x4 v = *reinterpret_cast<const f32x4*>(img + row * IMG_ROWB + ((ch ^ ((row & 7) << 1)) << 4));
    if (row < rows_valid) *reinterpret_cast<f32x4*>(hrow + (size_t)row * IMG_ROWB + (ch << 4)) = v;
}

__global__ __launch_bounds__(NTHREADS, 3)
void patch_coattn_fwd_kernel(const __bf16* __restrict__ x,        // [total_rows][1024] patch features
                             const __bf16* __restrict__ wb,       // W_H rounded to bf16, packed in fragment order (see load_w)
                             const float* __restrict__ bias,      // [256]
                             const int* __restrict__ cu,
                             const float* __restrict__ qk2,       // [n_slides][n_q][256], log2 units
                             __bf16* __restrict__ h_out,          // [total_rows][256]
                             float* __restrict__ part_ml, float* __restrict__ part_ctx,
                             float* __restrict__ s_out,           // nullable: raw log2 logits, ragged [n_q][M_b] blocks
                             int n_q, float drop_p, unsigned long long seed, unsigned long long offset_,
                             const unsigned long long* __restrict__ epoch, BagPlan plan) {
    __shared__ __attribute__((aligned(1024))) char lds[LDS_TOTAL];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const WgGeom wg = wg_geom(cu, plan);
    const int b = wg.b, row_begin = wg.row_begin, m_rows = wg.m_rows, r0 = wg.r0, r1 = wg.r1;
    const int nblocks = r1 > r0 ? (r1 - r0 + BM - 1) / BM : 0;
    const int n_stages = nblocks * KSTEPS;
    const bool tail_on = qk2 != nullptr;      // NULL: patch layer only (H_bag is the only output); same barriers, no co-attention slices

    // ---- prologue: bias and the query fragments (three bf16 terms, compact [term][k-step][group][slot]) into LDS
    {
        float* lb = reinterpret_cast<float*>(lds + OFF_BIAS);
        if (tid < PE) lb[tid] = bias[tid];
        const float* qrow = tail_on ? qk2 + (size_t)b * n_q * PE : nullptr;
        for (int e = tid; tail_on && e < 8 * 4 * QCAP; e += NTHREADS) {
            const int slot = e % QCAP, sg = e / QCAP;             // sg = 4 s + g
            bf16x8 t0, t1, t2;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = slot < n_q ? qrow[slot * PE + 8 * sg + j] : 0.f;      // k = 32 s + 8 g + j
                const __bf16 hi = (__bf16)v;
                const float r1f = v - (float)hi;
                const __bf16 mid = (__bf16)r1f;
                t0[j] = hi;
                t1[j] = mid;
                t2[j] = (__bf16)(r1f - (float)mid);
            }
            char* dst = lds + OFF_Q + ((sg * QCAP + slot) << 4);
            *reinterpret_cast<bf16x8*>(dst) = t0;
            *reinterpret_cast<bf16x8*>(dst + 8 * 4 * QCAP * 16) = t1;
            *reinterpret_cast<bf16x8*>(dst + 2 * 8 * 4 * QCAP * 16) = t2;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // (the plain loads above must not sit in front of the ring)
    const unsigned long long offset = epoch_offset(offset_, epoch);
    const uint32_t drop_key = hash_stream_key(seed, offset);          // (the stream offset is in the key: the counter below is the element group alone)
    const size_t pbase = wg.part;
    // Barrier sequence of EVERY role: 1 (prologue) + per block [16 step barriers + 1 image barrier] + 1 (states in LDS).

    if (wave < 8) {
        // ================================================================ GEMM role: embed columns 32 wave .. + 31, all 128 patch rows
        const int g = lane >> 4;
        const int frow = (lane & 15) * SROWB;                     // fragment row inside a 16-row tile of a stage image
        const int fswz = (lane >> 1) & 7;                         // its chunk swizzle: ((row >> 1) & 7) depends on the lane only
        const int fc0 = ((0 + g) ^ fswz) << 4, fc1 = ((4 + g) ^ fswz) << 4;
        const uint32_t thr8 = (uint32_t)(drop_p * 256.0f + 0.5f); // keep iff byte >= thr8: realised p = thr8 / 256
        const float inv_keep = drop_p > 0.f ? 256.0f / (256.0f - (float)thr8) : 1.0f;
        // W_H fragments come from a PACKED copy of the weight (pack_patch_weight_kernel below): the 64 fragments that one
        // wave-instruction needs -- lane (i, g): row 32 wave + 16 dt + i, k = 64 step + 32 s + 8 g .. + 7 -- are 1 KiB
        // contiguous, [wave][step][dt][s][lane][8].  Read in place from the row-major matrix the same instruction touches
        // 16 rows x 64 B, which costs the address/texture path several times a contiguous KiB (measured: the GEMM waves
        // then stalled at issue on their own weight loads).
        const char* wpk = reinterpret_cast<const char*>(wb) + (size_t)wave * (KSTEPS * 4 * 1024) + lane * 16;
        auto load_w = [&](int k, bf16x8 (&w)[2][2]) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    w[dt][s] = *reinterpret_cast<const bf16x8*>(wpk + ((k * 2 + dt) * 2 + s) * 1024);
        };
        int cslot = 0;
        f32x4 acc[2][8];
        auto step = [&](const bf16x8 (&w)[2][2], bf16x8 (&wn)[2][2], int kn) {
            wg_barrier();                                         // this step's X stage has landed
            load_w(kn, wn);                                       // W_H fragments of three steps ahead (spread below)
            const char* xa = lds + OFF_A + cslot * A_STAGE + frow;
            cslot = cslot + 1 == A_SLOTS ? 0 : cslot + 1;
            {
                // 16 X fragments, two MFMAs each (the wave's two embed tiles).  The reads run FIVE fragments ahead of the
                // MFMAs that consume them: LDS latency is ~100+ cycles against 32 cycles of matrix work per fragment, and
                // left to itself hipcc keeps two reads in flight (measured: the matrix pipe idled 60 % of the step)
                bf16x8 xf[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(xa + (j & 7) * 16 * SROWB + (j < 8 ? fc0 : fc1));
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    acc[0][j & 7] = mfma_bf16(w[0][j >> 3], xf[j], acc[0][j & 7]);
                    acc[1][j & 7] = mfma_bf16(w[1][j >> 3], xf[j], acc[1][j & 7]);
                }
                {
                    __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);        // 5 DS reads
#pragma unroll
                    for (int j = 0; j < 11; ++j) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);    // the two MFMAs of fragment j
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // the read of fragment j + 5
                        // one W_H load after every sixth MFMA: a load that finds the memory pipeline backed up holds the
                        // wave (in-order issue) -- with matrix work already queued behind it, not in front of an idle pipe
                        if (j % 3 == 2 && j < 12) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 10, 0);
                }
            }
        };
        wg_barrier();                                             // (bias + query fragments visible)
        // W_H fragments are requested THREE steps ahead (four register sets, named statically: 4 steps per trip): under
        // the streaming load of the whole chip an L2 hit takes 2-3 us, several steps -- with one step of lookahead every
        // step waited for its fragments and the matrix work and the HBM stream ran one after the other (measured).
        bf16x8 w0[2][2], w1[2][2], w2[2][2], w3[2][2];
        if (nblocks > 0) {
            load_w(0, w0);
            load_w(1, w1);
            load_w(2, w2);
        }
        for (int blk = 0; blk < nblocks; ++blk) {
            const int rb = r0 + blk * BM;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < KSTEPS; k += 4) {                 // (steps past 15 wrap to the next block: W_H is the same for all)
                step(w0, w3, (k + 3) & (KSTEPS - 1));
                step(w1, w0, (k + 4) & (KSTEPS - 1));
                step(w2, w1, (k + 5) & (KSTEPS - 1));
                step(w3, w2, (k + 6) & (KSTEPS - 1));
            }
            // epilogue: acc[dt][pt] holds H^T: embed column 32 wave + 16 dt + 4 g + r of patch row 16 pt + (lane & 15).
            // The previous image was consumed before the helpers reached barrier 15 of this block.
            // (`el` is the lane id made opaque once per block: the per-lane addresses below are then recomputed here
            //  instead of being hoisted out of the block loop, where they would sit in registers through the main loop)
            int el = lane;
            asm volatile("" : "+v"(el));
            {
                const int eg = el >> 4;
                const float* lb = reinterpret_cast<const float*>(lds + OFF_BIAS) + 32 * wave + 4 * eg;
                const f32x4 bv0 = *reinterpret_cast<const f32x4*>(lb), bv1 = *reinterpret_cast<const f32x4*>(lb + 16);
#pragma unroll
                for (int u = 0; u < 4; ++u) {                     // patch tiles 2 u, 2 u + 1 share one draw of the counter hash: 16 elements, 8 bits each
                    uint4 rnd = {0u, 0u, 0u, 0u};
                    if (drop_p > 0.f) {
                        const unsigned long long ctr = (unsigned long long)(row_begin + rb + 32 * u + (el & 15)) * 32ull
                                                       + (unsigned)(4 * wave + eg);
                        rnd = hash4x32(drop_key, ctr);
                    }
                    const uint32_t rw[4] = {rnd.x, rnd.y, rnd.z, rnd.w};
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int pt = 2 * u + half;
                        const int p = 16 * pt + (el & 15);
                        char* rowp = lds + OFF_IMG + p * IMG_ROWB + 8 * (eg & 1);
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt) {
                            bf16x4 o;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float v = fmaxf(acc[dt][pt][r] + (dt ? bv1[r] : bv0[r]), 0.f);
                                if (drop_p > 0.f) v = (((rw[2 * half + dt] >> (8 * r)) & 0xFFu) >= thr8) ? v * inv_keep : 0.f;
                                o[r] = (__bf16)v;
                            }
                            const int c = 4 * wave + 2 * dt + (eg >> 1);
                            *reinterpret_cast<bf16x4*>(rowp + ((c ^ ((p & 7) << 1)) << 4)) = o;
                        }
                    }
                }
            }
            wg_barrier();                                         // the image is complete
        }
        wg_barrier();                                             // (the helpers' states are in LDS)
    } else {
        // ================================================================ X loaders + co-attention
        // The loaders outrank the GEMM waves they share a SIMD with: their few instructions per step (wait, request the
        // next stage, one co-attention slice) must issue at once, not in whatever slots two MFMA streams leave over.
        __builtin_amdgcn_s_setprio(3);
        const int h = wave - 8;                                   // pieces 4 h .. 4 h + 3 of every X stage
        const int pair = h >> 1, dhalf = h & 1;                   // image rows 64 pair .. + 63; context columns 128 dhalf .. + 127
        const char* xs = reinterpret_cast<const char*>(x) + (size_t)row_begin * (PK * 2);     // this slide's rows
        const unsigned lds0 = lds_addr(lds);
        // X stage n (block n / 16, k-step n % 16) -> ring slot n % 5
        auto issue_x = [&](int n, int slot) {
            const int rb = r0 + (n >> 4) * BM, kb = (n & 15) * SROWB;
            int el = lane;                                        // opaque per call: the per-piece address parts are
            asm volatile("" : "+v"(el));                          // recomputed here, not hoisted out of the loops and spilled
            const int lrow = el >> 3;                             // row of this lane inside an 8-row x 128-B piece
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int piece = 4 * h + t;
                const int row = 8 * piece + lrow;
                const int c = (el & 7) ^ ((row >> 1) & 7);
                const int grow = min(rb + row, m_rows - 1);       // rows past the slide: clamped (finite; masked later)
                glds16(xs + (size_t)grow * (PK * 2) + kb + (c << 4), lds0 + OFF_A + slot * A_STAGE + piece * 1024);
            }
        };
        int issued = 0, islot = 0;                                // X stages requested so far, slot of the next one
        for (; issued < A_AHEAD && issued < n_stages; ++issued) {
            issue_x(issued, islot);
            islot = islot + 1 == A_SLOTS ? 0 : islot + 1;
        }
        TailState st;
        st.m_run = -INFINITY;
        st.l_run = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) st.cacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        st.s0 = st.s1 = f32x4{0.f, 0.f, 0.f, 0.f};
        const int q = lane & 15, g = lane >> 4;
        const bool q_live = q < n_q;
        const int qslot = q_live ? q : n_q;                       // slot n_q holds zeros
        // (the raw scores of an inference call are written by one wave of a pair)
        float* s_row = (s_out != nullptr && dhalf == 0) ? s_out + (size_t)n_q * row_begin + (size_t)q * m_rows : nullptr;
        const char* img64 = lds + OFF_IMG + 64 * pair * IMG_ROWB;
        const char* qf = lds + OFF_Q;
        wg_barrier();                                             // (bias + query fragments visible)
        for (int blk = 0; blk < nblocks; ++blk) {
            const int rows_prev = blk > 0 ? BM - 64 * pair : 0;   // rows of the PREVIOUS block's image from this pair's first row on
            float* s_prev = s_row ? s_row + r0 + (blk - 1) * BM + 64 * pair : nullptr;     // (a previous block is never short)
            const bool copy_prev = blk > 0;
            char* hprev = reinterpret_cast<char*>(h_out) + ((size_t)row_begin + r0 + (blk - 1) * BM) * IMG_ROWB;
#pragma unroll 1
            for (int k = 0; k < KSTEPS; ++k) {
                const int n = blk * KSTEPS + k;
                // This wave's own requests for stage n have landed.  gfx950 retires a wave's memory operations in issue
                // order, so the count is the number of its operations YOUNGER than stage n (a count above their true
                // number would let stage n itself slip through): the up to three X stages requested after it, 4 loads
                // each (score stores of an inference call only add to the true number).
                // Plus the row stores of the copy-out below, two per step in steps 0..7 of every block but the first: those of
                // the last four steps are younger than stage n as well (it was requested at step n - 4, before that step's
                // stores).  gfx950 has ONE counter
                // for loads and stores, which is also why these stores live here and not in the GEMM waves: there every
                // wait for the next step's W_H fragments would also wait for a 2-us store acknowledgement.
                int younger_stores = 0;
                if (copy_prev)
                    for (int j = k - 4; j < k; ++j) younger_stores += (j >= 0 && j < 8) ? 2 : 0;
                wait_vm(4 * min(A_AHEAD - 1, n_stages - 1 - n) + younger_stores);
                wg_barrier();                                     // ... and everybody's; stage n - 1 has been read by all
                if (issued < n_stages) {
                    issue_x(issued, islot);
                    islot = islot + 1 == A_SLOTS ? 0 : islot + 1;
                    ++issued;
                }
                if (copy_prev && k < 8) {                         // image rows 32 h + 4 k .. + 3 of the previous block -> H_bag
                    copy_out_rows(lds + OFF_IMG, hprev, BM, 32 * h + 4 * k, lane);
                    copy_out_rows(lds + OFF_IMG, hprev, BM, 32 * h + 4 * k + 2, lane);
                }
                if (tail_on && rows_prev > 0 && k < 2 * TAIL_STEPS) {        // one slice of the previous block's co-attention (k is wave-uniform)
                    int el = lane;                                // (opaque per slice: its LDS addresses are computed here)
                    asm volatile("" : "+v"(el));
#define MPO_TAIL_CASE(SL) if (k == SL) tail_step<SL>(st, img64, rows_prev, qf, qslot, dhalf, s_prev, q_live, el);
                    MPO_TAIL_CASE(0) MPO_TAIL_CASE(1) MPO_TAIL_CASE(2) MPO_TAIL_CASE(3)
                    MPO_TAIL_CASE(4) MPO_TAIL_CASE(5) MPO_TAIL_CASE(6) MPO_TAIL_CASE(7)
#undef MPO_TAIL_CASE
                }
            }
            wg_barrier();                                         // the image is complete
        }
        if (nblocks > 0) {                                        // the last block's image: nobody rewrites it
            const int rows_last = min(BM, r1 - (r0 + (nblocks - 1) * BM)) - 64 * pair;
            float* s_last = s_row ? s_row + r0 + (nblocks - 1) * BM + 64 * pair : nullptr;
            if (tail_on && rows_last > 0) tail_steps_from<0>(st, img64, rows_last, qf, qslot, dhalf, s_last, q_live, lane);
            const int rb = r0 + (nblocks - 1) * BM;
            char* hlast = reinterpret_cast<char*>(h_out) + ((size_t)row_begin + rb) * IMG_ROWB;
            for (int i = 0; i < 16; ++i) copy_out_rows(lds + OFF_IMG, hlast, min(BM, r1 - rb), 32 * h + 2 * i, lane);
        }
        // the two co-attention states of this workgroup -> LDS (the ring region, idle now: 2 x 16 KiB); a wave brings its
        // half of the context columns
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float l_tot = st.l_run + __shfl_xor(st.l_run, 16);
        l_tot += __shfl_xor(l_tot, 32);
        float* wctx = reinterpret_cast<float*>(lds + OFF_A + pair * 16384);         // [16][256] floats
        float* wml = reinterpret_cast<float*>(lds + OFF_ML) + pair * 32;
#pragma unroll
        for (int t = 0; t < 8; ++t) *reinterpret_cast<f32x4*>(wctx + q * PE + 16 * (8 * dhalf + t) + 4 * g) = st.cacc[t];
        if (g == 0 && dhalf == 0) {
            wml[2 * q] = st.m_run;
            wml[2 * q + 1] = l_tot;
        }
        wg_barrier();
    }

    // ---- merge into ONE partial per workgroup (all threads)
    if (!tail_on) return;
    const float* ml = reinterpret_cast<const float*>(lds + OFF_ML);
    for (int idx = tid; idx < n_q * PE; idx += NTHREADS) {
        const int qq = idx / PE;
        float mt = -INFINITY;
#pragma unroll
        for (int w = 0; w < 2; ++w) mt = fmaxf(mt, ml[w * 32 + 2 * qq]);
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const float mw = ml[w * 32 + 2 * qq];
            const float wgt = mw == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mw - mt);
            a += wgt * reinterpret_cast<const float*>(lds + OFF_A + w * 16384)[idx];
        }
        part_ctx[pbase * n_q * PE + idx] = a;
    }
    if (tid < 16) {
        const int qq = tid;
        float mt = -INFINITY, lt = 0.f;
#pragma unroll
        for (int w = 0; w < 2; ++w) mt = fmaxf(mt, ml[w * 32 + 2 * qq]);
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const float mw = ml[w * 32 + 2 * qq];
            lt += (mw == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mw - mt)) * ml[w * 32 + 2 * qq + 1];
        }
        part_ml[pbase * 32 + 2 * qq] = mt;
        part_ml[pbase * 32 + 2 * qq + 1] = lt;
    }
}

// W_H [256][1024] fp32 -> bf16 in the fragment order of the GEMM waves: block ((wave * 16 + step) * 2 + dt) * 2 + s of
// 1 KiB holds, for lane (i = lane & 15, g = lane >> 4), W_H[32 wave + 16 dt + i][64 step + 32 s + 8 g .. + 7].
__global__ void pack_patch_weight_kernel(const float* __restrict__ w, bf16x8* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;          // one 16-byte fragment per thread: 256 * 1024 / 8 of them
    if (t >= PE * PK / 8) return;
    const int lane = t & 63, blk = t >> 6;
    const int s = blk & 1, dt = (blk >> 1) & 1, step = (blk >> 2) & (KSTEPS - 1), wave = blk >> 6;
    const int row = 32 * wave + 16 * dt + (lane & 15), k0 = 64 * step + 32 * s + 8 * (lane >> 4);
    const f32x4 a = *reinterpret_cast<const f32x4*>(w + (size_t)row * PK + k0);
    const f32x4 b = *reinterpret_cast<const f32x4*>(w + (size_t)row * PK + k0 + 4);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        o[j] = (__bf16)a[j];
        o[4 + j] = (__bf16)b[j];
    }
    out[t] = o;
}

}  // namespace

int mpo_launch_pack_patch_weight(const float* w, void* out, int embed, int patch_dim, hipStream_t stream) {
    MPO_CHECK(embed == PE && patch_dim == PK, "patch weight packing is built for %d x %d (got %d x %d)", PE, PK, embed, patch_dim);
    pack_patch_weight_kernel<<<PE * PK / 8 / 256, 256, 0, stream>>>(w, reinterpret_cast<bf16x8*>(out));
    MPO_LAUNCH_CHECK();
    return 0;
}

int mpo_launch_patch_coattn_fwd(const void* x, const void* w_bf16, const float* bias, const int* cu, const float* qk2,
                                void* h_out, float* part_ml, float* part_ctx, float* s_out, int n_q, float drop_p,
                                unsigned long long seed, unsigned long long offset, const unsigned long long* epoch,
                                const BagPlan& plan, hipStream_t stream) {
    MPO_CHECK(qk2 == nullptr || (n_q >= 1 && n_q <= QCAP - 1), "fused patch layer + co-attention: 1..%d queries (got %d)", QCAP - 1, n_q);
    MPO_CHECK(qk2 != nullptr || (part_ml == nullptr && part_ctx == nullptr && s_out == nullptr), "patch layer only: no co-attention outputs");
    MPO_CHECK(drop_p >= 0.f && drop_p < 1.f, "patch-layer dropout p must be in [0,1) (got %f)", (double)drop_p);
patch_coattn_fwd_kernel<<<plan_grid(plan), NTHREADS, 0, stream>>>(
        reinterpret_cast<const __bf16*>(x), reinterpret_cast<const __bf16*>(w_bf16), bias, cu, qk2,
        reinterpret_cast<__bf16*>(h_out), part_ml, part_ctx, s_out, n_q, drop_p, seed, offset, epoch, plan);
    MPO_LAUNCH_CHECK();
    return 0;
}
