// Row f1 of SURVEY.md section 8(f): the patch layer fused with the forward of MCAT's co-attention (K1), ONE pass
// over the raw patch matrix.
//
// Replaces, for a bf16-stored window at embed_dim 256 (models/mcat/mcat.py:24-29,87,97):
//     H_bag = Dropout(ReLU(X W_H^T + b_H))                  X [rows, 1024], W_H [256, 1024]
//     S = qk H_bag^T, online softmax, ctx = A H_bag        (the folded co-attention of coattn_fwd.hip)
// as a persistent kernel, one workgroup of 8 waves per CU; a workgroup owns a contiguous row range of ONE slide (the
// window's work plan, coattn_tile.h) and walks it in blocks of 128 patch rows:
//
//   main loop, 16 steps of K = 64 per block: H^T[256 x 128] += W_H[256 x 64] X^T[64 x 128] on the bf16 MFMA
//     (v_mfma_f32_16x16x32_bf16, fp32 accumulate; wave (wm, wn) owns 64 embed columns x 64 patch rows = 4 x 4 tiles).
//     The product is taken TRANSPOSED (A operand = W_H, B operand = X) so that a lane ends up with four consecutive
//     embed columns of one patch row: the epilogue packs them into one 8-byte LDS store of a row-major H image.
//     Both operands travel global -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave-instruction, hand-counted
//     s_waitcnt): X through a ring of 5 stages x 16 KiB filled FOUR steps ahead by waves 4-7 (HBM: ~48 KiB must be in
//     flight per CU to cover the latency at full bandwidth), W_H through a ring of 2 stages x 32 KiB filled one step
//     ahead by waves 0-3 (512 KiB re-read per block, out of L2).  Loads and waits of a wave are of ONE kind, because
//     gfx950 retires a wave's memory operations in issue order: a wave that waited for its W_H stage would also wait
//     for every X stage it requested before it.  One s_barrier per step.  Stage images are [row][64 k] with 128-byte
//     rows, 16-byte chunk c of row r stored at c ^ ((r >> 1) & 7) (applied to the GLOBAL chunk a lane fetches; the
//     LDS side of an LDS-DMA is linear): conflict-free ds_read_b128 fragments.
//   epilogue per block: + bias, ReLU, Philox dropout on the accumulators, bf16, into the 128 x 512 B H image (it aliases
//     the W_H ring, idle between blocks; same swizzle as coattn_tile.h so that row_frag / col_frag read it).  Then
//     waves 0-3 run the co-attention tile step of K1 on 32 rows each (scores with the query operand split in THREE
//     bf16 terms, online softmax in log2 units, context accumulation; state in registers for the whole range) while
//     waves 4-7 copy the image out to H_bag in whole rows (needed by the backward pass) -- the bag is never re-read.
//   end of range: the four (max, sum, ctx) states are merged through LDS into one partial per workgroup, combined per
//     slide by coattn_combine_kernel as for K1.
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
constexpr int A_STAGE = BM * BK * 2;            // 16 KiB: X stage
constexpr int B_STAGE = PE * BK * 2;            // 32 KiB: W_H stage
constexpr int A_SLOTS = 5;
constexpr int A_AHEAD = A_SLOTS - 1;            // X stages requested ahead of the one being consumed
constexpr int OFF_A = 0;
constexpr int OFF_B = A_SLOTS * A_STAGE;        // 81 920
constexpr int OFF_IMG = OFF_B;                  // the H image (128 x 512 B = 64 KiB) aliases both W_H slots
constexpr int OFF_Q = OFF_B + 2 * B_STAGE;      // 147 456
constexpr int QCAP = 9;                         // query slots per fragment row: n_q <= 8 live + one zero slot
constexpr int Q_BYTES = 3 * 8 * 4 * QCAP * 16;  // [term][k-step][lane group][slot] x 16 B = 13 824
constexpr int OFF_BIAS = OFF_Q + Q_BYTES;       // 161 280
constexpr int OFF_ML = OFF_BIAS + PE * 4;       // 162 304
constexpr int LDS_TOTAL = OFF_ML + 4 * 128;     // 162 816 <= 163 840
constexpr int IMG_ROWB = PE * 2;                // 512
static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");
static_assert(2 * B_STAGE == BM * IMG_ROWB, "the H image must fit the W_H ring exactly");

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
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;       // (waiting for everything is always safe)
    }
}

// tile step of the co-attention on 32 image rows (coattn_fwd.hip's fwd_tile with the query fragments in LDS, three
// bf16 terms: qk = hi + mid + lo carries all 24 mantissa bits, so a logit of magnitude 100 is good to ~1e-5)
__device__ __forceinline__ void tail_tile(const char* tile, int nvalid, const char* qf, int qslot, float& m_run, float& l_run,
                                          f32x4 (&cacc)[16], float* s_out, bool q_live, int lane) {
    const int g = lane >> 4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const bf16x8 a0 = row_frag<PE>(tile, 0, s, lane);
        const bf16x8 a1 = row_frag<PE>(tile, 1, s, lane);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const bf16x8 q = *reinterpret_cast<const bf16x8*>(qf + ((((t * 8 + s) * 4 + g) * QCAP + qslot) << 4));
            s0 = mfma_bf16(a0, q, s0);
            s1 = mfma_bf16(a1, q, s1);
        }
    }
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
    const float m_new = fmaxf(m_run, mx);                        // finite: row 0 of a processed tile is valid
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
        for (int t = 0; t < 16; ++t) cacc[t] *= alpha;
    }
    bf16x8 ph, pl;
    pack_hi_lo(pv, ph, pl);
    tile_accum_cols<PE, 1>(tile, tile, ph, pl, cacc, lane);
}

__global__ __launch_bounds__(512, 2)
void patch_coattn_fwd_kernel(const __bf16* __restrict__ x,        // [total_rows][1024] patch features
                             const __bf16* __restrict__ wb,       // [256][1024] W_H rounded to bf16
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

    // ---- prologue: bias and the query fragments (three bf16 terms, compact [term][k-step][group][slot]) into LDS
    {
        float* lb = reinterpret_cast<float*>(lds + OFF_BIAS);
        if (tid < PE) lb[tid] = bias[tid];
        const float* qrow = qk2 + (size_t)b * n_q * PE;
        for (int e = tid; e < 8 * 4 * QCAP; e += 512) {
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // (the plain loads above must not sit in front of the rings)
    const unsigned long long offset = epoch_offset(offset_, epoch);
    const size_t pbase = wg.part;

    // Two roles, two code paths with the SAME sequence of workgroup barriers (1 + per block 16 + 3, + 1): the accumulators of
    // one role are never live in the other, so each fits the 256 registers of two waves per SIMD.
    if (wave < 4) {
        // ================================================================ GEMM role: 64 embed columns x all 128 patch rows
        const int g = lane >> 4;
        const int frow = (lane & 15) * (BK * 2);                  // fragment row inside a 16-row tile of a stage image
        const int fswz = (lane >> 1) & 7;                         // its chunk swizzle: ((row >> 1) & 7) depends on the lane only
        const int fc0 = ((0 + g) ^ fswz) << 4, fc1 = ((4 + g) ^ fswz) << 4;
        const uint32_t thr8 = (uint32_t)(drop_p * 256.0f + 0.5f); // keep iff byte >= thr8: realised p = thr8 / 256
        const float inv_keep = drop_p > 0.f ? 256.0f / (256.0f - (float)thr8) : 1.0f;
        wg_barrier();                                             // (bias + query fragments visible)
        int cslot = 0;
        for (int blk = 0; blk < nblocks; ++blk) {
            const int rb = r0 + blk * BM;
            const int rows_here = min(BM, r1 - rb);
            f32x4 acc[4][8];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < KSTEPS; ++k) {
                wg_barrier();                                     // stage (blk, k) of both rings has landed
                const char* xa = lds + OFF_A + cslot * A_STAGE + frow;
                const char* wa = lds + OFF_B + (k & 1) * B_STAGE + (64 * wave) * (BK * 2) + frow;
                cslot = cslot + 1 == A_SLOTS ? 0 : cslot + 1;
                // all 24 fragment reads of the step are issued up front (12 for k 0..31, 12 for k 32..63): the second
                // half's LDS latency hides behind the first half's 32 MFMAs (one GEMM wave per SIMD: nobody else covers it)
                bf16x8 wf0[4], xf0[8], wf1[4], xf1[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) wf0[i] = *reinterpret_cast<const bf16x8*>(wa + i * 16 * (BK * 2) + fc0);
#pragma unroll
                for (int j = 0; j < 8; ++j) xf0[j] = *reinterpret_cast<const bf16x8*>(xa + j * 16 * (BK * 2) + fc0);
#pragma unroll
                for (int i = 0; i < 4; ++i) wf1[i] = *reinterpret_cast<const bf16x8*>(wa + i * 16 * (BK * 2) + fc1);
#pragma unroll
                for (int j = 0; j < 8; ++j) xf1[j] = *reinterpret_cast<const bf16x8*>(xa + j * 16 * (BK * 2) + fc1);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[i][j] = mfma_bf16(wf0[i], xf0[j], acc[i][j]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[i][j] = mfma_bf16(wf1[i], xf1[j], acc[i][j]);
                __builtin_amdgcn_sched_group_barrier(0x100, 24, 0);       // 24 DS reads
                __builtin_amdgcn_sched_group_barrier(0x008, 64, 0);       // 64 MFMAs
            }
            wg_barrier();                                         // every GEMM wave is done with the W_H ring: it becomes the H image
            // epilogue: acc[dt][pt] holds H^T: embed column 64 wave + 16 dt + 4 g + r of patch row 16 pt + (lane & 15)
            // (`el` is the lane id made opaque once per block: the ~50 per-lane addresses below are then recomputed here
            //  instead of being hoisted out of the block loop, where they would sit in registers through the main loop)
            int el = lane;
            asm volatile("" : "+v"(el));
            {
                const int g = el >> 4;
                const float* lb = reinterpret_cast<const float*>(lds + OFF_BIAS) + 64 * wave + 4 * g;
                f32x4 bv[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) bv[dt] = *reinterpret_cast<const f32x4*>(lb + 16 * dt);
#pragma unroll
                for (int pt = 0; pt < 8; ++pt) {
                    const int p = 16 * pt + (el & 15);
                    uint4 rnd = {0u, 0u, 0u, 0u};
                    if (drop_p > 0.f) {                           // the 16 elements of this lane share one Philox draw, 8 bits each
                        const unsigned long long ctr = offset + (unsigned long long)(row_begin + rb + p) * 16ull + (unsigned)(4 * wave + g);
                        rnd = philox4x32((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
                    }
                    const uint32_t rw[4] = {rnd.x, rnd.y, rnd.z, rnd.w};
                    char* rowp = lds + OFF_IMG + p * IMG_ROWB + 8 * (g & 1);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        bf16x4 o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = fmaxf(acc[dt][pt][r] + bv[dt][r], 0.f);
                            if (drop_p > 0.f) v = (((rw[dt] >> (8 * r)) & 0xFFu) >= thr8) ? v * inv_keep : 0.f;
                            o[r] = (__bf16)v;
                        }
                        const int c = 8 * wave + 2 * dt + (g >> 1);
                        *reinterpret_cast<bf16x4*>(rowp + ((c ^ ((p & 7) << 1)) << 4)) = o;
                    }
                }
            }
            wg_barrier();                                         // the image is complete
            {   // whole rows of the image -> H_bag (the backward pass reads it): a wave-instruction stores two rows (1 KiB)
                char* hrow = reinterpret_cast<char*>(h_out) + (size_t)(row_begin + rb) * IMG_ROWB;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = kTileRows * wave + 2 * i + (el >> 5), ch = el & 31;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(lds + OFF_IMG + row * IMG_ROWB + ((ch ^ ((row & 7) << 1)) << 4));
                    if (row < rows_here) *reinterpret_cast<f32x4*>(hrow + (size_t)row * IMG_ROWB + (ch << 4)) = v;
                }
            }
            wg_barrier();                                         // the image has been consumed: the ring is free again
        }
        wg_barrier();                                             // (the helpers' states are in LDS)
    } else {
        // ================================================================ helper role: the two rings + the co-attention
        const int h = wave - 4;                                   // h = 0, 1: X ring; h = 2, 3: W_H ring; all four: 32 image rows each
        const bool x_loader = h < 2;
        const char* xs = reinterpret_cast<const char*>(x) + (size_t)row_begin * (PK * 2);     // this slide's rows
        const char* wbytes = reinterpret_cast<const char*>(wb);
        const unsigned lds0 = lds_addr(lds);
        const int lrow = lane >> 3;                               // row of this lane inside an 8-row x 128-B piece
        // X stage n (block n / 16, k-step n % 16) -> ring slot n % 5: wave h requests pieces 8 h .. 8 h + 7
        auto issue_x = [&](int n, int slot) {
            const int rb = r0 + (n >> 4) * BM, kb = (n & 15) * (BK * 2);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int piece = 8 * h + t;
                const int row = 8 * piece + lrow;
                const int c = (lane & 7) ^ ((row >> 1) & 7);
                const int grow = min(rb + row, m_rows - 1);       // rows past the slide: clamped (finite; masked later)
                glds16(xs + (size_t)grow * (PK * 2) + kb + (c << 4), lds0 + OFF_A + slot * A_STAGE + piece * 1024);
            }
        };
        // W_H stage k -> slot k & 1: wave h requests pieces 16 (h - 2) .. + 15
        auto issue_w = [&](int k) {
            const int kb = k * (BK * 2);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int piece = 16 * (h - 2) + t;
                const int row = 8 * piece + lrow;
                const int c = (lane & 7) ^ ((row >> 1) & 7);
                glds16(wbytes + (size_t)row * (PK * 2) + kb + (c << 4), lds0 + OFF_B + (k & 1) * B_STAGE + piece * 1024);
            }
        };
        int issued = 0, islot = 0;                                // X stages requested so far, slot of the next one
        if (x_loader) {
            for (; issued < A_AHEAD && issued < n_stages; ++issued) {
                issue_x(issued, islot);
                islot = islot + 1 == A_SLOTS ? 0 : islot + 1;
            }
        } else if (n_stages > 0) {
            issue_w(0);
        }
        float m_run = -INFINITY, l_run = 0.f;
        f32x4 cacc[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) cacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int q = lane & 15, g = lane >> 4;
        const bool q_live = q < n_q;
        const int qslot = q_live ? q : n_q;                       // slot n_q holds zeros
        float* s_row = (s_out != nullptr) ? s_out + (size_t)n_q * row_begin + (size_t)q * m_rows : nullptr;
        wg_barrier();                                             // (bias + query fragments visible)
        for (int blk = 0; blk < nblocks; ++blk) {
            const int rb = r0 + blk * BM;
            const int rows_here = min(BM, r1 - rb);
            for (int k = 0; k < KSTEPS; ++k) {
                const int n = blk * KSTEPS + k;
                // This wave's own requests for stage n have landed.  gfx950 retires a wave's memory operations in issue
                // order, so the count is the number of its operations YOUNGER than stage n (a count above their true
                // number would let stage n itself slip through): the up to three X stages requested after it, 8 loads each.
                if (x_loader) wait_vm(8 * min(A_AHEAD - 1, n_stages - 1 - n));
                else wait_vm(0);
                wg_barrier();                                     // ... and everybody's; stage n - 1 has been read by all
                if (x_loader) {
                    if (issued < n_stages) {
                        issue_x(issued, islot);
                        islot = islot + 1 == A_SLOTS ? 0 : islot + 1;
                        ++issued;
                    }
                } else if (k + 1 < KSTEPS) {
                    issue_w(k + 1);
                }
            }
            wg_barrier();                                         // (the GEMM waves leave the W_H ring)
            wg_barrier();                                         // the image is complete
            const int nvalid = min(kTileRows, rows_here - kTileRows * h);
            if (nvalid > 0)
                tail_tile(lds + OFF_IMG + kTileRows * h * IMG_ROWB, nvalid, lds + OFF_Q, qslot, m_run, l_run, cacc,
                          s_row ? s_row + rb + kTileRows * h : nullptr, q_live, lane);
            wg_barrier();                                         // the image has been consumed: the ring is free again
            if (!x_loader && blk + 1 < nblocks) issue_w(0);
        }
        // the four co-attention states of this workgroup -> LDS (the image region: 4 x 16 KiB)
        float l_tot = l_run + __shfl_xor(l_run, 16);
        l_tot += __shfl_xor(l_tot, 32);
        float* wctx = reinterpret_cast<float*>(lds + OFF_IMG + h * 16384);          // [16][256] floats
        float* wml = reinterpret_cast<float*>(lds + OFF_ML) + h * 32;
#pragma unroll
        for (int t = 0; t < 16; ++t) *reinterpret_cast<f32x4*>(wctx + q * PE + 16 * t + 4 * g) = cacc[t];
        if (g == 0) {
            wml[2 * q] = m_run;
            wml[2 * q + 1] = l_tot;
        }
        wg_barrier();
    }

    // ---- merge into ONE partial per workgroup (all 512 threads)
    const float* ml = reinterpret_cast<const float*>(lds + OFF_ML);
    for (int idx = tid; idx < n_q * PE; idx += 512) {
        const int qq = idx / PE;
        float mt = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) mt = fmaxf(mt, ml[w * 32 + 2 * qq]);
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float mw = ml[w * 32 + 2 * qq];
            const float wgt = mw == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mw - mt);
            a += wgt * reinterpret_cast<const float*>(lds + OFF_IMG + w * 16384)[idx];
        }
        part_ctx[pbase * n_q * PE + idx] = a;
    }
    if (tid < 16) {
        const int qq = tid;
        float mt = -INFINITY, lt = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) mt = fmaxf(mt, ml[w * 32 + 2 * qq]);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float mw = ml[w * 32 + 2 * qq];
            lt += (mw == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mw - mt)) * ml[w * 32 + 2 * qq + 1];
        }
        part_ml[pbase * 32 + 2 * qq] = mt;
        part_ml[pbase * 32 + 2 * qq + 1] = lt;
    }
}

__global__ void cast_f32_bf16_kernel(const f32x4* __restrict__ in, bf16x4* __restrict__ out, size_t n4) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 v = in[i];
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (__bf16)v[j];
    out[i] = o;
}

}  // namespace

int mpo_launch_cast_bf16(const float* in, void* out, size_t n, hipStream_t stream) {
    MPO_CHECK(n % 4 == 0, "cast: element count %zu is not a multiple of 4", n);
    const size_t n4 = n / 4;
    cast_f32_bf16_kernel<<<(unsigned)((n4 + 255) / 256), 256, 0, stream>>>(reinterpret_cast<const f32x4*>(in),
                                                                           reinterpret_cast<bf16x4*>(out), n4);
    MPO_LAUNCH_CHECK();
    return 0;
}

int mpo_launch_patch_coattn_fwd(const void* x, const void* w_bf16, const float* bias, const int* cu, const float* qk2,
                                void* h_out, float* part_ml, float* part_ctx, float* s_out, int n_q, float drop_p,
                                unsigned long long seed, unsigned long long offset, const unsigned long long* epoch,
                                const BagPlan& plan, hipStream_t stream) {
    MPO_CHECK(n_q >= 1 && n_q <= QCAP - 1, "fused patch layer + co-attention: 1..%d queries (got %d)", QCAP - 1, n_q);
    MPO_CHECK(drop_p >= 0.f && drop_p < 1.f, "patch-layer dropout p must be in [0,1) (got %f)", (double)drop_p);
    patch_coattn_fwd_kernel<<<plan_grid(plan), 512, 0, stream>>>(
        reinterpret_cast<const __bf16*>(x), reinterpret_cast<const __bf16*>(w_bf16), bias, cu, qk2,
        reinterpret_cast<__bf16*>(h_out), part_ml, part_ctx, s_out, n_q, drop_p, seed, offset, epoch, plan);
    MPO_LAUNCH_CHECK();
    return 0;
}
