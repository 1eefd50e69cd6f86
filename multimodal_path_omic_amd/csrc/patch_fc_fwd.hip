// The patch layer of a bf16-stored window, H_bag = Dropout(ReLU(X W_H^T + b_H))   (models/mcat/mcat.py:24-29,87;
// models/nacagat/nacagat.py the same layer), X [rows, 1024] bf16, W_H [256, 1024], H_bag [rows, 256] bf16: ONE pass over the
// raw patch matrix, H_bag written once.  Row H2 / f1 of SURVEY.md section 8.
//
// r02-r03 ran this layer fused with K1's forward (8 GEMM waves on 128-row blocks + 4 loader / co-attention waves: 335-349 us
// per 32 x 15 000-row window); r04 rebuilt that as one output-stationary 256 x 256 block per workgroup with the co-attention
// in the block epilogue (358-361 us) and counted cycles inside it: the main loop runs the matrix pipes at ~1 PF/s, which is
// what the chip sustains under this mix of matrix and memory work, and EVERY cycle of epilogue stands on top of it --
// bias / ReLU / dropout / conversion 7-9 k cycles per block, the H_bag stores 9.4 k (issue-bound, ~14 B/clk per CU), the fused
// co-attention tile steps + merge + their barriers 14 k (~70 us per launch: more than K1's own forward kernel takes, 46 us,
// because they run on SIMDs whose matrix pipes then idle), the ring refill 2-6 k -- against 45 k of main loop (NOTES.md r04-f1).
// So the co-attention is K1's forward launch again (coattn_fwd.hip) and this kernel hides its epilogue under matrix work:
//
//   * persistent, one workgroup of 8 waves per CU = TWO streams of four waves (one wave of each stream per SIMD).  A
//     stream walks 128-row CHUNKS of the workgroup's row range (alternate chunks) with a 128 x 256 fp32 accumulator block in
//     registers (wave cq = embed columns 64 cq .. + 63: 4 x 8 tiles of v_mfma_f32_16x16x32_bf16, 128 registers) over 32
//     K-stages of 32, then spends 8 stages on its epilogue: one patch tile (16 rows) per stage -- bias, ReLU, dropout
//     (counter hash, one draw per 16 elements), bf16, two 16-byte stores straight to H_bag (the W_H tiles are packed with
//     their rows permuted so that a lane holds two runs of 8 consecutive embed columns and the four lane groups of a row
//     store 64 contiguous bytes).  The second stream runs 20 stages behind the first, so on every SIMD a wave in its
//     epilogue (vector ALU + stores) sits beside a wave in its main loop (matrix pipe) -- 16 of the 40 stages of a period --
//     and the H_bag stores trickle out at 8 KiB per stage instead of 128 KiB per block.
//   * both streams consume the SAME W_H stage at the same time: the W_H stream runs on through the K index cyclically
//     (k = (stage + k0) mod 32) and a chunk simply starts at whatever k the stream is at -- its K-steps are taken in ROTATED
//     order.  The rotation of a chunk is a pure function of its index inside the slide, g(c) = 8 (c >> 1) + 20 (c & 1) mod 32
//     (period 40 = 8 mod 32 between the chunks of a stream, skew 20 between the streams; k0 = g(first chunk)): it does not
//     depend on how the window is cut into workgroups, so window == slide by slide holds bit for bit.  Workgroup row ranges
//     are multiples of 128 rows to that end (the bag plan's range rounded up inside the kernel).
//   * operands travel global -> LDS directly (global_load_lds_dwordx4, hand-counted s_waitcnt): W_H through a ring of three
//     16-KiB stages (a straight copy of the fragment-ordered packed weight, pack_patch_weight_kernel; two in flight: L2 hits),
//     each stream's X through its own ring of six 8-KiB stages ([128 rows][64 B], chunk-swizzled on the global side so that
//     the ds_read_b128 fragment reads are conflict-free; five in flight, across chunk boundaries and epilogues: HBM latency).
//     Every wave requests two pieces of each W_H stage and two of each X stage of its own stream; the H_bag stores are
//     buffer stores whose range ends at the workgroup's last row (out-of-range lanes are dropped by the hardware: every
//     wave issues the same number of vector-memory instructions per stage, which is what the hand-counted waits rely on).
//     One workgroup barrier per stage; the fragment reads of stage n + 1 are issued in the shadow of stage n's 32 MFMAs.
//
// Roofline: HBM.  Algorithmic bytes per patch row: 2048 read + 512 written; 524 288 flop per row put the MFMA floor at
// about half the HBM floor at the dense peak -- and above it at the ~1 PF/s the chip holds under load.
#include <type_traits>

#include "coattn_tile.h"
#include "mpo_kernels.h"

namespace {

constexpr int PE = 256;                         // embed_dim
constexpr int PK = 1024;                        // patch feature width (models/mcat/mcat.py:25)
constexpr int CH = 128;                         // patch rows per chunk (one stream's accumulator block)
constexpr int BK = 32;                          // k per stage (one MFMA k-step)
constexpr int NK = PK / BK;                     // 32 main stages per chunk
constexpr int EPI = 8;                          // epilogue stages per chunk: one 16-row patch tile each
constexpr int PERIOD = NK + EPI;                // 40
constexpr int SKEW = 20;                        // the second stream's delay: 2 SKEW = EPI (mod 32), see the rotation rule above
constexpr int X_IMG = CH * 2 * BK * 2;          // 16 KiB: [128 rows][128 B] = TWO k-steps (whole 128-byte lines of X)
constexpr int W_IMG = PE * BK * 2;              // 16 KiB: [cq 4][dt 4][lane 64][16 B]
#ifdef MPO_PF_W4
constexpr int XSLOTS = 3, WSLOTS = 4;           // X: the pair being read + two pairs (four stages) in flight per stream; W_H: THREE stages in flight (160 KiB)
#else
constexpr int XSLOTS = 3, WSLOTS = 3;           // X: the pair being read + two pairs (four stages) in flight per stream; W_H: two stages in flight
#endif
constexpr int WAHEAD = WSLOTS - 1;
constexpr int OFF_X = 0;                        // stream s: OFF_X + s * XSLOTS * X_IMG
constexpr int OFF_W = 2 * XSLOTS * X_IMG;       // 96 KiB
constexpr int LDS_TOTAL = OFF_W + WSLOTS * W_IMG;   // 144 KiB
constexpr int NTHREADS = 512;
static_assert((2 * SKEW) % NK == EPI % NK && SKEW >= EPI && SKEW % 2 == 0 && EPI % 2 == 0, "rotation rule / disjoint epilogues / even rotations");
static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");

// ---- the static schedule of a stream (position p of its 40-stage period; all counts are vector-memory instructions of ONE wave)
// Order inside a stage, right behind its barrier: the W_H request (2: stage + 2), the X request (4: a 16-KiB pair of k-steps, at
// even positions, for the pair two pairs ahead -- positions 28..34 have none: the chunk's 16 pairs are out, the next chunk's
// first two pairs go out at 36 and 38); then the stage's work, which in the epilogue ends with two H_bag stores.  (Requests
// behind the work were measured: the memory pipeline then idles for the length of every stage's work -- DMA alone 223 us,
// matrix work alone 138 us, together 329 us.)
constexpr int pmod(int p) { return ((p % PERIOD) + PERIOD) % PERIOD; }
constexpr int x_ops(int p) { return (pmod(p) % 2 == 0 && (pmod(p) <= NK - 6 || pmod(p) >= PERIOD - 4)) ? 4 : 0; }
constexpr int s_ops(int p) { return pmod(p) >= NK ? 2 : 0; }
// instructions younger than this wave's W_H pieces of the stage it is about to read (requested first thing two stages before):
// the rest of that stage, and everything of the stage in between.  (The X pair of the moment was requested earlier still.)
constexpr int n_younger(int p) {
    int n = 0;
    for (int d = 1; d <= WAHEAD; ++d) n += x_ops(p - d) + s_ops(p - d) + (d < WAHEAD ? 2 : 0);
    return n;
}
// the same where there is no epilogue behind the stream yet (its first chunk) and in stages without work of its own: only what
// is certain to be younger -- the W_H requests of the stages in between (waiting for more than necessary is safe, for less is not)
constexpr int N_W_ONLY = 2 * (WAHEAD - 1);
// first chunk, positions 1 and 2: the W_H requests in between and the X pair requested at position 0
constexpr int N_FIRST_1 = WAHEAD == 2 ? 6 : 8, N_FIRST_2 = WAHEAD == 2 ? 6 : 8;

__device__ __forceinline__ unsigned lds_addr(const char* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
// one LDS-DMA wave-instruction: lane l's 16 bytes at base + off land at lds_dst + 16 l (base, lds_dst wave-uniform)
__device__ __forceinline__ void glds16(const char* base, unsigned off, unsigned lds_dst) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"       // m0 is "reserved": nothing else in this kernel uses it
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(off), "s"(base), "s"(lds_dst) : "memory", "m0");
#pragma clang diagnostic pop
}
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// wait until at most N of this wave's vector-memory operations are outstanding
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 2 && N <= 20 && N % 2 == 0, "see n_younger()");      // (the counter holds 63)
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}
__device__ __forceinline__ int wrap_inc(int v, int n) { return v + 1 == n ? 0 : v + 1; }
// The 16 mask bytes of one (row, 16-column group) of the patch layer's dropout: a counter hash like hash4x32 (mpo_common.h)
// with fewer multiplies -- 32-bit integer multiplies run at a quarter of the vector rate, and this runs beside a wave that
// streams MFMAs (which leave the vector ALU half of its issue slots): one murmur3 finaliser of (key ^ counter), then per word
// one multiply and a fold of the product's high half into its low one.  The mask lives in H_bag as zeros: nothing regenerates it.
__device__ __forceinline__ uint4 hash16(uint32_t key, uint32_t inc, uint32_t ctr) {
    const uint32_t x = fmix32(key ^ ctr);
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t t = (x + (uint32_t)(i + 1) * inc) * 0x9E3779B1u;
        w[i] = t ^ (t >> 15);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}
// K rotation of chunk c of a slide (see the header)
__device__ __forceinline__ int chunk_rotation(int c) { return (8 * (c >> 1) + SKEW * (c & 1)) & (NK - 1); }

#ifdef MPO_PF_STAMPS
__device__ float mpo_pf_stamps[1024 * 16];     // per workgroup: wave 0 -> [0..7], wave 4 -> [8..15]
#endif

// WIDTH = embed_dim of the model (models/mcat/mcat.py:16-21: 128 'small', 256, 512 'big'), the row pitch of H_bag.  The pass
// always works on a 256-column block of W_H: WIDTH 128 = the block's upper half is zero rows of the packed weight and the
// waves that own it (cq 2, 3) aim their stores past the buffer's range; WIDTH 512 = two passes, columns col_off = 0 and 256.
template <int WIDTH>
__global__ __launch_bounds__(NTHREADS, 2)
void patch_fc_fwd_kernel(const __bf16* __restrict__ x,            // [total_rows][1024] patch features
                         const __bf16* __restrict__ wb,           // the pass's 256 rows of W_H as bf16, packed in stage order (pack_patch_weight_kernel)
                         const float* __restrict__ bias,          // [WIDTH]
                         const int* __restrict__ cu,
                         __bf16* __restrict__ h_out,              // [total_rows][WIDTH]
                         int col_off,                             // first embed column of this pass
                         float drop_p, unsigned long long seed, unsigned long long offset_,
                         const unsigned long long* __restrict__ epoch, BagPlan plan) {
    constexpr int NCQ = WIDTH < PE ? WIDTH / 64 : 4;              // waves of a stream whose 64 columns exist
    __shared__ __attribute__((aligned(1024))) char lds[LDS_TOTAL];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- this workgroup's row range: the plan's, with the range length rounded up to whole chunks
    int b, split, rps;
    if (plan.wg_start != nullptr) {
        const int wg = blockIdx.x;
        int lo = 0, hi = plan.n_slides;                           // wg_start[lo] <= wg < wg_start[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (plan.wg_start[mid] <= wg) lo = mid; else hi = mid;
        }
        b = lo;
        split = wg - plan.wg_start[lo];
        rps = plan.rows_per_wg;
    } else {
        b = blockIdx.y;
        split = blockIdx.x;
        rps = 0;
    }
    const int row_begin = cu[b], m_rows = cu[b + 1] - row_begin;
    if (plan.wg_start == nullptr) rps = (m_rows + plan.splits - 1) / plan.splits;
    rps = (rps + CH - 1) / CH * CH;
    const int r0 = split * rps, r1 = min(m_rows, r0 + rps);
    if (r1 <= r0) return;                                         // (rounding the ranges up can leave a slide's last workgroups without rows)
    const int n_ch = (r1 - r0 + CH - 1) / CH;                     // chunks r0 / CH .. : the streams take alternate ones
    const int st = wave >> 2, cq = wave & 3;                      // stream; embed columns 64 cq .. + 63
    const int n_mine = (n_ch - st + 1) >> 1;                      // chunks of this wave's stream
    const int k0 = chunk_rotation(r0 / CH);
    const int n_lead = (n_ch + 1) >> 1, n_lag = n_ch >> 1;
    const int G = max(n_lead * PERIOD, n_lag > 0 ? n_lag * PERIOD + SKEW : 0);      // global stages

    // this lane's 16 embed columns (64 cq + 32 (dt >> 1) + 8 g + 4 (dt & 1) + r): bias / keep, kept in registers
    f32x4 bk[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
        bk[dt] = (wave & 3) < NCQ
            ? *reinterpret_cast<const f32x4*>(bias + col_off + 64 * (wave & 3) + 32 * (dt >> 1) + 8 * (lane >> 4) + 4 * (dt & 1))
            : f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // (the plain loads above must not sit in front of the rings)
    const unsigned long long offset = epoch_offset(offset_, epoch);
    const uint32_t drop_key = hash_stream_key(seed, offset);      // (the stream offset is in the key: the counter below is the element group alone)
    const uint32_t thr8 = (uint32_t)(drop_p * 256.0f + 0.5f);     // keep iff byte >= thr8: realised p = thr8 / 256
    const float inv_keep = drop_p > 0.f ? 256.0f / (256.0f - (float)thr8) : 1.0f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) bk[dt] *= inv_keep;
    const uint32_t drop_inc = hash_word_stride(drop_key);

    const char* xs = reinterpret_cast<const char*>(x) + (size_t)row_begin * (PK * 2);     // this slide's rows
    const char* wpk = reinterpret_cast<const char*>(wb);
    const unsigned lds0 = lds_addr(lds);
    // H_bag of this slide as a buffer whose range ends at the workgroup's last row: stores to rows >= r1 are dropped
    // (a pass over columns col_off .. + 255 of a wider H_bag: base moved by col_off, range shortened by as much)
    const __amdgpu_buffer_rsrc_t hbuf = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(h_out) + (size_t)row_begin * (WIDTH * 2) + col_off * 2, 0,
        (int)((unsigned)r1 * (WIDTH * 2) - (unsigned)col_off * 2), 0x00020000);

    int gs = 0;                                                   // global stage
    // X pair q (main stages 2 q, 2 q + 1) of chunk j of this stream: rows r0 + CH (2 j + st) .. + 127 x 128 B at k-steps
    // (j PERIOD + 2 q + st SKEW + k0) mod 32 and the next (rotations are even: the pair is one aligned 128-byte line per row).
    // Piece t = rows 8 t .. + 7 x 128 B: lane l fetches the 16-byte chunk that belongs at position l & 7 of row 8 t + (l >> 3) --
    // chunk c of row r is stored at position c ^ ((r >> 1) & 7): conflict-free ds_read_b128 fragment reads (lane & 15 = row,
    // lane >> 4 = chunk of a k-step half).  This wave: pieces 4 cq .. 4 cq + 3.
    auto issue_x = [&](int j, int q, int slot) {
        int el = lane;                                            // opaque per call: the per-piece address parts are
        asm volatile("" : "+v"(el));                              // recomputed here, not hoisted out of the loop and kept in registers
        const int rb = r0 + CH * (2 * j + st);
        const int k = (j * PERIOD + 2 * q + st * SKEW + k0) & (NK - 1);
        const char* xblk = xs + (size_t)rb * (PK * 2);
        const int lim = m_rows - 1 - rb;                          // >= 0: rows past the slide are clamped (finite; never stored)
        const unsigned dst0 = lds0 + OFF_X + (st * XSLOTS + slot) * X_IMG + cq * 4096;
        const int lrow = el >> 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 32 * cq + 8 * i + lrow;
            const unsigned c16 = (unsigned)(((el & 7) ^ ((row >> 1) & 7)) << 4) + (unsigned)k * (BK * 2);
            glds16(xblk, (unsigned)min(row, lim) * (PK * 2) + c16, dst0 + i * 1024);
        }
    };
    // W_H stage g: 16 KiB of the packed weight at k-step (g + k0) mod 32 as it stands.  This wave: pieces 2 wave, 2 wave + 1.
    auto issue_w = [&](int g, int slot) {
        int el = lane;
        asm volatile("" : "+v"(el));
        const int k = (g + k0) & (NK - 1);
        const unsigned dst0 = lds0 + OFF_W + slot * W_IMG + wave * 2048;
        const unsigned o = (unsigned)k * W_IMG + (unsigned)wave * 2048 + (unsigned)el * 16;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(wpk, o + i * 1024, dst0 + i * 1024);
    };

    f32x4 acc[4][8];
    // fragment reads of one stage: 4 W_H fragments (embed tiles 4 cq + dt) and 8 X fragments (patch tiles pt of the chunk)
    const int xrd = OFF_X + st * XSLOTS * X_IMG + (lane & 15) * (2 * BK * 2);          // + the swizzled chunk of the k-step half, below
    const int xsw = (lane >> 1) & 7;                              // ((row >> 1) & 7) depends on the lane only: tiles are 16 rows apart
    const int xc0 = (((lane >> 4) ^ xsw) << 4), xc1 = (((4 + (lane >> 4)) ^ xsw) << 4);
    const int wrd = OFF_W + cq * 4096 + lane * 16;
    // One stage of a stream's main loop for this wave: the 32 MFMAs of the stage whose fragments are in registers (cw, fx) with
    // the fragment reads of the NEXT stage (already landed in the rings) issued in their shadow -- its W_H fragments into the
    // other W_H register set, each X fragment into the registers its predecessor has just left (the MFMAs walk the patch
    // tiles in order: one X register set suffices).  FIRST: reads only; LAST: MFMAs only.
    auto stage = [&](auto first_tag, auto last_tag, auto zero_tag, int xslot, int khalf, int wslot, bf16x8 (&lw)[4], const bf16x8 (&cw)[4], bf16x8 (&fx)[8]) {
        constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value, ZEROC = decltype(zero_tag)::value;
        const char* wst = lds + wrd + wslot * W_IMG;
        const char* xst = lds + xrd + xslot * X_IMG + (khalf ? xc1 : xc0);
        // The MFMAs are inline asm with the accumulator tied to the destination: left to the register allocator the
        // accumulators wander (a result is written over the dying X fragment, the old accumulator's registers take the
        // next fragment read), the rotation fragments the register file and one accumulator tile ends up in scratch --
        // whose store sits in the vector-memory queue the hand-counted waits below count.  Program order is the
        // schedule: every group is closed by a scheduling barrier.
#pragma unroll
        for (int pt = 0; pt < 8; ++pt) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                if constexpr (!FIRST) {
                    if constexpr (ZEROC)                          // a chunk's first k-step starts the accumulators (no clearing pass)
                        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(acc[dt][pt]) : "v"(cw[dt]), "v"(fx[pt]));
                    else
                        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[dt][pt]) : "v"(cw[dt]), "v"(fx[pt]));
                }
                if constexpr (!LAST) {
                    if (pt == 0) {                                // the next stage's W_H fragments among the first MFMAs
                        lw[dt] = *reinterpret_cast<const bf16x8*>(wst + dt * 1024);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if constexpr (!LAST) fx[pt] = *reinterpret_cast<const bf16x8*>(xst + pt * 2048);      // into the registers the tile's fragment has just left
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (LAST) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // (the compiler does not know these were matrix instructions: their results are read by vector instructions next)
    };
    // Epilogue stage PT of chunk j: acc[dt][PT][r] = H^T: embed column 64 cq + 32 (dt >> 1) + 8 g + 4 (dt & 1) + r of chunk row
    // 16 PT + (lane & 15) -> + bias, ReLU, dropout, bf16 -> H_bag; the accumulators of the tile are cleared for the next chunk.
    auto convert = [&](auto pt_tag, int j) {
        constexpr int PT = decltype(pt_tag)::value;
        int el = lane;
        asm volatile("" : "+v"(el));
        const int eg = el >> 4, ei = el & 15;
        const int row = r0 + CH * (2 * j + st) + 16 * PT + ei;    // slide-relative
        uint32_t rw[4] = {0u, 0u, 0u, 0u};
        if (drop_p > 0.f) {                                       // one draw = the 16 bytes of this lane's 16 elements of the row
            const uint4 rnd = hash16(drop_key, drop_inc, (uint32_t)(row_begin + row) * (uint32_t)(WIDTH / 16) + (uint32_t)(col_off / 16 + 4 * cq + eg));
            rw[0] = rnd.x; rw[1] = rnd.y; rw[2] = rnd.z; rw[3] = rnd.w;
        }
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        u32x4 o[2];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            // (acc + b) / keep as one packed fma per two elements (bk = b / keep), ReLU, the keep bit of the element's byte
            // (no dropout: threshold 0 keeps every byte), one conversion per two elements
            const f32x2 lo = f32x2{acc[dt][PT][0], acc[dt][PT][1]} * f32x2{inv_keep, inv_keep} + f32x2{bk[dt][0], bk[dt][1]};
            const f32x2 hi = f32x2{acc[dt][PT][2], acc[dt][PT][3]} * f32x2{inv_keep, inv_keep} + f32x2{bk[dt][2], bk[dt][3]};
            float e[4] = {lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
            for (int r = 0; r < 4; ++r) e[r] = (((rw[dt] >> (8 * r)) & 0xFFu) >= thr8) ? __builtin_elementwise_maximum(e[r], 0.f) : 0.f;      // (v_maximum3_f32: a NaN stays a NaN, as in torch.relu)
            o[dt >> 1][2 * (dt & 1)] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{e[0], e[1]}, bf16x2));
            o[dt >> 1][2 * (dt & 1) + 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{e[2], e[3]}, bf16x2));
        }
        const unsigned voff = cq < NCQ ? (unsigned)row * (WIDTH * 2) + (unsigned)((8 * cq + eg) << 4)
                                       : 0xFFFFFF00u;            // (columns that do not exist: out of the buffer's range, dropped -- the store still counts)
        __builtin_amdgcn_raw_buffer_store_b128(o[0], hbuf, voff, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(o[1], hbuf, voff + 64, 0, 0);
    };

    // ---- initial fill: this stream's first two X pairs, then W_H stages 0 and 1
    int xw = 0, xr = 0;                                           // this stream's X ring: slot requested next / read next
    if (n_mine > 0) {
        issue_x(0, 0, 0);
        issue_x(0, 1, 1);
        xw = 2;
    }
#pragma unroll
    for (int i = 0; i < WAHEAD; ++i) issue_w(i, i);
    int ww = WAHEAD % WSLOTS, wr = 0;                             // W_H ring: slot requested next / read next
    // Every wave goes through the same G stages (one workgroup barrier each); what it does between the barrier and its requests
    // depends on where its stream is.  pre<N>(): the stage has landed.  Own pieces: gfx950 retires a wave's vector-memory
    // operations in issue order, N = the number of this wave's operations younger than its W_H pieces of this stage (static:
    // n_younger()); everybody's: the barrier, after which the slots read in the previous stage may be overwritten.
    // post(): the requests of this stage -- always the W_H stage two ahead (past the end it is fetched for nobody: the counts
    // stay static), at even positions this stream's X pair two pairs ahead (past the stream's last chunk: that chunk again).
#ifdef MPO_PF_STAMPS
    unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = __builtin_amdgcn_s_memtime();
    int st_kind = 3;                                              // what the stage that ends now was: 0 main, 1 epilogue, 2 first / last, 3 idle
#define MPO_KIND(k) st_kind = k;
#else
#define MPO_KIND(k)
#endif
    auto pre = [&](auto n_tag) {
        __builtin_amdgcn_sched_barrier(0);                        // (nothing of one stage is scheduled into another)
#ifdef MPO_PF_STAMPS
        { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_sum[st_kind] += t_ - st_t; st_t = t_; }
#endif
        wait_vm<decltype(n_tag)::value>();
#ifdef MPO_PF_STAMPS
        { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_sum[4] += t_ - st_t; st_t = t_; }
#endif
        wg_barrier();
#ifdef MPO_PF_STAMPS
        { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_sum[5] += t_ - st_t; st_t = t_; }
#endif
        __builtin_amdgcn_sched_barrier(0);
    };
    auto req_w = [&]() {                                          // right behind the barrier
        issue_w(gs + WAHEAD, ww);
        ww = wrap_inc(ww, WSLOTS);
    };
    auto req_x = [&](int j, int q) {                              // (after req_w: the order the counts assume)
        issue_x(min(j, n_mine - 1), q, xw);
        xw = wrap_inc(xw, XSLOTS);
    };
    auto adv = [&]() {
        wr = wrap_inc(wr, WSLOTS);
        ++gs;
    };
    using std::integral_constant;
    using T = integral_constant<bool, true>;
    using F = integral_constant<bool, false>;
#define MPO_N(p) integral_constant<int, n_younger(p)>{}
    using NW = integral_constant<int, N_W_ONLY>;
    using NF1 = integral_constant<int, N_FIRST_1>;
    using NF2 = integral_constant<int, N_FIRST_2>;

    for (int i = 0; i < st * SKEW && gs < G; ++i) {               // the lagging stream's head start for the other one
        pre(NW{});
        MPO_KIND(3)
        req_w();
        adv();
    }
    for (int cj = 0; cj < n_mine; ++cj) {
        bf16x8 pw[4], qw[4], fx[8];
        // position 0: reads only
        if (cj == 0) pre(NW{}); else pre(MPO_N(0));               // (first chunk: nothing but the W_H stages in between is younger)
        MPO_KIND(2)
        req_w();
        req_x(cj, 2);
        stage(T{}, F{}, F{}, xr, 0, wr, pw, qw, fx);
        adv();
        static_assert(n_younger(3) == n_younger(5) && n_younger(3) == n_younger(25) && n_younger(4) == n_younger(6) &&
                      n_younger(4) == n_younger(26), "main loop counts");
        static_assert(WAHEAD == 2 ? n_younger(2) == N_FIRST_2 : n_younger(2) == N_FIRST_2 + 2, "position 2: the epilogue's last stores are in the window at three ahead");
        if (cj == 0) pre(NF1{}); else pre(MPO_N(1));              // (first chunk: no epilogue stores before it)
        MPO_KIND(0)
        req_w();
        stage(F{}, F{}, T{}, xr, 1, wr, qw, pw, fx);              // position 1: the chunk's first MFMAs (accumulators start from zero)
        xr = wrap_inc(xr, XSLOTS);
        adv();
        if (cj == 0) pre(NF2{}); else pre(MPO_N(2));
        req_w();
        req_x(cj, 3);
        stage(F{}, F{}, F{}, xr, 0, wr, pw, qw, fx);
        adv();
        for (int q = 1; q < 13; ++q) {                            // positions 3 .. 26 in pairs (odd, even); even positions request pair q + 3
            pre(MPO_N(3));
            req_w();
            stage(F{}, F{}, F{}, xr, 1, wr, qw, pw, fx);
            xr = wrap_inc(xr, XSLOTS);
            adv();
            pre(MPO_N(4));
            req_w();
            req_x(cj, q + 3);
            stage(F{}, F{}, F{}, xr, 0, wr, pw, qw, fx);
            adv();
        }
#define MPO_MAIN_STAGE(P, LOADSET, USESET) pre(MPO_N(P)); req_w(); stage(F{}, F{}, F{}, xr, (P) & 1, wr, LOADSET, USESET, fx); \
        if ((P) & 1) xr = wrap_inc(xr, XSLOTS); adv();
        MPO_MAIN_STAGE(27, qw, pw)
        MPO_MAIN_STAGE(28, pw, qw)                                // (the chunk's last pair, 15, was requested at 26)
        MPO_MAIN_STAGE(29, qw, pw)
        MPO_MAIN_STAGE(30, pw, qw)
        MPO_MAIN_STAGE(31, qw, pw)
#undef MPO_MAIN_STAGE
        // epilogue: position 32 finishes the MFMAs (stage 31 was read into the second W_H set), then one patch tile per stage;
        // positions 36 and 38 request the next chunk's first two pairs
        pre(MPO_N(32));
        MPO_KIND(2)
        req_w();
        stage(F{}, T{}, F{}, 0, 0, 0, pw, qw, fx);
        __builtin_amdgcn_sched_barrier(0);                        // (the tile's draw and conversions after the last MFMAs, not among them: registers)
        convert(integral_constant<int, 0>{}, cj);
        adv();
#define MPO_EPI_STAGE(PT, XREQ) pre(MPO_N(NK + PT)); MPO_KIND(1) req_w(); XREQ convert(integral_constant<int, PT>{}, cj); adv();
        MPO_EPI_STAGE(1, ) MPO_EPI_STAGE(2, ) MPO_EPI_STAGE(3, )
        MPO_EPI_STAGE(4, req_x(cj + 1, 0);)
        MPO_EPI_STAGE(5, )
        MPO_EPI_STAGE(6, req_x(cj + 1, 1);)
        MPO_EPI_STAGE(7, )
#undef MPO_EPI_STAGE
    }
    if (n_mine > 0 && gs < G) {                                   // the other stream is still at work: stage "40" of the last chunk, ..
        pre(MPO_N(0));
        MPO_KIND(3)
        req_w();
        adv();
        if (gs < G) {
            pre(integral_constant<int, 4>{});                     // (.. "41": at least the last epilogue stage's stores and the W_H request of "40")
            req_w();
            adv();
        }
    }
    while (gs < G) {
        pre(NW{});
        req_w();
        adv();
    }
#undef MPO_N
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // (requests for nobody are still landing in this workgroup's LDS)
#ifdef MPO_PF_STAMPS
    if (lane == 0 && (wave & 3) == 0 && blockIdx.x < 1024)
        for (int i = 0; i < 8; ++i) mpo_pf_stamps[blockIdx.x * 16 + 8 * (wave >> 2) + i] = (float)st_sum[i] * 1e-3f;
#endif
}

// W_H [256][1024] fp32 -> bf16 in the stage order of the kernel above: 16-byte fragment t = ((s * 4 + cq) * 4 + dt) * 64 + lane,
// lane (i = lane & 15, g = lane >> 4), holds W_H[64 cq + 32 (dt >> 1) + 8 (i >> 2) + 4 (dt & 1) + (i & 3)][32 s + 8 g .. + 7]: the
// A operand (row i of embed tile (cq, dt), k-group g) of k-step s, with the tile's rows permuted so that the MFMA result
// leaves lane group g with embed columns 64 cq + 32 (dt >> 1) + 8 g + 4 (dt & 1) + r.
// Rows >= rows_valid (the 'small' model: 128 rows of W_H) are packed as zeros.
__global__ void pack_patch_weight_kernel(const float* __restrict__ w, bf16x8* __restrict__ out, int rows_valid) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;          // one 16-byte fragment per thread: 256 * 1024 / 8 of them
    if (t >= PE * PK / 8) return;
    const int lane = t & 63, blk = t >> 6;
    const int dt = blk & 3, cq = (blk >> 2) & 3, s = blk >> 4;
    const int i = lane & 15, g = lane >> 4;
    const int row = 64 * cq + 32 * (dt >> 1) + 8 * (i >> 2) + 4 * (dt & 1) + (i & 3), k0 = 32 * s + 8 * g;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
    if (row < rows_valid) {
        a = *reinterpret_cast<const f32x4*>(w + (size_t)row * PK + k0);
        b = *reinterpret_cast<const f32x4*>(w + (size_t)row * PK + k0 + 4);
    }
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        o[j] = (__bf16)a[j];
        o[4 + j] = (__bf16)b[j];
    }
    out[t] = o;
}

}  // namespace

#ifdef MPO_PF_STAMPS
extern "C" int mpo_debug_patch_fc_stamps(float* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(mpo_pf_stamps), sizeof(float) * 1024 * 16);
}
#endif

// out: one 512-KiB packed block per 256 rows of W_H (embed 128: one block, upper half zero; 512: two blocks)
int mpo_launch_pack_patch_weight(const float* w, void* out, int embed, int patch_dim, hipStream_t stream) {
    MPO_CHECK((embed == 128 || embed == 256 || embed == 512) && patch_dim == PK,
              "patch weight packing is built for {128, 256, 512} x %d (got %d x %d)", PK, embed, patch_dim);
    for (int c0 = 0; c0 < embed; c0 += PE) {
        pack_patch_weight_kernel<<<PE * PK / 8 / 256, 256, 0, stream>>>(w + (size_t)c0 * PK, reinterpret_cast<bf16x8*>(out) + (size_t)c0 * PK / 8,
                                                                      min(PE, embed - c0));
        MPO_LAUNCH_CHECK();
    }
    return 0;
}

int mpo_launch_patch_fc_fwd(const void* x, const void* w_packed, const float* bias, const int* cu, void* h_out, int embed, float drop_p,
                            unsigned long long seed, unsigned long long offset, const unsigned long long* epoch,
                            const BagPlan& plan, hipStream_t stream) {
    MPO_CHECK(embed == 128 || embed == 256 || embed == 512, "patch layer: embed_dim %d not in {128, 256, 512}", embed);
    MPO_CHECK(drop_p >= 0.f && drop_p < 1.f, "patch-layer dropout p must be in [0,1) (got %f)", (double)drop_p);
    MPO_CHECK(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w_packed) | reinterpret_cast<uintptr_t>(h_out)) & 15) == 0,
              "patch layer: operands must be 16-byte aligned");
    const __bf16* xb = reinterpret_cast<const __bf16*>(x);
    const __bf16* wp = reinterpret_cast<const __bf16*>(w_packed);
    __bf16* hb = reinterpret_cast<__bf16*>(h_out);
    if (embed == 128) {
        patch_fc_fwd_kernel<128><<<plan_grid(plan), NTHREADS, 0, stream>>>(xb, wp, bias, cu, hb, 0, drop_p, seed, offset, epoch, plan);
    } else if (embed == 256) {
        patch_fc_fwd_kernel<256><<<plan_grid(plan), NTHREADS, 0, stream>>>(xb, wp, bias, cu, hb, 0, drop_p, seed, offset, epoch, plan);
    } else {
        for (int c0 = 0; c0 < 512; c0 += PE) {                    // (the patch matrix is read once per column half)
            patch_fc_fwd_kernel<512><<<plan_grid(plan), NTHREADS, 0, stream>>>(xb, wp + (size_t)c0 * PK, bias, cu, hb, c0, drop_p, seed, offset,
                                                                              epoch, plan);
            MPO_LAUNCH_CHECK();
        }
    }
    MPO_LAUNCH_CHECK();
    return 0;
}
