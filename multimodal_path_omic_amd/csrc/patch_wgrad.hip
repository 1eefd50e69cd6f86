// Weight gradient of the patch layer self.H (models/mcat/mcat.py:24-29): dW_H = g^T X for the whole window,
//   g [T, 256] bf16  = d(pre-activation) as K1's backward emits it (ReLU / dropout derivative already applied),
//   X [T, 1024] bf16 = the raw patch matrix,  dW_H [256, 1024] fp32.
// T = 480 000 rows at the headline configuration: 251 GFLOP over 1.23 GB of operands, both read once (PMC: 1.229 GB
// fetched).  The library ran it as a 64-batch split-K GEMM: 257-326 us + a 15 us reduction; this kernel: 251-283 us + 11 us.
// Neither HBM (the same stream without the MFMAs: 210 us) nor the MFMA count (240k of 374k cycles busy) is the wall: under
// this mix of matrix and memory work the chip clocks down (GRBM cycles / time: 2.4 GHz streaming only, 2.05 GHz MFMA
// only, 1.3-1.6 GHz both), so fewer cycles bought less time than they should (NOTES.md r02-dW_H).
//
// One workgroup per CU, 256 of them = 64 row ranges x 4 column blocks of X (256 x 1 for a 256-wide X: the key-projection
// weight gradient of NaCAGaT, d_k^T H_bag); a workgroup accumulates the 256 x 256 block
//   dW[:, 256 cb ..] += g[rows]^T X[rows, 256 cb ..]
// of its row range in registers (8 waves x 32 accumulator tiles) and writes one fp32 partial; a second launch sums the 64
// partials.  The four column blocks of a row range read the same g rows: they sit on the SAME XCD (blockIdx -> XCD is
// round-robin, see the id mapping) and run in step, so three of those four reads are L2 hits.
//
// Both MFMA operands contract over the patch-row index, i.e. both are "transposed" reads of row-major tiles: the
// 32 x 256 bf16 images of coattn_tile.h (512-byte rows, chunk swizzle) and its col_frag (ds_read_b64_tr_b16) deliver
// them; the k order col_frag produces is a permutation of the 32 rows, the same for both operands.  Tiles travel
// global -> LDS directly (global_load_lds_dwordx4, swizzle applied to the source chunk, waits counted by hand) into a
// ring of four stages, three chunks of 32 rows ahead; one workgroup barrier per chunk.
#include "coattn_tile.h"
#include "mpo_kernels.h"

namespace {

constexpr int WG_E = 256;                    // rows of dW (= columns of g)
constexpr int WG_CB = 256;                   // columns of X per workgroup
constexpr int WG_BK = 32;                    // patch rows per chunk (one MFMA k-step)
constexpr int WG_STAGES = 4;
constexpr int WG_TILE = WG_BK * 512;         // 16 KiB image
constexpr int WG_STAGE = 2 * WG_TILE;        // g image | x image
constexpr int WG_LDS = WG_STAGES * WG_STAGE; // 128 KiB
constexpr int WG_WAVES = 8;
constexpr int WG_WGS = 256;                  // workgroups per launch: row ranges x column blocks, one per CU

__device__ __forceinline__ void wait_vm(int n) {
    // n = wave-instructions that may stay outstanding; 4 per chunk and wave
    switch (n) {
        case 0: __builtin_amdgcn_s_waitcnt(0x0F70); break;
        case 4: __builtin_amdgcn_s_waitcnt(0x0F74); break;
        default: __builtin_amdgcn_s_waitcnt(0x0F78); break;      // 8
    }
}

__global__ __launch_bounds__(WG_WAVES * 64, 1)
void patch_wgrad_kernel(const __bf16* __restrict__ g, const __bf16* __restrict__ x, int total_rows, int patch_dim,
                        int g_dim,          // columns of g (its row pitch): 128, 256 or 512; this pass takes 256 of them from g on
                        int e_rows,         // of which exist: 128 (g_dim 128: the image's upper half repeats the lower, its dW rows are not written) or 256
                        int rows_per_range, float* __restrict__ part) {
    using G = TileGeom<WG_E>;
    __shared__ __attribute__((aligned(1024))) char lds[WG_LDS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // workgroup id -> (row range, column block): the column blocks of a row range share an XCD (id mod 8)
    const int n_cb = patch_dim < WG_CB ? 1 : patch_dim / WG_CB;    // (a 128-wide X: one block whose upper half repeats the lower and is not written)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int rr = xcd + 8 * (slot / n_cb), cb = slot % n_cb;
    const int rb = min(total_rows, rr * rows_per_range), re = min(total_rows, rb + rows_per_range);
    const int n_chunks = (re - rb + WG_BK - 1) / WG_BK;

    // this wave's four wave-instructions of a chunk: instruction q = 4 wave + i fills KiB (q & 15) of image (q >> 4)
    const bool x_tile = wave >= 4;
    const char* src_base = x_tile ? reinterpret_cast<const char*>(x) + (size_t)cb * WG_CB * 2 : reinterpret_cast<const char*>(g);
    const size_t src_row = x_tile ? (size_t)patch_dim * 2 : (size_t)g_dim * 2;
    const int cmask = x_tile ? (patch_dim < WG_CB ? patch_dim / 8 - 1 : 31) : e_rows / 8 - 1;      // 16-byte chunks of a source row that exist
    // (32-bit byte offsets against a scalar base: the launcher checks that the operands stay below 4 GiB)
    const unsigned src_row32 = (unsigned)src_row;
    auto issue = [&](int chunk) {
        char* stage = lds + (chunk % WG_STAGES) * WG_STAGE + (x_tile ? WG_TILE : 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = 4 * (wave & 3) + i;
            const int r = 2 * k + (lane >> 5), cs = lane & 31;
            const int c = (cs ^ ((r & 7) << 1)) & cmask;
            const int grow = min(rb + chunk * WG_BK + r, total_rows - 1);      // past the end: clamped, the g rows are zeroed below
            const unsigned off = (unsigned)grow * src_row32 + (unsigned)(c * 16);
            const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(stage + k * 1024);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                         :: "v"(off), "s"(src_base), "s"(dst) : "memory", "m0");
#pragma clang diagnostic pop
        }
    };

    const int wm = wave & 3, wn = wave >> 2;           // 64 rows of dW x 128 columns per wave
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // make chunk m visible: own pieces landed (hand-counted wait), rows past the range cleared, workgroup barrier (after
    // which chunk m-1 -- read into registers by every wave before it arrived here -- may be overwritten), next request
    auto arrive = [&](int m) {
        wait_vm(4 * min(WG_STAGES - 2, n_chunks - 1 - m));
        asm volatile("" ::: "memory");
        char* stage = lds + (m % WG_STAGES) * WG_STAGE;
        if (!x_tile && rb + (m + 1) * WG_BK > re) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = 4 * (wave & 3) + i;
                const int r = 2 * k + (lane >> 5);
                if (rb + m * WG_BK + r >= re)
                    *reinterpret_cast<f32x4*>(stage + k * 1024 + lane * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (m + WG_STAGES - 1 < n_chunks) issue(m + WG_STAGES - 1);
    };
    auto load_frags = [&](int m, bf16x8 (&fa)[4], bf16x8 (&fb)[8]) {
        const char* gt = lds + (m % WG_STAGES) * WG_STAGE;
        const char* xt = gt + WG_TILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = col_frag<WG_E>(gt, 4 * wm + i, lane);
#pragma unroll
        for (int j = 0; j < 8; ++j) fb[j] = col_frag<WG_E>(xt, 8 * wn + j, lane);
    };
    auto mma = [&](const bf16x8 (&fa)[4], const bf16x8 (&fb)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = mfma_bf16(fa[i], fb[j], acc[i][j]);
    };

    // Two fragment sets: the transposing LDS reads of chunk n+1 are in flight while the 32 MFMAs of chunk n run.
    for (int c = 0; c < min(WG_STAGES - 1, n_chunks); ++c) issue(c);
    bf16x8 pa[4], pb[8], qa[4], qb[8];
    if (n_chunks > 0) {
        arrive(0);
        load_frags(0, pa, pb);
    }
    int n = 0;
    // steady state, two chunks per trip, straight-line: constant wait count, no tail handling, every request valid.
    // The 24 transposing reads of the next chunk and their address arithmetic are issued in the shadow of the current
    // chunk's MFMAs (forced interleave) -- all eight waves leave each barrier together, so whatever a wave issues before
    // its first MFMA is time the matrix pipes of all four SIMDs idle (measured: 1834 cycles per chunk against 1024 of MFMA)
    auto half = [&](int m, bf16x8 (&la)[4], bf16x8 (&lb)[8], const bf16x8 (&ca)[4], const bf16x8 (&cb_)[8]) {
        __builtin_amdgcn_s_waitcnt(0x0F78);                               // vmcnt(8): chunk m landed (two younger in flight)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        load_frags(m, la, lb);
        mma(ca, cb_);
        issue(m + WG_STAGES - 1);                                         // (after the MFMAs in program order: not urgent)
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);            // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);            // 1 LDS read of the next chunk
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    };
    for (; n + 5 < n_chunks; n += 2) {
        half(n + 1, qa, qb, pa, pb);
        half(n + 2, pa, pb, qa, qb);
    }
    for (; n < n_chunks; n += 2) {                                        // the last chunks: general form
        if (n + 1 < n_chunks) {
            arrive(n + 1);
            load_frags(n + 1, qa, qb);
        }
        mma(pa, pb);
        if (n + 1 < n_chunks) {
            if (n + 2 < n_chunks) {
                arrive(n + 2);
                load_frags(n + 2, pa, pb);
            }
            mma(qa, qb);
        }
    }
    // partial [rr][e_rows][patch_dim]: lane holds rows 16 (4 wm + i) + 4 (lane >> 4) + r, column 16 (8 wn + j) + (lane & 15)
    if (64 * wm >= e_rows || 128 * wn >= patch_dim) return;
    float* out = part + ((size_t)rr * e_rows) * patch_dim + (size_t)cb * WG_CB;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(size_t)(16 * (4 * wm + i) + 4 * (lane >> 4) + r) * patch_dim + 16 * (8 * wn + j) + (lane & 15)] = acc[i][j][r];
}

// dW[i] = sum_rr part[rr][i]: one float4 per thread, the 64 loads in flight eight at a time
__global__ __launch_bounds__(256)
void patch_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int n4, int n_parts, int accumulate) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (accumulate) a = *reinterpret_cast<const f32x4*>(out + (size_t)i * 4);      // (a later row segment of the same window)
#pragma unroll 8
    for (int s = 0; s < n_parts; ++s) a += *reinterpret_cast<const f32x4*>(part + ((size_t)s * n4 + i) * 4);
    *reinterpret_cast<f32x4*>(out + (size_t)i * 4) = a;
}

}  // namespace

size_t mpo_patch_wgrad_partial_floats(int embed, int patch_dim) {
    const int n_cb = patch_dim / WG_CB > 0 ? patch_dim / WG_CB : 1;
    return (size_t)(WG_WGS / n_cb) * (embed < WG_E ? embed : WG_E) * patch_dim;
}

// g [total_rows][embed] bf16, x [total_rows][patch_dim] bf16 -> d_weight [embed][patch_dim] fp32 (overwritten).  embed 512: one pass
// per 256 columns of g; a patch matrix of 4 GiB or more (the kernel's DMA offsets are 32-bit): one pass per row segment.
int mpo_launch_patch_wgrad(const void* g, const void* x, int64_t total_rows, int embed, int patch_dim, float* part, float* d_weight,
                           int workgroups, hipStream_t stream) {
    MPO_CHECK((embed == 128 || embed == 256 || embed == 512) &&
              (patch_dim == 128 || (patch_dim >= WG_CB && patch_dim % WG_CB == 0 && (256 % (patch_dim / WG_CB)) == 0)),
              "patch weight gradient: built for embed in {128, 256, 512} and patch_dim in {128, 256, 512, 1024, 2048} (got %d, %d)", embed, patch_dim);
    MPO_CHECK(total_rows >= 1, "patch weight gradient: no rows");
    MPO_CHECK(((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(x)) & 15) == 0, "patch weight gradient: operands must be 16-byte aligned");
    const int n_cb = patch_dim < WG_CB ? 1 : patch_dim / WG_CB;
    // workgroups = 0: one per CU.  Fewer (a multiple of 8 n_cb, e.g. 224 of 256 at patch_dim 1024) leave CUs to a kernel of
    // another stream -- the gradient all-reduce of a data-parallel step: a persistent kernel that owns
    // every CU's LDS and registers lets nothing else run until its workgroups retire.
    MPO_CHECK(workgroups == 0 || (workgroups >= 8 * n_cb && workgroups <= WG_WGS && workgroups % (8 * n_cb) == 0),
              "patch weight gradient: workgroups must be 0 or a multiple of %d up to %d (got %d)", 8 * n_cb, WG_WGS, workgroups);
    const int ranges = (workgroups ? workgroups : WG_WGS) / n_cb;   // 64 row ranges at patch_dim 1024, 256 at 256
    const int e_rows = embed < WG_E ? embed : WG_E;
    const int wide = patch_dim > embed ? patch_dim : embed;
    const int64_t seg_rows = ((((int64_t)1 << 32) - 1) / ((int64_t)wide * 2)) / WG_BK * WG_BK;      // rows whose byte offsets stay below 4 GiB
    const __bf16* gb = static_cast<const __bf16*>(g);
    const __bf16* xb = static_cast<const __bf16*>(x);
    for (int64_t s0 = 0; s0 < total_rows; s0 += seg_rows) {
        const int rows = (int)(total_rows - s0 < seg_rows ? total_rows - s0 : seg_rows);
        const int rpr = ((rows + ranges - 1) / ranges + WG_BK - 1) / WG_BK * WG_BK;
        for (int c0 = 0; c0 < embed; c0 += WG_E) {
            patch_wgrad_kernel<<<ranges * n_cb, WG_WAVES * 64, 0, stream>>>(gb + (size_t)s0 * embed + c0, xb + (size_t)s0 * patch_dim, rows, patch_dim,
                                                                          embed, e_rows, rpr, part);
            MPO_LAUNCH_CHECK();
            const int n4 = e_rows * patch_dim / 4;
            patch_wgrad_reduce_kernel<<<(n4 + 255) / 256, 256, 0, stream>>>(part, d_weight + (size_t)c0 * patch_dim, n4, ranges, s0 > 0);
            MPO_LAUNCH_CHECK();
        }
    }
    return 0;
}
