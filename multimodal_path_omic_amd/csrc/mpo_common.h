// Shared device helpers for the gfx950 (MI355X, CDNA4) kernels of the fusion path.
// Wave = 64 lanes everywhere; no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))
#define GLOBAL_PTR(T, p) ((const __attribute__((address_space(1))) T*)(p))

static constexpr float kLog2e = 1.4426950408889634f;
static constexpr float kLn2 = 0.6931471805599453f;

__device__ __forceinline__ __bf16 f2bf(float x) { return (__bf16)x; }          // v_cvt_pk_bf16_f32 (RNE, NaN-safe)
__device__ __forceinline__ float bf2f(__bf16 h) { return (float)h; }

// x = hi + lo with both parts bf16: ~16 mantissa bits survive an MFMA that takes bf16 operands.
__device__ __forceinline__ void split_bf16(float x, __bf16& hi, __bf16& lo) {
    hi = (__bf16)x;
    lo = (__bf16)(x - (float)hi);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ---- counter-based RNG for dropout masks (seed + 64-bit element counter) ----
// philox4x32 (Philox4x32-10) is kept for the one pass-bound kernel that still calls it (the bf16 patch epilogue of the library
// path); every other mask is cut from draw4x32 below.
// Forward and backward regenerate the same mask from (seed, offset, element index); nothing is stored.
__device__ __forceinline__ uint4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}
// ---- counter hash for masks that only ONE kernel ever generates (the fused patch layer writes its mask into H_bag as zeros;
// consumers read it back from there).  32-bit integer multiplies run at a quarter of the vector rate: Philox's 40 per draw were
// 640 cycles per 16 elements in that kernel's epilogue; this is the murmur3 finaliser (full avalanche, 2 multiplies per word).
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ uint32_t hash_stream_key(unsigned long long seed, unsigned long long offset) {      // once per kernel
    uint32_t k = fmix32((uint32_t)seed ^ 0x5A17u);
    k = fmix32(k ^ (uint32_t)(seed >> 32));
    k = fmix32(k ^ (uint32_t)offset);
    return fmix32(k ^ (uint32_t)(offset >> 32));
}
// The 128 bits of counter `ctr` under a 32-bit stream key.  The key enters twice: XOR-ed into the counter before the first
// finaliser round AND as the (odd) stride between the four output words, so two streams are not the same set of draws at
// permuted counters (with the key in the first round only, stream B's counter c ^ kA ^ kB reproduced stream A's counter c:
// per-rank seeds would have shared their masks up to a permutation).  What remains: all 128 bits hang off one 32-bit
// intermediate x, so inside ONE stream two counters collide on x with probability 2^-32 per pair (~1e-4 of the draws of a
// 1M-counter stream share their 16 bytes with another draw) -- irrelevant for dropout, and the reason this is not called Philox.
__device__ __forceinline__ uint32_t hash_word_stride(uint32_t key) { return fmix32(key ^ 0x9E3779B9u) | 1u; }
__device__ __forceinline__ uint4 hash4x32(uint32_t key, unsigned long long ctr) {
    const uint32_t x = fmix32(key ^ (uint32_t)ctr) + (uint32_t)(ctr >> 32) * 0x85EBCA77u;
    const uint32_t inc = hash_word_stride(key);                 // (wave-uniform: the compiler keeps it on the scalar unit)
    return make_uint4(fmix32(x + inc), fmix32(x + 2u * inc), fmix32(x + 3u * inc), fmix32(x + 4u * inc));
}
// The 128 bits of counter (c_lo, c_hi) of the stream `seed` -- what every dropout mask of the tail and of K2 is cut from
// (forward and backward call it with the same arguments).  A counter hash (murmur3 finaliser per word: 11 vector integer
// multiplies per draw), not Philox4x32-10 (40): 32-bit multiplies run at a quarter of the vector rate, and the tail's kernels
// are short enough for that to show -- same-box A/B of the whole MCAT window step in r02: 1.143-1.150 -> 1.126-1.130 ms.
// Masks stay a pure function of (seed, counter).
__device__ __forceinline__ uint4 draw4x32(uint32_t c_lo, uint32_t c_hi, uint32_t s_lo, uint32_t s_hi) {
    return hash4x32(fmix32(s_lo ^ fmix32(s_hi ^ 0x5A17u)), ((unsigned long long)c_hi << 32) | c_lo);
}
// keep-scale for element `idx` of a stream identified by (seed, stream): 0 or 1/(1-p)
__device__ __forceinline__ float dropout_keep(uint64_t seed, uint64_t offset, uint64_t idx, float p, float inv_keep) {
    uint64_t ctr = offset + (idx >> 2);
    uint4 r = draw4x32((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
    uint32_t w = (idx & 3) == 0 ? r.x : (idx & 3) == 1 ? r.y : (idx & 3) == 2 ? r.z : r.w;
    float u = (float)(w >> 8) * (1.0f / 16777216.0f);
    return u >= p ? inv_keep : 0.0f;
}

// nn.AlphaDropout(p) constants: y = a * (keep ? x : alpha') + b  (self-normalising dropout of the omic SNNs)
static constexpr float kAlphaPrime = -1.7580993408473766f;
__host__ __device__ __forceinline__ float alpha_drop_a(float p) { return 1.0f / sqrtf((1.0f - p) * (1.0f + p * kAlphaPrime * kAlphaPrime)); }
__host__ __device__ __forceinline__ float alpha_drop_b(float p) { return -alpha_drop_a(p) * kAlphaPrime * p; }

// Dropout streams are (seed, offset) by value plus an optional DEVICE-resident epoch: under HIP-graph replay
// the by-value part is frozen, so the host bumps *epoch (a captured device op) once per step and every
// stream moves by kEpochStride counters.  epoch == nullptr (eager mode) leaves the offset as passed.
static constexpr unsigned long long kEpochStride = 1ull << 40;
__device__ __forceinline__ unsigned long long epoch_offset(unsigned long long offset, const unsigned long long* epoch) {
    return epoch ? offset + (*epoch) * kEpochStride : offset;
}

// ---- host-side error plumbing (one definition in capi.hip) ----
void mpo_set_error(const char* fmt, ...);
#define MPO_CHECK(cond, ...) do { if (!(cond)) { mpo_set_error(__VA_ARGS__); return 1; } } while (0)
#define MPO_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    mpo_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define MPO_LAUNCH_CHECK() MPO_HIP(hipGetLastError())
