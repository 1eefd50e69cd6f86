// Epilogue activations and the element-wise "gate" applied to an operand while it is loaded (dropout masks, activation
// derivatives): shared by the LDS-staged fp32 GEMM kernels (gemm_f32.hip) and the direct ones (gemm_f32_direct.h).
#pragma once
#include "mpo_common.h"
#include "mpo_kernels.h"

namespace {

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case MPO_ACT_RELU: return fmaxf(v, 0.f);
        case MPO_ACT_ELU: return v > 0.f ? v : expm1f(v);
        case MPO_ACT_TANH: return tanhf(v);
        case MPO_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
        default: return v;
    }
}

struct GateFn {
    const float* g;
    int mode;
    float p, inv_keep;
    uint64_t seed, off;
    __device__ __forceinline__ float operator()(float gv, size_t idx) const {
        switch (mode) {
            case MPO_GATE_RELU: return gv > 0.f ? inv_keep : 0.f;
            case MPO_GATE_ELU: return gv > 0.f ? 1.0f : gv + 1.0f;
            case MPO_GATE_TANH: {
                if (gv == 0.f) return p > 0.f ? 0.f : 1.0f;
                const float t = gv * (1.0f - p);
                return (1.0f - t * t) * inv_keep;
            }
            case MPO_GATE_SIGMOID: {
                if (gv == 0.f) return 0.f;
                const float sg = gv * (1.0f - p);
                return sg * (1.0f - sg) * inv_keep;
            }
            case MPO_GATE_RNG: return dropout_keep(seed, off, idx, p, inv_keep);
            case MPO_GATE_ELU_ADROP: {
                if (p <= 0.f) return gv > 0.f ? 1.0f : gv + 1.0f;
                if (dropout_keep(seed, off, idx, p, 1.0f) == 0.f) return 0.f;
                const float a = alpha_drop_a(p), u = (gv - alpha_drop_b(p)) / a;
                return a * (u > 0.f ? 1.0f : u + 1.0f);
            }
            case MPO_GATE_MUL: return gv;
            default: return 1.0f;
        }
    }
    // same result as operator() when the element's random word is already at hand (one draw serves four elements)
    __device__ __forceinline__ float with_word(float gv, uint32_t w) const {
        const bool keep = (float)(w >> 8) * (1.0f / 16777216.0f) >= p;
        if (mode == MPO_GATE_RNG) return keep ? inv_keep : 0.f;
        if (p <= 0.f) return gv > 0.f ? 1.0f : gv + 1.0f;                    // MPO_GATE_ELU_ADROP
        if (!keep) return 0.f;
        const float a = alpha_drop_a(p), u = (gv - alpha_drop_b(p)) / a;
        return a * (u > 0.f ? 1.0f : u + 1.0f);
    }
    __device__ __forceinline__ bool draws() const { return mode == MPO_GATE_RNG || mode == MPO_GATE_ELU_ADROP; }
};

}  // namespace
