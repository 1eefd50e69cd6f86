// Fast path of the direct fp32 GEMM, grouped launches (see gemm_f32_fast.h).
#include "gemm_f32_fast.h"

void mpo_fast_group(const GemmGroup& grp, int gate_classes, int nbmax, dim3 grid, hipStream_t stream) {
    if (gate_classes <= 1) {
        if (nbmax == 4) gemm_f32_fast_group_kernel<1, 4><<<grid, 256, 0, stream>>>(grp);
        else gemm_f32_fast_group_kernel<1, 8><<<grid, 256, 0, stream>>>(grp);
    } else {
        if (nbmax == 4) gemm_f32_fast_group_kernel<2, 4><<<grid, 256, 0, stream>>>(grp);
        else gemm_f32_fast_group_kernel<2, 8><<<grid, 256, 0, stream>>>(grp);
    }
}
