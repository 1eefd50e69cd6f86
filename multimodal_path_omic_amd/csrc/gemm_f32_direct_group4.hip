// Direct fp32 GEMM, grouped launches (members of any layout) keeping 4 k-blocks per wave in flight (see gemm_f32_direct.h).
#include "gemm_f32_direct.h"

void mpo_direct_group_nb4(const GemmGroup& grp, dim3 grid, hipStream_t stream) {
    gemm_f32_direct_kernel<4><<<grid, 256, 0, stream>>>(grp);
}
