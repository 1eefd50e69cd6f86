// Fast path of the direct fp32 GEMM, grouped launches (see gemm_f32_fast.h).
#include "gemm_f32_fast.h"

namespace {
template <int GCL>
void launch(const GemmGroup& grp, int nbmax, dim3 grid, hipStream_t stream) {
    if (nbmax == 4) gemm_f32_fast_group_kernel<GCL, 4><<<grid, 256, 0, stream>>>(grp);
    else gemm_f32_fast_group_kernel<GCL, 8><<<grid, 256, 0, stream>>>(grp);
}
}  // namespace

void mpo_fast_group(const GemmGroup& grp, int gate_class, int nbmax, dim3 grid, hipStream_t stream) {
    switch (gate_class) {
        case 3: launch<3>(grp, nbmax, grid, stream); break;
        case 2: launch<2>(grp, nbmax, grid, stream); break;
        default: launch<1>(grp, nbmax, grid, stream); break;
    }
}
