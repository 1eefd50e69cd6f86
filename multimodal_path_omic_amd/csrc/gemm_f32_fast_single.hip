// Fast path of the direct fp32 GEMM, single-product launches (see gemm_f32_fast.h).
#include "gemm_f32_fast.h"

namespace {
template <int GCL, int NB>
void launch(const GemmArgs& g, int layout, dim3 grid, hipStream_t stream) {
    switch (layout) {
        case 3: gemm_f32_fast_single_kernel<true, true, GCL, NB><<<grid, 256, 0, stream>>>(g); break;
        case 2: gemm_f32_fast_single_kernel<true, false, GCL, NB><<<grid, 256, 0, stream>>>(g); break;
        case 1: gemm_f32_fast_single_kernel<false, true, GCL, NB><<<grid, 256, 0, stream>>>(g); break;
        default: gemm_f32_fast_single_kernel<false, false, GCL, NB><<<grid, 256, 0, stream>>>(g); break;
    }
}
}  // namespace

void mpo_fast_single(const GemmArgs& g, int layout, int gate_classes, int nbmax, dim3 grid, hipStream_t stream) {
    if (gate_classes <= 1) {
        if (nbmax == 4) launch<1, 4>(g, layout, grid, stream);
        else launch<1, 8>(g, layout, grid, stream);
    } else {
        if (nbmax == 4) launch<2, 4>(g, layout, grid, stream);
        else launch<2, 8>(g, layout, grid, stream);
    }
}
