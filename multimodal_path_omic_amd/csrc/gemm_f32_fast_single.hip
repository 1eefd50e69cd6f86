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
template <int GCL>
void launch_nb(const GemmArgs& g, int layout, int nbmax, dim3 grid, hipStream_t stream) {
    if (nbmax == 4) launch<GCL, 4>(g, layout, grid, stream);
    else launch<GCL, 8>(g, layout, grid, stream);
}
}  // namespace

void mpo_fast_single(const GemmArgs& g, int layout, int gate_class, int nbmax, dim3 grid, hipStream_t stream) {
    switch (gate_class) {
        case 3: launch_nb<3>(g, layout, nbmax, grid, stream); break;
        case 2: launch_nb<2>(g, layout, nbmax, grid, stream); break;
        default: launch_nb<1>(g, layout, nbmax, grid, stream); break;
    }
}
