// Direct fp32 GEMM, single-product launches keeping 8 k-blocks per wave in flight (see gemm_f32_direct.h).
#include "gemm_f32_direct.h"

void mpo_direct_single_nb8(const GemmArgs& g, int layout, dim3 grid, hipStream_t stream) {
    direct_launch_single<8>(g, layout, grid, stream);
}
