// Stand-in for a collective's kernel on a second stream (tools/gpu_probe_dp_overlap.py): `wgs` persistent workgroups of 256
// threads that sweep a buffer `iters` times (read-modify-write, 16 bytes per lane) and sleep ~1 us after every 4-KiB piece --
// a few CUs held for 100-200 us with little HBM traffic of their own (17 MB per sweep), the shape an RCCL ring all-reduce of
// the 17 MB gradient bucket has on the device: its pace is the links', not the memory's.  (A first version without the sleep
// moved 400 MB from 32 CUs and measured its own bandwidth contention with the weight gradient instead.)  hipcc --offload-arch=gfx950 -shared -fPIC -o
// tools/_bin/liboccupy.so tools/probes/occupy.hip
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ __launch_bounds__(256) void occupy_kernel(f32x4* buf, long n4, int iters) {
    for (int it = 0; it < iters; ++it)
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
            f32x4 v = buf[i];
            v += 1.0f;
            buf[i] = v;
            __builtin_amdgcn_s_sleep(32);
        }
}
extern "C" int occupy_launch(float* buf, long n_floats, int iters, int wgs, hipStream_t stream) {
    occupy_kernel<<<wgs, 256, 0, stream>>>(reinterpret_cast<f32x4*>(buf), n_floats / 4, iters);
    return (int)hipGetLastError();
}
