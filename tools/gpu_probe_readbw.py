"""Probe: what does a pure read-once stream of 245.76 MB (one bench window of bf16 H) reach on this box?
A grid-stride kernel with U independent 16-byte loads in flight per thread sums the buffer; two buffers alternate so
that no launch finds its data in the 256 MB infinity cache.  Sets the practical ceiling K1's forward is judged against."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
SRC = r'''
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int U>
__global__ __launch_bounds__(256) void rd(const f32x4* __restrict__ p, size_t n16, float* out) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    f32x4 acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (; i + (U - 1) * stride < n16; i += U * stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] += p[i + u * stride];
    }
    for (; i < n16; i += stride) acc[0] += p[i];
    f32x4 s = acc[0];
#pragma unroll
    for (int u = 1; u < U; ++u) s += acc[u];
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[0] = 1.f;     // never true: keeps the loads alive
}
// K1's pattern: every workgroup streams its OWN contiguous region (n16 / gridDim.x chunks of 16 bytes)
template <int U>
__global__ __launch_bounds__(256) void rd_blocked(const f32x4* __restrict__ p, size_t n16, float* out) {
    const size_t per = n16 / gridDim.x;
    const f32x4* q = p + per * blockIdx.x;
    f32x4 acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    size_t i = threadIdx.x;
    for (; i + (U - 1) * 256 < per; i += U * 256) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] += q[i + u * 256];
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int u = 1; u < U; ++u) s += acc[u];
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[0] = 1.f;
}
extern "C" void launch_blocked(int u, const void* p, size_t n16, float* out, int grid, hipStream_t st) {
    const f32x4* q = (const f32x4*)p;
    if (u == 4) rd_blocked<4><<<grid, 256, 0, st>>>(q, n16, out);
    else rd_blocked<8><<<grid, 256, 0, st>>>(q, n16, out);
}
extern "C" void launch(int u, const void* p, size_t n16, float* out, int grid, hipStream_t st) {
    const f32x4* q = (const f32x4*)p;
    if (u == 2) rd<2><<<grid, 256, 0, st>>>(q, n16, out);
    else if (u == 4) rd<4><<<grid, 256, 0, st>>>(q, n16, out);
    else if (u == 8) rd<8><<<grid, 256, 0, st>>>(q, n16, out);
    else rd<16><<<grid, 256, 0, st>>>(q, n16, out);
}
'''
open("/tmp/rd.hip", "w").write(SRC)
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "/tmp/rd.hip", "-o", "/tmp/rd.so"], check=True)
lib = ctypes.CDLL("/tmp/rd.so")
lib.launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
nbytes = 32 * 15000 * 256 * 2
bufs = [torch.randn(nbytes // 4, device=dev) for _ in range(2)]
out = torch.zeros(4, device=dev)
st = torch.cuda.current_stream().cuda_stream
lib.launch_blocked.argtypes = lib.launch.argtypes
for blocked, u, wg_per_cu in [(0, 4, 1), (0, 4, 2), (0, 8, 1), (1, 4, 1), (1, 4, 2), (1, 8, 1), (1, 8, 2), (1, 8, 4)]:
    if True:
        grid = 256 * wg_per_cu
        fn = lib.launch_blocked if blocked else lib.launch
        run = lambda i: fn(u, bufs[i & 1].data_ptr(), nbytes // 16, out.data_ptr(), grid, st)
        for i in range(4): run(i)
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(20): run(i)
        e.record(); torch.cuda.synchronize()
        us = a.elapsed_time(e) / 20 * 1e3
        print(f"{'per-WG contiguous regions' if blocked else 'one global sweep         '} U={u:2d} loads in flight/thread, {wg_per_cu:2d} WG/CU: {us:6.1f} us  {nbytes / us / 1e6:5.2f} TB/s", flush=True)
