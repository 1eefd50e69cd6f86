// Diagnostic: what starting workgroups costs on gfx950 -- an (almost) empty kernel of G workgroups x 256 threads, 200
// back-to-back launches on one stream, time per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* p, int n) {
    if ((int)(blockIdx.x * 256 + threadIdx.x) == n) p[0] = 1.f;       // never true: n = -1
}
int main() {
    float* p; (void)hipMalloc(&p, 4);
    hipStream_t s; (void)hipStreamCreate(&s);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int g : {1, 64, 256, 512, 1024, 2048, 3072, 4096, 8192}) {
        for (int i = 0; i < 20; ++i) k<<<g, 256, 0, s>>>(p, -1);
        (void)hipStreamSynchronize(s);
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < 200; ++i) k<<<g, 256, 0, s>>>(p, -1);
        (void)hipEventRecord(e1, s);
        (void)hipStreamSynchronize(s);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%5d workgroups x 256 threads: %.2f us per launch\n", g, ms * 1000.f / 200);
    }
    return 0;
}
