// Diagnostic: what starting workgroups costs on gfx950 -- an (almost) empty kernel of G workgroups x 256 threads, 200
// back-to-back launches on one stream, time per launch; the same with static LDS in use, a __syncthreads, and a large
// by-value argument (the shapes of this library's small kernels).
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { float v[340]; };                                           // 1360 bytes, like GemmGroup
__global__ void k_plain(float* p, int n) {
    if ((int)(blockIdx.x * 256 + threadIdx.x) == n) p[0] = 1.f;       // never true: n = -1
}
template <int BYTES>
__global__ void k_lds(float* p, int n) {
    __shared__ float s[BYTES / 4];
    s[threadIdx.x] = (float)n;
    __syncthreads();
    if ((int)(blockIdx.x * 256 + threadIdx.x) == n) p[0] = s[(threadIdx.x + 1) & 255];
}
__global__ void k_bigarg(float* p, int n, Big b) {
    if ((int)(blockIdx.x * 256 + threadIdx.x) == n) p[0] = b.v[threadIdx.x & 255];
}
template <typename F>
void run(const char* name, F launch) {
    hipStream_t s; (void)hipStreamCreate(&s);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int g : {64, 1024, 4096}) {
        for (int i = 0; i < 20; ++i) launch(g, s);
        (void)hipStreamSynchronize(s);
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < 200; ++i) launch(g, s);
        (void)hipEventRecord(e1, s);
        (void)hipStreamSynchronize(s);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %5d workgroups: %.2f us per launch\n", name, g, ms * 1000.f / 200);
    }
}
int main() {
    float* p; (void)hipMalloc(&p, 4);
    Big b{};
    run("plain", [&](int g, hipStream_t s) { k_plain<<<g, 256, 0, s>>>(p, -1); });
    run("4 KiB LDS + barrier", [&](int g, hipStream_t s) { k_lds<4096><<<g, 256, 0, s>>>(p, -1); });
    run("32 KiB LDS + barrier", [&](int g, hipStream_t s) { k_lds<32768><<<g, 256, 0, s>>>(p, -1); });
    run("1360-byte argument", [&](int g, hipStream_t s) { k_bigarg<<<g, 256, 0, s>>>(p, -1, b); });
    return 0;
}
