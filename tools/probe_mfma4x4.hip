// Diagnostic: lane / register mapping of v_mfma_f32_4x4x1_16B_f32 on gfx950, with BLGP variants.
// A lane l carries a = 100 + l, B lane l carries b = 1 (then one-hot probes); prints which (A lane, B lane) feeds each D slot.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int BLGP>
__global__ void probe(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, BLGP);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}
template <int BLGP>
void run(const char* tag) {
    float *a, *b, *d;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    float ha[64], hb[64], hd[256];
    // D = A_lane_value * B_lane_value: choose A = 1 + la (la = A lane), B = 1000^... use two runs with primes
    for (int i = 0; i < 64; ++i) { ha[i] = 1.f + i; hb[i] = 1.f; }
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    probe<BLGP><<<1, 64>>>(a, b, d); hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    int alane[256];
    for (int i = 0; i < 256; ++i) alane[i] = (int)(hd[i] + 0.5f) - 1;
    for (int i = 0; i < 64; ++i) { ha[i] = 1.f; hb[i] = 1.f + i; }
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    probe<BLGP><<<1, 64>>>(a, b, d); hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    printf("== %s: D[lane][reg] <- (A lane, B lane)\n", tag);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int r = 0; r < 4; ++r) printf(" (%2d,%2d)", alane[l * 4 + r], (int)(hd[l * 4 + r] + 0.5f) - 1);
        printf("\n");
    }
    hipFree(a); hipFree(b); hipFree(d);
}
int main() {
    run<0>("blgp0"); run<1>("blgp1"); run<2>("blgp2");
    return 0;
}
