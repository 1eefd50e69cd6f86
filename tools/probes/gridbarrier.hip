// Probe: what does one grid-wide barrier cost inside a persistent kernel on gfx950, against ~4.6 us per dependent launch
// in a captured graph?  G workgroups x 256 threads; each round every workgroup publishes a value, crosses the barrier and
// checks a neighbour's value (so the fences are the ones a real producer/consumer chain needs).  Three barriers:
//   0  one counter, release add, acquire polling
//   1  one counter, release fence + relaxed add, relaxed polling, acquire fence once
//   2  two levels: one counter per group of workgroups with the same blockIdx % 8 (one XCD if dispatch is round-robin),
//      the last arrival of a group adds to the global counter; everybody polls a generation word, relaxed
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/gridbarrier tools/probes/gridbarrier.hip && tools/_bin/gridbarrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr long kSpinCap = 1L << 22;                                               // exit condition every wave reaches

template <int MODE>
__device__ __forceinline__ void grid_barrier(unsigned* w, unsigned round) {
    // w[0] global counter, w[32] generation, w[64 + 32 g] group counters
    const unsigned G = gridDim.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        long spins = 0;
        if (MODE == 0) {
            __atomic_fetch_add(w, 1u, __ATOMIC_RELEASE);
            while (__atomic_load_n(w, __ATOMIC_ACQUIRE) < G * (round + 1) && ++spins < kSpinCap) __builtin_amdgcn_s_sleep(1);
        } else if (MODE == 1) {
            __atomic_thread_fence(__ATOMIC_RELEASE);
            __atomic_fetch_add(w, 1u, __ATOMIC_RELAXED);
            while (__atomic_load_n(w, __ATOMIC_RELAXED) < G * (round + 1) && ++spins < kSpinCap) __builtin_amdgcn_s_sleep(1);
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        } else {
            const unsigned g = blockIdx.x & 7, members = (G - g + 7) / 8, groups = G < 8 ? G : 8;
            __atomic_thread_fence(__ATOMIC_RELEASE);
            const unsigned prev = __atomic_fetch_add(w + 64 + 32 * g, 1u, __ATOMIC_RELAXED);
            if (prev + 1 == members * (round + 1)) {
                const unsigned p2 = __atomic_fetch_add(w, 1u, __ATOMIC_RELAXED);
                if (p2 + 1 == groups * (round + 1)) __atomic_store_n(w + 32, round + 1, __ATOMIC_RELAXED);
            }
            while (__atomic_load_n(w + 32, __ATOMIC_RELAXED) < round + 1 && ++spins < kSpinCap) __builtin_amdgcn_s_sleep(1);
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
    }
    __syncthreads();
}

template <int MODE>
__global__ void __launch_bounds__(256) probe_kernel(unsigned* w, float* slots, int rounds, int payload, int* bad) {
    const int G = gridDim.x, b = blockIdx.x;
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        for (int i = threadIdx.x; i < payload; i += 256) slots[(size_t)b * payload + i] = (float)(r * 1000 + b);
        grid_barrier<MODE>(w, 2 * r);
        const int nb = (b + 1 + r) % G;
        for (int i = threadIdx.x; i < payload; i += 256) {
            const float v = slots[(size_t)nb * payload + i];
            if (v != (float)(r * 1000 + nb)) atomicAdd(bad, 1);
            acc += v;
        }
        grid_barrier<MODE>(w, 2 * r + 1);                                         // readers done before the next overwrite
    }
    if (acc == -1.f) slots[0] = acc;
}

int main() {
    unsigned* w; float* slots; int* bad;
    const int payload = 1024;
    hipMalloc(&w, 4096); hipMalloc(&slots, 256 * payload * 4); hipMalloc(&bad, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode)
        for (int G : {8, 32, 64, 128, 256}) {
            const int rounds = 200, pl = 256;
            float best = 1e9f; int hb = 0;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemset(w, 0, 4096); hipMemset(bad, 0, 4);
                hipEventRecord(e0);
                if (mode == 0) probe_kernel<0><<<G, 256>>>(w, slots, rounds, pl, bad);
                else if (mode == 1) probe_kernel<1><<<G, 256>>>(w, slots, rounds, pl, bad);
                else probe_kernel<2><<<G, 256>>>(w, slots, rounds, pl, bad);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
                hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
            }
            printf("barrier %d  G %3d: %.2f us per barrier, stale reads %d\n", mode, G, best * 1000.f / (2 * rounds), hb);
        }
    return 0;
}
