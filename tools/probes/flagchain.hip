// Probe: can the ~4.6 us a DEPENDENT launch costs inside a captured graph be hidden by taking the dependency out of the graph?
// A chain of N small kernels (G workgroups x 256 threads, kernel k reads what kernel k - 1 wrote) captured three ways:
//   A  one stream: every launch behind a barrier packet (what the tail of the step is today)
//   B  one stream + device-side flags (every kernel also waits on / signals a counter): what the flags themselves cost
//   C  two streams, kernels alternate, NO graph edge between neighbours: kernel k spins on kernel k - 1's completion counter
//      (release add by every workgroup's thread 0 after its stores, acquire polling by every workgroup's thread 0 before its
//      loads), so the command processor can set up kernel k while k - 1 runs.  At most two kernels are in flight (k + 1 follows
//      k - 1 in stream order) and both grids are small enough to be resident together: no spinning workgroup can keep its
//      producer off the chip.
// The counters are cumulative over replays (target = replay number x G); a one-thread kernel bumps the replay number.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/flagchain tools/probes/flagchain.hip && tools/_bin/flagchain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr long kSpinCap = 1L << 22;                                               // exit condition every wave reaches

__global__ void bump(unsigned* epoch) { *epoch += 1; }

// buf: [N + 1][G * 256] floats; kernel k writes row k + 1 = row k + 1.0 (row 0 is constant input)
template <bool FLAGS>
__global__ void __launch_bounds__(256) link(float* buf, const unsigned* epoch, unsigned* flags, int k, int work, int* bad) {
    const int G = gridDim.x, n = G * 256;
    if (FLAGS && k > 0) {
        if (threadIdx.x == 0) {
            const unsigned target = *epoch * (unsigned)G;
            long spins = 0;
            while (__atomic_load_n(flags + 32 * (k - 1), __ATOMIC_ACQUIRE) < target && ++spins < kSpinCap) __builtin_amdgcn_s_sleep(1);
            if (spins >= kSpinCap) atomicAdd(bad, 1);
        }
        __syncthreads();
    }
    // every workgroup reads a DIFFERENT workgroup's slice of the previous row (a real cross-workgroup dependency)
    const int src = ((blockIdx.x + 1 + k) % G) * 256 + threadIdx.x, dst = blockIdx.x * 256 + threadIdx.x;
    float v = k == 0 ? (float)*epoch : buf[(size_t)k * n + src];          // (row 0 = the replay number: a read that overtakes its producer sees last replay's value)
    for (int i = 0; i < work; ++i) v = v * 1.0000001f + 1e-9f;                   // a little arithmetic (a 1-2 us kernel)
    buf[(size_t)(k + 1) * n + dst] = v + 1.0f;
    if (FLAGS) {
        __syncthreads();
        if (threadIdx.x == 0) __atomic_fetch_add(flags + 32 * k, 1u, __ATOMIC_RELEASE);
    }
}

int main(int argc, char** argv) {
    const int N = 64, reps = 50;
    const int work = argc > 1 ? atoi(argv[1]) : 200;
    for (int G : {16, 64, 192}) {
        float* buf; unsigned *epoch, *flags; int* bad;
        CK(hipMalloc(&buf, sizeof(float) * (N + 1) * G * 256));
        CK(hipMalloc(&epoch, 4)); CK(hipMalloc(&flags, 4 * 32 * N)); CK(hipMalloc(&bad, 4));
        hipStream_t s1, s2;
        CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
        hipEvent_t fork, join, e0, e1;
        CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int mode = 0; mode < 3; ++mode) {
            CK(hipMemset(buf, 0, sizeof(float) * (N + 1) * G * 256));
            CK(hipMemset(epoch, 0, 4)); CK(hipMemset(flags, 0, 4 * 32 * N)); CK(hipMemset(bad, 0, 4));
            CK(hipDeviceSynchronize());
            hipGraph_t graph; hipGraphExec_t exec;
            CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
            bump<<<1, 1, 0, s1>>>(epoch);
            if (mode == 2) { CK(hipEventRecord(fork, s1)); CK(hipStreamWaitEvent(s2, fork, 0)); }
            for (int k = 0; k < N; ++k) {
                hipStream_t s = (mode == 2 && (k & 1)) ? s2 : s1;
                if (mode == 0) link<false><<<G, 256, 0, s>>>(buf, epoch, flags, k, work, bad);
                else link<true><<<G, 256, 0, s>>>(buf, epoch, flags, k, work, bad);
            }
            if (mode == 2) { CK(hipEventRecord(join, s2)); CK(hipStreamWaitEvent(s1, join, 0)); }
            CK(hipStreamEndCapture(s1, &graph));
            CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(exec, s1));
            CK(hipStreamSynchronize(s1));
            CK(hipEventRecord(e0, s1));
            for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(exec, s1));
            CK(hipEventRecord(e1, s1));
            CK(hipStreamSynchronize(s1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<float> last(G * 256);
            CK(hipMemcpy(last.data(), buf + (size_t)N * G * 256, sizeof(float) * G * 256, hipMemcpyDeviceToHost));
            int hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
            int wrong = 0;
            const float want = (float)(N + 5 + reps);
            for (float v : last) wrong += !(v > want - 0.5f && v < want + 0.5f);      // (the arithmetic filler drifts by ~1.5e-3 relative over the chain; a stale read is off by 1)
            printf("G %3d work %d mode %c: %.2f us per replay of %d kernels = %.2f us per kernel   (wrong values %d, spin timeouts %d)\n",
                   G, work, "ABC"[mode], ms * 1e3 / reps, N, ms * 1e3 / reps / N, wrong, hb);
            fflush(stdout);
            CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
            if (hb) { printf("spin timeout: stopping\n"); return 1; }
        }
        CK(hipFree(buf)); CK(hipFree(epoch)); CK(hipFree(flags)); CK(hipFree(bad));
    }
    return 0;
}
