// Developer micro-benchmark (GPU box): latency of a dependent vector load (one wave chasing pointers through a random cycle)
// by working-set size - what a step of a walk through tables in L2 / Infinity Cache / HBM costs.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_latency.hip -o build/ubench_latency && gpurun -- build/ubench_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include <numeric>
#include <algorithm>
__global__ void chase(const uint32_t *next, uint32_t start, int steps, uint32_t *out, unsigned long long *cycles)
{
    uint32_t p = start + threadIdx.x; // 64 lanes, 64 different chains
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < steps; ++i) p = next[p];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = p;
    if (threadIdx.x == 0) *cycles = t1 - t0;
}
int main()
{
    uint32_t *d_out;
    unsigned long long *d_cyc;
    hipMalloc(&d_out, 256), hipMalloc(&d_cyc, 8);
    for (size_t kb : {16, 256, 1024, 3072, 8192, 32768, 131072, 1048576}) {
        const size_t n = kb * 1024 / 4;
        std::vector<uint32_t> perm(n), next(n);
        std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937 rng(7);
        std::shuffle(perm.begin(), perm.end(), rng);
        for (size_t i = 0; i < n; ++i) next[perm[i]] = perm[(i + 1) % n]; // one cycle through everything
        uint32_t *d_next;
        hipMalloc(&d_next, n * 4);
        hipMemcpy(d_next, next.data(), n * 4, hipMemcpyHostToDevice);
        const int steps = 20000;
        for (int lanes : {1, 64}) {
            unsigned long long cyc = 0;
            for (int rep = 0; rep < 3; ++rep) { // (the last repetition counts: the first warms the caches where the set fits)
                hipLaunchKernelGGL(chase, dim3(1), dim3(lanes), 0, 0, d_next, 0u, steps, d_out, d_cyc);
                hipDeviceSynchronize();
                hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
            }
            printf("working set %8zu KB, %2d lanes: %7.1f cycles (s_memtime ticks) per dependent load\n", kb, lanes, (double)cyc / steps);
        }
        hipFree(d_next);
    }
    return 0;
}
