// Developer check (GPU box): the wave64 DPP scans the dense (ray, entry) pairing of rrtx_kernels.hip is built on - inclusive
// prefix sum and inclusive prefix max over the 64 lanes - against a plain loop, on random and on structured inputs.
//   hipcc --offload-arch=gfx950 -O3 -Irrt_amd/csrc tools/ubench_scan.hip -o tools/ubench_scan && tools/ubench_scan
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "rrtx_wave.h"

__global__ void scan_kernel(const uint32_t *in, uint32_t *sum, uint32_t *mx, int n_waves)
{
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w >= n_waves) return;
    const uint32_t v = in[blockIdx.x * blockDim.x + threadIdx.x];
    sum[blockIdx.x * blockDim.x + threadIdx.x] = rrtx::wave_scan_add(v);
    mx[blockIdx.x * blockDim.x + threadIdx.x] = rrtx::wave_scan_max(v);
}

int main()
{
    const int n_waves = 4096, n = n_waves * 64;
    std::vector<uint32_t> h(n), s(n), m(n);
    srand(7);
    for (int i = 0; i < n; ++i) {
        const int w = i >> 6;
        h[i] = w == 0 ? 1u : (w == 1 ? (uint32_t)(i & 63) : (w == 2 ? ((i & 63) == 0 ? 77u : 0u) : (w % 3 == 0 ? (rand() % 5 == 0 ? (uint32_t)(rand() % 61) : 0u) : (uint32_t)(rand() % 256))));
    }
    uint32_t *di, *ds, *dm;
    hipMalloc(&di, n * 4), hipMalloc(&ds, n * 4), hipMalloc(&dm, n * 4);
    hipMemcpy(di, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(scan_kernel, dim3(n / 256), dim3(256), 0, 0, di, ds, dm, n_waves);
    hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(m.data(), dm, n * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int w = 0; w < n_waves; ++w) {
        uint32_t acc = 0, mm = 0;
        for (int l = 0; l < 64; ++l) {
            acc += h[w * 64 + l], mm = h[w * 64 + l] > mm ? h[w * 64 + l] : mm;
            if (s[w * 64 + l] != acc || m[w * 64 + l] != mm) {
                if (bad < 5) printf("wave %d lane %d: sum %u (want %u) max %u (want %u)\n", w, l, s[w * 64 + l], acc, m[w * 64 + l], mm);
                bad += 1;
            }
        }
    }
    printf("wave scans: %ld wrong of %d\n", bad, n);
    return bad != 0;
}
