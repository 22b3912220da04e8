// Microbenchmark (developer tool): the scan's conservative filter as ONE matrix product per tile.  With the discriminant written as a single
// dot product - the six monomials c_i c_j against n_i n_j, the three b_i c_i, g and the threshold, every f32 operand split into two f16
// pieces with the three leading cross products kept: 31 terms - a (16 spheres x 16 rays) tile is one v_mfma_f32_16x16x32_f16 whose result
// IS the filter's value: per result register only a compare is left.  Per block of 16 spheres x 64 rays: one ds_read_b128 (the spheres'
// operand), four MFMAs, 16 compares.  Prints (ray, sphere) pairs per clock per SIMD next to the VALU filter's (tools/ubench_mfma.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kBlocks = 31; // 488 spheres padded to 496

template <int THREADS, int PIPE> __global__ void __launch_bounds__(THREADS) k_mfma1(const uint4 *atab, float *out, int segments, float seed)
{
    __shared__ uint4 a_lds[kBlocks * 64]; // [block][lane]: 31 KB
    for (int i = threadIdx.x; i < kBlocks * 64; i += THREADS) a_lds[i] = atab[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f16x8 br[4]; // the rays' operands of four 16-ray tiles (in the real kernel: split and transposed once per segment)
    for (int t = 0; t < 4; ++t)
        for (int j = 0; j < 8; ++j) br[t][j] = (_Float16)(seed + 0.01f * (lane + t + j));
    unsigned hits = 0;
    const f32x4 zero = {0, 0, 0, 0};
    for (int s = 0; s < segments; ++s) {
        if (PIPE == 0) {
            for (int b = 0; b < kBlocks; ++b) {
                const f16x8 a = __builtin_bit_cast(f16x8, a_lds[b * 64 + lane]);
                f32x4 f[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) f[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, br[t], zero, 0, 0, 0);
                unsigned long long any = 0;
#pragma unroll
                for (int q = 0; q < 16; ++q) any |= __ballot(!(f[q >> 2][q & 3] < 0.0f));
                if (__builtin_expect(any != 0ull, 0))
                    for (int q = 0; q < 16; ++q) hits += !(f[q >> 2][q & 3] < 0.0f) ? 1u : 0u;
            }
        }
        else { // software pipeline: block b + 1's products are under way while block b's results are compared
            f32x4 f[4], g[4];
            f16x8 a = __builtin_bit_cast(f16x8, a_lds[lane]);
#pragma unroll
            for (int t = 0; t < 4; ++t) f[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, br[t], zero, 0, 0, 0);
            for (int b = 0; b < kBlocks; ++b) {
                a = __builtin_bit_cast(f16x8, a_lds[((b + 1 < kBlocks ? b + 1 : 0)) * 64 + lane]);
#pragma unroll
                for (int t = 0; t < 4; ++t) g[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, br[t], zero, 0, 0, 0);
                unsigned long long any = 0;
#pragma unroll
                for (int q = 0; q < 16; ++q) any |= __ballot(!(f[q >> 2][q & 3] < 0.0f));
                if (__builtin_expect(any != 0ull, 0))
                    for (int q = 0; q < 16; ++q) hits += !(f[q >> 2][q & 3] < 0.0f) ? 1u : 0u;
#pragma unroll
                for (int t = 0; t < 4; ++t) f[t] = g[t];
            }
        }
        br[0][0] = (_Float16)((float)br[0][0] + 1e-3f); // (keeps the loop from being hoisted)
    }
    out[blockIdx.x * THREADS + threadIdx.x] = (float)hits;
}

template <typename K> double time_ms(K launch)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    const int segments = 2000;
    std::vector<uint4> ha(kBlocks * 64);
    for (size_t i = 0; i < ha.size(); ++i) ha[i] = uint4{0xBC00BC00u, 0xBC00BC00u, 0xBC00BC00u, 0xBC00BC00u}; // f16 -1.0: the value stays negative ... mostly
    uint4 *da;
    float *dout;
    (void)hipMalloc(&da, ha.size() * sizeof(uint4)), (void)hipMalloc(&dout, 256 * 8 * 1024 * sizeof(float));
    (void)hipMemcpy(da, ha.data(), ha.size() * sizeof(uint4), hipMemcpyHostToDevice);
    const double clk = 2.4e9, pairs_per_wave = (double)segments * kBlocks * 16 * 64;
    for (int wps : {4, 5, 6}) {
        double ms = 0;
        if (wps == 4) ms = time_ms([&] { hipLaunchKernelGGL((k_mfma1<1024, 0>), dim3(256), dim3(1024), 0, 0, da, dout, segments, 0.5f); });
        if (wps == 5) ms = time_ms([&] { hipLaunchKernelGGL((k_mfma1<256, 0>), dim3(256 * 5), dim3(256), 0, 0, da, dout, segments, 0.5f); });
        if (wps == 6) ms = time_ms([&] { hipLaunchKernelGGL((k_mfma1<256, 0>), dim3(256 * 6), dim3(256), 0, 0, da, dout, segments, 0.5f); });
        printf("one f16 MFMA per tile, %d waves/SIMD:              %.3f ms  %.2f pairs/clk/SIMD\n", wps, ms, pairs_per_wave * wps / (ms * 1e-3 * clk));
        if (wps == 4) ms = time_ms([&] { hipLaunchKernelGGL((k_mfma1<1024, 1>), dim3(256), dim3(1024), 0, 0, da, dout, segments, 0.5f); });
        if (wps == 5) ms = time_ms([&] { hipLaunchKernelGGL((k_mfma1<256, 1>), dim3(256 * 5), dim3(256), 0, 0, da, dout, segments, 0.5f); });
        if (wps == 6) ms = time_ms([&] { hipLaunchKernelGGL((k_mfma1<256, 1>), dim3(256 * 6), dim3(256), 0, 0, da, dout, segments, 0.5f); });
        printf("one f16 MFMA per tile, pipelined, %d waves/SIMD:   %.3f ms  %.2f pairs/clk/SIMD\n", wps, ms, pairs_per_wave * wps / (ms * 1e-3 * clk));
    }
    return 0;
}
