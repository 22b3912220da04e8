// Microbenchmark (developer tool, not part of the product): would the scan's conservative filter run
// faster on the matrix cores with bf16 x 3 split operands?  Per block of 16 spheres x 64 rays the VALU
// filter costs 16 x 8 = 128 instructions; the MFMA form costs 8 x v_mfma_f32_16x16x32_bf16 (u = C.N and
// w = C.B + g - thr for four 16-ray tiles), two ds_read_b128 for the spheres' operands, and per output
// register one v_fma (u*u + w), one v_cmp and a branch (the push path is left out: 22 % of the registers
// hold a candidate in the real scan).  Prints (ray, sphere) pairs per clock per SIMD for both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kBlocks = 31; // 488 spheres padded to 496

template <int THREADS, int GROUP> __global__ void __launch_bounds__(THREADS) k_mfma(const uint4 *btab, float *out, int segments, float seed)
{
    __shared__ uint4 b_lds[kBlocks * 2 * 64]; // [block][u|w][lane]: 62 KB
    for (int i = threadIdx.x; i < kBlocks * 2 * 64; i += THREADS) b_lds[i] = btab[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // ray operands of four 16-ray tiles, u and w (in the real kernel: split and transposed through LDS once per segment)
    bf16x8 au[4], aw[4];
    for (int t = 0; t < 4; ++t)
        for (int j = 0; j < 8; ++j) au[t][j] = (__bf16)(seed + 0.01f * (lane + t + j)), aw[t][j] = (__bf16)(seed - 0.02f * (lane + t + j));
    unsigned hits = 0;
    for (int s = 0; s < segments; ++s) {
        for (int b = 0; b < kBlocks; ++b) {
            const uint4 bu4 = b_lds[(b * 2 + 0) * 64 + lane], bw4 = b_lds[(b * 2 + 1) * 64 + lane];
            const bf16x8 bu = __builtin_bit_cast(bf16x8, bu4), bw = __builtin_bit_cast(bf16x8, bw4);
            // all eight MFMAs first, then the VALU work on their 32 result registers: the matrix pipe of this
            // wave's next block and the other waves' VALU work overlap
            f32x4 u[4], w[4];
            const f32x4 zero = {0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                u[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(au[t], bu, zero, 0, 0, 0);
                w[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[t], bw, zero, 0, 0, 0);
            }
            if (GROUP == 16) { // one branch per block of 16 result registers
                float f[16];
                unsigned long long any = 0;
#pragma unroll
                for (int q = 0; q < 16; ++q) f[q] = __builtin_fmaf(u[q >> 2][q & 3], u[q >> 2][q & 3], w[q >> 2][q & 3]), any |= __ballot(!(f[q] < 0.0f));
                if (__builtin_expect(any != 0ull, 0))
                    for (int q = 0; q < 16; ++q) hits += !(f[q] < 0.0f) ? 1u : 0u;
            }
            else
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (GROUP) { // one branch per four result registers
                    float f[4];
                    unsigned long long any = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) f[r] = __builtin_fmaf(u[t][r], u[t][r], w[t][r]), any |= __ballot(!(f[r] < 0.0f));
                    if (__builtin_expect(any != 0ull, 0))
                        for (int r = 0; r < 4; ++r) hits += !(f[r] < 0.0f) ? 1u : 0u;
                }
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float f = __builtin_fmaf(u[t][r], u[t][r], w[t][r]);
                        if (__builtin_expect(__ballot(!(f < 0.0f)) != 0ull, 0)) hits += !(f < 0.0f) ? 1u : 0u; // (stands for the push)
                    }
                }
            }
        }
        au[0][0] = (__bf16)((float)au[0][0] + 1e-3f); // (keeps the loop from being hoisted)
    }
    out[blockIdx.x * THREADS + threadIdx.x] = (float)hits;
}

// the VALU filter as the product runs it: 7 FMAs + compare per sphere, sphere operands in SGPRs (here: LDS broadcast reads)
template <int THREADS, bool GROUP> __global__ void __launch_bounds__(THREADS) k_valu(const float4 *stab, float *out, int segments, float seed)
{
    __shared__ float4 s_lds[kBlocks * 16];
    for (int i = threadIdx.x; i < kBlocks * 16; i += THREADS) s_lds[i] = stab[i];
    __syncthreads();
    float nx = seed + threadIdx.x * 1e-3f, ny = seed * 0.5f, nz = 0.3f, bx = 0.1f, by = 0.2f, bz = 0.3f, g = -1e9f;
    unsigned hits = 0;
    for (int s = 0; s < segments; ++s) {
        for (int b = 0; b < kBlocks * 16; b += 4) {
            float f[4], thr[4];
            unsigned long long any = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 c = s_lds[b + q];
                const float u = __builtin_fmaf(c.z, nz, __builtin_fmaf(c.y, ny, c.x * nx));
                const float w = __builtin_fmaf(bz, c.z, __builtin_fmaf(by, c.y, __builtin_fmaf(bx, c.x, g)));
                f[q] = __builtin_fmaf(u, u, w), thr[q] = c.w;
                if (!GROUP && __builtin_expect(__ballot(!(f[q] < thr[q])) != 0ull, 0)) hits += !(f[q] < thr[q]) ? 1u : 0u;
                if (GROUP) any |= __ballot(!(f[q] < thr[q]));
            }
            if (GROUP && __builtin_expect(any != 0ull, 0))
                for (int q = 0; q < 4; ++q) hits += !(f[q] < thr[q]) ? 1u : 0u;
        }
        nx += 1e-3f;
    }
    out[blockIdx.x * THREADS + threadIdx.x] = (float)hits;
}

// the matrix pipe alone: the same loads and MFMAs, results only chained through the accumulators
template <int THREADS> __global__ void __launch_bounds__(THREADS) k_mfma_only(const uint4 *btab, float *out, int segments, float seed)
{
    __shared__ uint4 b_lds[kBlocks * 2 * 64];
    for (int i = threadIdx.x; i < kBlocks * 2 * 64; i += THREADS) b_lds[i] = btab[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    bf16x8 au[4], aw[4];
    for (int t = 0; t < 4; ++t)
        for (int j = 0; j < 8; ++j) au[t][j] = (__bf16)(seed + 0.01f * (lane + t + j)), aw[t][j] = (__bf16)(seed - 0.02f * (lane + t + j));
    f32x4 u[4] = {}, w[4] = {};
    for (int s = 0; s < segments; ++s)
        for (int b = 0; b < kBlocks; ++b) {
            asm volatile("" ::: "memory"); // (the loads stay in the loop)
            const uint4 bu4 = b_lds[(b * 2 + 0) * 64 + lane], bw4 = b_lds[(b * 2 + 1) * 64 + lane];
            const bf16x8 bu = __builtin_bit_cast(bf16x8, bu4), bw = __builtin_bit_cast(bf16x8, bw4);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                u[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(au[t], bu, u[t], 0, 0, 0);
                w[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[t], bw, w[t], 0, 0, 0);
            }
        }
    float acc = 0;
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 4; ++r) acc += u[t][r] + w[t][r];
    out[blockIdx.x * THREADS + threadIdx.x] = acc;
}

template <typename K> double time_ms(K launch)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    const int segments = 2000;
    std::vector<uint4> hb(kBlocks * 2 * 64);
    for (size_t i = 0; i < hb.size(); ++i) hb[i] = uint4{0xBF80BF80u, 0xBF80BF80u, 0xBF80BF80u, 0xBF80BF80u}; // bf16 -1.0: f = u*u + w stays negative... mostly
    std::vector<float4> hs(kBlocks * 16, float4{1.f, 2.f, 3.f, 1e30f});
    uint4 *db;
    float4 *ds;
    float *dout;
    hipMalloc(&db, hb.size() * sizeof(uint4)), hipMalloc(&ds, hs.size() * sizeof(float4)), hipMalloc(&dout, 256 * 8 * 1024 * sizeof(float));
    hipMemcpy(db, hb.data(), hb.size() * sizeof(uint4), hipMemcpyHostToDevice);
    hipMemcpy(ds, hs.data(), hs.size() * sizeof(float4), hipMemcpyHostToDevice);
    const double clk = 2.4e9, pairs_per_wave = (double)segments * kBlocks * 16 * 64;
    {
        double ms = time_ms([&] { hipLaunchKernelGGL((k_mfma<1024, 0>), dim3(256), dim3(1024), 0, 0, db, dout, segments, 0.5f); });
        printf("mfma bf16 filter, 4 waves/SIMD, branch per result register: %.3f ms  %.2f pairs/clk/SIMD\n", ms, pairs_per_wave * 4 / (ms * 1e-3 * clk));
        ms = time_ms([&] { hipLaunchKernelGGL((k_mfma<1024, 4>), dim3(256), dim3(1024), 0, 0, db, dout, segments, 0.5f); });
        printf("mfma bf16 filter, 4 waves/SIMD, branch per four registers:  %.3f ms  %.2f pairs/clk/SIMD\n", ms, pairs_per_wave * 4 / (ms * 1e-3 * clk));
        ms = time_ms([&] { hipLaunchKernelGGL((k_mfma<1024, 16>), dim3(256), dim3(1024), 0, 0, db, dout, segments, 0.5f); });
        printf("mfma bf16 filter, 4 waves/SIMD, branch per block (16 registers): %.3f ms  %.2f pairs/clk/SIMD\n", ms, pairs_per_wave * 4 / (ms * 1e-3 * clk));
        ms = time_ms([&] { hipLaunchKernelGGL((k_mfma_only<1024>), dim3(256), dim3(1024), 0, 0, db, dout, segments, 0.5f); });
        printf("the MFMAs and their LDS reads alone, 4 waves/SIMD:           %.3f ms  %.2f pairs/clk/SIMD\n", ms, pairs_per_wave * 4 / (ms * 1e-3 * clk));
        ms = time_ms([&] { hipLaunchKernelGGL((k_mfma<512, 4>), dim3(256), dim3(512), 0, 0, db, dout, segments, 0.5f); });
        printf("mfma bf16 filter, 2 waves/SIMD, branch per four registers:  %.3f ms  %.2f pairs/clk/SIMD\n", ms, pairs_per_wave * 2 / (ms * 1e-3 * clk));
    }
    for (int wps : {4, 6}) {
        double ms = time_ms([&] { hipLaunchKernelGGL((k_valu<256, false>), dim3(256 * wps), dim3(256), 0, 0, ds, dout, segments, 0.5f); });
        printf("valu filter (LDS operands), %d waves/SIMD, branch per sphere:       %.3f ms  %.2f pairs/clk/SIMD\n", wps, ms, pairs_per_wave * wps / (ms * 1e-3 * clk));
        ms = time_ms([&] { hipLaunchKernelGGL((k_valu<256, true>), dim3(256 * wps), dim3(256), 0, 0, ds, dout, segments, 0.5f); });
        printf("valu filter (LDS operands), %d waves/SIMD, branch per four spheres: %.3f ms  %.2f pairs/clk/SIMD\n", wps, ms, pairs_per_wave * wps / (ms * 1e-3 * clk));
    }
    return 0;
}
