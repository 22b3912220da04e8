// VALU issue-rate microbenchmark for gfx950 (developer tool, not part of the product).
// Each kernel runs ITER iterations of 32 independent instructions of one kind in every wave;
// 256 CUs x 8 blocks x 256 threads keep 8 waves per SIMD busy.  Prints wave-instructions per
// nanosecond per SIMD (x 1/clock = per cycle).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 4096
typedef float float2v __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

template <int KIND> __global__ void __launch_bounds__(256) k(float *out, float seed, const float *cptr)
{
    float a0 = seed + threadIdx.x, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f, a4 = a0 * 1.4f, a5 = a0 * 1.5f, a6 = a0 * 1.6f, a7 = a0 * 1.7f;
    float b = 1.0001f, c = 0.0001f;
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
    unsigned u0 = threadIdx.x * 2654435761u, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    float sc = __builtin_amdgcn_readfirstlane(cptr[0]);
    for (int i = 0; i < ITER; ++i) {
        if (KIND == 0) { // v_fma_f32
            REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
        }
        else if (KIND == 1) { // v_mul_f32 / v_add_f32 alternating
            REP8(asm volatile("v_mul_f32 %0, %0, %4\n v_add_f32 %1, %1, %5\n v_mul_f32 %2, %2, %4\n v_add_f32 %3, %3, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
        }
        else if (KIND == 2) { // v_pk_fma_f32
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));)
        }
        else if (KIND == 3) { // v_pk_mul_f32 / v_pk_add_f32
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));)
        }
        else if (KIND == 4) { // v_mul_lo_u32
            REP8(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u0 | 1));)
        }
        else if (KIND == 5) { // v_fma_f32 with an SGPR operand
            REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(sc), "v"(c));)
        }
        else if (KIND == 6) { // v_cmp writing an SGPR pair + s_or
            unsigned long long m;
            REP8(asm volatile("v_cmp_nlt_f32 %0, %1, %2\n v_cmp_nlt_f32 %0, %3, %2\n v_cmp_nlt_f32 %0, %4, %2\n v_cmp_nlt_f32 %0, %5, %2" : "=s"(m) : "v"(a0), "v"(b), "v"(a1), "v"(a2), "v"(a3));)
            if (m == 12345) a0 += 1;
        }
        else if (KIND == 7) { // dependent chain v_fma_f32 (latency with 8 waves/SIMD)
            REP32(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));)
        }
        else if (KIND == 8) { // v_sqrt_f32 + v_rcp_f32 transcendental rate
            REP8(asm volatile("v_sqrt_f32 %0, %0\n v_rcp_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + p2.x + p3.y + (float)(u0 + u1 + u2 + u3);
}

template <int KIND> void run(const char *name, float *d, const float *dc, int waves_per_simd)
{
    int blocks = 256 * waves_per_simd; // 4 waves per block -> waves_per_simd per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f, dc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f, dc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)blocks * 4 * ITER * 32; // wave-instructions
    double per_simd_per_ns = insts / 1024.0 / (ms * 1e6);
    printf("%-34s waves/SIMD=%d  %.3f ms  %.4f wave-inst/ns/SIMD  (= %.3f /cycle at 2.4 GHz, %.2f cycles each)\n", name, waves_per_simd, ms, per_simd_per_ns, per_simd_per_ns / 2.4, 2.4 / per_simd_per_ns);
}

int main()
{
    float *d, *dc;
    hipMalloc(&d, 256 * 8 * 256 * sizeof(float) * 2);
    hipMalloc(&dc, 64);
    float one = 1.0001f;
    hipMemcpy(dc, &one, 4, hipMemcpyHostToDevice);
    for (int w : {1, 2, 8}) {
        run<0>("v_fma_f32", d, dc, w);
        run<1>("v_mul_f32 + v_add_f32", d, dc, w);
        run<2>("v_pk_fma_f32", d, dc, w);
        run<3>("v_pk_mul_f32 + v_pk_add_f32", d, dc, w);
        run<4>("v_mul_lo_u32", d, dc, w);
        run<5>("v_fma_f32 (SGPR operand)", d, dc, w);
        run<6>("v_cmp_nlt_f32 -> SGPR", d, dc, w);
        run<7>("v_fma_f32 dependent chain", d, dc, w);
        run<8>("v_sqrt_f32 + v_rcp_f32", d, dc, w);
    }
    return 0;
}
