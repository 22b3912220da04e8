// Developer check (GPU box): v_mfma_f32_32x32x16_f16 keeps SUBNORMAL f16 operands (the low pieces of the matrix filter's split operands often are:
// tests/test_filter_mfma.py models them as kept) and adds in round-to-nearest: products of a subnormal by a normal, sums that cancel, a sum of -0 terms.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float *av, const float *bv, float *out)
{
    // every lane: a[j] = av[j], b[j] = bv[j]  ->  every result = sum over the 16 k of av[k % 8] * bv[k % 8] (both halves of the lanes hold the same 8)
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) a[j] = (_Float16)av[j], b[j] = (_Float16)bv[j];
    f32x16 d = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = d[0];
}
static float run(const float a[8], const float b[8])
{
    float *da, *db, *dout, r = 0;
    (void)hipMalloc(&da, 32), (void)hipMalloc(&db, 32), (void)hipMalloc(&dout, 4);
    (void)hipMemcpy(da, a, 32, hipMemcpyHostToDevice), (void)hipMemcpy(db, b, 32, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dout);
    (void)hipMemcpy(&r, dout, 4, hipMemcpyDeviceToHost);
    (void)hipFree(da), (void)hipFree(db), (void)hipFree(dout);
    return r;
}
int main()
{
    int bad = 0;
    {   // a subnormal (2^-20) times 1024, once per half of the lanes: 2 * 2^-10
        const float a[8] = {0x1p-20f, 0, 0, 0, 0, 0, 0, 0}, b[8] = {1024.0f, 0, 0, 0, 0, 0, 0, 0};
        const float r = run(a, b);
        printf("subnormal x normal: %g (kept: %g, flushed: 0)\n", r, 2 * 0x1p-10f);
        bad += r != 2 * 0x1p-10f;
    }
    {   // the smallest subnormal (2^-24) times 60000
        const float a[8] = {0x1p-24f, 0, 0, 0, 0, 0, 0, 0}, b[8] = {60000.0f, 0, 0, 0, 0, 0, 0, 0};
        const float r = run(a, b);
        printf("smallest subnormal x 60000: %g (kept: %g)\n", r, 2 * 60000.0f * 0x1p-24f);
        bad += r != 2 * 60000.0f * 0x1p-24f;
    }
    {   // exact cancellation gives +0, a sum of -0 products gives... (the kernel reads signs: -0 would read as negative)
        const float a[8] = {3.0f, -3.0f, 0, 0, 0, 0, 0, 0}, b[8] = {5.0f, 5.0f, 0, 0, 0, 0, 0, 0};
        const float r = run(a, b);
        printf("cancellation: %g, sign bit %d (+0 wanted)\n", r, (int)std::signbit(r));
        bad += r != 0.0f || std::signbit(r);
        const float a2[8] = {-0.0f, -0.0f, -0.0f, -0.0f, -0.0f, -0.0f, -0.0f, -0.0f}, b2[8] = {1, 1, 1, 1, 1, 1, 1, 1};
        const float r2 = run(a2, b2);
        printf("sixteen -0 products on a +0 accumulator: %g, sign bit %d (+0 wanted; the filter's sums always hold a +0 or positive square term)\n", r2, (int)std::signbit(r2));
        bad += std::signbit(r2);
    }
    printf("%d unexpected\n", bad);
    return bad != 0;
}
