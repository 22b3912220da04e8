// Developer check (GPU box): the operand and result layout of v_mfma_f32_32x32x16_f16 (gfx950) as rrtx_kernels.hip's matrix filter assumes it:
//   A (32 x 16): lane l holds A[l % 32][8 (l / 32) + j], j = 0..7;   B (16 x 32): lane l holds B[8 (l / 32) + j][l % 32];
//   D (32 x 32): lane l holds D[8 (v / 4) + 4 (l / 32) + v % 4][l % 32], v = 0..15;   two instructions chained over K = 32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const _Float16 *A, const _Float16 *B, float *D)
{
    const int l = threadIdx.x;
    f32x16 d = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int kh = 0; kh < 2; ++kh) {
        f16x8 a, b;
        for (int j = 0; j < 8; ++j) a[j] = A[(l % 32) * 32 + 16 * kh + 8 * (l / 32) + j], b[j] = B[(16 * kh + 8 * (l / 32) + j) * 32 + l % 32];
        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d, 0, 0, 0);
    }
    for (int v = 0; v < 16; ++v) D[(8 * (v / 4) + 4 * (l / 32) + v % 4) * 32 + l % 32] = d[v];
}
int main()
{
    std::vector<_Float16> A(32 * 32), B(32 * 32);
    for (int i = 0; i < 32; ++i)
        for (int kx = 0; kx < 32; ++kx) A[i * 32 + kx] = (_Float16)(0.25f * ((i * 7 + kx * 3) % 13) - 1.0f), B[kx * 32 + i] = (_Float16)(0.5f * ((i * 5 + kx * 11) % 9) - 2.0f);
    _Float16 *dA, *dB;
    float *dD;
    (void)hipMalloc(&dA, A.size() * 2), (void)hipMalloc(&dB, B.size() * 2), (void)hipMalloc(&dD, 1024 * 4);
    (void)hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice), (void)hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    std::vector<float> D(1024);
    (void)hipMemcpy(D.data(), dD, 1024 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            float want = 0;
            for (int kx = 0; kx < 32; ++kx) want += (float)A[i * 32 + kx] * (float)B[kx * 32 + j];
            if (D[i * 32 + j] != want) {
                if (bad < 5) printf("D[%d][%d] = %g, want %g\n", i, j, D[i * 32 + j], want);
                bad += 1;
            }
        }
    printf("mfma_f32_32x32x16_f16 x 2 layout: %d of 1024 wrong\n", bad);
    return bad != 0;
}
