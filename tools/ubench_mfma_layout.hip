// Developer check (GPU box): the operand and result layout of v_mfma_f32_16x16x32_f16 (gfx950) as rrtx_kernels.hip's matrix filter assumes it:
//   A (16 x 32): lane l holds A[l % 16][8 (l / 16) + j], j = 0..7;   B (32 x 16): lane l holds B[8 (l / 16) + j][l % 16];
//   D (16 x 16): lane l holds D[4 (l / 16) + r][l % 16], r = 0..3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const _Float16 *A, const _Float16 *B, float *D)
{
    const int l = threadIdx.x;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) a[j] = A[(l % 16) * 32 + 8 * (l / 16) + j], b[j] = B[(8 * (l / 16) + j) * 16 + l % 16];
    const f32x4 zero = {0, 0, 0, 0};
    const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, zero, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l / 16) + r) * 16 + l % 16] = d[r];
}
int main()
{
    std::vector<_Float16> A(16 * 32), B(32 * 16);
    for (int i = 0; i < 16; ++i)
        for (int kx = 0; kx < 32; ++kx) A[i * 32 + kx] = (_Float16)(0.25f * ((i * 7 + kx * 3) % 13) - 1.0f), B[kx * 16 + i] = (_Float16)(0.5f * ((i * 5 + kx * 11) % 9) - 2.0f);
    _Float16 *dA, *dB;
    float *dD;
    (void)hipMalloc(&dA, A.size() * 2), (void)hipMalloc(&dB, B.size() * 2), (void)hipMalloc(&dD, 256 * 4);
    (void)hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice), (void)hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    std::vector<float> D(256);
    (void)hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            float want = 0;
            for (int kx = 0; kx < 32; ++kx) want += (float)A[i * 32 + kx] * (float)B[kx * 16 + j];
            if (D[i * 16 + j] != want) {
                if (bad < 5) printf("D[%d][%d] = %g, want %g\n", i, j, D[i * 16 + j], want);
                bad += 1;
            }
        }
    printf("mfma_f32_16x16x32_f16 layout: %d of 256 wrong\n", bad);
    return bad != 0;
}
