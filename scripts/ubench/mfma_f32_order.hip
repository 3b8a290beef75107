// Is v_mfma_f32_16x16x4_f32 / 32x32x2_f32 bitwise an fmaf chain, and in which k order?  (needed to put AC-1 on the matrix cores)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k16(const float* A, const float* B, const float* C, float* D)   // A[16][4], B[4][16], C/D[16][16]
{
    const int l = threadIdx.x;
    const float a = A[(l & 15) * 4 + (l >> 4)], b = B[(l >> 4) * 16 + (l & 15)];
    f32x4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[((l >> 4) * 4 + r) * 16 + (l & 15)];
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
__global__ void k32(const float* A, const float* B, const float* C, float* D)   // A[32][2], B[2][32], C/D[32][32]
{
    const int l = threadIdx.x;
    const float a = A[(l & 31) * 2 + (l >> 5)], b = B[(l >> 5) * 32 + (l & 31)];
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)];
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
int main()
{
    srand(1);
    auto rnd = []() { return (float)((rand() / (double)RAND_MAX - 0.5) * exp((rand() % 20 - 10) * 0.7)); };
    float *dA, *dB, *dC, *dD; hipMalloc(&dA, 4096); hipMalloc(&dB, 4096); hipMalloc(&dC, 8192); hipMalloc(&dD, 8192);
    int asc = 0, desc = 0, tot = 0, asc32 = 0, desc32 = 0, tot32 = 0, zc = 0;
    for (int trial = 0; trial < 200; ++trial) {
        float A[128], B[128], C[1024], D[1024];
        for (int i = 0; i < 128; ++i) { A[i] = rnd(); B[i] = rnd(); }
        for (int i = 0; i < 1024; ++i) C[i] = (trial & 1) ? 0.0f : rnd();
        hipMemcpy(dA, A, 512, hipMemcpyHostToDevice); hipMemcpy(dB, B, 512, hipMemcpyHostToDevice); hipMemcpy(dC, C, 4096, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD); hipMemcpy(D, dD, 1024, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            float u = C[i * 16 + j], d = C[i * 16 + j];
            for (int k = 0; k < 4; ++k) u = fmaf(A[i * 4 + k], B[k * 16 + j], u);
            for (int k = 3; k >= 0; --k) d = fmaf(A[i * 4 + k], B[k * 16 + j], d);
            ++tot; asc += (u == D[i * 16 + j]); desc += (d == D[i * 16 + j]);
        }
        hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD); hipMemcpy(D, dD, 4096, hipMemcpyDeviceToHost);
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            float u = C[i * 32 + j], d = C[i * 32 + j];
            for (int k = 0; k < 2; ++k) u = fmaf(A[i * 2 + k], B[k * 32 + j], u);
            for (int k = 1; k >= 0; --k) d = fmaf(A[i * 2 + k], B[k * 32 + j], d);
            ++tot32; asc32 += (u == D[i * 32 + j]); desc32 += (d == D[i * 32 + j]);
        }
        (void)zc;
    }
    {   // denormals: inputs, products and sums in the subnormal range
        float A[128], B[128], C[1024], D[1024];
        for (int i = 0; i < 128; ++i) { A[i] = ldexpf(1.0f + (rand() % 100) / 128.0f, -70 - rand() % 10); B[i] = ldexpf(1.0f + (rand() % 100) / 128.0f, -60 - rand() % 10); }
        for (int i = 0; i < 1024; ++i) C[i] = (i & 1) ? ldexpf(1.5f, -140) : 0.0f;
        A[0] = ldexpf(1.25f, -130);  B[0] = 0.75f;      // subnormal input
        hipMemcpy(dA, A, 512, hipMemcpyHostToDevice); hipMemcpy(dB, B, 512, hipMemcpyHostToDevice); hipMemcpy(dC, C, 4096, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD); hipMemcpy(D, dD, 1024, hipMemcpyDeviceToHost);
        int ok = 0, nz = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            float u = C[i * 16 + j];
            for (int k = 0; k < 4; ++k) u = fmaf(A[i * 4 + k], B[k * 16 + j], u);
            ok += (u == D[i * 16 + j]); nz += (u != 0.0f);
        }
        printf("16x16x4 subnormal range: %d / 256 match the host fmaf chain (%d nonzero expected), D[0][0]=%g expected %g\n", ok, nz, D[0], fmaf(A[3], B[48], fmaf(A[2], B[32], fmaf(A[1], B[16], fmaf(A[0], B[0], C[0])))));
    }
    printf("16x16x4 : ascending-k fma chain matches %d / %d, descending %d / %d\n", asc, tot, desc, tot);
    printf("32x32x2 : ascending-k fma chain matches %d / %d, descending %d / %d\n", asc32, tot32, desc32, tot32);
    return 0;
}
