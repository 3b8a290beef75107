// Do VALU instructions execute while a v_mfma_f32_32x32x2_f32 is in flight on the same SIMD?
//   A: one wave, N dependent MFMAs (one accumulator)                      B: one wave, M dependent-free VALU fmas (8 chains)
//   C: one wave, the two interleaved in ONE instruction stream (8 fmas after every MFMA)
//   D: two waves on one SIMD (waves 0 and 4 of a 512-thread workgroup): wave 0 runs A's loop, wave 4 runs B's loop, both timed
//   E: as D but wave 4 also runs MFMAs (two MFMA streams on one SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int N = 2048;
__device__ __forceinline__ f32x16 mfma_loop(f32x16 acc, float a, float b, int n)
{
#pragma unroll 8
    for (int i = 0; i < n; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ void valu_loop(float (&v)[8], float m, int n)
{
#pragma unroll 4
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], m, 1.0f);
    }
}
__global__ void __launch_bounds__(512) k(float* out, unsigned long long* cyc, int mode)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x16 acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float v[8]; for (int j = 0; j < 8; ++j) v[j] = (float)lane * 1e-3f + j;
    const float a = 1.0f + lane * 1e-4f, b = 0.5f, m = 0.999f;
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0;
    if (mode == 0 && wave == 0) { t0 = __builtin_amdgcn_s_memtime(); acc = mfma_loop(acc, a, b, N); asm volatile("" :: "v"(acc)); t1 = __builtin_amdgcn_s_memtime(); }
    if (mode == 1 && wave == 0) { t0 = __builtin_amdgcn_s_memtime(); valu_loop(v, m, N); asm volatile("" :: "v"(v[0])); t1 = __builtin_amdgcn_s_memtime(); }
    if (mode == 2 && wave == 0) {
        t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 4
        for (int i = 0; i < N; ++i) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], m, 1.0f);
        }
        asm volatile("" :: "v"(acc), "v"(v[0]));
        t1 = __builtin_amdgcn_s_memtime();
    }
    if (mode == 3 || mode == 4) {
        if (wave == 0) { t0 = __builtin_amdgcn_s_memtime(); acc = mfma_loop(acc, a, b, N); asm volatile("" :: "v"(acc)); t1 = __builtin_amdgcn_s_memtime(); }
        if (wave == 4 && mode == 3) { t0 = __builtin_amdgcn_s_memtime(); valu_loop(v, m, N); asm volatile("" :: "v"(v[0])); t1 = __builtin_amdgcn_s_memtime(); }
        if (wave == 4 && mode == 4) { t0 = __builtin_amdgcn_s_memtime(); acc = mfma_loop(acc, a, b, N); asm volatile("" :: "v"(acc)); t1 = __builtin_amdgcn_s_memtime(); }
    }
    float s = 0.f; for (int i = 0; i < 16; ++i) s += acc[i]; for (int j = 0; j < 8; ++j) s += v[j];
    out[threadIdx.x] = s;
    if (lane == 0) cyc[mode * 8 + wave] = t1 - t0;
    { unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); if (lane == 0 && mode == 0) cyc[40 + wave] = (hw >> 4) & 3; }
}
int main()
{
    float* out; unsigned long long* dc, hc[48];
    (void)hipMalloc(&out, 512 * 4); (void)hipMalloc(&dc, sizeof hc); (void)hipMemset(dc, 0, sizeof hc);
    for (int r = 0; r < 2; ++r)
        for (int mode = 0; mode < 5; ++mode) hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, out, dc, mode);
    (void)hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost);
    printf("per loop trip (1 MFMA 32x32x2 f32 and/or 8 independent fmas), s_memtime ticks:\n");
    printf("A  lone wave, MFMAs only            %.1f\n", hc[0] / (double)N);
    printf("B  lone wave, 8 fmas only           %.1f\n", hc[8] / (double)N);
    printf("C  lone wave, MFMA + 8 fmas mixed   %.1f\n", hc[16] / (double)N);
    printf("D  two waves on one SIMD: MFMA wave %.1f, VALU wave %.1f\n", hc[24] / (double)N, hc[28] / (double)N);
    printf("E  two MFMA waves on one SIMD:      %.1f and %.1f\n", hc[32] / (double)N, hc[36] / (double)N);
    printf("SIMD of waves 0..7:"); for (int w = 0; w < 8; ++w) printf(" %llu", hc[40 + w]); printf("\n");
    return 0;
}
