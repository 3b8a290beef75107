// What is the issue rate of v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 on gfx950 -- per dependent chain, per wave with several
// independent accumulators, per SIMD with several waves -- and what does the whole chip sustain (FLOP/s by the wall clock)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(256) k32(float* out, unsigned long long* cyc, int n)
{
#ifdef __HIP_DEVICE_COMPILE__
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    const float a = 1.0f + threadIdx.x * 1e-4f, b = 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    for (int j = 0; j < NACC; ++j) asm volatile("" :: "v"(acc[j]));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f; for (int j = 0; j < NACC; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
#endif
}
template <int NACC>
__global__ void __launch_bounds__(256) k16(float* out, unsigned long long* cyc, int n)
{
#ifdef __HIP_DEVICE_COMPILE__
    f32x4 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int i = 0; i < 4; ++i) acc[j][i] = 0.f;
    const float a = 1.0f + threadIdx.x * 1e-4f, b = 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
    }
    for (int j = 0; j < NACC; ++j) asm volatile("" :: "v"(acc[j]));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f; for (int j = 0; j < NACC; ++j) for (int i = 0; i < 4; ++i) s += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
#endif
}
template __global__ void k32<1>(float*, unsigned long long*, int);
template __global__ void k32<2>(float*, unsigned long long*, int);
template __global__ void k32<4>(float*, unsigned long long*, int);
template __global__ void k16<1>(float*, unsigned long long*, int);
template __global__ void k16<4>(float*, unsigned long long*, int);
#define RUN(name, kern, blocks, threads, nacc, flop_per_mfma)                                                                              \
    do {                                                                                                                               \
        const int n = 4096; void (*kp)(float*, unsigned long long*, int) = kern;                                                      \
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);                                                       \
        hipLaunchKernelGGL(kp, dim3(blocks), dim3(threads), 0, 0, out, dc, n);                                                       \
        (void)hipEventRecord(e0);                                                                                                      \
        hipLaunchKernelGGL(kp, dim3(blocks), dim3(threads), 0, 0, out, dc, n);                                                       \
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);                                                                       \
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);                                                                              \
        unsigned long long hc; (void)hipMemcpy(&hc, dc, 8, hipMemcpyDeviceToHost);                                                     \
        const double mfmas = (double)n * nacc;                                                                                         \
        const double total = mfmas * (threads / 64) * blocks * flop_per_mfma;                                                          \
        printf("%-48s blocks %4d x %d waves: %6.1f ticks per MFMA (wave 0), %7.3f ms -> %7.1f TFLOP/s, tick rate %.0f MHz\n", name, blocks, \
               threads / 64, hc / mfmas, ms, total / ms / 1e9, hc / (ms * 1e3));                                                       \
    } while (0)
int main()
{
    float* out; unsigned long long* dc;
    (void)hipMalloc(&out, 4096 * 1024 * 4); (void)hipMalloc(&dc, 64);
    RUN("32x32x2, 1 accumulator, 1 wave", k32<1>, 1, 64, 1, 4096.0);
    RUN("32x32x2, 2 accumulators, 1 wave", k32<2>, 1, 64, 2, 4096.0);
    RUN("32x32x2, 4 accumulators, 1 wave", k32<4>, 1, 64, 4, 4096.0);
    RUN("32x32x2, 1 accumulator, 4 waves (1 per SIMD)", k32<1>, 1, 256, 1, 4096.0);
    RUN("32x32x2, 1 accumulator, chip, 1 wave per SIMD", k32<1>, 256, 256, 1, 4096.0);
    RUN("32x32x2, 1 accumulator, chip, 2 waves per SIMD", k32<1>, 512, 256, 1, 4096.0);
    RUN("32x32x2, 4 accumulators, chip, 1 wave per SIMD", k32<4>, 256, 256, 4, 4096.0);
    RUN("32x32x2, 4 accumulators, chip, 2 waves per SIMD", k32<4>, 512, 256, 4, 4096.0);
    RUN("16x16x4, 1 accumulator, 1 wave", k16<1>, 1, 64, 1, 2048.0);
    RUN("16x16x4, 4 accumulators, 1 wave", k16<4>, 1, 64, 4, 2048.0);
    RUN("16x16x4, 4 accumulators, chip, 2 waves per SIMD", k16<4>, 512, 256, 4, 2048.0);
    return 0;
}
