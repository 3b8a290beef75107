// Do VALU instructions of one wave execute while ANOTHER wave's v_mfma_f32_32x32x2_f32 is in flight on the same SIMD?
// 512 workgroups x 4 waves on 256 CUs = 2 waves per SIMD (mfma_f32_rate.hip: two MFMA waves per SIMD see 128 ticks per MFMA each).
// role(block): 0 = MFMA loop, 1 = VALU loop (8 independent fma chains), 2 = VALU loop with v_exp_f32 (transcendental)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int N = 4096;
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, int role_lo, int role_hi)
{
#ifdef __HIP_DEVICE_COMPILE__
    const int role = blockIdx.x < 256 ? role_lo : role_hi;
    f32x16 acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float v[8]; for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 1e-3f + j;
    const float a = 1.0f + threadIdx.x * 1e-4f, b = 0.5f, m = 0.999f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 0) {
#pragma unroll 8
        for (int i = 0; i < N; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        asm volatile("" :: "v"(acc));
    } else if (role == 1) {
#pragma unroll 4
        for (int i = 0; i < N; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], m, 1.0f);
        }
        asm volatile("" :: "v"(v[0]));
    } else {
#pragma unroll 4
        for (int i = 0; i < N; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_exp2f(v[j] * m);
        }
        asm volatile("" :: "v"(v[0]));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f; for (int i = 0; i < 16; ++i) s += acc[i]; for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 256)) cyc[blockIdx.x / 256] = t1 - t0;
#endif
}
int main()
{
    float* out; unsigned long long* dc, hc[2];
    (void)hipMalloc(&out, 512 * 256 * 4); (void)hipMalloc(&dc, 16);
    const char* nm[3] = {"MFMA", "fma x8", "exp x8 (2 instr each)"};
    const int combos[6][2] = {{0, 0}, {1, 1}, {2, 2}, {0, 1}, {0, 2}, {1, 2}};
    for (int c = 0; c < 6; ++c) {
        hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, out, dc, combos[c][0], combos[c][1]);
        hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, out, dc, combos[c][0], combos[c][1]);
        (void)hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost);
        printf("SIMD-mates %-22s + %-22s : %7.1f and %7.1f ticks per loop trip\n", nm[combos[c][0]], nm[combos[c][1]], hc[0] / (double)N, hc[1] / (double)N);
    }
    return 0;
}
