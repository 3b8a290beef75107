// Micro-benchmarks of the chain wave's primitives (tuning aid, not product code).  One wave, s_memtime around N iterations.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define N 2000
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
extern __shared__ __attribute__((aligned(16))) float lds[];
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDS4(o) (((__attribute__((address_space(3))) f32x4*)lds)[(o)])

template <int MODE>
__global__ void k(float* out, unsigned long long* cyc, const float* w)
{
    const int lane = threadIdx.x & 63;
    float wr[32];
    for (int i = 0; i < 32; ++i) wr[i] = w[i * 64 + lane];
    float x = w[lane] * 0.01f;
    lds[lane] = x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < N; ++it) {
        float acc = 0.f;
        if (MODE == 0) {            // 32 dependent fmac, VGPR operands only
#pragma unroll
            for (int c = 0; c < 32; ++c) acc = fma_(wr[c], x, acc);
        } else if (MODE == 1) {     // readlane + dependent fmac (current)
#pragma unroll
            for (int c = 0; c < 32; ++c) acc = fma_(wr[c], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), c)), acc);
        } else if (MODE == 2) {     // 4 interleaved chains with readlane
            float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
            for (int c = 0; c < 32; c += 4) {
                a0 = fma_(wr[c], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), c)), a0);
                a1 = fma_(wr[c + 1], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), c + 1)), a1);
                a2 = fma_(wr[c + 2], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), c + 2)), a2);
                a3 = fma_(wr[c + 3], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), c + 3)), a3);
            }
            acc = (a0 + a1) + (a2 + a3);
        } else if (MODE == 3) {     // LDS write + broadcast b128 reads + dependent chain
            lds[64 + lane] = x;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int kq = 0; kq < 8; ++kq) {
                const f32x4 q = LDS4(16 + kq);
                acc = fma_(wr[4 * kq], q.x, acc); acc = fma_(wr[4 * kq + 1], q.y, acc);
                acc = fma_(wr[4 * kq + 2], q.z, acc); acc = fma_(wr[4 * kq + 3], q.w, acc);
            }
        } else if (MODE == 4) {     // LDS write + broadcast reads + 4 interleaved chains
            lds[64 + lane] = x;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
            for (int kq = 0; kq < 8; ++kq) {
                const f32x4 q = LDS4(16 + kq);
                a0 = fma_(wr[4 * kq], q.x, a0); a1 = fma_(wr[4 * kq + 1], q.y, a1);
                a2 = fma_(wr[4 * kq + 2], q.z, a2); a3 = fma_(wr[4 * kq + 3], q.w, a3);
            }
            acc = (a0 + a1) + (a2 + a3);
        } else if (MODE == 5) {     // 32 readlanes only (no fma dependency on them)
            float s = 0;
#pragma unroll
            for (int c = 0; c < 32; ++c) s += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), c));
            acc = s;
        } else if (MODE == 6) {     // rational activation + permlane swap + mul (the gated unit)
            float v = x;
            float x2 = v * v, p = fma_(x2, 1e-3f, 2e-3f);
            p = fma_(x2, p, 1e-3f); p = fma_(x2, p, 1e-3f); p = fma_(x2, p, 1e-3f); p = fma_(x2, p, 1e-3f); p = fma_(x2, p, 1e-3f); p = v * p;
            float q = fma_(x2, 1e-3f, 2e-3f); q = fma_(x2, q, 1e-3f); q = fma_(x2, q, 1e-3f); q = fma_(x2, q, 1e-3f); q = fma_(x2, q, 1.0f);
            float r = __fdiv_rn(p, q);
            auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(r), __float_as_uint(r), false, false);
            acc = __uint_as_float(sw[0]) * __uint_as_float(sw[1]);
        } else if (MODE == 8) {     // all 32 readlanes first (kept ahead by a scheduling barrier), then the sequential chain
            float sv[32];
#pragma unroll
            for (int c = 0; c < 32; ++c) sv[c] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), c));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 32; ++c) acc = fma_(wr[c], sv[c], acc);
        } else if (MODE == 9) {     // 2 interleaved chains + readlane
            float a0 = 0, a1 = 0;
#pragma unroll
            for (int c = 0; c < 32; c += 2) {
                a0 = fma_(wr[c], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), c)), a0);
                a1 = fma_(wr[c + 1], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), c + 1)), a1);
            }
            acc = a0 + a1;
        } else if (MODE == 10) {    // 8 interleaved chains + readlane
            float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < 32; c += 8)
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = fma_(wr[c + j], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), c + j)), a[j]);
            acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        } else if (MODE == 11) {    // sequential chain, readlanes software-pipelined 4 ahead
            float s0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 0)), s1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 1));
            float s2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 2)), s3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 3));
#pragma unroll
            for (int c = 0; c < 32; c += 4) {
                const float n0 = c + 4 < 32 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), (c + 4) & 31)) : 0.f;
                acc = fma_(wr[c], s0, acc);
                const float n1 = c + 4 < 32 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), (c + 5) & 31)) : 0.f;
                acc = fma_(wr[c + 1], s1, acc);
                const float n2 = c + 4 < 32 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), (c + 6) & 31)) : 0.f;
                acc = fma_(wr[c + 2], s2, acc);
                const float n3 = c + 4 < 32 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), (c + 7) & 31)) : 0.f;
                acc = fma_(wr[c + 3], s3, acc);
                __builtin_amdgcn_sched_barrier(0);
                s0 = n0; s1 = n1; s2 = n2; s3 = n3;
            }
        } else if (MODE == 7) {     // 16 x ds_read_b128 (a tile + a half tile), dependent use
            float s = 0;
#pragma unroll
            for (int kq = 0; kq < 16; ++kq) { const f32x4 q = LDS4(64 + kq * 64 + lane); s += q.x + q.w; }
            acc = s;
        }
        x = acc * 1e-3f + 0.01f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
    float *out, *w; unsigned long long* cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 64); hipMalloc(&w, 64 * 2048 * 4);
    std::vector<float> hw(64 * 2048);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 7919) % 1000) * 1e-3f - 0.5f;
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    const char* names[] = {"32 dep fmac (VGPR)", "32 x (readlane+dep fmac)", "4 interleaved chains + readlane", "LDS bcast + dep chain",
                           "LDS bcast + 4 chains", "32 readlanes + adds", "gated unit (rational, div, swap)", "16 ds_read_b128 + use", "readlanes hoisted + seq chain", "2 chains + readlane", "8 chains + readlane", "seq chain, readlane 4 ahead"};
#define RUN(M) { hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 65536, 0, out, cyc, w); hipDeviceSynchronize(); \
    hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 65536, 0, out, cyc, w); hipDeviceSynchronize(); \
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-36s %8.1f ticks/iter\n", names[M], (double)c / N); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(8) RUN(9) RUN(10) RUN(11)
    return 0;
}
