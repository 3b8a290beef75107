// What does a SEQUENTIAL float64 sum over 256 class values cost on a lone wave (the one-hot sampler's softmax sum / cdf cumsum)?
// A: dependent v_add_f64 chain, operands already uniform in registers        B: v_readlane_b32 x2 (SGPR lane index) + v_add_f64
// C: v_readlane_b32 x2 (constant lane index, unrolled) + v_add_f64           D: LDS broadcast reads (8 ahead) + v_add_f64
// E: dependent v_add_f32 chain (calibration)                                 F: B with the cumsum's per-lane capture (v_cmp + 2 v_cndmask)
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double rl64(double v, int l)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__global__ void __launch_bounds__(64) k(const double* in, double* out, unsigned long long* cyc, int Q)
{
    __shared__ double tab[1024];
    const int lane = threadIdx.x;
    double v[4];
    for (int kk = 0; kk < 4; ++kk) { v[kk] = in[lane + 64 * kk]; tab[lane + 64 * kk] = v[kk]; }
    __syncthreads();
    unsigned long long t0, t1;
    double acc;
    // A
    {
        double x0 = v[0], x1 = v[1];
        acc = 0.0;
        t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
        for (int i = 0; i < Q; i += 8) {
            acc += x0; acc += x1; acc += x0; acc += x1; acc += x0; acc += x1; acc += x0; acc += x1;
            asm volatile("" : "+v"(acc));
        }
        t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { cyc[0] = t1 - t0; out[0] = acc; }
    }
    // B
    {
        acc = 0.0;
        t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll 1
            for (int l = 0; l < 64; l += 8) {
                double x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = rl64(v[kk], l + j);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += x[j];
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { cyc[1] = t1 - t0; out[1] = acc; }
    }
    // C
    {
        acc = 0.0;
        t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int l = 0; l < 64; ++l) acc += rl64(v[kk], l);
        }
        t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { cyc[2] = t1 - t0; out[2] = acc; }
    }
    // D
    {
        acc = 0.0;
        t0 = __builtin_amdgcn_s_memtime();
        double nx[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) nx[j] = tab[j];
#pragma unroll 1
        for (int i = 0; i < Q; i += 8) {
            double cur[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { cur[j] = nx[j]; nx[j] = tab[i + 8 + j]; }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += cur[j];
        }
        t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { cyc[3] = t1 - t0; out[3] = acc; }
    }
    // E
    {
        float a = 0.0f, x0 = (float)v[0], x1 = (float)v[1];
        t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
        for (int i = 0; i < Q; i += 8) {
            a += x0; a += x1; a += x0; a += x1; a += x0; a += x1; a += x0; a += x1;
            asm volatile("" : "+v"(a));
        }
        t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { cyc[4] = t1 - t0; out[4] = a; }
    }
    // F
    {
        acc = 0.0;
        double c[4] = {0, 0, 0, 0};
        t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll 1
            for (int l = 0; l < 64; l += 8) {
                double x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = rl64(v[kk], l + j);
#pragma unroll
                for (int j = 0; j < 8; ++j) { acc += x[j]; c[kk] = lane == l + j ? acc : c[kk]; }
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) cyc[5] = t1 - t0;
        out[64 + lane] = c[0] + c[1] + c[2] + c[3];
    }
}
int main()
{
    double h[256], *din, *dout; unsigned long long* dc, hc[8];
    for (int i = 0; i < 256; ++i) h[i] = 1.0 / (1 + i);
    hipMalloc(&din, sizeof h); hipMalloc(&dout, 256 * 8); hipMalloc(&dc, 64);
    hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, dc, 256);
        hipMemcpy(hc, dc, 64, hipMemcpyDeviceToHost);
        printf("run %d, s_memtime ticks per element (100 MHz?):  A regs %.2f | B readlane(sgpr) %.2f | C readlane(const) %.2f | D lds-batched %.2f | E f32 regs %.2f | F B+capture %.2f\n",
               r, hc[0] / 256.0, hc[1] / 256.0, hc[2] / 256.0, hc[3] / 256.0, hc[4] / 256.0, hc[5] / 256.0);
    }
    return 0;
}
