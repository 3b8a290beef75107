// Does a VGPR bank conflict between src0 and src2 of v_pk_fma_f32 cost issue cycles on gfx950?  (the generation kernel's dot chains have one
// on every instruction: accumulators hop through the dead weight registers, which all sit at 4m+2).
// Two dependent chains of 16 v_pk_fma each, as in dot_readlane; variant 0: src0 and src2 in the same banks, variant 1: different banks.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int VAR>
__global__ void k(unsigned long long* out, float* sink, int iters)
{
    float r = 0.0f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (VAR == 0) {
            asm volatile(
                ".rept 8\n"
                "v_pk_fma_f32 v[10:11], v[10:11], s[4:5], v[18:19]\n"     // banks (2,3) <- (2,3), (2,3)
                "v_pk_fma_f32 v[12:13], v[12:13], s[4:5], v[20:21]\n"     // banks (0,1) <- (0,1), (0,1)
                "v_pk_fma_f32 v[18:19], v[18:19], s[4:5], v[10:11]\n"
                "v_pk_fma_f32 v[20:21], v[20:21], s[4:5], v[12:13]\n"
                ".endr\n" ::: "v10", "v11", "v12", "v13", "v18", "v19", "v20", "v21", "s4", "s5");
        } else {
            asm volatile(
                ".rept 8\n"
                "v_pk_fma_f32 v[10:11], v[10:11], s[4:5], v[20:21]\n"     // banks (2,3) <- (2,3), (0,1)
                "v_pk_fma_f32 v[12:13], v[12:13], s[4:5], v[18:19]\n"     // banks (0,1) <- (0,1), (2,3)
                "v_pk_fma_f32 v[20:21], v[20:21], s[4:5], v[10:11]\n"
                "v_pk_fma_f32 v[18:19], v[18:19], s[4:5], v[12:13]\n"
                ".endr\n" ::: "v10", "v11", "v12", "v13", "v18", "v19", "v20", "v21", "s4", "s5");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[VAR] = t1 - t0;
    sink[threadIdx.x] = r;
}
int main()
{
    unsigned long long* d; float* s;
    hipMalloc(&d, 64); hipMalloc(&s, 4096);
    const int iters = 4096;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, s, iters);
        hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d, s, iters);
        hipDeviceSynchronize();
    }
    unsigned long long h[2];
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("v_pk_fma_f32 dependent pairs, 32 instructions per iteration: same-bank src0/src2 %.2f ticks/instr, different banks %.2f ticks/instr (s_memtime ticks)\n",
           (double)h[0] / (iters * 32.0), (double)h[1] / (iters * 32.0));
    return 0;
}
