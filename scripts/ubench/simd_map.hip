// which SIMD does wave i of a 512-thread workgroup land on?  (HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh[12] se[15:13])
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = v;
}
int main() {
    unsigned* d; hipMalloc(&d, 64 * 16 * 4);
    for (int nw : {8, 10, 12}) {
        hipMemset(d, 0, 64 * 16 * 4);
        hipLaunchKernelGGL(k, dim3(4), dim3(nw * 64), 0, 0, d);
        unsigned h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        for (int b = 0; b < 4; ++b) { printf("waves=%d block %d simd:", nw, b); for (int w = 0; w < nw; ++w) printf(" %u", (h[b * 16 + w] >> 4) & 3); printf("  cu %u\n", (h[b * 16] >> 8) & 15); }
    }
    return 0;
}
