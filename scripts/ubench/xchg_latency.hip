// round-trip latency of the {epoch,value} granule exchange between two workgroups: same XCD vs different XCD, agent- vs workgroup-scope
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(1))) unsigned long long gu64;
template <int SCOPE>
__global__ void k(unsigned long long* X, int peer_a, int peer_b, int reps, unsigned long long* out, unsigned* xcc)
{
    const int me = blockIdx.x;
    unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if (threadIdx.x == 0) xcc[me] = id;
    if (me != peer_a && me != peer_b) return;
    const bool first = me == peer_a;
    unsigned long long* mine = X + (first ? 0 : 64);
    unsigned long long* theirs = X + (first ? 64 : 0);
    unsigned long long t0 = 0, t1 = 0;
    for (int r = 1; r <= reps; ++r) {
        if (r == 2) t0 = __builtin_amdgcn_s_memtime();
        if (first) __hip_atomic_store((gu64*)(mine + threadIdx.x), ((unsigned long long)r << 32) | threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int it = 0; it < (1 << 12); ++it) {
            unsigned long long v;
            if (SCOPE == __HIP_MEMORY_SCOPE_AGENT) v = __hip_atomic_load((gu64*)(theirs + threadIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else {   // group-scope load: sc0 = miss-always in the CU's L1, served by the XCD's L2
                const unsigned long long* p = theirs + threadIdx.x;
                asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
            }
            if (__all((unsigned)(v >> 32) == (unsigned)r)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (!first) __hip_atomic_store((gu64*)(mine + threadIdx.x), ((unsigned long long)r << 32) | threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (first && threadIdx.x == 0) out[0] = (t1 - t0) / (reps - 1);
}
int main()
{
    unsigned long long *X, *out; unsigned* xcc;
    hipMalloc(&X, 4096); hipMalloc(&out, 64); hipMalloc(&xcc, 64 * 4);
    unsigned hx[64]; unsigned long long h;
    struct { int a, b; const char* nm; } cs[] = {{0, 8, "WG 0 <-> 8 (same XCD if round-robin)"}, {0, 1, "WG 0 <-> 1 (different XCD)"}, {0, 16, "WG 0 <-> 16"}};
    for (auto c : cs) {
        hipMemset(X, 0, 4096);
        hipLaunchKernelGGL(k<__HIP_MEMORY_SCOPE_AGENT>, dim3(32), dim3(64), 0, 0, X, c.a, c.b, 300, out, xcc);
        hipDeviceSynchronize(); hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost); hipMemcpy(hx, xcc, 128, hipMemcpyDeviceToHost);
        fflush(stdout); printf("%-40s agent scope    : round trip %llu ticks (xcc %u / %u)\n", c.nm, h, hx[c.a], hx[c.b]);
        if (hx[c.a] == hx[c.b]) {
            hipMemset(X, 0, 4096);
            hipLaunchKernelGGL(k<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(32), dim3(64), 0, 0, X, c.a, c.b, 300, out, xcc);
            hipDeviceSynchronize(); hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
            printf("%-40s sc0 loads (L2): round trip %llu ticks\n", c.nm, h); fflush(stdout);
        }
    }
    printf("xcc ids of WG 0..15:"); for (int i = 0; i < 16; ++i) printf(" %u", hx[i]); printf("\n");
    return 0;
}
