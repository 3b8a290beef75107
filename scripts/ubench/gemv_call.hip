// cost of one decg_gemv call (tacotron decoder, csrc/twv_tacotron.hip) in isolation: K=512, N=256 (proj stage), G=8, g=0 -> 16 tiles per WG
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../tacotron-wavenet-vocoder-korean_amd/csrc/twv_dev.hpp"
using namespace twv;
// chunk partials of this workgroup's blocks: lds[o_part + i*64 + lane], i = local tile index = m*nchunk + ch; wave w takes tiles
// w, w+8, ...  (a real call, not inlined: the register allocator then sees one small live range instead of twelve copies).
// No integer division anywhere: (m, ch) advance incrementally, G is a power of two (lg = log2 G).
struct DecgPos { int m, ch; };
__device__ __forceinline__ void decg_adv(DecgPos& p, int step, int nchunk)
{
    p.ch += step;
    while (p.ch >= nchunk) { p.ch -= nchunk; ++p.m; }
}
__device__ __forceinline__ int decg_off(int wt_bytes, const DecgPos& p, int nchunk, int g, int lg)
{
    return wt_bytes + ((((p.m << lg) + g) * nchunk + p.ch) << 13);        // kTile * 4 = 8192 bytes per tile
}
__device__ __noinline__ void decg_gemv(rsrc_t rs, int wt_bytes, int K, int N, int xo, int o_part, int wave, int lane, int g, int lg)
{
    const int nchunk = (K + 31) >> 5, nblk = (N + 63) >> 6, G = 1 << lg;
    const int ntile = nblk > g ? ((nblk - g + G - 1) >> lg) * nchunk : 0;
    const int vo = lane * 16;
    Tile t0, t1, t2;
    DecgPos p0{0, 0}, p1, p2;
    decg_adv(p0, wave, nchunk);
    p1 = p0; decg_adv(p1, 8, nchunk);
    p2 = p1; decg_adv(p2, 8, nchunk);
    if (wave < ntile) load_tile_b(t0, rs, vo, decg_off(wt_bytes, p0, nchunk, g, lg));
    if (wave + 8 < ntile) load_tile_b(t1, rs, vo, decg_off(wt_bytes, p1, nchunk, g, lg));
    if (wave + 16 < ntile) load_tile_b(t2, rs, vo, decg_off(wt_bytes, p2, nchunk, g, lg));
    for (int i = wave; i < ntile; i += 24) {
        {
            const float r = dot_ldso(t0, xo + p0.ch * 32);
            __builtin_amdgcn_sched_barrier(0);
            lds[o_part + i * 64 + lane] = r;
            decg_adv(p0, 24, nchunk);
            if (i + 24 < ntile) load_tile_b(t0, rs, vo, decg_off(wt_bytes, p0, nchunk, g, lg));
        }
        if (i + 8 < ntile) {
            const float r = dot_ldso(t1, xo + p1.ch * 32);
            __builtin_amdgcn_sched_barrier(0);
            lds[o_part + (i + 8) * 64 + lane] = r;
            decg_adv(p1, 24, nchunk);
            if (i + 32 < ntile) load_tile_b(t1, rs, vo, decg_off(wt_bytes, p1, nchunk, g, lg));
        }
        if (i + 16 < ntile) {
            const float r = dot_ldso(t2, xo + p2.ch * 32);
            __builtin_amdgcn_sched_barrier(0);
            lds[o_part + (i + 16) * 64 + lane] = r;
            decg_adv(p2, 24, nchunk);
            if (i + 40 < ntile) load_tile_b(t2, rs, vo, decg_off(wt_bytes, p2, nchunk, g, lg));
        }
    }
}
__global__ void __launch_bounds__(512) k(const float* P, int bytes, int K, int N, int G, int reps, unsigned long long* out)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P), 0, bytes, 0x00020000);
    for (int i = threadIdx.x; i < 1024; i += 512) lds[i] = 1.0f;
    __syncthreads();
    unsigned long long tot = 0;
    for (int r = 0; r < reps; ++r) {
        __syncthreads();
        const unsigned long long a = __builtin_amdgcn_s_memtime();
        decg_gemv(rs, 0, K, N, 0, 2048, wave, lane, blockIdx.x & (G - 1), G == 8 ? 3 : (G == 4 ? 2 : (G == 2 ? 1 : 0)));
        const unsigned long long b = __builtin_amdgcn_s_memtime();
        __syncthreads();
        const unsigned long long c = __builtin_amdgcn_s_memtime();
        if (r) { tot += b - a; if (threadIdx.x == 0) out[64 + blockIdx.x] += c - a; }
    }
    if (lane == 0) out[blockIdx.x * 8 + wave] = tot / (reps - 1);
}
int main()
{
    const int bytes = 8 << 20;
    float* P; hipMalloc(&P, bytes); hipMemset(P, 0, bytes);
    unsigned long long* out; hipMalloc(&out, 4096 * 8);
    unsigned long long h[4096];
    struct { int K, N, G; } cs[] = {{512, 256, 8}, {640, 512, 8}, {256, 256, 1}, {96, 256, 1}};
    for (auto c : cs) {
        hipMemset(out, 0, 4096 * 8);
        hipLaunchKernelGGL(k, dim3(8), dim3(512), 48 * 1024, 0, P, bytes, c.K, c.N, c.G, 101, out);
        hipDeviceSynchronize();
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        printf("K %d N %d G %d : wave0..7 of WG0 ticks:", c.K, c.N, c.G);
        for (int w = 0; w < 8; ++w) printf(" %llu", h[w]);
        printf("   incl. barrier (thread 0): %llu\n", h[64] / 100);
    }
    return 0;
}
