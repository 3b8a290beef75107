// In-isolation micro-benchmark of the chain wave's per-layer body (tuning aid, not product code).
// One wave, NSLOT LDS slots pre-filled, no loaders/workers: cycles per layer for several code shapes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../../tacotron-wavenet-vocoder-korean_amd/csrc/twv_math.hpp"
using namespace twv;
extern __shared__ __attribute__((aligned(16))) float lds[];
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDSI(off) (((__attribute__((address_space(3))) int*)lds)[(off)])
#define LDSVI(off) (((__attribute__((address_space(3))) volatile int*)lds)[(off)])
#define LDS4(o) (((__attribute__((address_space(3))) f32x4*)lds)[(o)])
struct Tile { float w[32]; };
constexpr int SLOTF = 3520, NSLOT = 8, NL = 30, STEPS = 200;
constexpr int O_Z = 0, O_X = 1024, O_GC = 2048, O_CTL = 4096, O_SLOTS = 4160;
__device__ __forceinline__ void lds_tile(Tile& t, int fo, int lane) {
#pragma unroll
    for (int kq = 0; kq < 8; ++kq) { const f32x4 q = LDS4((fo >> 2) + kq * 64 + lane); t.w[4*kq]=q.x; t.w[4*kq+1]=q.y; t.w[4*kq+2]=q.z; t.w[4*kq+3]=q.w; }
}
__device__ __forceinline__ void lds_half_tile(Tile& t, int fo, int lane) {
#pragma unroll
    for (int kq = 0; kq < 8; ++kq) { const f32x4 q = LDS4((fo >> 2) + kq * 32 + (lane & 31)); t.w[4*kq]=q.x; t.w[4*kq+1]=q.y; t.w[4*kq+2]=q.z; t.w[4*kq+3]=q.w; }
}
__device__ __forceinline__ float dot_readlane(const Tile& t, float xv) {
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
    for (int c = 0; c < 32; c += 4) {
        s0 = fma_(t.w[c], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), c)), s0);
        s1 = fma_(t.w[c+1], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), c+1)), s1);
        s2 = fma_(t.w[c+2], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), c+2)), s2);
        s3 = fma_(t.w[c+3], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), c+3)), s3);
    }
    return (s0 + s1) + (s2 + s3);
}
// MODE 0: current product shape (flags, ready check, single-buffered fetch); 1: no flags/ready; 2: MODE1 + no LDS fetch (regs reused);
// 3: only the math (conv, act, dense) ; 4: MODE 0 but small vectors packed into one b128
// dot with the NEXT tile's LDS reads interleaved: 1 ds_read_b128 per 8 VALU
__device__ __forceinline__ float dot_readlane_fetch(const Tile& t, float xv, Tile& nt, int fo, int lane, bool half) {
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
    for (int c = 0; c < 32; c += 4) {
        const int kq = c >> 2;
        const f32x4 q = half ? LDS4((fo >> 2) + kq * 32 + (lane & 31)) : LDS4((fo >> 2) + kq * 64 + lane);
        s0 = fma_(t.w[c], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), c)), s0);
        s1 = fma_(t.w[c+1], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), c+1)), s1);
        s2 = fma_(t.w[c+2], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), c+2)), s2);
        s3 = fma_(t.w[c+3], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), c+3)), s3);
        nt.w[c] = q.x; nt.w[c+1] = q.y; nt.w[c+2] = q.z; nt.w[c+3] = q.w;
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
        __builtin_amdgcn_sched_group_barrier(0x2, 8, 0);     // 8 VALU
    }
    return (s0 + s1) + (s2 + s3);
}
template <int MODE>
__global__ void k(float* out, unsigned long long* cyc, const float* w)
{
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < O_SLOTS + NSLOT * SLOTF; i += 64) lds[i] = w[i % 65536] * 0.05f;
    for (int i = 0; i < 16; ++i) LDSI(O_CTL + 16 + i) = 1 << 30;   // ready words: always ready
    __syncthreads();
    const ActCoef coef = act_coef(lane >= 32);
    Tile w1, wd; float pre, bfg, gcv, lcv, bd;
    auto fetch_conv = [&](int sb, int l) {
        lds_tile(w1, sb, lane);
        if (MODE == 4) { const f32x4 q = LDS4(((sb + 3328) >> 2) + lane); pre = q.x; bfg = q.y; gcv = q.z; lcv = q.w; }
        else { pre = lds[sb + 3456 + lane]; bfg = lds[sb + 3072 + lane]; gcv = lds[O_GC + l * 64 + lane]; lcv = lds[sb + 3392 + lane]; }
    };
    auto fetch_dense = [&](int sb) { lds_half_tile(wd, sb + 2048, lane); bd = lds[sb + 3136 + (lane & 31)]; };
    fetch_conv(O_SLOTS, 0); fetch_dense(O_SLOTS);
    float x = w[lane] * 0.01f;
    int item = 0, slot = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 6) {
        // weights straight from global/L2 into registers, requested one layer ahead (no LDS weight slots); pre-activation inputs packed b128 from LDS
        auto gload = [&](Tile& a, Tile& b, int l) {
            const f32x4* p = reinterpret_cast<const f32x4*>(w + (l * 3072) % 60000);
#pragma unroll
            for (int kq = 0; kq < 8; ++kq) { const f32x4 q = p[kq * 64 + lane]; a.w[4*kq]=q.x; a.w[4*kq+1]=q.y; a.w[4*kq+2]=q.z; a.w[4*kq+3]=q.w; }
#pragma unroll
            for (int kq = 0; kq < 8; ++kq) { const f32x4 q = p[512 + kq * 32 + (lane & 31)]; b.w[4*kq]=q.x; b.w[4*kq+1]=q.y; b.w[4*kq+2]=q.z; b.w[4*kq+3]=q.w; }
        };
        Tile a0, b0, a1, b1;
        gload(a0, b0, 0);
        for (int t = 0; t < STEPS; ++t) {
            for (int l = 0; l < NL; l += 2) {
#define LAYER(A, B, AN, BN, LL)                                                                                   \
                {                                                                                                 \
                    const int slot_n = (slot + 1 == NSLOT) ? 0 : slot + 1;                                        \
                    const int sb = O_SLOTS + slot * SLOTF;                                                        \
                    const int rdy = LDSVI(O_CTL + 16 + slot_n);                                                   \
                    gload(AN, BN, (LL) + 1 < NL ? (LL) + 1 : 0);                                                  \
                    if (lane < 32) lds[O_X + (LL) * 32 + lane] = x;                                               \
                    const f32x4 q = LDS4(((sb + 3328) >> 2) + lane);                                              \
                    const float acc1 = dot_readlane(A, x);                                                        \
                    float v = q.x + acc1; v = v + q.y; v = v + q.z; v = v + q.w;                                  \
                    const float act = act_eval(coef, v);                                                          \
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(act), __float_as_uint(act), false, false); \
                    const float z = __uint_as_float(sw[0]) * __uint_as_float(sw[1]);                              \
                    if (lane < 32) lds[O_Z + (LL) * 32 + lane] = z;                                               \
                    asm volatile("" ::: "memory"); if (lane == 0) LDSVI(O_CTL + 0) = item + 1; asm volatile("" ::: "memory"); \
                    if (rdy < item + 2) { while (LDSVI(O_CTL + 16 + slot_n) < item + 2) __builtin_amdgcn_s_sleep(1); }       \
                    float tr = dot_readlane(B, z); tr = tr + lds[sb + 3136 + (lane & 31)]; x = x + tr;            \
                    ++item; slot = slot_n;                                                                        \
                }
                LAYER(a0, b0, a1, b1, l)
                LAYER(a1, b1, a0, b0, l + 1)
#undef LAYER
            }
            x = x * 0.5f;
        }
    } else
    if (MODE == 5) {
        // double-buffered tiles: w1 (cur) / w1n (being fetched during the dense dot); wd (cur) / wdn (fetched during the conv dot)
        Tile w1n, wdn;
        for (int t = 0; t < STEPS; ++t) {
            for (int l = 0; l < NL; ++l) {
                const int slot_n = (slot + 1 == NSLOT) ? 0 : slot + 1;
                const int sb = O_SLOTS + slot * SLOTF, sbn = O_SLOTS + slot_n * SLOTF;
                const int rdy = LDSVI(O_CTL + 16 + slot_n);
                if (lane < 32) lds[O_X + l * 32 + lane] = x;
                // conv with THIS layer's dense tile streaming in
                const float acc1 = dot_readlane_fetch(w1, x, wdn, sb + 2048, lane, true);
                float v = pre + acc1; v = v + bfg; v = v + gcv; v = v + lcv;
                const float bdc = lds[sb + 3136 + (lane & 31)];
                const float act = act_eval(coef, v);
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(act), __float_as_uint(act), false, false);
                const float z = __uint_as_float(sw[0]) * __uint_as_float(sw[1]);
                if (lane < 32) lds[O_Z + l * 32 + lane] = z;
                asm volatile("" ::: "memory"); if (lane == 0) LDSVI(O_CTL + 0) = item + 1; asm volatile("" ::: "memory");
                if (rdy < item + 2) { while (LDSVI(O_CTL + 16 + slot_n) < item + 2) __builtin_amdgcn_s_sleep(1); }
                // dense with the NEXT layer's conv tile streaming in
                float tr = dot_readlane_fetch(wdn, z, w1n, sbn, lane, false); tr = tr + bdc; x = x + tr;
                { const f32x4 q = LDS4(((sbn + 3328) >> 2) + lane); pre = q.x; bfg = q.y; gcv = q.z; lcv = q.w; }
                w1 = w1n;
                ++item; slot = slot_n;
            }
            x = x * 0.5f;
        }
    } else
    for (int t = 0; t < STEPS; ++t) {
        for (int l = 0; l < NL; ++l) {
            if (MODE != 3) { if (lane < 32) lds[O_X + l * 32 + lane] = x; }
            const int slot_n = (slot + 1 == NSLOT) ? 0 : slot + 1;
            const int sbn = O_SLOTS + slot_n * SLOTF;
            const int ln = (l + 1 < NL) ? l + 1 : 0;
            int rdy = 1 << 30;
            if (MODE == 0 || MODE == 4) rdy = LDSVI(O_CTL + 16 + slot_n);
            const float acc1 = dot_readlane(w1, x);
            float v = pre + acc1; v = v + bfg; v = v + gcv; v = v + lcv;
            if (MODE == 0 || MODE == 4) { if (rdy < item + 2) { while (LDSVI(O_CTL + 16 + slot_n) < item + 2) __builtin_amdgcn_s_sleep(1); } }
            if (MODE == 0 || MODE == 1 || MODE == 4) fetch_conv(sbn, ln);
            const float act = act_eval(coef, v);
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(act), __float_as_uint(act), false, false);
            const float z = __uint_as_float(sw[0]) * __uint_as_float(sw[1]);
            if (MODE != 3) { if (lane < 32) lds[O_Z + l * 32 + lane] = z; }
            if (MODE == 0 || MODE == 4) { asm volatile("" ::: "memory"); if (lane == 0) LDSVI(O_CTL + 0) = item + 1; asm volatile("" ::: "memory"); }
            float tr = dot_readlane(wd, z); tr = tr + bd; x = x + tr;
            if (MODE == 0 || MODE == 1 || MODE == 4) fetch_dense(sbn);
            if (MODE == 0 || MODE == 4) { asm volatile("" ::: "memory"); if (lane == 0) LDSVI(O_CTL + 3) = item + 1; asm volatile("" ::: "memory"); }
            ++item; slot = slot_n;
        }
        x = x * 0.5f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    float *out, *w; unsigned long long* cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 64); hipMalloc(&w, 65536 * 4);
    std::vector<float> hw(65536);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 7919) % 1000) * 1e-3f - 0.5f;
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    const size_t shm = (O_SLOTS + NSLOT * SLOTF) * 4;
    const char* names[] = {"product shape", "no flags/ready", "no flags, no LDS fetch", "math only", "product shape, packed small vectors", "interleaved fetch, packed, single flag", "weights from global one layer ahead, packed, single flag"};
#define RUN(M) { hipFuncSetAttribute((const void*)k<M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
    hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), shm, 0, out, cyc, w); hipDeviceSynchronize(); \
    hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), shm, 0, out, cyc, w); hipDeviceSynchronize(); \
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-40s %8.1f ticks/layer\n", names[M], (double)c / (STEPS * NL)); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6)
    return 0;
}
