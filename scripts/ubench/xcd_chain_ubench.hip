// Round-2 design probes for the XCD-per-stream generation kernel (tuning aid, not product code).
// (Round 5: sections 3 / 3b follow the product's arithmetic contract AC-1b / AC-2 as adopted in round 5 -- conditioning addend and dense
//  bias as chain start values, Estrin rational with a software reciprocal; the numbers of profiles/r02_xcd_chain_ubench*.txt were taken
//  under rounds 1-4's contract, which scripts/ubench/chain_contract_ubench.hip keeps as its base line C0.)
//   1. lane-crossing primitives: DPP row_newbcast on a 32-bit VOP2, v_permlane16_swap, v_permlane32_swap (semantics on gfx950)
//   2. a 32-term AC-1 chunk as 32 v_fmac_f32_dpp (no v_readlane): bit-exactness vs fmaf chains, cycles per dot
//   3. the whole residual-layer body in the row-broadcast lane layout with register-resident weights: bits vs the canonical
//      (shuffle) formulation, cycles per layer
//   4. granule hop between two workgroups of ONE XCD: plain / sc1 stores x sc1 / sc0 sc1 loads, 8- and 16-byte granules
//   5. wave -> wave hand-off through LDS inside a workgroup (tagged ds_write_b64 / ds_read_b64)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off scripts/ubench/xcd_chain_ubench.hip -o scripts/ubench/xcd_chain_ubench.exe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../tacotron-wavenet-vocoder-korean_amd/csrc/twv_math.hpp"
#include "../../tacotron-wavenet-vocoder-korean_amd/csrc/twv_dpp.hpp"
using namespace twv;

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(r_), __LINE__); exit(1); } } while (0)

// ---------------------------------------------------------------- 1. primitives
__global__ void prim_kernel(int* out)
{
    const int lane = threadIdx.x;
    int v = lane, r = -1;
    asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    out[lane] = r;
    float acc = 100.0f, xv = (float)lane, wv = 2.0f;
    asm volatile("s_nop 1\n v_fmac_f32_dpp %0, %1, %2 row_newbcast:7 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(xv), "v"(wv));
    out[64 + lane] = (int)acc;
    const auto s16 = __builtin_amdgcn_permlane16_swap((unsigned)lane, (unsigned)(lane + 100), false, false);
    out[128 + lane] = (int)s16[0];
    out[192 + lane] = (int)s16[1];
    const auto s32 = __builtin_amdgcn_permlane32_swap((unsigned)lane, (unsigned)(lane + 100), false, false);
    out[256 + lane] = (int)s32[0];
    out[320 + lane] = (int)s32[1];
}

// ---------------------------------------------------------------- 2. dot
// lane L owns output L; W[k][64]; x[32]
__global__ void __launch_bounds__(64) dot_kernel(const float* W, const float* x, float* out, unsigned long long* cyc, int reps)
{
    const int lane = threadIdx.x;
    float w[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) w[k] = W[k * 64 + lane];
    float xa = x[lane & 15], xb = x[16 + (lane & 15)];
    out[lane] = dot32_dpp(w, xa, xb);
    // timing: dependent dots (the result feeds the next operand)
    float acc = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long w0 = wall_clock64();
    for (int r = 0; r < reps; ++r) {
        const float d = dot32_dpp(w, xa, xb);
        acc += d;
        xa = xa + d * 1e-30f; xb = xb - d * 1e-30f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long w1 = wall_clock64();
    if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
    out[64 + lane] = acc;
}

// ---------------------------------------------------------------- 3. layer body
constexpr int NLU = 4;     // layers held by one wave
struct LayerCanon {        // canonical layouts
    float Wc[32][64];      // tap-1 conv kernel [k][f 0..31 | g 0..31]
    float Wd[32][32];      // dense [k][o]
    float pre[64], bfg[64], gcv[64], lcv[64], bd[32];
};
// canonical formulation (lane j: filter j / gate j-32; shuffles)
__global__ void __launch_bounds__(64) layer_ref_kernel(const LayerCanon* Lc, const float* x0, float* xout, float* zout, int steps)
{
    const int lane = threadIdx.x;
    const ActCoef coef = act_coef(lane >= 32);
    float x = x0[lane & 31];
    for (int t = 0; t < steps; ++t) {
        for (int l = 0; l < NLU; ++l) {
            const LayerCanon& P = Lc[l];
            float s[4] = {((P.pre[lane] + P.bfg[lane]) + P.gcv[lane]) + P.lcv[lane], -0.0f, -0.0f, -0.0f};      // AC-1b: chain 0 starts from the addend, the others from their first product
            for (int k = 0; k < 32; ++k) s[k & 3] = fma_(P.Wc[k][lane], __shfl(x, k), s[k & 3]);
            const float v = (s[0] + s[1]) + (s[2] + s[3]);
            const float act = act_eval(coef, v);
            const float z = __shfl(act, lane & 31) * __shfl(act, 32 + (lane & 31));
            float q[4] = {P.bd[lane & 31], -0.0f, -0.0f, -0.0f};                                    // AC-1b: the bias is the start value
            for (int k = 0; k < 32; ++k) q[k & 3] = fma_(P.Wd[k][lane & 31], __shfl(z, k), q[k & 3]);
            const float tr = (q[0] + q[1]) + (q[2] + q[3]);
            x = x + tr;
            if (t == steps - 1) zout[l * 32 + (lane & 31)] = z;
        }
        x = x * 0.25f;      // keep the recursion bounded
    }
    if (lane < 32) xout[lane] = x;
}
// row-broadcast formulation, register-resident weights
__global__ void __launch_bounds__(64) layer_dpp_kernel(const LayerCanon* Lc, const float* x0, float* xout, float* zout, int steps, unsigned long long* cyc)
{
    const int lane = threadIdx.x;
    const ActCoef coef = act_coef(lane >= 32);
    LayerRegs W[NLU];
    float pre[NLU], lcv[NLU];
    const int oc = dpp_conv_out(lane), od = dpp_dense_out(lane);
#pragma unroll
    for (int l = 0; l < NLU; ++l) {
#pragma unroll
        for (int k = 0; k < 32; ++k) W[l].wc[k] = Lc[l].Wc[k][oc];
#pragma unroll
        for (int i = 0; i < 16; ++i) W[l].wd[i] = Lc[l].Wd[dpp_dense_k(lane, i)][od];
        W[l].bd_init = dense_bias_init(lane, Lc[l].bd[od]);
        pre[l] = ((Lc[l].pre[oc] + Lc[l].bfg[oc]) + Lc[l].gcv[oc]) + Lc[l].lcv[oc]; lcv[l] = 0.0f;
    }
    float X = x0[od];
    float z = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long w0 = wall_clock64();
    for (int t = 0; t < steps; ++t) {
#pragma unroll
        for (int l = 0; l < NLU; ++l) {
            z = layer_body_dpp(W[l], coef, X, pre[l]);
            if (t == steps - 1 && lane < 32) zout[l * 32 + dpp_z_index(lane)] = z;
        }
        X = X * 0.25f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long w1 = wall_clock64();
    if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
    if ((lane & 16) == 0) xout[od] = X;
}


// ---------------------------------------------------------------- 3b. what each piece of the product's layer loop costs
// MODE bit0: dense kernel from LDS (4 ds_read_b128 per layer); bit1: one 16-byte granule store + wait states per layer;
// bit2: run-time layer count (uniform branch per layer); bit3: two 8-byte granule stores instead of bit1's one
typedef unsigned u32x4t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2t __attribute__((ext_vector_type(2)));
typedef float f32x4t __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
#define ULDS4(o) (((__attribute__((address_space(3))) f32x4t*)dyn_lds)[(o)])
template <int MODE>
__global__ void __launch_bounds__(64) layer_var_kernel(const LayerCanon* Lc, const float* x0, float* xout, int steps, unsigned long long* cyc, unsigned long long* gran, int nl_rt)
{
    const int lane = threadIdx.x;
    const ActCoef coef = act_coef(lane >= 32);
    LayerRegs W[NLU];
    float pre[NLU], lcv[NLU];
    const int oc = dpp_conv_out(lane), od = dpp_dense_out(lane);
#pragma unroll
    for (int l = 0; l < NLU; ++l) {
#pragma unroll
        for (int k = 0; k < 32; ++k) W[l].wc[k] = Lc[l].Wc[k][oc];
#pragma unroll
        for (int i = 0; i < 16; ++i) { W[l].wd[i] = Lc[l].Wd[dpp_dense_k(lane, i)][od]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) ULDS4((l * 4 + q) * 64 + lane) = f32x4t{W[l].wd[4 * q], W[l].wd[4 * q + 1], W[l].wd[4 * q + 2], W[l].wd[4 * q + 3]};
        W[l].bd_init = dense_bias_init(lane, Lc[l].bd[od]);
        pre[l] = ((Lc[l].pre[oc] + Lc[l].bfg[oc]) + Lc[l].gcv[oc]) + Lc[l].lcv[oc]; lcv[l] = 0.0f;
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(gran, 0, 1 << 20, 0x00020000);
    float X = x0[od];
    const int nl = (MODE & 4) ? nl_rt : NLU;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long w0 = wall_clock64();
    for (int t = 0; t < steps; ++t) {
        const unsigned tag = (unsigned)t + 1u;
#pragma unroll
        for (int l = 0; l < NLU; ++l) {
            if (l < nl) {
                float wd[16];
                if (MODE & 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const f32x4t v = ULDS4((l * 4 + q) * 64 + lane); wd[4 * q] = v.x; wd[4 * q + 1] = v.y; wd[4 * q + 2] = v.z; wd[4 * q + 3] = v.w; }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) wd[i] = W[l].wd[i];
                }
                const float z = layer_front_dpp(W[l].wc, coef, X, pre[l]);
                if (MODE & 2) {
                    const u32x4t d = {__float_as_uint(z), tag, __float_as_uint(X), tag};
                    __builtin_amdgcn_raw_buffer_store_b128(d, rs, lane * 16, l * 1024, 0);
                    asm volatile("s_nop 1" ::"v"(d) : "memory");
                }
                if (MODE & 8) {
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2t{__float_as_uint(z), tag}, rs, lane * 8, l * 1024, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2t{__float_as_uint(X), tag}, rs, lane * 8, l * 1024 + 512, 0);
                    asm volatile("" ::: "memory");
                }
                layer_back_dpp(wd, W[l].bd_init, z, X);
            }
        }
        X = X * 0.25f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long w1 = wall_clock64();
    if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
    if ((lane & 16) == 0) xout[od] = X;
}

// ---------------------------------------------------------------- 4. same-XCD hop
// MODE bit0: sc1 store (else plain), bit1: "sc0 sc1" load (else sc1), bit2: 16-byte granules
template <int MODE>
__global__ void hop_kernel(unsigned long long* X, int peer_a, int peer_b, int reps, unsigned long long* out, unsigned* xcc)
{
    const int me = blockIdx.x;
    unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if (threadIdx.x == 0) xcc[me] = id;
    if (me != peer_a && me != peer_b) return;
    const bool first = me == peer_a;
    constexpr int GW = (MODE & 4) ? 2 : 1;      // u64 words per granule
    unsigned long long* mine = X + (first ? 0 : 256) + threadIdx.x * GW;
    unsigned long long* theirs = X + (first ? 256 : 0) + threadIdx.x * GW;
    unsigned long long t0 = 0, t1 = 0, w0 = 0, w1 = 0;
    auto store = [&](unsigned r) {
        const unsigned long long v = ((unsigned long long)r << 32) | threadIdx.x;
        if (MODE & 4) {
            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
            const u64x2 vv = {v, v};
            if (MODE & 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(mine), "v"(vv) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(mine), "v"(vv) : "memory");
        } else {
            if (MODE & 1) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(mine), "v"(v) : "memory");
            else asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(mine), "v"(v) : "memory");
        }
    };
    for (int r = 1; r <= reps; ++r) {
        if (r == 2) { t0 = __builtin_amdgcn_s_memtime(); w0 = wall_clock64(); }
        if (first) store((unsigned)r);
        bool ok = false;
        for (int it = 0; it < (1 << 14) && !ok; ++it) {
            unsigned long long v;
            if (MODE & 4) {
                typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                u64x2 vv;
                if (MODE & 2) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(vv) : "v"(theirs) : "memory");
                else asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(vv) : "v"(theirs) : "memory");
                v = vv[0];
            } else {
                if (MODE & 2) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(theirs) : "memory");
                else asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(theirs) : "memory");
            }
            ok = __all((unsigned)(v >> 32) == (unsigned)r);
        }
        if (!ok) { if (threadIdx.x == 0) out[2] = 1; break; }
        if (!first) store((unsigned)r);
    }
    t1 = __builtin_amdgcn_s_memtime(); w1 = wall_clock64();
    if (first && threadIdx.x == 0) { out[0] = (t1 - t0) / (reps - 1); out[1] = (w1 - w0) * 1000 / (reps - 1); }
}

// ---------------------------------------------------------------- 5. LDS wave -> wave
__global__ void lds_hop_kernel(int reps, unsigned long long* out)
{
    __shared__ unsigned long long box[2][64];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    box[0][lane] = 0; box[1][lane] = 0;
    __syncthreads();
    volatile unsigned long long* mine = &box[wv][lane];
    volatile unsigned long long* theirs = &box[1 - wv][lane];
    unsigned long long t0 = 0, w0 = 0;
    for (int r = 1; r <= reps; ++r) {
        if (r == 2) { t0 = __builtin_amdgcn_s_memtime(); w0 = wall_clock64(); }
        if (wv == 0) *mine = ((unsigned long long)r << 32) | lane;
        for (int it = 0; it < (1 << 14); ++it) {
            const unsigned long long v = *theirs;
            if (__all((unsigned)(v >> 32) == (unsigned)r)) break;
        }
        if (wv == 1) *mine = ((unsigned long long)r << 32) | lane;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = (t1 - t0) / (reps - 1); out[1] = (w1 - w0) * 1000 / (reps - 1); }
}

static float frand() { return (float)rand() / (float)RAND_MAX * 2.0f - 1.0f; }

int main()
{
    // ---- 1
    {
        int* d; CHECK(hipMalloc(&d, 384 * 4));
        hipLaunchKernelGGL(prim_kernel, dim3(1), dim3(64), 0, 0, d);
        int h[384]; CHECK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        bool ok1 = true, ok2 = true, ok16 = true, ok32 = true;
        for (int l = 0; l < 64; ++l) {
            ok1 = ok1 && h[l] == (l & 48) + 5;
            ok2 = ok2 && h[64 + l] == 100 + 2 * ((l & 48) + 7);
            const int row = l >> 4;
            // expected: v0' = [A0,B0,A2,B2], v1' = [A1,B1,A3,B3]  (A = lane, B = lane + 100)
            const int e0 = (row & 1) ? (l - 16 + 100) : l, e1 = (row & 1) ? (l + 100) : (l + 16);
            ok16 = ok16 && h[128 + l] == e0 && h[192 + l] == e1;
            const int f0 = l < 32 ? l : (l - 32 + 100), f1 = l < 32 ? (l + 32) : (l + 100);
            ok32 = ok32 && h[256 + l] == f0 && h[320 + l] == f1;
        }
        printf("[1] v_mov_dpp row_newbcast %s | v_fmac_f32_dpp row_newbcast %s | permlane16_swap %s | permlane32_swap %s\n",
               ok1 ? "OK" : "WRONG", ok2 ? "OK" : "WRONG", ok16 ? "OK" : "WRONG", ok32 ? "OK" : "WRONG");
        if (!ok16) { printf("    p16 v0':"); for (int l = 0; l < 64; ++l) printf(" %d", h[128 + l]); printf("\n    p16 v1':"); for (int l = 0; l < 64; ++l) printf(" %d", h[192 + l]); printf("\n"); }
        if (!ok1) { printf("    bcast:"); for (int l = 0; l < 64; ++l) printf(" %d", h[l]); printf("\n"); }
        fflush(stdout);
    }
    // ---- 2
    {
        std::vector<float> W(32 * 64), x(32);
        srand(1);
        for (auto& v : W) v = frand();
        for (auto& v : x) v = frand();
        W[5] = 1e-39f; x[3] = 3e-40f; W[3 * 64 + 7] = 2e-20f; x[9] = -1e-25f;     // subnormal operands / products
        float *dW, *dx, *dout; unsigned long long* dc;
        CHECK(hipMalloc(&dW, W.size() * 4)); CHECK(hipMalloc(&dx, 128)); CHECK(hipMalloc(&dout, 512)); CHECK(hipMalloc(&dc, 64));
        CHECK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dx, x.data(), 128, hipMemcpyHostToDevice));
        const int reps = 20000;
        hipLaunchKernelGGL(dot_kernel, dim3(1), dim3(64), 0, 0, dW, dx, dout, dc, reps);
        float out[128]; unsigned long long c[2];
        CHECK(hipMemcpy(out, dout, 512, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int j = 0; j < 64; ++j) {
            float s[4] = {0, 0, 0, 0};
            for (int k = 0; k < 32; ++k) s[k & 3] = fmaf(W[k * 64 + j], x[k], s[k & 3]);
            const float want = (s[0] + s[1]) + (s[2] + s[3]);
            if (memcmp(&want, &out[j], 4)) { if (bad < 4) printf("    dot mismatch lane %d: %a vs %a\n", j, out[j], want); ++bad; }
        }
        printf("[2] dot32_dpp vs fmaf chains: %d / 64 mismatches; %.1f memtime ticks, %.2f ns per dependent dot (+ ~5 VALU of loop glue)\n",
               bad, (double)c[0] / reps, (double)c[1] * 10.0 / reps);
        fflush(stdout);
    }
    // ---- 3
    {
        std::vector<LayerCanon> L(NLU);
        srand(2);
        for (auto& P : L) {
            for (int k = 0; k < 32; ++k) for (int o = 0; o < 64; ++o) P.Wc[k][o] = frand() * 0.3f;
            for (int k = 0; k < 32; ++k) for (int o = 0; o < 32; ++o) P.Wd[k][o] = frand() * 0.3f;
            for (int o = 0; o < 64; ++o) { P.pre[o] = frand(); P.bfg[o] = frand() * 0.1f; P.gcv[o] = frand() * 0.1f; P.lcv[o] = frand() * 0.1f; }
            for (int o = 0; o < 32; ++o) P.bd[o] = frand() * 0.1f;
        }
        std::vector<float> x0(32);
        for (auto& v : x0) v = frand();
        LayerCanon* dL; float *dx0, *dxa, *dxb, *dza, *dzb; unsigned long long* dc;
        CHECK(hipMalloc(&dL, sizeof(LayerCanon) * NLU)); CHECK(hipMalloc(&dx0, 128)); CHECK(hipMalloc(&dxa, 128)); CHECK(hipMalloc(&dxb, 128));
        CHECK(hipMalloc(&dza, NLU * 128)); CHECK(hipMalloc(&dzb, NLU * 128)); CHECK(hipMalloc(&dc, 64));
        CHECK(hipMemcpy(dL, L.data(), sizeof(LayerCanon) * NLU, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dx0, x0.data(), 128, hipMemcpyHostToDevice));
        const int steps = 5000;
        hipLaunchKernelGGL(layer_ref_kernel, dim3(1), dim3(64), 0, 0, dL, dx0, dxa, dza, steps);
        hipLaunchKernelGGL(layer_dpp_kernel, dim3(1), dim3(64), 0, 0, dL, dx0, dxb, dzb, steps, dc);
        CHECK(hipDeviceSynchronize());
        float xa[32], xb[32], za[NLU * 32], zb[NLU * 32]; unsigned long long c[2];
        CHECK(hipMemcpy(xa, dxa, 128, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(xb, dxb, 128, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(za, dza, sizeof(za), hipMemcpyDeviceToHost)); CHECK(hipMemcpy(zb, dzb, sizeof(zb), hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost));
        int badx = 0, badz = 0;
        for (int i = 0; i < 32; ++i) badx += memcmp(&xa[i], &xb[i], 4) != 0;
        for (int i = 0; i < NLU * 32; ++i) badz += memcmp(&za[i], &zb[i], 4) != 0;
        printf("[3] layer body (row-broadcast layout, register weights) vs canonical: x mismatches %d/32, z mismatches %d/%d after %d steps x %d layers (x[0] = %a)\n",
               badx, badz, NLU * 32, steps, NLU, xb[0]);
        printf("    %.1f memtime ticks = %.1f ns per layer\n", (double)c[0] / steps / NLU, (double)c[1] * 10.0 / steps / NLU);
        fflush(stdout);

        {
            unsigned long long* gran; CHECK(hipMalloc(&gran, 1 << 20));
            const char* nm[16] = {"registers only", "+ dense kernel from LDS", "+ 16-B store + s_nop", "+ LDS + 16-B store", "+ run-time layer count", "+ LDS + count", "+ store + count", "+ LDS + store + count (product shape)",
                                  "+ two 8-B stores", "+ LDS + two 8-B stores", "", "", "+ two 8-B stores + count", "+ LDS + two 8-B stores + count", "", ""};
            for (int m = 0; m < 14; ++m) {
                if (m == 10 || m == 11) continue;
#define LV(M) hipLaunchKernelGGL(layer_var_kernel<M>, dim3(1), dim3(64), 32768, 0, dL, dx0, dxb, steps, dc, gran, NLU)
                switch (m) { case 0: LV(0); break; case 1: LV(1); break; case 2: LV(2); break; case 3: LV(3); break; case 4: LV(4); break; case 5: LV(5); break; case 6: LV(6); break; case 7: LV(7); break;
                             case 8: LV(8); break; case 9: LV(9); break; case 12: LV(12); break; default: LV(13); break; }
#undef LV
                CHECK(hipDeviceSynchronize());
                CHECK(hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(xb, dxb, 128, hipMemcpyDeviceToHost));
                int bad = 0; for (int i = 0; i < 32; ++i) bad += memcmp(&xa[i], &xb[i], 4) != 0;
                printf("    [3b] %-40s %6.1f ticks = %6.1f ns per layer (x mismatches %d)\n", nm[m], (double)c[0] / steps / NLU, (double)c[1] * 10.0 / steps / NLU, bad);
            }
            fflush(stdout);
        }
    }
    // ---- 4
    {
        unsigned long long *X, *out; unsigned* xcc;
        CHECK(hipMalloc(&X, 8192)); CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&xcc, 64 * 4));
        unsigned hx[64]; unsigned long long h[3];
        const char* nm[8] = {"plain store, sc1 load, 8 B", "sc1 store, sc1 load, 8 B", "plain store, sc0 sc1 load, 8 B", "sc1 store, sc0 sc1 load, 8 B",
                             "plain store, sc1 load, 16 B", "sc1 store, sc1 load, 16 B", "plain store, sc0 sc1 load, 16 B", "sc1 store, sc0 sc1 load, 16 B"};
        struct { int a, b; } pairs[] = {{0, 8}, {0, 1}};
        for (auto pr : pairs) {
            for (int m = 0; m < 8; ++m) {
                CHECK(hipMemset(X, 0, 8192)); CHECK(hipMemset(out, 0, 64));
                const int reps = 400;
#define LAUNCH(M) hipLaunchKernelGGL(hop_kernel<M>, dim3(32), dim3(64), 0, 0, X, pr.a, pr.b, reps, out, xcc)
                switch (m) { case 0: LAUNCH(0); break; case 1: LAUNCH(1); break; case 2: LAUNCH(2); break; case 3: LAUNCH(3); break;
                             case 4: LAUNCH(4); break; case 5: LAUNCH(5); break; case 6: LAUNCH(6); break; default: LAUNCH(7); break; }
#undef LAUNCH
                CHECK(hipDeviceSynchronize());
                CHECK(hipMemcpy(h, out, 24, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hx, xcc, 128, hipMemcpyDeviceToHost));
                printf("[4] WG %d <-> %d (xcc %u / %u) %-32s: round trip %5llu ticks = %6.0f ns%s\n", pr.a, pr.b, hx[pr.a], hx[pr.b], nm[m], h[0],
                       (double)h[1] / 100.0, h[2] ? "   NEVER SEEN (timeout)" : "");
                fflush(stdout);
            }
        }
    }
    // ---- 5
    {
        unsigned long long* out; CHECK(hipMalloc(&out, 64));
        hipLaunchKernelGGL(lds_hop_kernel, dim3(1), dim3(128), 0, 0, 2000, out);
        unsigned long long h[2]; CHECK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
        printf("[5] LDS wave<->wave tagged granule: round trip %llu ticks = %.0f ns\n", h[0], (double)h[1] / 100.0);
    }
    return 0;
}
