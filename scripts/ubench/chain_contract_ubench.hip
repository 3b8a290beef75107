// Round-5 pricing of alternative ARITHMETIC CONTRACTS for the generation chain's layer body (tuning aid, not
// product code).  Same harness as xcd_chain_ubench.hip section 3 / 3b: one wave, four layers with register-resident tap-1 kernels,
// the layer body in the row-broadcast lane layout (twv_dpp.hpp), 5000 steps, s_memtime ticks (= core clocks) per layer.
//
// Contracts (each is checked bit for bit against its own canonical "shuffle" formulation, i.e. against what a C restatement would
// compute with fmaf chains -- a contract that a CPU cannot reproduce is of no use here):
//   C0  the product's AC-1 / AC-2: v = (((tap0 + tap1) + bias) + gc) + lc; Horner rational with one IEEE division; dense + bias
//   C1  (b)  pre-summed addend: A = ((tap0 + bias) + gc) + lc comes from the service workgroup; v = tap1 + A        (3 adds fewer)
//   C2  (b') the addend is the START VALUE of chain 0 of the tap-1 chunk (no add at all); likewise the dense bias
//   C3  (c)  activation: the SAME rational (Eigen's coefficients), numerator and denominator by Estrin's scheme (depth 4 instead
//            of 6), the division replaced by a software-specified reciprocal of the denominator -- integer seed 0x7EF311C7 - bits(Q)
//            (5 % off), one cubic and one quadratic Newton step, all fmaf -- that runs NEXT TO the numerator, result fma(x P, r, half)
//   C4  C1 + C3        C5  C2 + C3
//   C7  C5 + chains 1..3 (dense: the lane's second chain, and chain 2) start from their first product instead of +0 (priced after C5 was adopted)
//   C6  (a)  C0's bits with the tap-1 chunk as v_readlane + v_pk_fma_f32 (two accumulators per register pair) instead of DPP fmacs
// Shapes: R registers only | P product shape (dense kernel from LDS, one 16-byte granule store per layer, run-time layer count)
//         | D product shape with the granule stores DEFERRED to behind the wave's last layer (a scheduling change, not a contract)
//         | G product shape with the dense kernels in registers | N product shape without the store (what the chain would see if the
//         store left the sample path) | M = N with the dense kernels in registers
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off scripts/ubench/chain_contract_ubench.hip -o scripts/ubench/chain_contract_ubench.exe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../tacotron-wavenet-vocoder-korean_amd/csrc/twv_math.hpp"
#include "../../tacotron-wavenet-vocoder-korean_amd/csrc/twv_dpp.hpp"
using namespace twv;

// ---------------------------------------------------------------- rounds 1-4's activation (contract C0): Horner rational + IEEE division.
// (the product's twv_math.hpp carries the candidate of this file since round 5 adopted it; the old form lives on here as the base line)
namespace ac2_r04 {
struct ActCoef {
    float clampv, a13, a11, a9, a7, a5, a3, a1, b10, b8, b6, b4, b2, b0;
    int is_sig;
};
__device__ __forceinline__ ActCoef act_coef(bool sig)
{
    ActCoef c;
    c.is_sig = sig ? 1 : 0;
    c.clampv = sig ? 18.0f : 9.0f;
    c.a13 = sig ? 0.0f : -2.76076847742355e-16f;
    c.a11 = sig ? 0.0f : 2.00018790482477e-13f;
    c.a9 = sig ? 4.37031012579801e-11f : -8.60467152213735e-11f;
    c.a7 = sig ? 1.15627324459942e-07f : 5.12229709037114e-08f;
    c.a5 = sig ? 6.08574864600143e-05f : 1.48572235717979e-05f;
    c.a3 = sig ? 8.51377133304701e-03f : 6.37261928875436e-04f;
    c.a1 = sig ? 2.48287947061529e-01f : 4.89352455891786e-03f;
    c.b10 = sig ? 6.10247389755681e-13f : 0.0f;
    c.b8 = sig ? 5.76102136993427e-09f : 0.0f;
    c.b6 = sig ? 6.29106785017040e-06f : 1.19825839466702e-06f;
    c.b4 = sig ? 1.70198817374094e-03f : 1.18534705686654e-04f;
    c.b2 = sig ? 1.16817656904453e-01f : 2.26843463243900e-03f;
    c.b0 = sig ? 9.93151921023180e-01f : 4.89352518554385e-03f;
    return c;
}
__device__ __forceinline__ float act_eval(const ActCoef& c, float x)
{
    x = x < -c.clampv ? -c.clampv : (x > c.clampv ? c.clampv : x);
    const float x2 = x * x;
    float p = fma_(x2, c.a13, c.a11);
    p = fma_(x2, p, c.a9);
    p = fma_(x2, p, c.a7);
    p = fma_(x2, p, c.a5);
    p = fma_(x2, p, c.a3);
    p = fma_(x2, p, c.a1);
    p = x * p;
    float q = fma_(x2, c.b10, c.b8);
    q = fma_(x2, q, c.b6);
    q = fma_(x2, q, c.b4);
    q = fma_(x2, q, c.b2);
    q = fma_(x2, q, c.b0);
    const float r = div_(p, q);
    return c.is_sig ? r + 0.5f : r;
}
// Latency-oriented form for the generation chain wave (one evaluation per layer on a lone wave, where instruction count is
// what matters): the odd numerator polynomial p (6 Horner steps) and the even denominator q (5 steps) share the multiplier
// x^2, so steps 2..6 of p run packed with steps 1..5 of q (v_pk_fma_f32) -- the same fmas, half the instructions.  In
// throughput code (many independent evaluations per thread, e.g. the Tacotron attention scores) the scalar form above is
// faster (measured: 13.7 vs 15.3 ms per Tacotron pass).
typedef float f32x2m __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float act_eval_pk(const ActCoef& c, float x)
{
    x = x < -c.clampv ? -c.clampv : (x > c.clampv ? c.clampv : x);
    const float x2 = x * x;
    const f32x2m xx = {x2, x2};
    f32x2m pq = {fma_(x2, c.a13, c.a11), c.b10};
    pq = __builtin_elementwise_fma(xx, pq, f32x2m{c.a9, c.b8});
    pq = __builtin_elementwise_fma(xx, pq, f32x2m{c.a7, c.b6});
    pq = __builtin_elementwise_fma(xx, pq, f32x2m{c.a5, c.b4});
    pq = __builtin_elementwise_fma(xx, pq, f32x2m{c.a3, c.b2});
    pq = __builtin_elementwise_fma(xx, pq, f32x2m{c.a1, c.b0});
    const float p = x * pq[0];
    const float r = div_(p, pq[1]);
    return c.is_sig ? r + 0.5f : r;
}
// the same with the clamp as ONE v_med3_f32 (three dependent instructions fewer per layer of the generation chain); identical for
// every non-NaN input (a NaN input stays NaN in both forms' consumers: the rational of a NaN is NaN)
__device__ __forceinline__ float act_eval_pk_med3(const ActCoef& c, float x)
{
    x = __builtin_amdgcn_fmed3f(x, -c.clampv, c.clampv);
    const float x2 = x * x;
    const f32x2m xx = {x2, x2};
    f32x2m pq = {fma_(x2, c.a13, c.a11), c.b10};
    pq = __builtin_elementwise_fma(xx, pq, f32x2m{c.a9, c.b8});
    pq = __builtin_elementwise_fma(xx, pq, f32x2m{c.a7, c.b6});
    pq = __builtin_elementwise_fma(xx, pq, f32x2m{c.a5, c.b4});
    pq = __builtin_elementwise_fma(xx, pq, f32x2m{c.a3, c.b2});
    pq = __builtin_elementwise_fma(xx, pq, f32x2m{c.a1, c.b0});
    const float p = x * pq[0];
    const float r = div_(p, pq[1]);
    return c.is_sig ? r + 0.5f : r;
}
}  // namespace ac2_r04

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(r_), __LINE__); exit(1); } } while (0)

constexpr int NLU = 4;
struct LayerCanon {
    float Wc[32][64];      // tap-1 conv kernel [k][f 0..31 | g 0..31]
    float Wd[32][32];      // dense [k][o]
    float pre[64], bfg[64], gcv[64], lcv[64], bd[32];
};

// ---------------------------------------------------------------- the candidate activation (contract C3)
constexpr unsigned kRcpMagicU = 0x7EF311C7u;
struct ActCoef2 { float cl, a1, a3, a5, a7, a9, a11, a13, b0, b2, b4, b6, b8, b10, half; };
__host__ __device__ inline ActCoef2 act_coef2(bool sig)
{
    ActCoef2 c;
    c.cl = sig ? 18.0f : 9.0f;
    c.a1 = sig ? 2.48287947061529e-01f : 4.89352455891786e-03f;
    c.a3 = sig ? 8.51377133304701e-03f : 6.37261928875436e-04f;
    c.a5 = sig ? 6.08574864600143e-05f : 1.48572235717979e-05f;
    c.a7 = sig ? 1.15627324459942e-07f : 5.12229709037114e-08f;
    c.a9 = sig ? 4.37031012579801e-11f : -8.60467152213735e-11f;
    c.a11 = sig ? 0.0f : 2.00018790482477e-13f;
    c.a13 = sig ? 0.0f : -2.76076847742355e-16f;
    c.b0 = sig ? 9.93151921023180e-01f : 4.89352518554385e-03f;
    c.b2 = sig ? 1.16817656904453e-01f : 2.26843463243900e-03f;
    c.b4 = sig ? 1.70198817374094e-03f : 1.18534705686654e-04f;
    c.b6 = sig ? 6.29106785017040e-06f : 1.19825839466702e-06f;
    c.b8 = sig ? 5.76102136993427e-09f : 0.0f;
    c.b10 = sig ? 6.10247389755681e-13f : 0.0f;
    c.half = sig ? 0.5f : -0.0f;          // x + (-0) == x for every x, signed zeros included
    return c;
}
// plain form (what oracle/mathfn.c would hold); host and device
__host__ __device__ inline float act2_plain(const ActCoef2& c, float x)
{
    x = x < -c.cl ? -c.cl : (x > c.cl ? c.cl : x);
    const float t = x * x, t2 = t * t, t4 = t2 * t2;
    const float p01 = fmaf(c.a3, t, c.a1), p23 = fmaf(c.a7, t, c.a5), p45 = fmaf(c.a11, t, c.a9);
    const float q01 = fmaf(c.b2, t, c.b0), q23 = fmaf(c.b6, t, c.b4), q45 = fmaf(c.b10, t, c.b8);
    const float p456 = fmaf(c.a13, t2, p45);
    const float p03 = fmaf(p23, t2, p01), q03 = fmaf(q23, t2, q01);
    const float P = fmaf(p456, t4, p03), Q = fmaf(q45, t4, q03);
    const float xp = x * P;
    unsigned qb; memcpy(&qb, &Q, 4);
    const unsigned rb = kRcpMagicU - qb;
    float r; memcpy(&r, &rb, 4);
    float e = fmaf(-Q, r, 1.0f);
    const float s = fmaf(e, e, e);
    r = fmaf(r, s, r);
    e = fmaf(-Q, r, 1.0f);
    r = fmaf(r, e, r);
    return fmaf(xp, r, c.half);
}
// latency-oriented device form: the same operations, packed where two of them share a multiplier
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float act2_pk(const ActCoef2& c, float x)
{
    x = __builtin_amdgcn_fmed3f(x, -c.cl, c.cl);
    const float t = x * x;
    const f2 tt = {t, t};
    const f2 pq01 = __builtin_elementwise_fma(f2{c.a3, c.b2}, tt, f2{c.a1, c.b0});
    const f2 pq23 = __builtin_elementwise_fma(f2{c.a7, c.b6}, tt, f2{c.a5, c.b4});
    const f2 pq45 = __builtin_elementwise_fma(f2{c.a11, c.b10}, tt, f2{c.a9, c.b8});
    const float t2 = t * t;
    const float p456 = fma_(c.a13, t2, pq45[0]);
    const float t4 = t2 * t2;
    const f2 pq03 = __builtin_elementwise_fma(pq23, f2{t2, t2}, pq01);
    const f2 PQ = __builtin_elementwise_fma(f2{p456, pq45[1]}, f2{t4, t4}, pq03);
    const float xp = x * PQ[0];
    const float Q = PQ[1];
    float r = __uint_as_float(kRcpMagicU - __float_as_uint(Q));
    float e = fma_(-Q, r, 1.0f);
    const float s = fma_(e, e, e);
    r = fma_(r, s, r);
    e = fma_(-Q, r, 1.0f);
    r = fma_(r, e, r);
    return fma_(xp, r, c.half);
}

// ---------------------------------------------------------------- dots with a start value in chain 0 (contract C2)
__device__ __forceinline__ float dot32_dpp_init(const float (&w)[32], float xa, float xb, float init)
{
    float c0 = init, c1 = 0.0f, c2 = 0.0f, c3 = 0.0f;
    asm volatile(
        "s_nop 1\n" TWV_ALIGN8
        "v_fmac_f32_dpp %0, %4, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %4, %6 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %2, %4, %7 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %3, %4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %4, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %4, %10 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %2, %4, %11 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %3, %4, %12 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %4, %13 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %4, %14 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %2, %4, %15 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %3, %4, %16 row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %4, %17 row_newbcast:12 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %4, %18 row_newbcast:13 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %2, %4, %19 row_newbcast:14 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %3, %4, %20 row_newbcast:15 row_mask:0xf bank_mask:0xf"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
        : "v"(xa), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "v"(w[9]), "v"(w[10]), "v"(w[11]), "v"(w[12]), "v"(w[13]), "v"(w[14]), "v"(w[15]));
    asm volatile(
        "s_nop 1\n" TWV_ALIGN8
        "v_fmac_f32_dpp %0, %4, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %4, %6 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %2, %4, %7 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %3, %4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %4, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %4, %10 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %2, %4, %11 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %3, %4, %12 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %4, %13 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %4, %14 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %2, %4, %15 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %3, %4, %16 row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %4, %17 row_newbcast:12 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %4, %18 row_newbcast:13 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %2, %4, %19 row_newbcast:14 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %3, %4, %20 row_newbcast:15 row_mask:0xf bank_mask:0xf"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
        : "v"(xb), "v"(w[16]), "v"(w[17]), "v"(w[18]), "v"(w[19]), "v"(w[20]), "v"(w[21]), "v"(w[22]), "v"(w[23]), "v"(w[24]), "v"(w[25]), "v"(w[26]), "v"(w[27]), "v"(w[28]), "v"(w[29]), "v"(w[30]), "v"(w[31]));
    return (c0 + c1) + (c2 + c3);
}
__device__ __forceinline__ float dot16_dpp_init(const float (&w)[16], float z, float init)
{
    float c0 = init, c1 = 0.0f;
    asm volatile(
        "s_nop 1\n" TWV_ALIGN8
        "v_fmac_f32_dpp %0, %2, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %2, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %2, %5 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %2, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %2, %7 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %2, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %2, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %2, %10 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %2, %11 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %2, %12 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %2, %13 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %2, %14 row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %2, %15 row_newbcast:12 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %2, %16 row_newbcast:13 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %2, %17 row_newbcast:14 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %2, %18 row_newbcast:15 row_mask:0xf bank_mask:0xf"
        : "+v"(c0), "+v"(c1)
        : "v"(z), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "v"(w[9]), "v"(w[10]), "v"(w[11]), "v"(w[12]), "v"(w[13]), "v"(w[14]), "v"(w[15]));
    return c0 + c1;
}
// contract C7: chain 0 from the addend, chains 1..3 from their FIRST PRODUCT (v_mul_f32_dpp: no zero-initialised accumulators;
// fma(w, x, -0) == w * x for every w, x, so a C restatement starts those chains from -0)
__device__ __forceinline__ float dot32_dpp_pf(const float (&w)[32], float xa, float xb, float init)
{
    float c0 = init, c1, c2, c3;
    asm volatile(
        "s_nop 1\n" TWV_ALIGN8
        "v_fmac_f32_dpp %0, %4, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
        "v_mul_f32_dpp %1, %4, %6 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
        "v_mul_f32_dpp %2, %4, %7 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
        "v_mul_f32_dpp %3, %4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %4, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %4, %10 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %2, %4, %11 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %3, %4, %12 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %4, %13 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %4, %14 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %2, %4, %15 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %3, %4, %16 row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %4, %17 row_newbcast:12 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %4, %18 row_newbcast:13 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %2, %4, %19 row_newbcast:14 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %3, %4, %20 row_newbcast:15 row_mask:0xf bank_mask:0xf"
        : "+v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3)
        : "v"(xa), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "v"(w[9]), "v"(w[10]), "v"(w[11]), "v"(w[12]), "v"(w[13]), "v"(w[14]), "v"(w[15]));
    asm volatile(
        TWV_ALIGN8
        "v_fmac_f32_dpp %0, %4, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %4, %6 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %2, %4, %7 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %3, %4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %4, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %4, %10 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %2, %4, %11 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %3, %4, %12 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %4, %13 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %4, %14 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %2, %4, %15 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %3, %4, %16 row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %4, %17 row_newbcast:12 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %4, %18 row_newbcast:13 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %2, %4, %19 row_newbcast:14 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %3, %4, %20 row_newbcast:15 row_mask:0xf bank_mask:0xf"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
        : "v"(xb), "v"(w[16]), "v"(w[17]), "v"(w[18]), "v"(w[19]), "v"(w[20]), "v"(w[21]), "v"(w[22]), "v"(w[23]), "v"(w[24]), "v"(w[25]), "v"(w[26]), "v"(w[27]), "v"(w[28]), "v"(w[29]), "v"(w[30]), "v"(w[31]));
    return (c0 + c1) + (c2 + c3);
}
// dense half chunk: the lane's first chain from `init` (even rows: the bias = chain 0; odd rows: -0 = chain 2's first product), its
// second chain from the first product
__device__ __forceinline__ float dot16_dpp_pf(const float (&w)[16], float z, float init)
{
    float c0 = init, c1;
    asm volatile(
        "s_nop 1\n" TWV_ALIGN8
        "v_fmac_f32_dpp %0, %2, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
        "v_mul_f32_dpp %1, %2, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %2, %5 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %2, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %2, %7 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %2, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %2, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %2, %10 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %2, %11 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %2, %12 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %2, %13 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %2, %14 row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %2, %15 row_newbcast:12 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %2, %16 row_newbcast:13 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %2, %17 row_newbcast:14 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %2, %18 row_newbcast:15 row_mask:0xf bank_mask:0xf"
        : "+v"(c0), "=&v"(c1)
        : "v"(z), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "v"(w[9]), "v"(w[10]), "v"(w[11]), "v"(w[12]), "v"(w[13]), "v"(w[14]), "v"(w[15]));
    return c0 + c1;
}
// (a) C0's chunk through scalar broadcasts: x[k] -> SGPR with v_readlane, two packed accumulators (chains 0|1 and 2|3)
__device__ __forceinline__ float dot32_readlane_pk(const float (&w)[32], float X)
{
    f2 s01 = {0.0f, 0.0f}, s23 = {0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 32; k += 4) {
        // X layout: x[j] in lane j (j < 16), lane 16 + j (j >= 16)
        const float x0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(X), k < 16 ? k : 16 + k));
        const float x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(X), k + 1 < 16 ? k + 1 : 17 + k));
        const float x2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(X), k + 2 < 16 ? k + 2 : 18 + k));
        const float x3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(X), k + 3 < 16 ? k + 3 : 19 + k));
        s01 = __builtin_elementwise_fma(f2{w[k], w[k + 1]}, f2{x0, x1}, s01);
        s23 = __builtin_elementwise_fma(f2{w[k + 2], w[k + 3]}, f2{x2, x3}, s23);
    }
    return (s01[0] + s01[1]) + (s23[0] + s23[1]);
}

// ---------------------------------------------------------------- the layer body under contract C
struct LayerIn { float pre, bfg, gcv, lcv, A, bd, bd_init, bd_init7; };
template <int C>
__device__ __forceinline__ float front(const float (&wc)[32], const LayerIn& p, const ac2_r04::ActCoef& co, const ActCoef2& co2, float X)
{
    float v;
    if (C == 6) {
        v = p.pre + dot32_readlane_pk(wc, X);
        v = v + p.bfg; v = v + p.gcv; v = v + p.lcv;
    } else {
        const auto xs = __builtin_amdgcn_permlane32_swap(__float_as_uint(X), __float_as_uint(X), false, false);
        const float xa = __uint_as_float(xs[0]), xb = __uint_as_float(xs[1]);
        if (C == 0 || C == 3) {
            v = p.pre + dot32_dpp(wc, xa, xb);
            v = v + p.bfg; v = v + p.gcv; v = v + p.lcv;
        } else if (C == 1 || C == 4) {
            v = dot32_dpp(wc, xa, xb) + p.A;
        } else if (C == 7) {
            v = dot32_dpp_pf(wc, xa, xb, p.A);
        } else {
            v = dot32_dpp_init(wc, xa, xb, p.A);
        }
    }
    const float act = ((C >= 3 && C <= 5) || C == 7) ? act2_pk(co2, v) : ac2_r04::act_eval_pk_med3(co, v);
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(act), __float_as_uint(act), false, false);
    return __uint_as_float(sw[0]) * __uint_as_float(sw[1]);
}
template <int C>
__device__ __forceinline__ void back(const float (&wd)[16], const LayerIn& p, float z, float& X)
{
    if (C == 7) {
        const float s = dot16_dpp_pf(wd, z, p.bd_init7);
        const auto ds = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
        X = X + (__uint_as_float(ds[0]) + __uint_as_float(ds[1]));
    } else if (C == 2 || C == 5) {
        const float s = dot16_dpp_init(wd, z, p.bd_init);
        const auto ds = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
        X = X + (__uint_as_float(ds[0]) + __uint_as_float(ds[1]));
    } else {
        const float s = dot16_dpp(wd, z);
        const auto ds = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
        float tr = __uint_as_float(ds[0]) + __uint_as_float(ds[1]);
        tr = tr + p.bd;
        X = X + tr;
    }
}

// canonical formulation of contract C (lane j: filter j / gate j-32; operands through shuffles; plain fmaf chains)
template <int C>
__global__ void __launch_bounds__(64) layer_ref_kernel(const LayerCanon* Lc, const float* x0, float* xout, int steps)
{
    const int lane = threadIdx.x;
    const ac2_r04::ActCoef coef = ac2_r04::act_coef(lane >= 32);
    const ActCoef2 coef2 = act_coef2(lane >= 32);
    constexpr bool presum = (C == 1 || C == 2 || C == 4 || C == 5 || C == 7), init = (C == 2 || C == 5 || C == 7), act2 = ((C >= 3 && C <= 5) || C == 7);
    constexpr float z0 = (C == 7) ? -0.0f : 0.0f;             // C7: the chains without a start value begin with their first product
    float x = x0[lane & 31];
    for (int t = 0; t < steps; ++t) {
        for (int l = 0; l < NLU; ++l) {
            const LayerCanon& P = Lc[l];
            const float A = ((P.pre[lane] + P.bfg[lane]) + P.gcv[lane]) + P.lcv[lane];
            float s[4] = {init ? A : 0.0f, z0, z0, z0};
            for (int k = 0; k < 32; ++k) s[k & 3] = fma_(P.Wc[k][lane], __shfl(x, k), s[k & 3]);
            const float chunk = (s[0] + s[1]) + (s[2] + s[3]);
            float v;
            if (init) v = chunk;
            else if (presum) v = chunk + A;
            else { v = P.pre[lane] + chunk; v = v + P.bfg[lane]; v = v + P.gcv[lane]; v = v + P.lcv[lane]; }
            const float act = act2 ? act2_plain(coef2, v) : ac2_r04::act_eval(coef, v);
            const float z = __shfl(act, lane & 31) * __shfl(act, 32 + (lane & 31));
            float q[4] = {init ? P.bd[lane & 31] : 0.0f, z0, z0, z0};
            for (int k = 0; k < 32; ++k) q[k & 3] = fma_(P.Wd[k][lane & 31], __shfl(z, k), q[k & 3]);
            float tr = (q[0] + q[1]) + (q[2] + q[3]);
            if (!init) tr = tr + P.bd[lane & 31];
            x = x + tr;
        }
        x = x * 0.25f;      // keep the recursion bounded
    }
    if (lane < 32) xout[lane] = x;
}

// SHAPE bit0: dense kernel from LDS; bit1: one 16-byte granule store per layer between the halves; bit2: run-time layer count;
//       bit4: the granule stores deferred to behind the last layer
typedef unsigned u32x4t __attribute__((ext_vector_type(4)));
typedef float f32x4t __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
#define ULDS4(o) (((__attribute__((address_space(3))) f32x4t*)dyn_lds)[(o)])
template <int C, int SHAPE>
__global__ void __launch_bounds__(64) layer_var_kernel(const LayerCanon* Lc, const float* x0, float* xout, int steps, unsigned long long* cyc, unsigned long long* gran, int nl_rt)
{
    const int lane = threadIdx.x;
    const ac2_r04::ActCoef coef = ac2_r04::act_coef(lane >= 32);
    const ActCoef2 coef2 = act_coef2(lane >= 32);
    float wc[NLU][32], wdr[NLU][16];
    LayerIn in[NLU];
    const int oc = dpp_conv_out(lane), od = dpp_dense_out(lane);
#pragma unroll
    for (int l = 0; l < NLU; ++l) {
#pragma unroll
        for (int k = 0; k < 32; ++k) wc[l][k] = Lc[l].Wc[k][oc];
#pragma unroll
        for (int i = 0; i < 16; ++i) wdr[l][i] = Lc[l].Wd[dpp_dense_k(lane, i)][od];
#pragma unroll
        for (int q = 0; q < 4; ++q) ULDS4((l * 4 + q) * 64 + lane) = f32x4t{wdr[l][4 * q], wdr[l][4 * q + 1], wdr[l][4 * q + 2], wdr[l][4 * q + 3]};
        in[l].pre = Lc[l].pre[oc]; in[l].bfg = Lc[l].bfg[oc]; in[l].gcv = Lc[l].gcv[oc]; in[l].lcv = Lc[l].lcv[oc];
        in[l].A = ((in[l].pre + in[l].bfg) + in[l].gcv) + in[l].lcv;
        in[l].bd = Lc[l].bd[od];
        in[l].bd_init = ((lane >> 4) & 1) ? 0.0f : in[l].bd;        // chain 0 of the dense chunk lives on the even rows
        in[l].bd_init7 = ((lane >> 4) & 1) ? -0.0f : in[l].bd;      // C7: the odd rows' first chain (chain 2) starts from its first product
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(gran, 0, 1 << 20, 0x00020000);
    float X = x0[od];
    const int nl = (SHAPE & 4) ? nl_rt : NLU;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long w0 = wall_clock64();
    for (int t = 0; t < steps; ++t) {
        const unsigned tag = (unsigned)t + 1u;
        float zs[NLU], xs[NLU];
#pragma unroll
        for (int l = 0; l < NLU; ++l) {
            if (l < nl) {
                float wd[16];
                if (SHAPE & 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const f32x4t v = ULDS4((l * 4 + q) * 64 + lane); wd[4 * q] = v.x; wd[4 * q + 1] = v.y; wd[4 * q + 2] = v.z; wd[4 * q + 3] = v.w; }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) wd[i] = wdr[l][i];
                }
                const float z = front<C>(wc[l], in[l], coef, coef2, X);
                if (SHAPE & 2) {
                    const u32x4t d = {__float_as_uint(z), tag, __float_as_uint(X), tag};
                    __builtin_amdgcn_raw_buffer_store_b128(d, rs, lane * 16, l * 1024, 0);
                    asm volatile("s_nop 1" ::"v"(d) : "memory");
                }
                zs[l] = z; xs[l] = X;
                back<C>(wd, in[l], z, X);
            }
        }
        if (SHAPE & 16) {
#pragma unroll
            for (int l = 0; l < NLU; ++l) {
                if (l < nl) {
                    const u32x4t d = {__float_as_uint(zs[l]), tag, __float_as_uint(xs[l]), tag};
                    __builtin_amdgcn_raw_buffer_store_b128(d, rs, lane * 16, l * 1024, 0);
                    asm volatile("s_nop 1" ::"v"(d) : "memory");
                }
            }
        }
        X = X * 0.25f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long w1 = wall_clock64();
    if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
    if ((lane & 16) == 0) xout[od] = X;
}

// activation sweep: device act2_pk / act2_plain against the host's act2_plain (gcc-equivalent fmaf code)
__global__ void act_sweep_kernel(const float* x, float* y_pk, float* y_plain, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool sig = (i & 1) != 0;
    const ActCoef2 c = act_coef2(sig);
    y_pk[i] = act2_pk(c, x[i]);
    y_plain[i] = act2_plain(c, x[i]);
}
// dependent-chain timing of the two activations alone
template <int WHICH>
__global__ void __launch_bounds__(64) act_time_kernel(float* out, unsigned long long* cyc, int reps)
{
    const int lane = threadIdx.x;
    const ac2_r04::ActCoef coef = ac2_r04::act_coef(lane >= 32);
    const ActCoef2 coef2 = act_coef2(lane >= 32);
    float v = 0.01f * (float)lane - 0.3f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        const float a = WHICH ? act2_pk(coef2, v) : ac2_r04::act_eval_pk_med3(coef, v);
        v = a * 0.9f + 0.05f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[0] = t1 - t0;
    out[lane] = v;
}

static float frand() { return (float)rand() / (float)RAND_MAX * 2.0f - 1.0f; }

template <int C> static void run_contract(const char* name, LayerCanon* dL, float* dx0, float* dxa, float* dxb, unsigned long long* dc, unsigned long long* gran, int steps, double base_p)
{
    float xa[32], xb[32]; unsigned long long c[2];
    hipLaunchKernelGGL(layer_ref_kernel<C>, dim3(1), dim3(64), 0, 0, dL, dx0, dxa, steps);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(xa, dxa, 128, hipMemcpyDeviceToHost));
    const char* sn[6] = {"R registers only", "P product shape", "D product, stores behind the last layer", "G product, dense kernel in registers",
                         "N product without the granule store", "M product, no store, dense in registers"};
    printf("%s\n", name);
    for (int s = 0; s < 6; ++s) {
        switch (s) {
            case 0: hipLaunchKernelGGL((layer_var_kernel<C, 0>), dim3(1), dim3(64), 32768, 0, dL, dx0, dxb, steps, dc, gran, NLU); break;
            case 1: hipLaunchKernelGGL((layer_var_kernel<C, 7>), dim3(1), dim3(64), 32768, 0, dL, dx0, dxb, steps, dc, gran, NLU); break;
            case 2: hipLaunchKernelGGL((layer_var_kernel<C, 21>), dim3(1), dim3(64), 32768, 0, dL, dx0, dxb, steps, dc, gran, NLU); break;
            case 3: hipLaunchKernelGGL((layer_var_kernel<C, 6>), dim3(1), dim3(64), 32768, 0, dL, dx0, dxb, steps, dc, gran, NLU); break;
            case 4: hipLaunchKernelGGL((layer_var_kernel<C, 5>), dim3(1), dim3(64), 32768, 0, dL, dx0, dxb, steps, dc, gran, NLU); break;
            default: hipLaunchKernelGGL((layer_var_kernel<C, 4>), dim3(1), dim3(64), 32768, 0, dL, dx0, dxb, steps, dc, gran, NLU); break;
        }
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(xb, dxb, 128, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < 32; ++i) bad += memcmp(&xa[i], &xb[i], 4) != 0;
        const double ticks = (double)c[0] / steps / NLU, ns = (double)c[1] * 10.0 / steps / NLU;
        printf("    %-38s %6.1f ticks = %6.1f ns per layer  (x mismatches vs canonical %d)", sn[s], ticks, ns, bad);
        if (s == 1 && base_p > 0) printf("   %+5.1f %% vs C0 product shape", 100.0 * (ns - base_p) / base_p);
        printf("\n");
        fflush(stdout);
    }
}

int main()
{
    // ---- activation: device forms vs the host's plain form
    {
        const int n = 1 << 22;
        std::vector<float> x(n), hp(n), hq(n);
        srand(7);
        for (int i = 0; i < n; ++i) {
            const int kind = i % 5;
            float v = frand() * (kind == 0 ? 20.0f : kind == 1 ? 9.5f : kind == 2 ? 1.0f : kind == 3 ? 1e-3f : 1e-20f);
            if (i < 64) { const float sp[8] = {0.0f, -0.0f, 9.0f, -9.0f, 18.0f, -18.0f, 1e-40f, -1e-40f}; v = sp[i & 7]; }
            x[i] = v;
        }
        float *dx, *dp, *dq;
        CHECK(hipMalloc(&dx, n * 4)); CHECK(hipMalloc(&dp, n * 4)); CHECK(hipMalloc(&dq, n * 4));
        CHECK(hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(act_sweep_kernel, dim3(n / 256), dim3(256), 0, 0, dx, dp, dq, n);
        CHECK(hipMemcpy(hp.data(), dp, n * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hq.data(), dq, n * 4, hipMemcpyDeviceToHost));
        long bad_pk = 0, bad_plain = 0; double maxerr[2] = {0, 0};
        for (int i = 0; i < n; ++i) {
            const bool sig = i & 1;
            const ActCoef2 c = act_coef2(sig);
            const float want = act2_plain(c, x[i]);
            bad_pk += memcmp(&want, &hp[i], 4) != 0; bad_plain += memcmp(&want, &hq[i], 4) != 0;
            const double ref = sig ? 1.0 / (1.0 + exp(-(double)x[i])) : tanh((double)x[i]);
            const double e = fabs((double)want - ref); if (e > maxerr[sig]) maxerr[sig] = e;
        }
        printf("[act] candidate activation, %d inputs: device packed form vs host %ld mismatches, device plain form vs host %ld; max abs error vs float64 tanh %.3g, logistic %.3g\n",
               n, bad_pk, bad_plain, maxerr[0], maxerr[1]);
        float* dout; unsigned long long* dc; CHECK(hipMalloc(&dout, 256)); CHECK(hipMalloc(&dc, 64));
        unsigned long long c0, c1; const int reps = 20000;
        hipLaunchKernelGGL(act_time_kernel<0>, dim3(1), dim3(64), 0, 0, dout, dc, reps); CHECK(hipMemcpy(&c0, dc, 8, hipMemcpyDeviceToHost));
        hipLaunchKernelGGL(act_time_kernel<1>, dim3(1), dim3(64), 0, 0, dout, dc, reps); CHECK(hipMemcpy(&c1, dc, 8, hipMemcpyDeviceToHost));
        printf("[act] dependent evaluations on a lone wave (+ one fma of glue): AC-2 rational with IEEE division %.1f ticks, candidate %.1f ticks\n", (double)c0 / reps, (double)c1 / reps);
        fflush(stdout);
    }
    // ---- the layer body
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
    LayerCanon* dL; float *dx0, *dxa, *dxb; unsigned long long *dc, *gran;
    CHECK(hipMalloc(&dL, sizeof(LayerCanon) * NLU)); CHECK(hipMalloc(&dx0, 128)); CHECK(hipMalloc(&dxa, 128)); CHECK(hipMalloc(&dxb, 128));
    CHECK(hipMalloc(&dc, 64)); CHECK(hipMalloc(&gran, 1 << 20));
    CHECK(hipMemcpy(dL, L.data(), sizeof(LayerCanon) * NLU, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dx0, x0.data(), 128, hipMemcpyHostToDevice));
    const int steps = 5000;
    // C0's product shape first, as the base of the percentages
    double base_p = 0;
    {
        unsigned long long c[2];
        hipLaunchKernelGGL((layer_var_kernel<0, 7>), dim3(1), dim3(64), 32768, 0, dL, dx0, dxb, steps, dc, gran, NLU);
        CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost));
        base_p = (double)c[1] * 10.0 / steps / NLU;
    }
    run_contract<0>("C0  product contract (AC-1 / AC-2)", dL, dx0, dxa, dxb, dc, gran, steps, base_p);
    run_contract<6>("C6  (a) same bits, tap-1 chunk as v_readlane + v_pk_fma_f32", dL, dx0, dxa, dxb, dc, gran, steps, base_p);
    run_contract<1>("C1  (b) pre-summed addend: v = tap1 + A", dL, dx0, dxa, dxb, dc, gran, steps, base_p);
    run_contract<2>("C2  (b') addend / dense bias as the start value of chain 0", dL, dx0, dxa, dxb, dc, gran, steps, base_p);
    run_contract<3>("C3  (c) Estrin rational + software reciprocal", dL, dx0, dxa, dxb, dc, gran, steps, base_p);
    run_contract<4>("C4  (b) + (c)", dL, dx0, dxa, dxb, dc, gran, steps, base_p);
    run_contract<5>("C5  (b') + (c)", dL, dx0, dxa, dxb, dc, gran, steps, base_p);
    run_contract<7>("C7  C5 + the chains without a start value begin with their first product   [= the product's contract since round 5: AC-1b / AC-2]", dL, dx0, dxa, dxb, dc, gran, steps, base_p);
    return 0;
}
