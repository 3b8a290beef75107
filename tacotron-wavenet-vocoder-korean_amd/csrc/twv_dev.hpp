// twv_dev.hpp -- device-side helpers shared by the gfx950 translation units: the 64x32 weight TILE and the chunk
// evaluators of the arithmetic contract, explicit LDS address-space accessors, buffer-descriptor loads.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include "twv_layout.hpp"
#include "twv_math.hpp"

using namespace twv;

struct Tile { float w[32]; };

__device__ __forceinline__ void load_tile(Tile& t, const float* base, int lane)
{
    const float4* p = reinterpret_cast<const float4*>(base) + lane;
#pragma unroll
    for (int kq = 0; kq < 8; ++kq) {
        const float4 q = p[kq * 64];
        t.w[4 * kq + 0] = q.x; t.w[4 * kq + 1] = q.y; t.w[4 * kq + 2] = q.z; t.w[4 * kq + 3] = q.w;
    }
}

// Packed fp32 fma (v_pk_fma_f32): two IEEE fmas per instruction.  The chains of AC-1 are written as two explicit 2-vectors
// (s0,s1) and (s2,s3) -- left to the auto-vectoriser, one of the chain wave's two dot products silently stayed scalar
// (32 v_fma instead of 16 v_pk_fma) whenever unrelated code changed, a 12 % swing of the whole generation kernel.
typedef float f32x2p __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2p pk_fma(f32x2p a, f32x2p b, f32x2p c) { return __builtin_elementwise_fma(a, b, c); }

// one chunk of AC-1 (four interleaved fma chains, (s0+s1)+(s2+s3)), operand vector distributed over lanes 0..31 of `xv`.
// The interleave is what hides v_readlane's ~17-cycle result latency (measured: 262 vs 540 ticks per chunk).
__device__ __forceinline__ float dot_readlane(const Tile& t, float xv)
{
    f32x2p s01 = {0.0f, 0.0f}, s23 = {0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < 32; c += 4) {
        const f32x2p x01 = {__int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), c + 0)),
                            __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), c + 1))};
        const f32x2p x23 = {__int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), c + 2)),
                            __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), c + 3))};
        s01 = pk_fma(f32x2p{t.w[c + 0], t.w[c + 1]}, x01, s01);
        s23 = pk_fma(f32x2p{t.w[c + 2], t.w[c + 3]}, x23, s23);
    }
    return (s01[0] + s01[1]) + (s23[0] + s23[1]);
}

// dot_readlane with the 32 v_readlane and the 16 v_pk_fma_f32 software-pipelined by hand (same operations, same order of the fma chains:
// bit-identical).  The compiler emits all readlanes first and then the two dependent fma chains (16 x ~6 cycles of latency-bound
// issue, scripts/ubench/vgpr_bank.hip); interleaved, the readlanes of pair i+3 fill the latency of fma i and the chunk is issue-bound.
// Each SGPR pair is consumed >= 6 instructions after it is written; eight pairs s[84:99] rotate.
__device__ __forceinline__ float dot_readlane_pipe_a(const float (&tw)[32], float xv)
{
    f32x2p a, b;
    asm volatile(
        "s_nop 1\n"                                    // the operand may come straight out of a VALU instruction: hazards inside inline asm are ours
        "\t.p2align 3\n"                              // 8-byte instructions at 8-byte addresses (see TWV_ALIGN8, twv_dpp.hpp)
        "v_readlane_b32 s84, %[x], 0\n"
        "v_readlane_b32 s85, %[x], 1\n"
        "v_readlane_b32 s86, %[x], 2\n"
        "v_readlane_b32 s87, %[x], 3\n"
        "v_readlane_b32 s88, %[x], 4\n"
        "v_readlane_b32 s89, %[x], 5\n"
        "v_pk_fma_f32 %[a], %[p0], s[84:85], 0 op_sel_hi:[1,1,0]\n"
        "v_readlane_b32 s90, %[x], 6\n"
        "v_readlane_b32 s91, %[x], 7\n"
        "v_pk_fma_f32 %[b], %[p1], s[86:87], 0 op_sel_hi:[1,1,0]\n"
        "v_readlane_b32 s92, %[x], 8\n"
        "v_readlane_b32 s93, %[x], 9\n"
        "v_pk_fma_f32 %[a], %[p2], s[88:89], %[a]\n"
        "v_readlane_b32 s94, %[x], 10\n"
        "v_readlane_b32 s95, %[x], 11\n"
        "v_pk_fma_f32 %[b], %[p3], s[90:91], %[b]\n"
        "v_readlane_b32 s96, %[x], 12\n"
        "v_readlane_b32 s97, %[x], 13\n"
        "v_pk_fma_f32 %[a], %[p4], s[92:93], %[a]\n"
        "v_readlane_b32 s98, %[x], 14\n"
        "v_readlane_b32 s99, %[x], 15\n"
        "v_pk_fma_f32 %[b], %[p5], s[94:95], %[b]\n"
        "v_readlane_b32 s84, %[x], 16\n"
        "v_readlane_b32 s85, %[x], 17\n"
        "v_pk_fma_f32 %[a], %[p6], s[96:97], %[a]\n"
        "v_readlane_b32 s86, %[x], 18\n"
        "v_readlane_b32 s87, %[x], 19\n"
        "v_pk_fma_f32 %[b], %[p7], s[98:99], %[b]\n"
        "v_readlane_b32 s88, %[x], 20\n"
        "v_readlane_b32 s89, %[x], 21\n"
        "v_pk_fma_f32 %[a], %[p8], s[84:85], %[a]\n"
        "v_readlane_b32 s90, %[x], 22\n"
        "v_readlane_b32 s91, %[x], 23\n"
        "v_pk_fma_f32 %[b], %[p9], s[86:87], %[b]\n"
        "v_readlane_b32 s92, %[x], 24\n"
        "v_readlane_b32 s93, %[x], 25\n"
        "v_pk_fma_f32 %[a], %[p10], s[88:89], %[a]\n"
        "v_readlane_b32 s94, %[x], 26\n"
        "v_readlane_b32 s95, %[x], 27\n"
        "v_pk_fma_f32 %[b], %[p11], s[90:91], %[b]\n"
        "v_readlane_b32 s96, %[x], 28\n"
        "v_readlane_b32 s97, %[x], 29\n"
        "v_pk_fma_f32 %[a], %[p12], s[92:93], %[a]\n"
        "v_readlane_b32 s98, %[x], 30\n"
        "v_readlane_b32 s99, %[x], 31\n"
        "v_pk_fma_f32 %[b], %[p13], s[94:95], %[b]\n"
        "v_pk_fma_f32 %[a], %[p14], s[96:97], %[a]\n"
        "v_pk_fma_f32 %[b], %[p15], s[98:99], %[b]"
        : [a] "=&v"(a), [b] "=&v"(b)
        : [x] "v"(xv),
          [p0] "v"(f32x2p{tw[0], tw[1]}),
          [p1] "v"(f32x2p{tw[2], tw[3]}),
          [p2] "v"(f32x2p{tw[4], tw[5]}),
          [p3] "v"(f32x2p{tw[6], tw[7]}),
          [p4] "v"(f32x2p{tw[8], tw[9]}),
          [p5] "v"(f32x2p{tw[10], tw[11]}),
          [p6] "v"(f32x2p{tw[12], tw[13]}),
          [p7] "v"(f32x2p{tw[14], tw[15]}),
          [p8] "v"(f32x2p{tw[16], tw[17]}),
          [p9] "v"(f32x2p{tw[18], tw[19]}),
          [p10] "v"(f32x2p{tw[20], tw[21]}),
          [p11] "v"(f32x2p{tw[22], tw[23]}),
          [p12] "v"(f32x2p{tw[24], tw[25]}),
          [p13] "v"(f32x2p{tw[26], tw[27]}),
          [p14] "v"(f32x2p{tw[28], tw[29]}),
          [p15] "v"(f32x2p{tw[30], tw[31]})
        : "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99");
    return (a[0] + a[1]) + (b[0] + b[1]);
}
// the same with chain 0 started from `init` and chains 1..3 from their first product (= from -0) instead of +0 (AC-1b: the last chunk of a contraction on the generation chain starts from
// the addend -- earlier chunks, bias, conditioning -- so no add follows the dot product on the sample-to-sample path)
__device__ __forceinline__ float dot_readlane_pipe_init_a(const float (&tw)[32], float xv, float init)
{
    f32x2p a, b;
    asm volatile(
        "s_nop 1\n"                                    // the operand may come straight out of a VALU instruction: hazards inside inline asm are ours
        "\t.p2align 3\n"                              // 8-byte instructions at 8-byte addresses (see TWV_ALIGN8, twv_dpp.hpp)
        "v_readlane_b32 s84, %[x], 0\n"
        "v_readlane_b32 s85, %[x], 1\n"
        "v_readlane_b32 s86, %[x], 2\n"
        "v_readlane_b32 s87, %[x], 3\n"
        "v_readlane_b32 s88, %[x], 4\n"
        "v_readlane_b32 s89, %[x], 5\n"
        "v_pk_fma_f32 %[a], %[p0], s[84:85], %[i]\n"
        "v_readlane_b32 s90, %[x], 6\n"
        "v_readlane_b32 s91, %[x], 7\n"
        "v_pk_mul_f32 %[b], %[p1], s[86:87]\n"
        "v_readlane_b32 s92, %[x], 8\n"
        "v_readlane_b32 s93, %[x], 9\n"
        "v_pk_fma_f32 %[a], %[p2], s[88:89], %[a]\n"
        "v_readlane_b32 s94, %[x], 10\n"
        "v_readlane_b32 s95, %[x], 11\n"
        "v_pk_fma_f32 %[b], %[p3], s[90:91], %[b]\n"
        "v_readlane_b32 s96, %[x], 12\n"
        "v_readlane_b32 s97, %[x], 13\n"
        "v_pk_fma_f32 %[a], %[p4], s[92:93], %[a]\n"
        "v_readlane_b32 s98, %[x], 14\n"
        "v_readlane_b32 s99, %[x], 15\n"
        "v_pk_fma_f32 %[b], %[p5], s[94:95], %[b]\n"
        "v_readlane_b32 s84, %[x], 16\n"
        "v_readlane_b32 s85, %[x], 17\n"
        "v_pk_fma_f32 %[a], %[p6], s[96:97], %[a]\n"
        "v_readlane_b32 s86, %[x], 18\n"
        "v_readlane_b32 s87, %[x], 19\n"
        "v_pk_fma_f32 %[b], %[p7], s[98:99], %[b]\n"
        "v_readlane_b32 s88, %[x], 20\n"
        "v_readlane_b32 s89, %[x], 21\n"
        "v_pk_fma_f32 %[a], %[p8], s[84:85], %[a]\n"
        "v_readlane_b32 s90, %[x], 22\n"
        "v_readlane_b32 s91, %[x], 23\n"
        "v_pk_fma_f32 %[b], %[p9], s[86:87], %[b]\n"
        "v_readlane_b32 s92, %[x], 24\n"
        "v_readlane_b32 s93, %[x], 25\n"
        "v_pk_fma_f32 %[a], %[p10], s[88:89], %[a]\n"
        "v_readlane_b32 s94, %[x], 26\n"
        "v_readlane_b32 s95, %[x], 27\n"
        "v_pk_fma_f32 %[b], %[p11], s[90:91], %[b]\n"
        "v_readlane_b32 s96, %[x], 28\n"
        "v_readlane_b32 s97, %[x], 29\n"
        "v_pk_fma_f32 %[a], %[p12], s[92:93], %[a]\n"
        "v_readlane_b32 s98, %[x], 30\n"
        "v_readlane_b32 s99, %[x], 31\n"
        "v_pk_fma_f32 %[b], %[p13], s[94:95], %[b]\n"
        "v_pk_fma_f32 %[a], %[p14], s[96:97], %[a]\n"
        "v_pk_fma_f32 %[b], %[p15], s[98:99], %[b]"
        : [a] "=&v"(a), [b] "=&v"(b)
        : [x] "v"(xv), [i] "v"(f32x2p{init, -0.0f}),
          [p0] "v"(f32x2p{tw[0], tw[1]}),
          [p1] "v"(f32x2p{tw[2], tw[3]}),
          [p2] "v"(f32x2p{tw[4], tw[5]}),
          [p3] "v"(f32x2p{tw[6], tw[7]}),
          [p4] "v"(f32x2p{tw[8], tw[9]}),
          [p5] "v"(f32x2p{tw[10], tw[11]}),
          [p6] "v"(f32x2p{tw[12], tw[13]}),
          [p7] "v"(f32x2p{tw[14], tw[15]}),
          [p8] "v"(f32x2p{tw[16], tw[17]}),
          [p9] "v"(f32x2p{tw[18], tw[19]}),
          [p10] "v"(f32x2p{tw[20], tw[21]}),
          [p11] "v"(f32x2p{tw[22], tw[23]}),
          [p12] "v"(f32x2p{tw[24], tw[25]}),
          [p13] "v"(f32x2p{tw[26], tw[27]}),
          [p14] "v"(f32x2p{tw[28], tw[29]}),
          [p15] "v"(f32x2p{tw[30], tw[31]})
        : "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99");
    return (a[0] + a[1]) + (b[0] + b[1]);
}

__device__ __forceinline__ float dot_readlane_pipe(const Tile& t, float xv) { return dot_readlane_pipe_a(t.w, xv); }
__device__ __forceinline__ float dot_readlane_pipe_init(const Tile& t, float xv, float init) { return dot_readlane_pipe_init_a(t.w, xv, init); }

// the same with the operand vector in lanes 32..63 of `xv` (helper workgroups: second chunk of a wave)
__device__ __forceinline__ float dot_readlane_pipe32(const Tile& t, float xv)
{
    f32x2p a, b;
    asm volatile(
        "s_nop 1\n"
        "\t.p2align 3\n"
        "v_readlane_b32 s84, %[x], 32\n"
        "v_readlane_b32 s85, %[x], 33\n"
        "v_readlane_b32 s86, %[x], 34\n"
        "v_readlane_b32 s87, %[x], 35\n"
        "v_readlane_b32 s88, %[x], 36\n"
        "v_readlane_b32 s89, %[x], 37\n"
        "v_pk_fma_f32 %[a], %[p0], s[84:85], 0 op_sel_hi:[1,1,0]\n"
        "v_readlane_b32 s90, %[x], 38\n"
        "v_readlane_b32 s91, %[x], 39\n"
        "v_pk_fma_f32 %[b], %[p1], s[86:87], 0 op_sel_hi:[1,1,0]\n"
        "v_readlane_b32 s92, %[x], 40\n"
        "v_readlane_b32 s93, %[x], 41\n"
        "v_pk_fma_f32 %[a], %[p2], s[88:89], %[a]\n"
        "v_readlane_b32 s94, %[x], 42\n"
        "v_readlane_b32 s95, %[x], 43\n"
        "v_pk_fma_f32 %[b], %[p3], s[90:91], %[b]\n"
        "v_readlane_b32 s96, %[x], 44\n"
        "v_readlane_b32 s97, %[x], 45\n"
        "v_pk_fma_f32 %[a], %[p4], s[92:93], %[a]\n"
        "v_readlane_b32 s98, %[x], 46\n"
        "v_readlane_b32 s99, %[x], 47\n"
        "v_pk_fma_f32 %[b], %[p5], s[94:95], %[b]\n"
        "v_readlane_b32 s84, %[x], 48\n"
        "v_readlane_b32 s85, %[x], 49\n"
        "v_pk_fma_f32 %[a], %[p6], s[96:97], %[a]\n"
        "v_readlane_b32 s86, %[x], 50\n"
        "v_readlane_b32 s87, %[x], 51\n"
        "v_pk_fma_f32 %[b], %[p7], s[98:99], %[b]\n"
        "v_readlane_b32 s88, %[x], 52\n"
        "v_readlane_b32 s89, %[x], 53\n"
        "v_pk_fma_f32 %[a], %[p8], s[84:85], %[a]\n"
        "v_readlane_b32 s90, %[x], 54\n"
        "v_readlane_b32 s91, %[x], 55\n"
        "v_pk_fma_f32 %[b], %[p9], s[86:87], %[b]\n"
        "v_readlane_b32 s92, %[x], 56\n"
        "v_readlane_b32 s93, %[x], 57\n"
        "v_pk_fma_f32 %[a], %[p10], s[88:89], %[a]\n"
        "v_readlane_b32 s94, %[x], 58\n"
        "v_readlane_b32 s95, %[x], 59\n"
        "v_pk_fma_f32 %[b], %[p11], s[90:91], %[b]\n"
        "v_readlane_b32 s96, %[x], 60\n"
        "v_readlane_b32 s97, %[x], 61\n"
        "v_pk_fma_f32 %[a], %[p12], s[92:93], %[a]\n"
        "v_readlane_b32 s98, %[x], 62\n"
        "v_readlane_b32 s99, %[x], 63\n"
        "v_pk_fma_f32 %[b], %[p13], s[94:95], %[b]\n"
        "v_pk_fma_f32 %[a], %[p14], s[96:97], %[a]\n"
        "v_pk_fma_f32 %[b], %[p15], s[98:99], %[b]"
        : [a] "=&v"(a), [b] "=&v"(b)
        : [x] "v"(xv),
          [p0] "v"(f32x2p{t.w[0], t.w[1]}),
          [p1] "v"(f32x2p{t.w[2], t.w[3]}),
          [p2] "v"(f32x2p{t.w[4], t.w[5]}),
          [p3] "v"(f32x2p{t.w[6], t.w[7]}),
          [p4] "v"(f32x2p{t.w[8], t.w[9]}),
          [p5] "v"(f32x2p{t.w[10], t.w[11]}),
          [p6] "v"(f32x2p{t.w[12], t.w[13]}),
          [p7] "v"(f32x2p{t.w[14], t.w[15]}),
          [p8] "v"(f32x2p{t.w[16], t.w[17]}),
          [p9] "v"(f32x2p{t.w[18], t.w[19]}),
          [p10] "v"(f32x2p{t.w[20], t.w[21]}),
          [p11] "v"(f32x2p{t.w[22], t.w[23]}),
          [p12] "v"(f32x2p{t.w[24], t.w[25]}),
          [p13] "v"(f32x2p{t.w[26], t.w[27]}),
          [p14] "v"(f32x2p{t.w[28], t.w[29]}),
          [p15] "v"(f32x2p{t.w[30], t.w[31]})
        : "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99");
    return (a[0] + a[1]) + (b[0] + b[1]);
}

// one chunk of AC-1, operand vector already in (uniform) registers
__device__ __forceinline__ float dot_regs(const Tile& t, const float (&x)[32])
{
    f32x2p s01 = {0.0f, 0.0f}, s23 = {0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < 32; c += 4) {
        s01 = pk_fma(f32x2p{t.w[c + 0], t.w[c + 1]}, f32x2p{x[c + 0], x[c + 1]}, s01);
        s23 = pk_fma(f32x2p{t.w[c + 2], t.w[c + 3]}, f32x2p{x[c + 2], x[c + 3]}, s23);
    }
    return (s01[0] + s01[1]) + (s23[0] + s23[1]);
}


// ---- LDS of the generation kernel: one file-scope symbol + integer offsets keeps every access a ds_* instruction
extern __shared__ __attribute__((aligned(16))) float lds[];

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
using rsrc_t = __amdgpu_buffer_rsrc_t;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
// explicit LDS (address space 3) views of lds[]: a C++ cast to a generic pointer would turn the access into a FLAT
// instruction (vector-memory path, microseconds under load) and hide the 16-byte alignment needed for ds_read_b128.
#define LDSI(off) (((__attribute__((address_space(3))) int*)lds)[(off)])
#define LDSVI(off) (((__attribute__((address_space(3))) volatile int*)lds)[(off)])
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDS4(off4) (((__attribute__((address_space(3))) f32x4*)lds)[(off4)])   /* offset in float4 units */

// weight tile through a buffer descriptor: ONE per-lane offset VGPR (lane*16), tile position in an SGPR
__device__ __forceinline__ void load_tile_b(Tile& t, rsrc_t r, int voff16, int soff_bytes)
{
    const int so = __builtin_amdgcn_readfirstlane(soff_bytes);
#pragma unroll
    for (int kq = 0; kq < 8; ++kq) {
        const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(r, voff16 + (kq & 3) * 1024, so + (kq >> 2) * 4096, 0);
        t.w[4 * kq + 0] = __uint_as_float(q.x); t.w[4 * kq + 1] = __uint_as_float(q.y);
        t.w[4 * kq + 2] = __uint_as_float(q.z); t.w[4 * kq + 3] = __uint_as_float(q.w);
    }
}
__device__ __forceinline__ float load_f32_b(rsrc_t r, int voff4, int soff_bytes)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff4, __builtin_amdgcn_readfirstlane(soff_bytes), 0));
}
// tile staged in LDS at float offset fo (same [kq][lane][4] image as in HBM): conflict-free ds_read_b128
__device__ __forceinline__ void lds_tile(Tile& t, int fo, int lane)
{
#pragma unroll
    for (int kq = 0; kq < 8; ++kq) {
        const f32x4 q = LDS4((fo >> 2) + kq * 64 + lane);
        t.w[4 * kq + 0] = q.x; t.w[4 * kq + 1] = q.y; t.w[4 * kq + 2] = q.z; t.w[4 * kq + 3] = q.w;
    }
}
// half tile [kq][32][4]: lanes l and l+32 read the same 16 bytes (broadcast)
__device__ __forceinline__ void lds_half_tile(Tile& t, int fo, int lane)
{
#pragma unroll
    for (int kq = 0; kq < 8; ++kq) {
        const f32x4 q = LDS4((fo >> 2) + kq * 32 + (lane & 31));
        t.w[4 * kq + 0] = q.x; t.w[4 * kq + 1] = q.y; t.w[4 * kq + 2] = q.z; t.w[4 * kq + 3] = q.w;
    }
}

// one chunk of AC-1, operand vector in LDS at float offset `xo` (same address in every lane: broadcast reads)
__device__ __forceinline__ float dot_ldso(const Tile& t, int xo)
{
    f32x2p s01 = {0.0f, 0.0f}, s23 = {0.0f, 0.0f};
#pragma unroll
    for (int kq = 0; kq < 8; ++kq) {
        const f32x4 q = LDS4((xo >> 2) + kq);
        s01 = pk_fma(f32x2p{t.w[4 * kq + 0], t.w[4 * kq + 1]}, f32x2p{q.x, q.y}, s01);
        s23 = pk_fma(f32x2p{t.w[4 * kq + 2], t.w[4 * kq + 3]}, f32x2p{q.z, q.w}, s23);
    }
    return (s01[0] + s01[1]) + (s23[0] + s23[1]);
}


// ---- host-side helpers implemented in twv_wavenet.hip -------------------------------------------------------------
// dst tiles [g][jblk][chunk][kq][lane][4]; element = src[base(half) + g*sg + k*rowlen + jj] (0 outside K x ncols)
struct PackTiles {
    long long dst_off, dst_gstride, baseA, baseB, src_gstride;
    int ngroups, njblk, nchunk, K, rowlen, ncols, halves, lanes;
};
struct PackVec {
    long long dst_off, dst_gstride, baseA, baseB, src_gstride;
    int ngroups, n, ncols, halves;
};
void twv_launch_pack_tiles(float* dst, const float* src, const PackTiles& p, hipStream_t st);
void twv_launch_copy(float* dst, const float* src, long long n, hipStream_t st);
int twv_fail(int code, const std::string& msg);
