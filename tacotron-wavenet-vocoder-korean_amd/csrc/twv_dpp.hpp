// twv_dpp.hpp -- the AC-1 chunk and the residual-layer body in the ROW-BROADCAST lane layout (gfx950).
//
// A 32-term chunk of the arithmetic contract (four interleaved fma chains j = k mod 4, each from +0 in ascending k,
// value (s0+s1)+(s2+s3)) needs every output lane to see operand x[k] at step k.  v_readlane + v_pk_fma_f32 spends 48 issue slots on
// it; here the operand vector sits in the lanes of one row (16 lanes) and `v_fmac_f32_dpp ... row_newbcast:n` (DPP control 0x150+n:
// lane n of each row to the whole row) feeds it to the fma directly: 32 instructions, no scalar registers, same fmas in the same order.
//
// Lane layouts (lane = 16*row + n):
//   X  (layer input / residual, 32 values):  rows 0,1: x[n]        rows 2,3: x[16+n]
//   XA = all rows x[n], XB = all rows x[16+n]:  ONE v_permlane32_swap of (X, X)
//   conv output of lane (model.py:68-69 conv_filter | conv_gate):  row 0: f[a(n)]  row 1: f[b(n)]  row 2: g[a(n)]  row 3: g[b(n)]
//        a(n) = 4*(n/2) + n%2  (k = 0,1,4,5,...: chains 0 and 1),   b(n) = a(n) + 2  (chains 2 and 3)
//   Z  (gated output, model.py:86) after v_permlane32_swap + multiply:  rows 0,2: z[a(n)]   rows 1,3: z[b(n)]
//   dense 1x1 (model.py:89), 32 outputs on 64 lanes: lane (row, n) owns output n + 16*(row/2) and HALF of its chunk -- even rows the
//        chains 0,1 (k = a(i)), odd rows the chains 2,3 (k = b(i)) -- 16 fmas, then (s0+s1) | (s2+s3) meet through ONE
//        v_permlane16_swap: exactly AC-1's (s0+s1)+(s2+s3), and the result lands in the X layout again.
#pragma once
#include <hip/hip_runtime.h>
#include "twv_math.hpp"

// 8-byte instructions (v_fmac_f32_dpp is VOP2 + a DPP dword) issue measurably slower when they sit at an address of 4 mod 8: the same
// chain loop ran at 10.42 or 10.60 us per generation step depending on whether unrelated code had shifted it by an odd number of
// dwords (scan over 16 paddings: every even one fast, every odd one slow; profiles/r02_chain_alignment_scan.txt).  Every block of
// DPP fmacs therefore starts 8-byte aligned (the assembler pads with one s_nop where needed).
#define TWV_ALIGN8 "\t.p2align 3\n"

namespace twv {

__host__ __device__ inline int dpp_a(int n) { return 4 * (n >> 1) + (n & 1); }
// conv output (0..31 filter, 32..63 gate) owned by a lane
__host__ __device__ inline int dpp_conv_out(int lane) { const int r = lane >> 4, n = lane & 15; return ((r >> 1) ? 32 : 0) + dpp_a(n) + ((r & 1) ? 2 : 0); }
// dense output / residual element owned by a lane
__host__ __device__ inline int dpp_dense_out(int lane) { return (lane & 15) + 16 * (lane >> 5); }
// reduction index of a lane's i-th dense fma
__host__ __device__ inline int dpp_dense_k(int lane, int i) { return dpp_a(i) + (((lane >> 4) & 1) ? 2 : 0); }
// z element held by a lane of the Z layout
__host__ __device__ inline int dpp_z_index(int lane) { return dpp_a(lane & 15) + (((lane >> 4) & 1) ? 2 : 0); }

// 32-term AC-1 chunk: acc_j += w[k] * x[k]  (k = 0..15 from XA, 16..31 from XB).  init: the start value of chain 0 (AC-1b: the
// contraction's last chunk on the generation chain starts from the addend; +0 everywhere else)
__device__ __forceinline__ float dot32_dpp(const float (&w)[32], float xa, float xb, float init = 0.0f)
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

// two independent 32-term chunks (different tiles, different operands) interleaved instruction by instruction: eight fma chains in
// flight instead of four, so neither dot waits on its own chains (conv1 workgroups: 2 chunk tiles per wave)
__device__ __forceinline__ void dot32_dpp_x2(const float (&wa)[32], float xa0, float xb0, const float (&wb)[32], float xa1, float xb1,
                                             float& ra, float& rb)
{
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, c3 = 0.0f, d0 = 0.0f, d1 = 0.0f, d2 = 0.0f, d3 = 0.0f;
    asm volatile(
        "s_nop 1\n" TWV_ALIGN8
        "v_fmac_f32_dpp %0, %8, %10 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %4, %9, %18 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %8, %11 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %5, %9, %19 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %2, %8, %12 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %6, %9, %20 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %3, %8, %13 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %7, %9, %21 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %8, %14 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %4, %9, %22 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %8, %15 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %5, %9, %23 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %2, %8, %16 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %6, %9, %24 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %3, %8, %17 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %7, %9, %25 row_newbcast:7 row_mask:0xf bank_mask:0xf"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)
        : "v"(xa0), "v"(xa1), "v"(wa[0]), "v"(wa[1]), "v"(wa[2]), "v"(wa[3]), "v"(wa[4]), "v"(wa[5]), "v"(wa[6]), "v"(wa[7]), "v"(wb[0]), "v"(wb[1]), "v"(wb[2]), "v"(wb[3]), "v"(wb[4]), "v"(wb[5]), "v"(wb[6]), "v"(wb[7]));
    asm volatile(
        "s_nop 1\n" TWV_ALIGN8
        "v_fmac_f32_dpp %0, %8, %10 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %4, %9, %18 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %8, %11 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %5, %9, %19 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %2, %8, %12 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %6, %9, %20 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %3, %8, %13 row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %7, %9, %21 row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %8, %14 row_newbcast:12 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %4, %9, %22 row_newbcast:12 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %8, %15 row_newbcast:13 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %5, %9, %23 row_newbcast:13 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %2, %8, %16 row_newbcast:14 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %6, %9, %24 row_newbcast:14 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %3, %8, %17 row_newbcast:15 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %7, %9, %25 row_newbcast:15 row_mask:0xf bank_mask:0xf"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)
        : "v"(xa0), "v"(xa1), "v"(wa[8]), "v"(wa[9]), "v"(wa[10]), "v"(wa[11]), "v"(wa[12]), "v"(wa[13]), "v"(wa[14]), "v"(wa[15]), "v"(wb[8]), "v"(wb[9]), "v"(wb[10]), "v"(wb[11]), "v"(wb[12]), "v"(wb[13]), "v"(wb[14]), "v"(wb[15]));
    asm volatile(
        "s_nop 1\n" TWV_ALIGN8
        "v_fmac_f32_dpp %0, %8, %10 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %4, %9, %18 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %8, %11 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %5, %9, %19 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %2, %8, %12 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %6, %9, %20 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %3, %8, %13 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %7, %9, %21 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %8, %14 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %4, %9, %22 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %8, %15 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %5, %9, %23 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %2, %8, %16 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %6, %9, %24 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %3, %8, %17 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %7, %9, %25 row_newbcast:7 row_mask:0xf bank_mask:0xf"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)
        : "v"(xb0), "v"(xb1), "v"(wa[16]), "v"(wa[17]), "v"(wa[18]), "v"(wa[19]), "v"(wa[20]), "v"(wa[21]), "v"(wa[22]), "v"(wa[23]), "v"(wb[16]), "v"(wb[17]), "v"(wb[18]), "v"(wb[19]), "v"(wb[20]), "v"(wb[21]), "v"(wb[22]), "v"(wb[23]));
    asm volatile(
        "s_nop 1\n" TWV_ALIGN8
        "v_fmac_f32_dpp %0, %8, %10 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %4, %9, %18 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %8, %11 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %5, %9, %19 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %2, %8, %12 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %6, %9, %20 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %3, %8, %13 row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %7, %9, %21 row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %0, %8, %14 row_newbcast:12 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %4, %9, %22 row_newbcast:12 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %1, %8, %15 row_newbcast:13 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %5, %9, %23 row_newbcast:13 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %2, %8, %16 row_newbcast:14 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %6, %9, %24 row_newbcast:14 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %3, %8, %17 row_newbcast:15 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f32_dpp %7, %9, %25 row_newbcast:15 row_mask:0xf bank_mask:0xf"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)
        : "v"(xb0), "v"(xb1), "v"(wa[24]), "v"(wa[25]), "v"(wa[26]), "v"(wa[27]), "v"(wa[28]), "v"(wa[29]), "v"(wa[30]), "v"(wa[31]), "v"(wb[24]), "v"(wb[25]), "v"(wb[26]), "v"(wb[27]), "v"(wb[28]), "v"(wb[29]), "v"(wb[30]), "v"(wb[31]));
    ra = (c0 + c1) + (c2 + c3);
    rb = (d0 + d1) + (d2 + d3);
}

// THREE independent 32-term chunks interleaved (twelve fma chains in flight): the three chunks of an 80-channel lc projection
// (many-streams kernel's lc role).  Four terms of each chunk per asm statement (an asm statement takes at most 30 operands):
// k = 4b .. 4b+3 goes to chains 0..3 in this order in every block, so each chain still sees its terms in ascending k.
#define TWV_FMAC12_DPP(c, d, e, xa, xb, xc, wa, wb, wc, base, n0, n1, n2, n3)                                     \
    asm volatile(                                                                                                  \
        "s_nop 1\n" TWV_ALIGN8                                                                                     \
        "v_fmac_f32_dpp %0, %12, %15 row_newbcast:" #n0 " row_mask:0xf bank_mask:0xf\n"                            \
        "v_fmac_f32_dpp %4, %13, %19 row_newbcast:" #n0 " row_mask:0xf bank_mask:0xf\n"                            \
        "v_fmac_f32_dpp %8, %14, %23 row_newbcast:" #n0 " row_mask:0xf bank_mask:0xf\n"                            \
        "v_fmac_f32_dpp %1, %12, %16 row_newbcast:" #n1 " row_mask:0xf bank_mask:0xf\n"                            \
        "v_fmac_f32_dpp %5, %13, %20 row_newbcast:" #n1 " row_mask:0xf bank_mask:0xf\n"                            \
        "v_fmac_f32_dpp %9, %14, %24 row_newbcast:" #n1 " row_mask:0xf bank_mask:0xf\n"                            \
        "v_fmac_f32_dpp %2, %12, %17 row_newbcast:" #n2 " row_mask:0xf bank_mask:0xf\n"                            \
        "v_fmac_f32_dpp %6, %13, %21 row_newbcast:" #n2 " row_mask:0xf bank_mask:0xf\n"                            \
        "v_fmac_f32_dpp %10, %14, %25 row_newbcast:" #n2 " row_mask:0xf bank_mask:0xf\n"                           \
        "v_fmac_f32_dpp %3, %12, %18 row_newbcast:" #n3 " row_mask:0xf bank_mask:0xf\n"                            \
        "v_fmac_f32_dpp %7, %13, %22 row_newbcast:" #n3 " row_mask:0xf bank_mask:0xf\n"                            \
        "v_fmac_f32_dpp %11, %14, %26 row_newbcast:" #n3 " row_mask:0xf bank_mask:0xf"                             \
        : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]),          \
          "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3])                                                           \
        : "v"(xa), "v"(xb), "v"(xc), "v"(wa[base]), "v"(wa[base + 1]), "v"(wa[base + 2]), "v"(wa[base + 3]),         \
          "v"(wb[base]), "v"(wb[base + 1]), "v"(wb[base + 2]), "v"(wb[base + 3]),                                   \
          "v"(wc[base]), "v"(wc[base + 1]), "v"(wc[base + 2]), "v"(wc[base + 3]))
__device__ __forceinline__ void dot32_dpp_x3(const float (&wa)[32], float xa0, float xb0, const float (&wb)[32], float xa1, float xb1,
                                             const float (&wc)[32], float xa2, float xb2, float& ra, float& rb, float& rc)
{
    float c[4] = {0.0f, 0.0f, 0.0f, 0.0f}, d[4] = {0.0f, 0.0f, 0.0f, 0.0f}, e[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    TWV_FMAC12_DPP(c, d, e, xa0, xa1, xa2, wa, wb, wc, 0, 0, 1, 2, 3);
    TWV_FMAC12_DPP(c, d, e, xa0, xa1, xa2, wa, wb, wc, 4, 4, 5, 6, 7);
    TWV_FMAC12_DPP(c, d, e, xa0, xa1, xa2, wa, wb, wc, 8, 8, 9, 10, 11);
    TWV_FMAC12_DPP(c, d, e, xa0, xa1, xa2, wa, wb, wc, 12, 12, 13, 14, 15);
    TWV_FMAC12_DPP(c, d, e, xb0, xb1, xb2, wa, wb, wc, 16, 0, 1, 2, 3);
    TWV_FMAC12_DPP(c, d, e, xb0, xb1, xb2, wa, wb, wc, 20, 4, 5, 6, 7);
    TWV_FMAC12_DPP(c, d, e, xb0, xb1, xb2, wa, wb, wc, 24, 8, 9, 10, 11);
    TWV_FMAC12_DPP(c, d, e, xb0, xb1, xb2, wa, wb, wc, 28, 12, 13, 14, 15);
    ra = (c[0] + c[1]) + (c[2] + c[3]);
    rb = (d[0] + d[1]) + (d[2] + d[3]);
    rc = (e[0] + e[1]) + (e[2] + e[3]);
}

// a lane's half of a 32-term chunk: two chains, 16 terms, operand vector in the Z layout.  init: start value of the lane's first
// chain (even rows hold the chunk's chains 0 and 1: their first chain is chain 0 of AC-1b)
__device__ __forceinline__ float dot16_dpp(const float (&w)[16], float z, float init = 0.0f)
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

// ---- the two contractions ON the generation chain (AC-1b): chain 0 starts from the addend, the other chains from their FIRST PRODUCT
// (v_mul_f32_dpp -- no zero-initialised accumulators: four v_mov fewer per layer; fma(w, x, -0) == w * x for every w and x, so the C
// restatement of the CPU checker starts those chains from -0).  profiles/r05_chain_contract_ubench.txt, contract C7.
__device__ __forceinline__ float dot32_dpp_chain(const float (&w)[32], float xa, float xb, float init)
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
__device__ __forceinline__ float dot16_dpp_chain(const float (&w)[16], float z, float init)
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
// model.py:66-101 for one step of one stream, in two halves so that the caller can publish z between them.
// front: X (X layout) -> z (Z layout).  AC-1b (round 5): `addend` = ((tap-0 chunk + bias) + gc) + lc -- the reference's statement
// order ((conv + bias) + gc) + lc (model.py:68-83) with conv = chunk(tap 0) + chunk(tap 1) -- is computed OFF the chain (service
// workgroup / loader waves: nothing in it depends on x[t]) and is the start value of chain 0 of the tap-1 chunk: no add follows the
// dot product on the sample-to-sample path (profiles/r05_chain_contract_ubench.txt: contracts C2 / C5).
__device__ __forceinline__ float layer_front_dpp(const float (&wc)[32], const ActCoef& coef, float X, float addend)
{
    const auto xs = __builtin_amdgcn_permlane32_swap(__float_as_uint(X), __float_as_uint(X), false, false);
    const float v = dot32_dpp_chain(wc, __uint_as_float(xs[0]), __uint_as_float(xs[1]), addend);
    const float act = act_eval_pk_med3(coef, v);                             // model.py:86: lanes 0-31 tanh, 32-63 logistic
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(act), __float_as_uint(act), false, false);
    return __uint_as_float(sw[0]) * __uint_as_float(sw[1]);
}
// back: model.py:89 dense 1x1 of z, model.py:98-101 residual; X in / out in the X layout.  bd_init: the dense bias on the lanes that
// hold chain 0 of the chunk (even rows), -0 on the odd rows, whose first chain is chain 2: it starts from its first product
// (dense_bias_init): the bias is the start value of chain 0 (AC-1b).
__device__ __forceinline__ float dense_bias_init(int lane, float bd) { return ((lane >> 4) & 1) ? -0.0f : bd; }
__device__ __forceinline__ void layer_back_dpp(const float (&wd)[16], float bd_init, float z, float& X)
{
    const float s = dot16_dpp_chain(wd, z, bd_init);
    const auto ds = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    X = X + (__uint_as_float(ds[0]) + __uint_as_float(ds[1]));
}
// one layer's chain operands, register-resident for the whole launch (micro-benchmarks)
struct LayerRegs {
    float wc[32];          // tap-1 conv kernel column of this lane's conv output
    float wd[16];          // this lane's half of the dense kernel column
    float bd_init;         // dense_bias_init
};
__device__ __forceinline__ float layer_body_dpp(const LayerRegs& W, const ActCoef& coef, float& X, float addend)
{
    const float z = layer_front_dpp(W.wc, coef, X, addend);
    layer_back_dpp(W.wd, W.bd_init, z, X);
    return z;
}

// The causal layer's chunk (model.py:41-46) with the newest sample split off: the queue holds the last 32 inputs, k = 31 the newest.
// Chain j of AC-1 takes k = j, j+4, ..; k = 31 is the LAST term of chain 3, so the other 31 terms are summed before the sample
// exists; what is left on the sample-to-sample path is one fma and the three adds.  ha / hb: all rows hist[n] / hist[16+n]
// (lane 15 of hb is not read).
__device__ __forceinline__ void causal_partial_dpp(const float (&w)[32], float ha, float hb, float (&c)[4])
{
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, c3 = 0.0f;
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
        : "v"(ha), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "v"(w[9]), "v"(w[10]), "v"(w[11]), "v"(w[12]), "v"(w[13]), "v"(w[14]), "v"(w[15]));
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
        "v_fmac_f32_dpp %2, %4, %19 row_newbcast:14 row_mask:0xf bank_mask:0xf"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
        : "v"(hb), "v"(w[16]), "v"(w[17]), "v"(w[18]), "v"(w[19]), "v"(w[20]), "v"(w[21]), "v"(w[22]), "v"(w[23]), "v"(w[24]), "v"(w[25]), "v"(w[26]), "v"(w[27]), "v"(w[28]), "v"(w[29]), "v"(w[30]));
    c[0] = c0; c[1] = c1; c[2] = c2; c[3] = c3;
}

}  // namespace twv
