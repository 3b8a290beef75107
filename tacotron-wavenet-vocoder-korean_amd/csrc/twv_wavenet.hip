// twv_wavenet.hip -- MI355X (gfx950) WaveNet-vocoder generation path + its C-ABI (include/twv_amd.h).
//
// Replaces, for hccho2/Tacotron-Wavenet-Vocoder-Korean (citations into /root/reference):
//   wavenet/model.py:41-167,215-245  (incremental network)        -> wn_generate_kernel (persistent, one launch per call)
//   wavenet/mixture.py:84-114        (MoL sampler)                -> fused into wn_generate_kernel
//   generate.py:199-233              (per-sample host loop)       -> the kernel's step loop
//   wavenet/model.py:102-111         (create_upsample)            -> wn_upsample_stage_kernel
//   wavenet/model.py:71-83,181-212   (gc/lc 1x1 projections)      -> hoisted: wn_gc_kernel / wn_lc_kernel
//   wavenet/ops.py:22-47             (mu-law codec)               -> wn_mulaw_*_kernel
//
// Design: one workgroup per utterance (stream).  Wave 0 is the CHAIN wave: it walks the dilated
// residual stack layer by layer (the strictly serial part), every lane owning one filter/gate output, operands
// broadcast with v_readlane, gated tanh*sigmoid evaluated as one instruction stream with per-half-wave
// coefficients.  Waves 1..W are WORKERS: they stream the wide 1x1 convolutions (skip, post) as 64x32 weight tiles
// straight into registers (coalesced 16 B/lane), following the chain through an LDS sequence flag.  The
// discretised-mixture-of-logistics sampler runs on worker 0 and feeds the next step through LDS.
// All arithmetic follows the arithmetic contract: results are bit-identical to the
// CPU checker for any launch geometry.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include "../../include/twv_amd.h"
#include "twv_layout.hpp"
#include "twv_math.hpp"

using namespace twv;

// =====================================================================================================
//  small device helpers
// =====================================================================================================
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

// one chunk of AC-1: fma chain from +0 over 32 terms, operand vector distributed over lanes 0..31 of `xv`
__device__ __forceinline__ float dot_readlane(const Tile& t, float xv)
{
    float acc = 0.0f;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        const float s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), c));
        acc = fma_(t.w[c], s, acc);
    }
    return acc;
}

// one chunk of AC-1, operand vector already in (uniform) registers
__device__ __forceinline__ float dot_regs(const Tile& t, const float (&x)[32])
{
    float acc = 0.0f;
#pragma unroll
    for (int c = 0; c < 32; ++c) acc = fma_(t.w[c], x[c], acc);
    return acc;
}

// one chunk of AC-1, operand vector in LDS (same address in every lane: broadcast reads)
__device__ __forceinline__ float dot_lds(const Tile& t, const float* xs)
{
    float acc = 0.0f;
    const float4* p = reinterpret_cast<const float4*>(xs);
#pragma unroll
    for (int kq = 0; kq < 8; ++kq) {
        const float4 q = p[kq];
        acc = fma_(t.w[4 * kq + 0], q.x, acc);
        acc = fma_(t.w[4 * kq + 1], q.y, acc);
        acc = fma_(t.w[4 * kq + 2], q.z, acc);
        acc = fma_(t.w[4 * kq + 3], q.w, acc);
    }
    return acc;
}

__device__ __forceinline__ void read_lds32(float (&x)[32], const float* xs)
{
    const float4* p = reinterpret_cast<const float4*>(xs);
#pragma unroll
    for (int kq = 0; kq < 8; ++kq) {
        const float4 q = p[kq];
        x[4 * kq + 0] = q.x; x[4 * kq + 1] = q.y; x[4 * kq + 2] = q.z; x[4 * kq + 3] = q.w;
    }
}

// bounded wait on an LDS sequence word written by another wave of the same workgroup.
// Never hangs: after ~2^22 polls it raises the workgroup's abort word and returns.
__device__ __forceinline__ bool wait_ge(volatile int* flag, int target, volatile int* abortf, int code)
{
    if (*flag >= target) return true;
    for (int it = 0; it < (1 << 22); ++it) {
        __builtin_amdgcn_s_sleep(1);
        if (*flag >= target) return true;
        if (*abortf) return false;
    }
    *abortf = code;
    return false;
}

// =====================================================================================================
//  pack: canonical checkpoint blob (TF layouts) -> streaming layout        (generate.py:157-161 Saver.restore)
// =====================================================================================================
// dst tiles [g][jblk][chunk][kq][lane][4]; element = src[base(half) + g*sg + k*rowlen + jj] (0 outside K x ncols)
struct PackTiles {
    long long dst_off, dst_gstride, baseA, baseB, src_gstride;
    int ngroups, njblk, nchunk, K, rowlen, ncols, halves;
};
__global__ void wn_pack_tiles_kernel(float* dst, const float* src, PackTiles p)
{
    const long long per_g = (long long)p.njblk * p.nchunk * kTile;
    const long long total = per_g * p.ngroups;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(i / per_g);
        long long r = i - (long long)g * per_g;
        const int jb = (int)(r / ((long long)p.nchunk * kTile));
        r -= (long long)jb * p.nchunk * kTile;
        const int ch = (int)(r / kTile);
        const int e = (int)(r - (long long)ch * kTile);
        const int kq = e >> 8, lane = (e >> 2) & 63, q = e & 3;
        const int k = ch * 32 + kq * 4 + q;
        const int j = jb * 64 + lane;
        float v = 0.0f;
        if (k < p.K) {
            if (p.halves) {
                const long long base = (lane < 32 ? p.baseA : p.baseB) + (long long)g * p.src_gstride;
                v = src[base + (long long)k * p.rowlen + (lane & 31)];
            } else if (j < p.ncols) {
                v = src[p.baseA + (long long)g * p.src_gstride + (long long)k * p.rowlen + j];
            }
        }
        dst[p.dst_off + (long long)g * p.dst_gstride + r + (long long)jb * p.nchunk * kTile] = v;
    }
}
// vectors: dst[g][j] (n entries per group)
struct PackVec {
    long long dst_off, dst_gstride, baseA, baseB, src_gstride;
    int ngroups, n, ncols, halves;
};
__global__ void wn_pack_vec_kernel(float* dst, const float* src, PackVec p)
{
    const long long total = (long long)p.ngroups * p.n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(i / p.n), j = (int)(i % p.n);
        float v = 0.0f;
        if (p.halves) v = src[((j & 63) < 32 ? p.baseA : p.baseB) + (long long)g * p.src_gstride + (j & 31)];
        else if (j < p.ncols) v = src[p.baseA + (long long)g * p.src_gstride + j];
        dst[p.dst_off + (long long)g * p.dst_gstride + j] = v;
    }
}
__global__ void wn_copy_kernel(float* dst, const float* src, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = src[i];
}

// =====================================================================================================
//  create_upsample: one transposed-conv stage        (model.py:102-111)
//  out[b, t*f + a, m] = K[a][0]*in[b,t,m] + K[a][1]*in[b,t,m-1]   as a 2-term AC-1 chain
// =====================================================================================================
__global__ void wn_upsample_stage_kernel(const float* K, const float* in, float* out, int B, long long Tin, int f, int Lc)
{
    const long long total = (long long)B * Tin * f * Lc;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i % Lc);
        const long long ta = i / Lc;              // b*Tin*f + t*f + a
        const int a = (int)(ta % f);
        const long long bt = ta / f;              // b*Tin + t
        const float x0 = in[bt * Lc + m];
        const float x1 = m > 0 ? in[bt * Lc + m - 1] : 0.0f;
        float acc = fma_(K[a * 2 + 0], x0, 0.0f);
        acc = fma_(K[a * 2 + 1], x1, acc);
        out[i] = acc;
    }
}

// =====================================================================================================
//  hoisted conditioning projections        (model.py:71-83, 181-212)
// =====================================================================================================
// gc: GCv[b][l][lane] = cdot(gc_{filter|gate}_l[:, lane], gc_embedding[gc_ids[b]])
__global__ void __launch_bounds__(64) wn_gc_kernel(const float* P, Layout L, const int32_t* gc_ids, float* GCv)
{
    __shared__ __attribute__((aligned(16))) float emb[64];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int id = gc_ids[b];
    emb[lane] = lane < L.G ? P[L.off_gcemb + (long long)id * L.G + lane] : 0.0f;
    __syncthreads();
    for (int l = 0; l < L.NL; ++l) {
        float res = 0.0f;
        for (int c = 0; c < L.NGC; ++c) {
            Tile t;
            load_tile(t, P + L.off_gcw + (long long)l * L.gcw_stride + (long long)c * kTile, lane);
            const int kn = min(32, L.G - c * 32);
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 32; ++k)
                if (k < kn) acc = fma_(t.w[k], emb[c * 32 + k], acc);
            res = c == 0 ? acc : res + acc;
        }
        GCv[((long long)b * L.NL + l) * 64 + lane] = res;
    }
}

// lc: LC[row][l][lane] = cdot(lc_{filter|gate}_l[:, lane], upsampled[row, :]),  rows = B*n_steps.
// grid (ceil(rows/64), ceil(NL/4)), 256 threads: wave w owns layer 4*blockIdx.y + w, 64 rows staged in LDS.
constexpr int kLcRows = 64;
__global__ void __launch_bounds__(256) wn_lc_kernel(const float* P, Layout L, const float* U, float* LC, long long rows)
{
    extern __shared__ __attribute__((aligned(16))) float us[];   // [kLcRows][LP], LP = NLC*32
    const int LP = L.NLC * 32;
    const long long row0 = (long long)blockIdx.x * kLcRows;
    for (int i = threadIdx.x; i < kLcRows * LP; i += 256) {
        const int r = i / LP, k = i - r * LP;
        us[i] = (row0 + r < rows && k < L.L) ? U[(row0 + r) * L.L + k] : 0.0f;
    }
    __syncthreads();
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int l = blockIdx.y * 4 + wv;
    if (l >= L.NL) return;
    Tile t[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (c < L.NLC) load_tile(t[c], P + L.off_lcw + (long long)l * L.lcw_stride + (long long)c * kTile, lane);
    const int nrow = (int)min((long long)kLcRows, rows - row0);
    for (int r = 0; r < nrow; ++r) {
        float res = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < L.NLC) {
                const int kqn = min(8, (L.L - c * 32) >> 2);   // L % 4 == 0 (validated on the host)
                const float4* p = reinterpret_cast<const float4*>(us + r * LP + c * 32);
                float acc = 0.0f;
#pragma unroll
                for (int kq = 0; kq < 8; ++kq) {
                    if (kq < kqn) {
                        const float4 q = p[kq];
                        acc = fma_(t[c].w[4 * kq + 0], q.x, acc);
                        acc = fma_(t[c].w[4 * kq + 1], q.y, acc);
                        acc = fma_(t[c].w[4 * kq + 2], q.z, acc);
                        acc = fma_(t[c].w[4 * kq + 3], q.w, acc);
                    }
                }
                res = c == 0 ? acc : res + acc;
            }
        }
        LC[((row0 + r) * L.NL + l) * 64 + lane] = res;
    }
}

// =====================================================================================================
//  the persistent generation kernel
// =====================================================================================================
struct GenArgs {
    const float* P;            // packed weights
    float* state;              // per-stream delay lines etc.
    const float* cond;         // [B][NL][64] gc  then  [B][T][NL][64] lc
    const void* first_input;   // (B)
    const void* uniforms;      // (B,T,nr_mix+1) f32  |  (B,T) f64
    void* out;                 // (B,T)
    int* status;               // [4]
    float* dbg;                // optional [B][dbg_steps][NL*64 + Opad]
    int dbg_steps;
    int B, T;
    float temperature;
    Layout lay;
};

// ---- LDS of the generation kernel: one file-scope symbol + integer offsets keeps every access a ds_* instruction
extern __shared__ __attribute__((aligned(16))) float lds[];

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
using rsrc_t = __amdgpu_buffer_rsrc_t;

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

// one chunk of AC-1, operand vector in LDS at float offset `xo` (same address in every lane: broadcast reads)
__device__ __forceinline__ float dot_ldso(const Tile& t, int xo)
{
    float acc = 0.0f;
#pragma unroll
    for (int kq = 0; kq < 8; ++kq) {
        const float4 q = *reinterpret_cast<const float4*>(&lds[xo + 4 * kq]);
        acc = fma_(t.w[4 * kq + 0], q.x, acc);
        acc = fma_(t.w[4 * kq + 1], q.y, acc);
        acc = fma_(t.w[4 * kq + 2], q.z, acc);
        acc = fma_(t.w[4 * kq + 3], q.w, acc);
    }
    return acc;
}

// bounded wait on an LDS sequence word (float-offset `fo`) written by another wave of the same workgroup.
// Never hangs: after ~2^22 polls it raises the workgroup's abort word and returns.
__device__ __forceinline__ bool wait_seq(int fo_flag, int target, int fo_abort, int code)
{
    volatile int* flag = reinterpret_cast<volatile int*>(&lds[fo_flag]);
    volatile int* abortf = reinterpret_cast<volatile int*>(&lds[fo_abort]);
    if (*flag >= target) return true;
    for (int it = 0; it < (1 << 22); ++it) {
        __builtin_amdgcn_s_sleep(1);
        if (*flag >= target) return true;
        if (*abortf) return false;
    }
    *abortf = code;
    return false;
}

// per-workgroup context shared by the wave roles (LDS positions are float offsets into lds[])
struct Ctx {
    int o_zbuf, o_h1, o_h2, o_cpart, o_hist, o_ringpos, o_ctrl;   // ctrl: +0 zseq, +1 abort, +2 sample
    int b, lane;
    float* stb;            // this stream's state
    const int* pmeta;      // dil[64] | ring_off[64]
    float* ring;
    const float *GCv, *LCb;
};
#define LDSI(off) (reinterpret_cast<int*>(lds)[(off)])
#define LDSVI(off) (reinterpret_cast<volatile int*>(lds)[(off)])

// =============================== CHAIN WAVE (wave 0) ===============================
// Walks model.py:112-149 for one stream: causal layer, then the dilated residual stack, one layer after the other.
template <bool SCALAR>
__device__ __forceinline__ void chain_main(const GenArgs& a, const Ctx& c, rsrc_t rs, int& hpos, int& prev_valid, int& qprev)
{
    const Layout& L = a.lay;
    const int NL = L.NL, T = a.T, lane = c.lane, b = c.b;
    const int v16 = lane * 16, v4 = lane * 4;
    const bool has_gc = L.G > 0, has_lc = L.L > 0, use_bias = L.use_bias != 0;
    const int lay0 = (int)L.off_layer0 * 4, lstride = (int)L.layer_stride * 4;   // bytes
    const ActCoef coef = act_coef(lane >= 32);
    Tile wt0, wt1, wdd;            // tap-0 / tap-1 conv tiles and dense tile of the NEXT layer to process
    float n_xo = 0.f, n_bfg = 0.f, n_bd = 0.f, n_gc = 0.f, n_lc = 0.f;

    auto prefetch_vec = [&](int l, int tt) {
        const int lb = lay0 + l * lstride;
        n_xo = c.ring[c.pmeta[64 + l] + LDSI(c.o_ringpos + l) * 32 + (lane & 31)];
        if (use_bias) { n_bfg = load_f32_b(rs, v4, lb + LayerOff::BFG * 4); n_bd = load_f32_b(rs, v4, lb + LayerOff::BD * 4); }
        if (has_gc) n_gc = c.GCv[l * 64 + lane];
        if (has_lc) {
            // lc frame used at step tt = frame PUSHED at step tt-1 (model.py:79-80: slice from the FRONT of the queue)
            const float* row = tt == 0 ? (c.stb + L.st_lcprev) : (c.LCb + (long long)(min(tt, T) - 1) * NL * 64);
            n_lc = row[l * 64 + lane];
        }
    };

    __builtin_amdgcn_s_setprio(3);
    load_tile_b(wt0, rs, v16, lay0 + LayerOff::T0 * 4);
    load_tile_b(wt1, rs, v16, lay0 + LayerOff::T1 * 4);
    load_tile_b(wdd, rs, v16, lay0 + LayerOff::WD * 4);
    prefetch_vec(0, 0);

    for (int t = 0; t < T; ++t) {
        float x;
        if (SCALAR) {
            // model.py:122 causal_queue shift+append; model.py:41-46 causal conv (k = ifw, no bias)
            const float s_in = (t == 0) ? reinterpret_cast<const float*>(a.first_input)[b] : lds[c.o_ctrl + 2];
            if (lane == 0) lds[c.o_hist + hpos] = s_in;
            hpos = (hpos + 1 == L.ifw) ? 0 : hpos + 1;   // now the position of the OLDEST sample
            x = 0.0f;
            for (int ca = 0; ca < L.NCA; ++ca) {
                Tile tc;
                load_tile_b(tc, rs, v16, ((int)L.off_causal + ca * kTile) * 4);
                float acc = 0.0f;
#pragma unroll
                for (int k0 = 0; k0 < 32; ++k0) {
                    const int k = ca * 32 + k0;
                    if (k < L.ifw) {
                        int idx = hpos + k;
                        idx = idx >= L.ifw ? idx - L.ifw : idx;
                        acc = fma_(tc.w[k0], lds[c.o_hist + idx], acc);
                    }
                }
                x = ca == 0 ? acc : x + acc;
            }
        } else {
            // one-hot input: the k=2 causal conv over one-hot rows is the sum of two kernel rows
            const int qcur = (t == 0) ? reinterpret_cast<const int*>(a.first_input)[b] : LDSI(c.o_ctrl + 2);
            const float w1 = a.P[L.off_causal + ((long long)L.Q + qcur) * 32 + (lane & 31)];
            if (prev_valid) {
                const float w0 = a.P[L.off_causal + (long long)qprev * 32 + (lane & 31)];
                x = w0 + w1;
            } else {
                x = w1;
            }
            qprev = qcur; prev_valid = 1;
        }

        for (int l = 0; l < NL; ++l) {
            // vectors of this layer were prefetched one layer (or one step) ago
            const float xo = n_xo, bfg = n_bfg, bd = n_bd, gcv = n_gc, lcv = n_lc;
            // model.py:145 dilation queue: slot holds x[t-d]; overwrite it with the layer INPUT x[t]
            const int d = c.pmeta[l];
            const int pos = LDSI(c.o_ringpos + l);
            if (lane < 32) c.ring[c.pmeta[64 + l] + pos * 32 + lane] = x;
            if (lane == 0) LDSI(c.o_ringpos + l) = (pos + 1 == d) ? 0 : pos + 1;
            const int ln = (l + 1 < NL) ? l + 1 : 0;
            const int tn = (l + 1 < NL) ? t : t + 1;
            prefetch_vec(ln, tn);
            const int lnb = lay0 + ln * lstride;

            // model.py:68-69 conv_filter | conv_gate: chunk(tap0) + chunk(tap1)
            const float acc0 = dot_readlane(wt0, xo);
            __builtin_amdgcn_sched_barrier(0);
            load_tile_b(wt0, rs, v16, lnb + LayerOff::T0 * 4);
            const float acc1 = dot_readlane(wt1, x);
            __builtin_amdgcn_sched_barrier(0);
            load_tile_b(wt1, rs, v16, lnb + LayerOff::T1 * 4);
            float v = acc0 + acc1;
            if (use_bias) v = v + bfg;
            if (has_gc) v = v + gcv;      // model.py:71-73
            if (has_lc) v = v + lcv;      // model.py:75-83
            // model.py:86 tanh(filter) * sigmoid(gate): lanes 0-31 hold tanh, lanes 32-63 the logistic
            const float act = act_eval(coef, v);
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(act), __float_as_uint(act), false, false);
            const float z = __uint_as_float(sw[0]) * __uint_as_float(sw[1]);   // every lane: z[lane & 31]
            if (lane < 32) lds[c.o_zbuf + l * 32 + lane] = z;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) LDSVI(c.o_ctrl + 0) = t * NL + l + 1;

            // model.py:89 dense 1x1, model.py:98-101 residual
            float tr = dot_readlane(wdd, z);
            __builtin_amdgcn_sched_barrier(0);
            load_tile_b(wdd, rs, v16, lnb + LayerOff::WD * 4);
            if (use_bias) tr = tr + bd;
            x = x + tr;
            if (a.dbg != nullptr && t < a.dbg_steps) {
                float* dp = a.dbg + ((long long)b * a.dbg_steps + t) * ((long long)NL * 64 + L.Opad) + (long long)l * 64;
                if (lane < 32) { dp[lane] = z; dp[32 + lane] = x; }
            }
        }
        __syncthreads();   // B1
        __syncthreads();   // B2
        __syncthreads();   // B3
        __syncthreads();   // B4: next input sample published
        if (LDSVI(c.o_ctrl + 1)) break;
    }
}

// =============================== WORKER WAVES (waves 1..W) ===============================
// model.py:94-96 skip 1x1 convs and their sum, model.py:150-165 postprocessing, mixture.py:84-114 sampling.
template <int W, int NTW, bool SCALAR>
__device__ __forceinline__ void worker_main(const GenArgs& a, const Ctx& c, rsrc_t rs, int w)
{
    const Layout& L = a.lay;
    const int NL = L.NL, T = a.T, NSJ = L.NSJ, NCH = L.NCH, lane = c.lane, b = c.b;
    const int v16 = lane * 16, v4 = lane * 4;
    const bool use_bias = L.use_bias != 0;
    const int lay0 = (int)L.off_layer0 * 4, lstride = (int)L.layer_stride * 4;   // bytes
    const int skb = LayerOff::SK * 4, bsb = (LayerOff::SK + NSJ * kTile) * 4;
    Tile tk[NTW];
    float n_bs[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        const int jb = w + n * W;
        n_bs[n] = 0.0f;
        if (jb < NSJ) {
            load_tile_b(tk[n], rs, v16, lay0 + skb + jb * kTile * 4);
            if (use_bias) n_bs[n] = load_f32_b(rs, v4, lay0 + bsb + jb * 256);
        }
    }

    for (int t = 0; t < T; ++t) {
        {
            float tot[NTW];
#pragma unroll
            for (int n = 0; n < NTW; ++n) tot[n] = 0.0f;
            for (int l = 0; l < NL; ++l) {
                wait_seq(c.o_ctrl + 0, t * NL + l + 1, c.o_ctrl + 1, 100 + l);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                float zz[32];
#pragma unroll
                for (int kq = 0; kq < 8; ++kq) {
                    const float4 q = *reinterpret_cast<const float4*>(&lds[c.o_zbuf + l * 32 + 4 * kq]);
                    zz[4 * kq + 0] = q.x; zz[4 * kq + 1] = q.y; zz[4 * kq + 2] = q.z; zz[4 * kq + 3] = q.w;
                }
                const int ln = (l + 1 < NL) ? l + 1 : 0;
                const int lnb = lay0 + ln * lstride;
#pragma unroll
                for (int n = 0; n < NTW; ++n) {
                    const int jb = w + n * W;
                    if (jb < NSJ) {
                        float v = dot_regs(tk[n], zz);               // model.py:96 skip 1x1
                        __builtin_amdgcn_sched_barrier(0);
                        if (use_bias) v = v + n_bs[n];
                        tot[n] = (l == 0) ? v : tot[n] + v;          // model.py:154 sum(outputs)
                        load_tile_b(tk[n], rs, v16, lnb + skb + jb * kTile * 4);
                        if (use_bias) n_bs[n] = load_f32_b(rs, v4, lnb + bsb + jb * 256);
                    }
                }
            }
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                const int jb = w + n * W;
                if (jb < NSJ) lds[c.o_h1 + jb * 64 + lane] = tot[n] > 0.0f ? tot[n] : 0.0f;   // model.py:157 relu
            }
        }
        __syncthreads();   // B1: h1 complete
        {
            // ---- model.py:158-160 conv1d_1 (S->S) + relu: worker owns output blocks jb = w + n*W, all chunks in order
            int nown = 0;
#pragma unroll
            for (int n = 0; n < NTW; ++n) if (w + n * W < NSJ) nown = n + 1;
            const int ntiles = nown * NCH;
            Tile ta, tb;
            auto tile_off = [&](int i) -> int {
                const int n = i / NCH, ch = i - n * NCH;
                return ((int)L.off_w1 + ((w + n * W) * NCH + ch) * kTile) * 4;
            };
            if (ntiles > 0) { load_tile_b(ta, rs, v16, tile_off(0)); load_tile_b(tb, rs, v16, tile_off(1)); }
            float r = 0.0f;
            for (int i = 0; i < ntiles; i += 2) {
                const int n = i / NCH, ch = i - n * NCH;
                float acc = dot_ldso(ta, c.o_h1 + ch * 32);
                __builtin_amdgcn_sched_barrier(0);
                r = (ch == 0) ? acc : r + acc;
                if (i + 2 < ntiles) load_tile_b(ta, rs, v16, tile_off(i + 2));
                acc = dot_ldso(tb, c.o_h1 + (ch + 1) * 32);
                __builtin_amdgcn_sched_barrier(0);
                r = r + acc;
                if (i + 3 < ntiles) load_tile_b(tb, rs, v16, tile_off(i + 3));
                if (ch + 2 == NCH) {
                    const int j = (w + n * W) * 64 + lane;
                    if (use_bias) r = r + load_f32_b(rs, v4, ((int)L.off_b1 + (w + n * W) * 64) * 4);
                    lds[c.o_h2 + j] = r > 0.0f ? r : 0.0f;
                }
            }
        }
        __syncthreads();   // B2: h2 complete
        {
            // ---- model.py:161-165 conv1d_2 (S->O): chunk partials, summed in order by the sampler wave
            for (int idx = w; idx < L.NOJ * NCH; idx += W) {
                const int ch = idx % NCH;
                Tile tq;
                load_tile_b(tq, rs, v16, ((int)L.off_w2 + idx * kTile) * 4);
                lds[c.o_cpart + idx * 64 + lane] = dot_ldso(tq, c.o_h2 + ch * 32);
            }
        }
        __syncthreads();   // B3: chunk partials complete
        if (w == 0) {
            if (SCALAR) {
                // raw network output y[lane] (lane < O), then mixture.py:84-114
                float y = 0.0f;
                for (int ch = 0; ch < NCH; ++ch) {
                    const float cp = lds[c.o_cpart + ch * 64 + lane];
                    y = (ch == 0) ? cp : y + cp;
                }
                if (use_bias && lane < L.O) y = y + a.P[L.off_b2 + lane];
                if (a.dbg != nullptr && t < a.dbg_steps)
                    a.dbg[((long long)b * a.dbg_steps + t) * ((long long)NL * 64 + L.Opad) + (long long)NL * 64 + lane] = y;
                const int nr = L.nr_mix;
                const float* up = reinterpret_cast<const float*>(a.uniforms) + ((long long)b * T + t) * (nr + 1);
                const float u = lane <= nr ? up[lane] : 0.5f;
                const float g = y - log_e(-log_e(u));                        // mixture.py:103 (lanes < nr)
                int k = 0;
                float best = __shfl(g, 0);
                for (int i = 1; i < nr; ++i) {
                    const float gi = __shfl(g, i);
                    if (gi > best) { best = gi; k = i; }
                }
                const float mean = __shfl(y, nr + k);                        // mixture.py:105
                float ls = __shfl(y, 2 * nr + k);                            // mixture.py:107
                const float uu = __shfl(u, nr);
                const float lsmin = (float)-32.23619130191664;
                ls = ls > lsmin ? ls : lsmin;
                const float tq = log_e(uu) - log_e(1.0f - uu);               // mixture.py:110-111
                const float e = exp_e(ls);
                const float prod = e * tq;
                float xs = mean + prod;
                xs = xs > -1.0f ? xs : -1.0f;                                // mixture.py:113
                xs = xs < 1.0f ? xs : 1.0f;
                if (lane == 0) {
                    reinterpret_cast<float*>(a.out)[(long long)b * T + t] = xs;
                    lds[c.o_ctrl + 2] = xs;
                }
            }
        }
        __syncthreads();   // B4: next input sample published
        if (LDSVI(c.o_ctrl + 1)) break;
    }
}

template <int W, int NTW, bool SCALAR>
__global__ void __launch_bounds__((1 + W) * 64) wn_generate_kernel(GenArgs a)
{
    const Layout& L = a.lay;
    const int NL = L.NL, S = L.S, NCH = L.NCH;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int T = a.T;
    Ctx c;
    c.lane = threadIdx.x & 63;
    c.b = blockIdx.x;
    c.o_zbuf = 0;                                   // [64][32]   gated outputs z_l of the current step
    c.o_h1 = c.o_zbuf + 64 * 32;                    // [S]        relu(sum of skips)
    c.o_h2 = c.o_h1 + S;                            // [S]        relu(post conv 1)
    c.o_cpart = c.o_h2 + S;                         // [NOJ][NCH][64] chunk partials of the last conv
    c.o_hist = c.o_cpart + L.NOJ * NCH * 64;        // [64]       causal_queue (circular)
    c.o_ringpos = c.o_hist + 64;                    // [64]       write position of every delay line
    c.o_ctrl = c.o_ringpos + 64;                    // [16]
    c.stb = a.state + (long long)c.b * L.state_stride;
    c.pmeta = reinterpret_cast<const int*>(a.P + L.off_meta);
    c.ring = c.stb + L.st_ring;
    c.GCv = a.cond + (long long)c.b * NL * 64;
    c.LCb = a.cond + (long long)a.B * NL * 64 + (long long)c.b * T * NL * 64;
    int* meta = reinterpret_cast<int*>(c.stb + L.st_meta);
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.P), 0, (int)(L.packed_floats * 4), 0x00020000);

    if (threadIdx.x < 64) {
        lds[c.o_hist + threadIdx.x] = c.stb[L.st_hist + threadIdx.x];
        LDSI(c.o_ringpos + threadIdx.x) = reinterpret_cast<int*>(c.stb + L.st_ringpos)[threadIdx.x];
    }
    if (threadIdx.x < 16) LDSI(c.o_ctrl + threadIdx.x) = 0;
    __syncthreads();

    int hpos = meta[M_HPOS], prev_valid = meta[M_PREV_VALID], qprev = meta[M_QPREV];
    if (wid == 0) chain_main<SCALAR>(a, c, rs, hpos, prev_valid, qprev);
    else worker_main<W, NTW, SCALAR>(a, c, rs, wid - 1);

    // ---------------- persist the per-stream state (model.py:49-64 queues) ----------------
    __syncthreads();
    if (threadIdx.x < 64) {
        c.stb[L.st_hist + threadIdx.x] = lds[c.o_hist + threadIdx.x];
        reinterpret_cast<int*>(c.stb + L.st_ringpos)[threadIdx.x] = LDSI(c.o_ringpos + threadIdx.x);
    }
    if (wid == 0 && c.lane == 0) {
        meta[M_TABS] = meta[M_TABS] + T;
        meta[M_HPOS] = hpos;
        meta[M_PREV_VALID] = prev_valid;
        meta[M_QPREV] = qprev;
        if (LDSI(c.o_ctrl + 1)) atomicMax(a.status, LDSI(c.o_ctrl + 1));
    }
    if (L.L > 0 && T > 0) {
        const float* last = c.LCb + (long long)(T - 1) * NL * 64;
        for (int i = threadIdx.x; i < NL * 64; i += blockDim.x) c.stb[L.st_lcprev + i] = last[i];
    }
}

// =====================================================================================================
//  mu-law codec (ops.py:22-47) and contract-function evaluators
// =====================================================================================================
__global__ void wn_mulaw_encode_kernel(const float* audio, long long n, int Q, int32_t* out)
{
    const float mu = (float)(Q - 1);
    const float log1p_mu = log1p_e(mu);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = audio[i];
        const float av = fabsf(v);
        const float safe = av < 1.0f ? av : 1.0f;
        const float mag = div_(log1p_e(mu * safe), log1p_mu);
        const float sgn = (float)((v > 0.0f) - (v < 0.0f));
        const float signal = sgn * mag;
        const float s1 = signal + 1.0f;
        const float s2 = div_(s1, 2.0f);
        const float s3 = s2 * mu;
        out[i] = (int32_t)(s3 + 0.5f);
    }
}
__device__ __forceinline__ float mulaw_expand_one(float s, int Q)
{
    const int mu = Q - 1;
    const float inv = div_(1.0f, (float)mu);
    const float pw = exp_e(fabsf(s) * log_e((float)(1 + mu)));
    const float mag = inv * (pw - 1.0f);
    const float sgn = (float)((s > 0.0f) - (s < 0.0f));
    return sgn * mag;
}
__global__ void wn_mulaw_decode_kernel(const int32_t* q, long long n, int Q, float* out)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float r = div_((float)q[i], (float)(Q - 1));
        const float s = 2.0f * r - 1.0f;
        out[i] = mulaw_expand_one(s, Q);
    }
}
__global__ void wn_mulaw_expand_kernel(const float* y, long long n, int Q, float* out)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = mulaw_expand_one(y[i], Q);
}
__global__ void wn_eval_kernel(int fn, const float* x, long long n, float* out)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = x[i];
        float r;
        switch (fn) {
            case 0: r = tanh_e(v); break;
            case 1: r = sigmoid_e(v); break;
            case 2: r = exp_e(v); break;
            case 3: r = log_e(v); break;
            default: r = log1p_e(v); break;
        }
        out[i] = r;
    }
}
__global__ void wn_eval64_kernel(int fn, const double* x, long long n, double* out)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = fn == 0 ? exp64_e(x[i]) : log64_e(x[i]);
}

// cross-lane primitive self-test: out[lane] = {permlane32_swap result 0, result 1, readlane(5), shfl(lane^1)}
__global__ void wn_selftest_kernel(float* out)
{
    const int lane = threadIdx.x & 63;
    const float v = (float)(lane + 1);
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    out[lane * 4 + 0] = __uint_as_float(sw[0]);
    out[lane * 4 + 1] = __uint_as_float(sw[1]);
    out[lane * 4 + 2] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 5));
    out[lane * 4 + 3] = __shfl(v, lane ^ 1);
}

// =====================================================================================================
//  host side: handle, layouts, C-ABI
// =====================================================================================================
struct twv_wavenet {
    twv_wavenet_dims dims;
    Layout lay;
    int dil[kMaxLayers];
    int ring_off[kMaxLayers];
    int workers;   // 4 or 8
};

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(expr)                                                                                              \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) return fail(TWV_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));         \
    } while (0)

extern "C" const char* twv_last_error(void) { return g_err.c_str(); }
extern "C" const char* twv_version(void) { return "twv_amd 0.1 (gfx950)"; }

static inline long long align_up(long long v, long long a) { return (v + a - 1) / a * a; }

static int build_layout(const twv_wavenet_dims& d, twv_wavenet* h)
{
    Layout& L = h->lay;
    memset(&L, 0, sizeof(L));
    if (d.n_layers < 1 || d.n_layers > kMaxLayers) return fail(TWV_E_INVALID, "n_layers must be in [1,64]");
    if (d.residual_channels != 32 || d.dilation_channels != 32)
        return fail(TWV_E_UNSUPPORTED, "residual_channels and dilation_channels must be 32 (hparams.py:71-72)");
    if (d.skip_channels < 64 || d.skip_channels > 1024 || d.skip_channels % 64)
        return fail(TWV_E_UNSUPPORTED, "skip_channels must be a multiple of 64 in [64,1024]");
    if (d.gc_channels < 0 || d.gc_channels > 64) return fail(TWV_E_UNSUPPORTED, "gc_channels must be <= 64");
    if (d.gc_channels > 0 && d.gc_cardinality < 1) return fail(TWV_E_INVALID, "gc_cardinality required with gc_channels");
    if (d.lc_channels < 0 || d.lc_channels > 128 || d.lc_channels % 4) return fail(TWV_E_UNSUPPORTED, "lc_channels must be a multiple of 4, <= 128");
    if (d.lc_channels > 0 && (d.n_upsample < 1 || d.n_upsample > 4)) return fail(TWV_E_INVALID, "1..4 upsample factors required with lc_channels");
    L.NL = d.n_layers; L.S = d.skip_channels; L.Q = d.quantization_channels; L.scalar = d.scalar_input ? 1 : 0;
    L.use_bias = d.use_biases ? 1 : 0; L.G = d.gc_channels; L.gc_card = d.gc_channels ? d.gc_cardinality : 0;
    L.L = d.lc_channels; L.n_up = d.lc_channels ? d.n_upsample : 0;
    for (int i = 0; i < L.n_up; ++i) { L.up[i] = d.upsample_factor[i]; if (L.up[i] < 1) return fail(TWV_E_INVALID, "upsample_factor must be >= 1"); }
    if (L.scalar) {
        if (d.out_channels < 3 || d.out_channels % 3 || d.out_channels > 63) return fail(TWV_E_UNSUPPORTED, "out_channels must be 3*nr_mix <= 63");
        if (d.initial_filter_width < 1 || d.initial_filter_width > 64) return fail(TWV_E_UNSUPPORTED, "initial_filter_width must be in [1,64]");
        L.O = d.out_channels; L.ifw = d.initial_filter_width; L.nr_mix = L.O / 3;
    } else {
        if (L.Q < 2 || L.Q > 1024) return fail(TWV_E_UNSUPPORTED, "quantization_channels must be in [2,1024]");
        L.O = L.Q; L.ifw = 2; L.nr_mix = 0;
    }
    L.Opad = (int)align_up(L.O, 64);
    L.NSJ = L.S / 64; L.NCH = L.S / 32; L.NOJ = L.Opad / 64;
    L.NCA = (L.ifw + 31) / 32; L.NLC = (L.L + 31) / 32; L.NGC = (L.G + 31) / 32;
    int ro = 0;
    for (int i = 0; i < L.NL; ++i) {
        if (d.dilations[i] < 1) return fail(TWV_E_INVALID, "dilations must be >= 1");
        h->dil[i] = d.dilations[i]; h->ring_off[i] = ro; ro += d.dilations[i] * 32;
    }
    L.ring_floats = ro;
    // ---- packed layout
    long long p = 0;
    L.off_meta = p; p += 128;
    L.off_causal = p; p += L.scalar ? (long long)L.NCA * kTile : (long long)2 * L.Q * 32;
    L.off_layer0 = p;
    L.layer_stride = LayerOff::SK + (long long)L.NSJ * kTile + L.S;
    p += L.layer_stride * L.NL;
    L.off_w1 = p; p += (long long)L.NSJ * L.NCH * kTile;
    L.off_b1 = p; p += L.S;
    L.off_w2 = p; p += (long long)L.NOJ * L.NCH * kTile;
    L.off_b2 = p; p += L.Opad;
    L.off_lcw = p; L.lcw_stride = (long long)L.NLC * kTile; p += L.lcw_stride * L.NL;
    L.off_gcw = p; L.gcw_stride = (long long)L.NGC * kTile; p += L.gcw_stride * L.NL;
    L.off_gcemb = p; p += align_up((long long)L.gc_card * L.G, 4);
    for (int i = 0; i < L.n_up; ++i) { L.off_up[i] = p; p += align_up((long long)L.up[i] * 2, 4); }
    L.packed_floats = p;
    // ---- canonical blob (must match DESIGN.md "canonical blob"; mirrored by the checker independently)
    long long c = 0;
    const int R = 32, D = 32;
    L.c_causal = c; c += L.scalar ? (long long)L.ifw * R : (long long)2 * L.Q * R;
    L.c_gcemb = c; c += (long long)L.gc_card * L.G;
    L.c_layer0 = c;
    long long q = 0;
    L.c_wf = q; q += 2 * R * D; L.c_bf = q; if (L.use_bias) q += D;
    L.c_wg = q; q += 2 * R * D; L.c_bg = q; if (L.use_bias) q += D;
    L.c_gcf = q; q += (long long)L.G * D; L.c_gcg = q; q += (long long)L.G * D;
    L.c_lcf = q; q += (long long)L.L * D; L.c_lcg = q; q += (long long)L.L * D;
    L.c_wd = q; q += D * R; L.c_bd = q; if (L.use_bias) q += R;
    L.c_ws = q; q += (long long)D * L.S; L.c_bs = q; if (L.use_bias) q += L.S;
    L.c_layer_stride = q;
    c += q * L.NL;
    L.c_w1 = c; c += (long long)L.S * L.S; L.c_b1 = c; if (L.use_bias) c += L.S;
    L.c_w2 = c; c += (long long)L.S * L.O; L.c_b2 = c; if (L.use_bias) c += L.O;
    for (int i = 0; i < L.n_up; ++i) { L.c_up[i] = c; c += (long long)L.up[i] * 2; }
    L.blob_floats = c;
    // ---- per-stream state
    long long s = 0;
    L.st_hist = s; s += 64;
    L.st_meta = s; s += 64;
    L.st_ringpos = s; s += 64;
    L.st_lcprev = s; s += (long long)L.NL * 64;
    L.st_ring = s; s += L.ring_floats;
    L.state_stride = align_up(s, 64);
    return TWV_OK;
}

extern "C" int twv_wavenet_create(const twv_wavenet_dims* dims, twv_wavenet** out)
{
    if (!dims || !out) return fail(TWV_E_INVALID, "null argument");
    twv_wavenet* h = new twv_wavenet();
    h->dims = *dims;
    h->workers = 4;
    const int rc = build_layout(*dims, h);
    if (rc != TWV_OK) { delete h; return rc; }
    *out = h;
    return TWV_OK;
}
extern "C" void twv_wavenet_destroy(twv_wavenet* h) { delete h; }

extern "C" int twv_wavenet_receptive_field(const twv_wavenet* h)
{
    int sum = 0;
    for (int i = 0; i < h->lay.NL; ++i) sum += h->dil[i];
    return sum + 1 + (h->lay.scalar ? h->lay.ifw - 1 : 1);   // model.py:31-39, filter_width = 2
}
extern "C" int twv_wavenet_hop_size(const twv_wavenet* h)
{
    int hop = 1;
    for (int i = 0; i < h->lay.n_up; ++i) hop *= h->lay.up[i];
    return hop;
}
extern "C" size_t twv_wavenet_blob_floats(const twv_wavenet* h) { return (size_t)h->lay.blob_floats; }
extern "C" size_t twv_wavenet_packed_bytes(const twv_wavenet* h) { return (size_t)h->lay.packed_floats * 4; }
extern "C" size_t twv_wavenet_state_bytes(const twv_wavenet* h, int batch) { return (size_t)h->lay.state_stride * 4 * (size_t)batch; }
extern "C" size_t twv_wavenet_cond_bytes(const twv_wavenet* h, int batch, int n_steps)
{
    return ((size_t)batch * h->lay.NL * 64 + (size_t)batch * (size_t)n_steps * h->lay.NL * 64) * 4;
}
extern "C" int twv_wavenet_set_option(twv_wavenet* h, const char* name, int value)
{
    if (!h || !name) return fail(TWV_E_INVALID, "null argument");
    if (!strcmp(name, "workers")) {
        if (value != 4 && value != 8) return fail(TWV_E_INVALID, "workers must be 4 or 8");
        h->workers = value;
        return TWV_OK;
    }
    return fail(TWV_E_INVALID, std::string("unknown option ") + name);
}

static inline int grid_for(long long n, int block) { long long g = (n + block - 1) / block; return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); }

extern "C" int twv_wavenet_pack(const twv_wavenet* h, const float* blob, void* packed, void* stream)
{
    if (!h || !blob || !packed) return fail(TWV_E_INVALID, "null argument");
    const Layout& L = h->lay;
    hipStream_t st = (hipStream_t)stream;
    float* dst = (float*)packed;
    HIPCHK(hipMemsetAsync(dst, 0, (size_t)L.packed_floats * 4, st));
    int metah[128];
    for (int i = 0; i < 64; ++i) { metah[i] = i < L.NL ? h->dil[i] : 1; metah[64 + i] = i < L.NL ? h->ring_off[i] : 0; }
    HIPCHK(hipMemcpyAsync(dst + L.off_meta, metah, sizeof(metah), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));   // metah is a stack buffer
    auto tiles = [&](long long dst_off, long long dst_gs, long long baseA, long long baseB, long long src_gs, int ng, int njb, int nch,
                     int K, int rowlen, int ncols, int halves) {
        PackTiles p{dst_off, dst_gs, baseA, baseB, src_gs, ng, njb, nch, K, rowlen, ncols, halves};
        const long long total = (long long)ng * njb * nch * kTile;
        hipLaunchKernelGGL(wn_pack_tiles_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, dst, blob, p);
    };
    auto vec = [&](long long dst_off, long long dst_gs, long long baseA, long long baseB, long long src_gs, int ng, int n, int ncols, int halves) {
        PackVec p{dst_off, dst_gs, baseA, baseB, src_gs, ng, n, ncols, halves};
        hipLaunchKernelGGL(wn_pack_vec_kernel, dim3(grid_for((long long)ng * n, 256)), dim3(256), 0, st, dst, blob, p);
    };
    const long long ls = L.layer_stride, cs = L.c_layer_stride, c0 = L.c_layer0, l0 = L.off_layer0;
    if (L.scalar) {
        tiles(L.off_causal, 0, L.c_causal, L.c_causal, 0, 1, 1, L.NCA, L.ifw, 32, 32, 1);      // wavenet/conv1d/kernel (ifw,1,R)
    } else {
        hipLaunchKernelGGL(wn_copy_kernel, dim3(grid_for((long long)2 * L.Q * 32, 256)), dim3(256), 0, st, dst + L.off_causal, blob + L.c_causal, (long long)2 * L.Q * 32);
    }
    // conv_filter|conv_gate kernel (2,R,D): tap 0 rows [0,32), tap 1 rows [32,64)
    tiles(l0 + LayerOff::T0, ls, c0 + L.c_wf, c0 + L.c_wg, cs, L.NL, 1, 1, 32, 32, 32, 1);
    tiles(l0 + LayerOff::T1, ls, c0 + L.c_wf + 32 * 32, c0 + L.c_wg + 32 * 32, cs, L.NL, 1, 1, 32, 32, 32, 1);
    tiles(l0 + LayerOff::WD, ls, c0 + L.c_wd, c0 + L.c_wd, cs, L.NL, 1, 1, 32, 32, 32, 1);    // dense kernel (1,D,R), duplicated
    tiles(l0 + LayerOff::SK, ls, c0 + L.c_ws, 0, cs, L.NL, L.NSJ, 1, 32, L.S, L.S, 0);         // skip kernel (1,D,S)
    if (L.use_bias) {
        vec(l0 + LayerOff::BFG, ls, c0 + L.c_bf, c0 + L.c_bg, cs, L.NL, 64, 32, 1);
        vec(l0 + LayerOff::BD, ls, c0 + L.c_bd, c0 + L.c_bd, cs, L.NL, 64, 32, 1);
        vec(l0 + LayerOff::SK + (long long)L.NSJ * kTile, ls, c0 + L.c_bs, 0, cs, L.NL, L.S, L.S, 0);
        vec(L.off_b1, 0, L.c_b1, 0, 0, 1, L.S, L.S, 0);
        vec(L.off_b2, 0, L.c_b2, 0, 0, 1, L.Opad, L.O, 0);
    }
    tiles(L.off_w1, 0, L.c_w1, 0, 0, 1, L.NSJ, L.NCH, L.S, L.S, L.S, 0);                       // conv1d_1 kernel (1,S,S)
    tiles(L.off_w2, 0, L.c_w2, 0, 0, 1, L.NOJ, L.NCH, L.S, L.O, L.O, 0);                       // conv1d_2 kernel (1,S,O)
    if (L.L) tiles(L.off_lcw, L.lcw_stride, c0 + L.c_lcf, c0 + L.c_lcg, cs, L.NL, 1, L.NLC, L.L, 32, 32, 1);
    if (L.G) {
        tiles(L.off_gcw, L.gcw_stride, c0 + L.c_gcf, c0 + L.c_gcg, cs, L.NL, 1, L.NGC, L.G, 32, 32, 1);
        hipLaunchKernelGGL(wn_copy_kernel, dim3(grid_for((long long)L.gc_card * L.G, 256)), dim3(256), 0, st, dst + L.off_gcemb, blob + L.c_gcemb, (long long)L.gc_card * L.G);
    }
    for (int i = 0; i < L.n_up; ++i)
        hipLaunchKernelGGL(wn_copy_kernel, dim3(1), dim3(256), 0, st, dst + L.off_up[i], blob + L.c_up[i], (long long)L.up[i] * 2);
    HIPCHK(hipGetLastError());
    return TWV_OK;
}

extern "C" int twv_wavenet_reset_state(const twv_wavenet* h, void* state, int batch, void* stream)
{
    if (!h || !state || batch < 1) return fail(TWV_E_INVALID, "bad argument");
    HIPCHK(hipMemsetAsync(state, 0, twv_wavenet_state_bytes(h, batch), (hipStream_t)stream));
    return TWV_OK;
}

extern "C" int twv_wavenet_upsample(const twv_wavenet* h, const void* packed, const float* mel, int batch, int t_mel,
                                    float* out, float* scratch, void* stream)
{
    if (!h || !packed || !mel || !out || !scratch || batch < 1 || t_mel < 1) return fail(TWV_E_INVALID, "bad argument");
    const Layout& L = h->lay;
    if (!L.L) return fail(TWV_E_INVALID, "model has no local conditioning");
    hipStream_t st = (hipStream_t)stream;
    const float* P = (const float*)packed;
    // ping-pong so that the LAST stage lands in `out`
    const float* cur = mel;
    long long T = t_mel;
    for (int i = 0; i < L.n_up; ++i) {
        const int remaining = L.n_up - 1 - i;
        float* dst = (remaining % 2 == 0) ? out : scratch;
        const long long total = (long long)batch * T * L.up[i] * L.L;
        hipLaunchKernelGGL(wn_upsample_stage_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, P + L.off_up[i], cur, dst, batch, T, L.up[i], L.L);
        cur = dst;
        T *= L.up[i];
    }
    HIPCHK(hipGetLastError());
    return TWV_OK;
}

extern "C" int twv_wavenet_condition(const twv_wavenet* h, const void* packed, const float* upsampled, const int32_t* gc_ids,
                                     int batch, int n_steps, void* cond, void* stream)
{
    if (!h || !packed || !cond || batch < 1 || n_steps < 0) return fail(TWV_E_INVALID, "bad argument");
    const Layout& L = h->lay;
    hipStream_t st = (hipStream_t)stream;
    const float* P = (const float*)packed;
    float* GCv = (float*)cond;
    float* LC = GCv + (size_t)batch * L.NL * 64;
    if (L.G) {
        if (!gc_ids) return fail(TWV_E_INVALID, "gc_ids required (generate.py:72-77)");
        hipLaunchKernelGGL(wn_gc_kernel, dim3(batch), dim3(64), 0, st, P, L, gc_ids, GCv);
    } else {
        HIPCHK(hipMemsetAsync(GCv, 0, (size_t)batch * L.NL * 64 * 4, st));
    }
    if (L.L && n_steps > 0) {
        if (!upsampled) return fail(TWV_E_INVALID, "upsampled local condition required");
        const long long rows = (long long)batch * n_steps;
        const size_t shm = (size_t)kLcRows * L.NLC * 32 * 4;
        dim3 grid((unsigned)((rows + kLcRows - 1) / kLcRows), (unsigned)((L.NL + 3) / 4));
        hipLaunchKernelGGL(wn_lc_kernel, grid, dim3(256), shm, st, P, L, upsampled, LC, rows);
    }
    HIPCHK(hipGetLastError());
    return TWV_OK;
}

template <int W, int NTW, bool SCALAR>
static int launch_generate(const GenArgs& a, size_t shm, hipStream_t st)
{
    auto kern = wn_generate_kernel<W, NTW, SCALAR>;
    if (shm > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL(kern, dim3(a.B), dim3((1 + W) * 64), shm, st, a);
    HIPCHK(hipGetLastError());
    return TWV_OK;
}

extern "C" int twv_wavenet_generate(const twv_wavenet* h, const void* packed, void* state, const void* cond,
                                    const void* first_input, const void* uniforms, double temperature,
                                    int batch, int n_steps, void* out, int32_t* status, float* debug, int debug_steps,
                                    void* stream)
{
    if (!h || !packed || !state || !cond || !first_input || !uniforms || !out || !status) return fail(TWV_E_INVALID, "null argument");
    if (batch < 1 || n_steps < 1) return fail(TWV_E_INVALID, "batch and n_steps must be >= 1");
    const Layout& L = h->lay;
    if ((long long)n_steps * L.NL > 2000000000LL) return fail(TWV_E_INVALID, "n_steps too large for one call");
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(status, 0, 16, st));
    GenArgs a;
    a.P = (const float*)packed; a.state = (float*)state; a.cond = (const float*)cond; a.first_input = first_input;
    a.uniforms = uniforms; a.out = out; a.status = status; a.dbg = debug; a.dbg_steps = debug ? debug_steps : 0;
    a.B = batch; a.T = n_steps; a.temperature = (float)temperature; a.lay = L;
    const size_t shm = ((size_t)64 * 32 + 2 * (size_t)L.S + (size_t)L.NOJ * L.NCH * 64 + 64 + 64 + 16) * 4;
    if (!L.scalar) return fail(TWV_E_UNSUPPORTED, "one-hot (mu-law softmax) generation is not built yet");
    const int W = h->workers;
    const int ntw = (L.NSJ + W - 1) / W;
    if (W == 8 && ntw == 1) return launch_generate<8, 1, true>(a, shm, st);
    if (W == 8 && ntw == 2) return launch_generate<8, 2, true>(a, shm, st);
    if (W == 4 && ntw <= 2) return launch_generate<4, 2, true>(a, shm, st);
    if (W == 4 && ntw <= 4) return launch_generate<4, 4, true>(a, shm, st);
    return fail(TWV_E_UNSUPPORTED, "skip_channels too large for the selected worker count");
}

extern "C" int twv_wavenet_status(const int32_t* status, void* stream)
{
    int32_t hst[4] = {0, 0, 0, 0};
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    HIPCHK(hipMemcpy(hst, status, sizeof(hst), hipMemcpyDeviceToHost));
    if (hst[0] != 0) return fail(TWV_E_KERNEL, "generation kernel watchdog code " + std::to_string(hst[0]));
    return TWV_OK;
}

extern "C" int twv_mu_law_encode(const float* audio, int64_t n, int Q, int32_t* out, void* stream)
{
    if (!audio || !out || n < 0 || Q < 2) return fail(TWV_E_INVALID, "bad argument");
    if (n) hipLaunchKernelGGL(wn_mulaw_encode_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, audio, (long long)n, Q, out);
    HIPCHK(hipGetLastError());
    return TWV_OK;
}
extern "C" int twv_mu_law_decode(const int32_t* q, int64_t n, int Q, float* out, void* stream)
{
    if (!q || !out || n < 0 || Q < 2) return fail(TWV_E_INVALID, "bad argument");
    if (n) hipLaunchKernelGGL(wn_mulaw_decode_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, q, (long long)n, Q, out);
    HIPCHK(hipGetLastError());
    return TWV_OK;
}
extern "C" int twv_mu_law_expand(const float* y, int64_t n, int Q, float* out, void* stream)
{
    if (!y || !out || n < 0 || Q < 2) return fail(TWV_E_INVALID, "bad argument");
    if (n) hipLaunchKernelGGL(wn_mulaw_expand_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, y, (long long)n, Q, out);
    HIPCHK(hipGetLastError());
    return TWV_OK;
}
extern "C" int twv_selftest(float* out256, void* stream)
{
    if (!out256) return fail(TWV_E_INVALID, "null argument");
    hipLaunchKernelGGL(wn_selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out256);
    HIPCHK(hipGetLastError());
    return TWV_OK;
}
extern "C" int twv_eval_elementwise(int fn, const float* x, int64_t n, float* out, void* stream)
{
    if (!x || !out || n < 0 || fn < 0 || fn > 4) return fail(TWV_E_INVALID, "bad argument");
    if (n) hipLaunchKernelGGL(wn_eval_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, fn, x, (long long)n, out);
    HIPCHK(hipGetLastError());
    return TWV_OK;
}
extern "C" int twv_eval_elementwise64(int fn, const double* x, int64_t n, double* out, void* stream)
{
    if (!x || !out || n < 0 || fn < 0 || fn > 1) return fail(TWV_E_INVALID, "bad argument");
    if (n) hipLaunchKernelGGL(wn_eval64_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, fn, x, (long long)n, out);
    HIPCHK(hipGetLastError());
    return TWV_OK;
}
