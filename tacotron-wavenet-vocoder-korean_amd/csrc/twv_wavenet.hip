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
// Design: G workgroups per utterance (stream), 8 waves each.  Wave 0 is the CHAIN wave: it walks the dilated
// residual stack layer by layer (the strictly serial part), every lane owning one filter/gate output, operands broadcast
// with v_readlane into packed fmas, the gated tanh*sigmoid evaluated as one instruction stream with per-half-wave
// coefficients.  Waves 1..3 are LOADERS: they stage each layer's chain weights, x[t-d] and conditioning row into a ring
// of LDS slots by LDS-DMA, compute the tap-0 chunk and keep the delay lines in HBM up to date.  Waves 4..7 are WORKERS:
// they stream the wide 1x1 convolutions (skip, post) as 64x32 weight tiles into registers and run the sampler.  All G
// workgroups of a stream run the same chain (identical bits) and own 1/G of the skip / conv1d_1 output blocks; two
// all-gathers per step travel as {epoch, value} granules.  Waves synchronise through LDS sequence words only.
// All arithmetic follows the arithmetic contract: results are bit-identical to the
// CPU checker for any launch geometry.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include "../../include/twv_amd.h"
#include "twv_layout.hpp"
#include "twv_math.hpp"
#include "twv_dev.hpp"
#include "twv_categorical.hpp"
#include "twv_xcd.hpp"

using namespace twv;

// =====================================================================================================
//  small device helpers
// =====================================================================================================
// =====================================================================================================
//  pack: canonical checkpoint blob (TF layouts) -> streaming layout        (generate.py:157-161 Saver.restore)
// =====================================================================================================
// dst tiles [g][jblk][chunk][kq][lane][4]; element = src[base(half) + g*sg + k*rowlen + jj] (0 outside K x ncols)
__global__ void wn_pack_tiles_kernel(float* dst, const float* src, PackTiles p)
{
    const int tile = 32 * p.lanes;   // floats per tile: [kq=8][lanes][4]
    const long long per_g = (long long)p.njblk * p.nchunk * tile;
    const long long total = per_g * p.ngroups;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(i / per_g);
        long long r = i - (long long)g * per_g;
        const int jb = (int)(r / ((long long)p.nchunk * tile));
        r -= (long long)jb * p.nchunk * tile;
        const int ch = (int)(r / tile);
        const int e = (int)(r - (long long)ch * tile);
        const int kq = e / (p.lanes * 4), lane = (e >> 2) % p.lanes, q = e & 3;
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
        dst[p.dst_off + (long long)g * p.dst_gstride + r + (long long)jb * p.nchunk * tile] = v;
    }
}
// vectors: dst[g][j] (n entries per group)
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
        const float s0 = fma_(K[a * 2 + 0], x0, 0.0f);
        const float s1 = fma_(K[a * 2 + 1], x1, 0.0f);
        out[i] = (s0 + s1) + (0.0f + 0.0f);   // AC-1 chunk with two terms
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
    // model.py:191-207: ids looked up in gc_embedding, or (no cardinality) the embedding itself as (B, G) floats through the same pointer
    if (L.gc_card > 0) {
        const int id = gc_ids[b];
        emb[lane] = lane < L.G ? P[L.off_gcemb + (long long)id * L.G + lane] : 0.0f;
    } else {
        emb[lane] = lane < L.G ? reinterpret_cast<const float*>(gc_ids)[(long long)b * L.G + lane] : 0.0f;
    }
    __syncthreads();
    for (int l = 0; l < L.NL; ++l) {
        float res = 0.0f;
        for (int c = 0; c < L.NGC; ++c) {
            Tile t;
            load_tile(t, P + L.off_gcw + (long long)l * L.gcw_stride + (long long)c * kTile, lane);
            const int kn = min(32, L.G - c * 32);
            float sj[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < 32; ++k)
                if (k < kn) sj[k & 3] = fma_(t.w[k], emb[c * 32 + k], sj[k & 3]);
            const float acc = (sj[0] + sj[1]) + (sj[2] + sj[3]);
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
                float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll
                for (int kq = 0; kq < 8; ++kq) {
                    if (kq < kqn) {
                        const float4 q = p[kq];
                        s0 = fma_(t[c].w[4 * kq + 0], q.x, s0);
                        s1 = fma_(t[c].w[4 * kq + 1], q.y, s1);
                        s2 = fma_(t[c].w[4 * kq + 2], q.z, s2);
                        s3 = fma_(t[c].w[4 * kq + 3], q.w, s3);
                    }
                }
                const float acc = (s0 + s1) + (s2 + s3);
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
    const void* forced;        // optional (B,T): teacher-forced inputs (priming, generate.py:177-180); no sampling then
    const void* uniforms;      // (B,T,nr_mix+1) f32  |  (B,T) f64
    void* out;                 // (B,T)
    int* status;               // [4]
    float* dbg;                // optional [B][dbg_steps][NL*64 + Opad]
    int dbg_steps;
    unsigned long long* prof;  // optional [prof_steps][80] s_memtime stamps of stream 0's chain wave
    int prof_steps;
    int B, T;
    int G;                       // workgroups per stream (each owns 1/G of the skip / conv1d_1 outputs)
    unsigned long long* exch;    // [B][2][S] {epoch,value} granules: all-gather of relu(skip sum) and relu(conv1d_1)
    float temperature;
    Layout lay;
};

// Intra-workgroup synchronisation is by monotonically increasing LDS sequence words only (no s_barrier inside the
// sample loop, so the wave roles run decoupled).  Every wait is BOUNDED: after ~2^22 polls it raises the workgroup's
// abort word, every other wait then falls through, and the launch ends with a watchdog code instead of hanging.
enum { C_ZSEQ = 0, C_ABORT = 1, C_SAMPLE = 2, C_CDONE = 3, C_H1CNT = 4, C_H2CNT = 5, C_CPCNT = 6, C_SSEQ = 7, C_P1CNT = 8, C_SKCNT = 9, C_LGCNT = 10, C_SKP = 11 /* ..14: per-worker skip progress */ };

__device__ __forceinline__ bool wait_seq(int fo_flag, int target, int fo_abort, int code)
{
    if (LDSVI(fo_flag) >= target) return true;
    // not unrolled: fifteen call sites x an 8-fold unrolled spin loop was a third of the kernel's code, and the three roles of a
    // workgroup share one instruction cache
#pragma nounroll
    for (int it = 0; it < (1 << 22); ++it) {
        __builtin_amdgcn_s_sleep(1);
        if (LDSVI(fo_flag) >= target) return true;
        if (LDSVI(fo_abort)) return false;
    }
    LDSVI(fo_abort) = code;
    return false;
}
// All payloads guarded by these words live in LDS, and one wave's LDS operations are performed in issue order, so the
// only ordering needed is "payload ds_write before flag ds_write" (program order) on the producer and "flag ds_read
// before payload ds_read" on the consumer.  A workgroup-scope C++ fence would also drain vmcnt -- i.e. wait for every
// outstanding HBM store/load of the wave (measured: +1.5 us per layer) -- so plain compiler barriers are used instead.
__device__ __forceinline__ void publish(int fo_flag, int value, int lane)
{
    asm volatile("" ::: "memory");
    if (lane == 0) LDSVI(fo_flag) = value;
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void arrive(int fo_cnt, int lane)
{
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(&LDSI(fo_cnt), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#define ACQUIRE_WG() asm volatile("" ::: "memory")

// ---- inter-workgroup all-gather (the G workgroups of one stream) -------------------------------------------------
// Data-tagged 8-byte granules {epoch, value}: ONE naturally aligned agent-scope (sc1, write-through) store per value,
// readers re-read with agent-scope loads until every tag equals the epoch -- no flag, no fence, placement independent
// (MI355X per-XCD L2s are not coherent; sc1 accesses bypass the L1 and meet at memory side).  Polls are bounded.
typedef __attribute__((address_space(1))) unsigned long long gu64;
__device__ __forceinline__ void granule_store(unsigned long long* p, unsigned epoch, float v)
{
    __hip_atomic_store((gu64*)p, ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the W worker waves sweep all S granules of X into lds[o_dst .. o_dst+S)
template <int W>
__device__ __forceinline__ bool gather_granules(unsigned long long* X, int S, unsigned epoch, int o_dst, int w, int lane,
                                                int fo_abort, int code)
{
    bool done[4] = {false, false, false, false};     // up to 4 granules per lane (S <= 1024 with W*64 = 256 lanes)
#pragma nounroll
    for (int it = 0; it < (1 << 20); ++it) {
        bool all_ok = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = w * 64 + lane + k * W * 64;
            if (idx < S && !done[k]) {
                const unsigned long long v = __hip_atomic_load((gu64*)(X + idx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(v >> 32) == epoch) { lds[o_dst + idx] = __uint_as_float((unsigned)v); done[k] = true; }
                else all_ok = false;
            }
        }
        if (__all(all_ok)) return true;
        if ((it & 15) == 15 && LDSVI(fo_abort)) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    LDSVI(fo_abort) = code;
    return false;
}

// per-workgroup context shared by the wave roles (LDS positions are float offsets into lds[])
struct Ctx {
    int o_zbuf, o_h1, o_h2, o_cpart, o_meta, o_ringpos, o_pos0, o_ready, o_ctrl, o_gc, o_ring1, o_causal, o_cpart1, o_skl, o_cat, o_slots;
    int b, g, lane;
    float* stb;            // this stream's state
    float* ring;
    const float *GCv, *LCb;
};

// =============================== CHAIN WAVE (wave 0) ===============================
// Walks model.py:112-149 for one stream: causal layer, then the dilated residual stack, one layer after the other.
// Everything a layer needs (conv/dense weights, biases, x[t-d], lc projection) has been staged in an LDS slot by the
// loader waves; the gc projections and the causal kernel are LDS-resident for the whole launch.
template <bool SCALAR, bool INSTR>
__device__ __forceinline__ void chain_main(const GenArgs& a, const Ctx& c, int& hpos, int& prev_valid, int& qprev)
{
    const Layout& L = a.lay;
    const int NL = L.NL, T = a.T, lane = c.lane, b = c.b, nslot = L.nslot;
    const bool has_gc = L.G > 0, has_lc = L.L > 0, use_bias = L.use_bias != 0;
    const ActCoef coef = act_coef(lane >= 32);
    const int ctl = c.o_ctrl;
    const int total = T * NL;
    int item = 0, slot = 0;
    // causal_queue (model.py:52): lane k holds the k-th oldest of the last ifw input samples
    float hv = SCALAR ? c.stb[L.st_hist + lane] : 0.0f;
    float x = 0.0f;

    // Operands of the layer about to run.  Single-buffered: each group is re-fetched from the NEXT item's LDS slot
    // right after its last use in the current layer, so the LDS latency hides under the rest of the layer.
    Tile w1;                           // tap-1 conv tile                (re-fetched after the conv)
    Tile wd;                           // dense half tile                (re-fetched after the dense conv)
    float pre, bd;                     // the conv's addend ((tap-0 chunk + bias) + gc) + lc (summed by a loader, AC-1b), dense bias
    auto fetch_conv = [&](int sb, int l) {
        lds_tile(w1, sb + SlotOff::T1, lane);
        pre = lds[sb + SlotOff::PK + lane];
        (void)l;
    };
    auto fetch_dense = [&](int sb) {
        lds_half_tile(wd, sb + SlotOff::WD, lane);
        bd = use_bias ? lds[sb + SlotOff::BD + (lane & 31)] : 0.0f;
    };

    __builtin_amdgcn_s_setprio(3);
    wait_seq(c.o_ready + 0, 1, ctl + C_ABORT, 199);     // operands of item 0
    ACQUIRE_WG();
    fetch_conv(c.o_slots, 0);
    fetch_dense(c.o_slots);

    // the first input sample is fetched once, outside the loop: a select between this global load and the LDS read of the later
    // samples is otherwise merged into ONE flat load through a selected pointer (vector-memory path) on every step
    const float first_in = (SCALAR && a.forced == nullptr) ? reinterpret_cast<const float*>(a.first_input)[b] : 0.0f;
    const bool full_causal = SCALAR && L.NCA == 1 && L.ifw == 32;     // one unmasked AC-1 chunk (hparams default)
    for (int t = 0; t < T; ++t) {
        const bool prof = INSTR && a.prof != nullptr && b == 0 && c.g == 0 && t < a.prof_steps;   // INSTR: instrumented build (phase stamps, dumps)
        unsigned long long* pp = a.prof + (long long)t * 80;
        // what does not depend on the new sample is done BEFORE waiting for it: the causal kernel tile, the queue shift
        Tile tc0;
        float sh = 0.0f;
        if (SCALAR) {
            if (full_causal) lds_half_tile(tc0, c.o_causal, lane);
            sh = __shfl_down(hv, 1);
        }
        if (t > 0 && a.forced == nullptr) { wait_seq(ctl + C_SSEQ, t, ctl + C_ABORT, 3); ACQUIRE_WG(); }   // sample t-1 published
        if (prof && lane == 0) { pp[0] = __builtin_amdgcn_s_memtime(); pp[7] = wall_clock64(); }
        if (SCALAR && full_causal) {
            // model.py:122 causal_queue shift+append; model.py:41-46 causal conv (k = ifw = 32, no bias): one AC-1 chunk
            const float s_lds = lds[ctl + C_SAMPLE];
            float s_in = (t == 0) ? first_in : s_lds;
            if (a.forced != nullptr) s_in = reinterpret_cast<const float*>(a.forced)[(long long)b * T + t];
            hv = (lane == 31) ? s_in : sh;
            x = dot_readlane_pipe(tc0, hv);
        } else if (SCALAR) {
            const float s_lds = lds[ctl + C_SAMPLE];
            float s_in = (t == 0) ? first_in : s_lds;
            if (a.forced != nullptr) s_in = reinterpret_cast<const float*>(a.forced)[(long long)b * T + t];
            hv = (lane == L.ifw - 1) ? s_in : sh;
            x = 0.0f;
            for (int ca = 0; ca < L.NCA; ++ca) {
                Tile tc;
                lds_half_tile(tc, c.o_causal + ca * 1024, lane);
                float sj[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int k0 = 0; k0 < 32; ++k0) {
                    const float hk = __shfl(hv, ca * 32 + k0);
                    const float nacc = fma_(tc.w[k0], hk, sj[k0 & 3]);
                    sj[k0 & 3] = (ca * 32 + k0 < L.ifw) ? nacc : sj[k0 & 3];
                }
                const float acc = (sj[0] + sj[1]) + (sj[2] + sj[3]);
                x = ca == 0 ? acc : x + acc;
            }
        } else {
            // one-hot input: the k=2 causal conv over one-hot rows is the sum of two kernel rows
            const int qcur = a.forced != nullptr ? reinterpret_cast<const int*>(a.forced)[(long long)b * T + t]
                             : ((t == 0) ? reinterpret_cast<const int*>(a.first_input)[b] : LDSI(ctl + C_SAMPLE));
            const float w1r = a.P[L.off_causal + ((long long)L.Q + qcur) * 32 + (lane & 31)];
            if (prev_valid) {
                const float w0r = a.P[L.off_causal + (long long)qprev * 32 + (lane & 31)];
                x = w0r + w1r;
            } else {
                x = w1r;
            }
            qprev = qcur; prev_valid = 1;
        }
        if (prof && lane == 0) pp[1] = __builtin_amdgcn_s_memtime();

        for (int l = 0; l < NL; ++l) {
            const bool fine = prof && l == 5 && lane == 0;
            if (fine) pp[72] = __builtin_amdgcn_s_memtime();
            // model.py:145 dilation queue <- the layer INPUT x[t].  The chain only drops it in LDS; a loader wave moves it
            // to the stream's delay line in HBM (under load every VMEM instruction issued here would stall the chain).
            if (lane < 32) lds[c.o_ring1 + l * 32 + lane] = x;
            // is the NEXT item staged?  (read early, consumed after the conv)
            const int slot_n = (slot + 1 == nslot) ? 0 : slot + 1;
            const int sbn = c.o_slots + slot_n * SlotOff::FLOATS;
            const int ln = (l + 1 < NL) ? l + 1 : 0;
            const bool have_next = item + 1 < total;
            const int rdy = have_next ? LDSVI(c.o_ready + slot_n) : 0x7fffffff;

            // model.py:68-69 conv_filter | conv_gate + model.py:71-83 conditioning: the tap-1 chunk, its chain 0 started from the addend
            // ((chunk(tap 0) + bias) + gc) + lc a loader has summed (AC-1b)
            const float v = dot_readlane_pipe_init(w1, x, pre);
            if (fine) pp[74] = __builtin_amdgcn_s_memtime();
            // conv operands of the NEXT item -> registers (latency hides under the gated unit and the dense conv)
            // (fetched unconditionally: guarding the fetch with `have_next` made the tile a two-way merge and cost 33 register
            //  moves per layer; after the very last item the slot's contents are simply never used)
            if (have_next && rdy < item + 2) wait_seq(c.o_ready + slot_n, item + 2, ctl + C_ABORT, 200 + l);
            ACQUIRE_WG();
            fetch_conv(sbn, ln);
            // model.py:86 tanh(filter) * sigmoid(gate): lanes 0-31 hold tanh, lanes 32-63 the logistic
            const float act = act_eval_pk(coef, v);
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(act), __float_as_uint(act), false, false);
            const float z = __uint_as_float(sw[0]) * __uint_as_float(sw[1]);   // every lane: z[lane & 31]
            if (lane < 32) lds[c.o_zbuf + l * 32 + lane] = z;
            publish(ctl + C_ZSEQ, item + 1, lane);
            if (fine) pp[75] = __builtin_amdgcn_s_memtime();

            // model.py:89 dense 1x1, model.py:98-101 residual
            const float tr = dot_readlane_pipe_init(wd, z, bd);      // the bias is the start value of chain 0 (AC-1b)
            x = x + tr;
            fetch_dense(sbn);
            if (INSTR && a.dbg != nullptr && c.g == 0 && t < a.dbg_steps) {
                float* dp = a.dbg + ((long long)b * a.dbg_steps + t) * ((long long)NL * 64 + L.Opad) + (long long)l * 64;
                if (lane < 32) { dp[lane] = z; dp[32 + lane] = x; }
            }
            if (fine) pp[76] = __builtin_amdgcn_s_memtime();
            if (prof && lane == 0) pp[8 + l] = __builtin_amdgcn_s_memtime();
            ++item;
            slot = slot_n;
        }
        if (prof && lane == 0) pp[2] = __builtin_amdgcn_s_memtime();
        if (LDSVI(ctl + C_ABORT)) break;
    }
    if (SCALAR) c.stb[L.st_hist + lane] = hv;
}

// =============================== LOADER WAVES ===============================
// Stage item i = (layer l, step t) into LDS slot i % nslot, NLD waves taking items round-robin:
//   the layer's chain block [T0|T1|WD|BFG|BD] as 21 one-KiB LDS-DMA pieces, x[t-d] from the stream's delay line in HBM
//   (the slot the chain wave will overwrite when it reaches this item), and the hoisted lc projection row.
__device__ __forceinline__ unsigned ring_pos(unsigned pos0, unsigned t, unsigned d)
{
    const unsigned v = pos0 + t;
    return (d & (d - 1)) == 0 ? (v & (d - 1)) : v % d;    // dilations are powers of two in every shipped config
}

template <int NLD, bool INSTR>
__device__ __forceinline__ void loader_main(const GenArgs& a, const Ctx& c, rsrc_t rs, int k)
{
    const Layout& L = a.lay;
    const int NL = L.NL, T = a.T, lane = c.lane, nslot = L.nslot;
    const bool has_lc = L.L > 0;
    const int ctl = c.o_ctrl;
    const int v16 = lane * 16;
    const int total = T * NL;          // < 2^31 (checked on the host)
    // all item bookkeeping is 32-bit and incremental: integer division is software on the GPU
    int nxt = k, nt = 0, nl = k, nslot_i = k % nslot;       // next item this wave stages = (layer nl, step nt), its slot
    while (nl >= NL) { nl -= NL; ++nt; }
    int wb = k - nslot, wt = 0, wl = k - nslot;             // item whose layer input goes back to HBM with it
    while (wl < 0) { wl += NL; --wt; }
    int pend = 0;                      // items in flight (at most 2: the 6-bit vmcnt holds 2 x 24 operations)
    int slotA = 0, itemA = 0, layA = 0, slotB = 0, itemB = 0, layB = 0;   // A = older, B = newer
    Tile t0A, t0B;                     // tap-0 conv tiles of the items in flight (never touch LDS)

    // delay-line write-back of item (lb, tb): x_lb[tb] was left in LDS by the chain wave
    auto write_back = [&](int lb, int tb) {
        const unsigned d = (unsigned)LDSI(c.o_meta + lb);
        if (d > 1) {
            const unsigned pos = ring_pos((unsigned)LDSI(c.o_pos0 + lb), (unsigned)tb, d);
            const float xv = lds[c.o_ring1 + lb * 32 + (lane & 31)];
            if (lane < 32) c.ring[LDSI(c.o_meta + 64 + lb) + pos * 32 + lane] = xv;
        }
    };
    auto advance = [&](int& item, int& l, int& t) {
        item += NLD; l += NLD;
        while (l >= NL) { l -= NL; ++t; }
    };

    for (;;) {
        // the slot's previous occupant, item nxt-nslot, was last read (its dense operands) BEFORE the chain wave published
        // ZSEQ = nxt-nslot+1 -- one wave's LDS operations are performed in order -- so that word also frees the slot
        const bool can_issue = nxt < total && pend < 2 && (nxt < nslot || LDSVI(ctl + C_ZSEQ) >= nxt - nslot + 1);
        if (can_issue) {
            const int t = nt, l = nl, slot = nslot_i;
            const int sb = c.o_slots + slot * SlotOff::FLOATS;
            const bool lprof = INSTR && a.prof != nullptr && c.b == 0 && c.g == 0 && k == 0 && l == 6 && t < a.prof_steps && lane == 0;
            if (lprof) a.prof[(long long)t * 80 + 40] = __builtin_amdgcn_s_memtime();
            // (1) ZSEQ >= nxt-nslot+1: the chain has run the conv of item nxt-nslot: its layer input sits in LDS -> HBM
            if (wb >= 0) write_back(wl, wt);
            // (2) tap-0 tile straight into registers, (3) the chain block, x[t-d] and the lc row by LDS-DMA
            const int lb4 = ((int)L.off_layer0 + l * (int)L.layer_stride) * 4;
            if (pend == 0) load_tile_b(t0A, rs, v16, lb4 + LayerOff::T0 * 4); else load_tile_b(t0B, rs, v16, lb4 + LayerOff::T0 * 4);
            const float* src = a.P + L.off_layer0 + (long long)l * L.layer_stride + LayerOff::T1 + lane * 4;
#pragma unroll
            for (int p = 0; p < SlotOff::PIECES; ++p)
                __builtin_amdgcn_global_load_lds((gptr_t)(src + p * 256), (lptr_t)(lds + sb + p * 256), 16, 0, 0);
            {   // x[t-d]: the delay line's write position at step t (it advances by one per step)
                const unsigned d = (unsigned)LDSI(c.o_meta + l);
                const unsigned pos = ring_pos((unsigned)LDSI(c.o_pos0 + l), (unsigned)t, d);
                const float* xs = c.ring + LDSI(c.o_meta + 64 + l) + pos * 32 + (lane & 31);
                __builtin_amdgcn_global_load_lds((gptr_t)xs, (lptr_t)(lds + sb + SlotOff::XO), 4, 0, 0);
            }
            {
                // lc frame used at step t = frame PUSHED at step t-1 (model.py:79-80: slice from the FRONT of the queue);
                // without local conditioning the piece is still issued (from the zeroed lcprev area) to keep counts fixed
                const float* row = (t == 0 || !has_lc) ? (c.stb + L.st_lcprev) : (c.LCb + (long long)(t - 1) * NL * 64);
                __builtin_amdgcn_global_load_lds((gptr_t)(row + l * 64 + lane), (lptr_t)(lds + sb + SlotOff::LC), 4, 0, 0);
            }
            if (lprof) a.prof[(long long)t * 80 + 41] = __builtin_amdgcn_s_memtime();
            if (pend == 0) { slotA = slot; itemA = nxt; layA = l; } else { slotB = slot; itemB = nxt; layB = l; }
            ++pend;
            advance(nxt, nl, nt);
            advance(wb, wl, wt);
            nslot_i += NLD; while (nslot_i >= nslot) nslot_i -= nslot;
            continue;
        }
        if (pend > 0) {
            // the OLDER item has landed once at most the newer item's 8 + 15 operations are outstanding
            // (an interleaved write-back store is older than those, so it is covered too)
            if (pend == 2) asm volatile("s_waitcnt vmcnt(23)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int tA = itemA / NL;
            const bool aprof = a.prof != nullptr && c.b == 0 && c.g == 0 && k == 0 && layA == 6 && tA < a.prof_steps && lane == 0;
            if (aprof) a.prof[(long long)tA * 80 + 42] = __builtin_amdgcn_s_memtime();
            {
                // tap-0 chunk of conv_filter|conv_gate (model.py:68-69): depends on x[t-d] only, so it is taken off the
                // chain wave's critical path: acc0[lane] = sum_c T0[c][lane] * x[t-d][c]   (one AC-1 chunk)
                const int sbA = c.o_slots + slotA * SlotOff::FLOATS;
                const int xsrc = LDSI(c.o_meta + layA) == 1 ? c.o_ring1 + layA * 32 : sbA + SlotOff::XO;
                // AC-1b: the addend the chain starts its tap-1 chunk from, in the reference's statement order ((conv + bias) + gc) + lc
                float ad = dot_ldso(t0A, xsrc);
                if (L.use_bias) ad = ad + lds[sbA + SlotOff::BFG + lane];
                if (L.G > 0) ad = ad + lds[c.o_gc + layA * 64 + lane];      // model.py:71-73
                if (has_lc) ad = ad + lds[sbA + SlotOff::LC + lane];        // model.py:75-83
                lds[sbA + SlotOff::PK + lane] = ad;
            }
            publish(c.o_ready + slotA, itemA + 1, lane);
            if (aprof) a.prof[(long long)tA * 80 + 43] = __builtin_amdgcn_s_memtime();
            slotA = slotB; itemA = itemB; layA = layB; t0A = t0B;
            --pend;
            continue;
        }
        if (nxt >= total) break;
        if (!wait_seq(ctl + C_ZSEQ, nxt - nslot + 1, ctl + C_ABORT, 300)) break;
    }
    // write-backs of the last nslot items: wait until the chain has run them
    for (; wb < total; advance(wb, wl, wt)) {
        if (wb < 0) continue;
        if (!wait_seq(ctl + C_ZSEQ, wb + 1, ctl + C_ABORT, 301)) break;
        write_back(wl, wt);
    }
}

// =============================== WORKER WAVES ===============================
// model.py:94-96 skip 1x1 convs and their sum, model.py:150-165 postprocessing, mixture.py:84-114 sampling.
// Workgroup g of a stream owns the output blocks jb with jb % G == g of the skip sum and of conv1d_1 (local index
// m = jb / G); the two 512-vectors in between are all-gathered across the G workgroups; conv1d_2 and the sampler run
// redundantly in every workgroup (identical bits), so each of them feeds its own chain wave without another hop.
template <int W, int NTW, bool SCALAR, bool SPLIT1, bool INSTR, int HELP>
__device__ __forceinline__ void worker_main(const GenArgs& a, const Ctx& c, rsrc_t rs, int w)
{
    const Layout& L = a.lay;
    const int NL = L.NL, T = a.T, NSJ = L.NSJ, NCH = L.NCH, S = L.S, lane = c.lane, b = c.b, G = a.G, g = c.g;
    if (a.forced != nullptr) return;                // priming only advances the delay lines (chain + loader waves)
    const int NSJL = NSJ / G;                       // output blocks owned by this workgroup
    constexpr bool split1 = SPLIT1;                 // == (NSJL < W), a launch-time fact made a template parameter: only one of the two
                                                    // post-phase shapes is compiled into a kernel (the other one's registers spilled)
    const int v16 = lane * 16, v4 = lane * 4;
    const bool use_bias = L.use_bias != 0;
    const int ctl = c.o_ctrl;
    const int lay0 = (int)L.off_layer0 * 4, lstride = (int)L.layer_stride * 4;   // bytes
    const int skb = LayerOff::SK * 4, bsb = (LayerOff::SK + NSJ * kTile) * 4;
    unsigned long long* X1 = a.exch + ((long long)b * 2 + 0) * S;
    unsigned long long* X2 = a.exch + ((long long)b * 2 + 1) * S;
    Tile tk[NTW];
    float n_bs[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        const int m = w + n * W, jb = m * G + g;
        n_bs[n] = 0.0f;
        if (!split1 && m < NSJL) {
            load_tile_b(tk[n], rs, v16, lay0 + skb + jb * kTile * 4);
            if (use_bias) n_bs[n] = load_f32_b(rs, v4, lay0 + bsb + jb * 256);
        }
    }
    if (split1 && w < NL) {
        load_tile_b(tk[0], rs, v16, lay0 + w * lstride + skb + g * kTile * 4);
        if (use_bias) n_bs[0] = load_f32_b(rs, v4, lay0 + w * lstride + bsb + g * 256);
    }

    for (int t = 0; t < T; ++t) {
        const bool wprof = INSTR && a.prof != nullptr && b == 0 && g == 0 && w == 0 && t < a.prof_steps && lane == 0;
        unsigned long long* wp = a.prof + (long long)t * 80;
        if (wprof) wp[44] = __builtin_amdgcn_s_memtime();
        // the sampler's noise terms depend only on the injected uniforms: evaluated here, a whole residual stack ahead of their use
        // (mixture.py:103 -log(-log u) per mixture lane; mixture.py:110-111 log u - log(1 - u) of the last draw)
        float s_lnl = 0.0f, s_tq = 0.0f, b2_pre = 0.0f;
        if (SCALAR && w == 0) {
            const float* up = reinterpret_cast<const float*>(a.uniforms) + ((long long)b * T + t) * (L.nr_mix + 1);
            const float u = lane <= L.nr_mix ? up[lane] : 0.5f;
            s_lnl = log_e(-log_e(u));
            const float uu = __shfl(u, L.nr_mix);
            s_tq = log_e(uu) - log_e(1.0f - uu);
            if (use_bias && lane < L.O) b2_pre = a.P[L.off_b2 + lane];
        }
        {
            float tot[NTW];
#pragma unroll
            for (int n = 0; n < NTW; ++n) tot[n] = 0.0f;
            if (!split1) {
                for (int l = 0; l < NL; ++l) {
                    wait_seq(ctl + C_ZSEQ, t * NL + l + 1, ctl + C_ABORT, 100 + l);
                    ACQUIRE_WG();
                    float zz[32];
#pragma unroll
                    for (int kq = 0; kq < 8; ++kq) {
                        const f32x4 q = LDS4(((c.o_zbuf + l * 32) >> 2) + kq);
                        zz[4 * kq + 0] = q.x; zz[4 * kq + 1] = q.y; zz[4 * kq + 2] = q.z; zz[4 * kq + 3] = q.w;
                    }
                    const int ln = (l + 1 < NL) ? l + 1 : 0;
                    const int lnb = lay0 + ln * lstride;
#pragma unroll
                    for (int n = 0; n < NTW; ++n) {
                        const int m = w + n * W, jb = m * G + g;
                        if (m < NSJL) {
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
                    const int m = w + n * W, jb = m * G + g;
                    if (m < NSJL) {
                        const float h = tot[n] > 0.0f ? tot[n] : 0.0f;       // model.py:157 relu
                        if (G == 1) lds[c.o_h1 + jb * 64 + lane] = h;
                        else granule_store(X1 + jb * 64 + lane, 2u * (unsigned)t + 1u, h);
                    }
                }
            } else {
                // few output blocks per workgroup: worker w takes layers l = w, w+W, ... (W-fold latency hiding for the tile
                // fetch), leaves each layer's skip value in LDS and advances its progress word; the owner of block m (worker m)
                // adds the values up IN LAYER ORDER as they appear, so after the last layer only one value is left to add
                int nextl = 0;
                float tsum = 0.0f;
                auto drain = [&](int upto) {                         // owner: layers nextl..upto of block w join the sum
                    for (; nextl <= upto; ++nextl) {
                        const int ow = nextl % W;
                        if (ow != w) wait_seq(ctl + C_SKP + ow, t * NL + nextl + 1, ctl + C_ABORT, 10);
                        ACQUIRE_WG();
                        const float v = lds[c.o_skl + (w * NL + nextl) * 64 + lane];
                        tsum = (nextl == 0) ? v : tsum + v;          // model.py:154 sum(outputs), in layer order
                    }
                };
                for (int l = w; l < NL; l += W) {
                    wait_seq(ctl + C_ZSEQ, t * NL + l + 1, ctl + C_ABORT, 100 + l);
                    ACQUIRE_WG();
                    float zz[32];
#pragma unroll
                    for (int kq = 0; kq < 8; ++kq) {
                        const f32x4 q = LDS4(((c.o_zbuf + l * 32) >> 2) + kq);
                        zz[4 * kq + 0] = q.x; zz[4 * kq + 1] = q.y; zz[4 * kq + 2] = q.z; zz[4 * kq + 3] = q.w;
                    }
                    int ln = l + W;
                    if (ln >= NL) ln = w < NL ? w : 0;               // first layer of this worker in the next step
                    const int lnb = lay0 + ln * lstride;
                    for (int m = 0; m < NSJL; ++m) {
                        const int jb = m * G + g;
                        if (m > 0) {                                 // (rare) more than one owned block: fetch on the spot
                            load_tile_b(tk[0], rs, v16, lay0 + l * lstride + skb + jb * kTile * 4);
                            if (use_bias) n_bs[0] = load_f32_b(rs, v4, lay0 + l * lstride + bsb + jb * 256);
                        }
                        float v = dot_regs(tk[0], zz);               // model.py:96 skip 1x1
                        __builtin_amdgcn_sched_barrier(0);
                        if (use_bias) v = v + n_bs[0];
                        lds[c.o_skl + (m * NL + l) * 64 + lane] = v;
                    }
                    publish(ctl + C_SKP + w, t * NL + l + 1, lane);
                    load_tile_b(tk[0], rs, v16, lnb + skb + g * kTile * 4);   // block m = 0 of this worker's next layer
                    if (use_bias) n_bs[0] = load_f32_b(rs, v4, lnb + bsb + g * 256);
                    if (w < NSJL) drain(l);
                }
                if (w < NSJL) {
                    drain(NL - 1);
                    const float h = tsum > 0.0f ? tsum : 0.0f;        // model.py:157 relu
                    const int jb = w * G + g;
                    if (G == 1) lds[c.o_h1 + jb * 64 + lane] = h;
                    else granule_store(X1 + jb * 64 + lane, 2u * (unsigned)t + 1u, h);
                }
            }
        }
        if (wprof) wp[45] = __builtin_amdgcn_s_memtime();
        // post-phase weights do not depend on data: the first two tiles of this worker's conv1d_1 share are requested before the
        // all-gather of h1 (and, further down, conv1d_2's before the all-gather of h2)
        Tile qa, qb;
        const int n1 = NSJL * NCH;
        auto w1_off = [&](int idx) -> int { const int m = idx / NCH, ch = idx - m * NCH; return ((int)L.off_w1 + ((m * G + g) * NCH + ch) * kTile) * 4; };
        // HELP: conv1d_1 of this slice runs in the stream's helper workgroup (helper_main), straight from the h1 granules to the h2
        // granules; this workgroup goes on to wait for h2
        if (!HELP) {
        if (split1) {
            if (w < n1) load_tile_b(qa, rs, v16, w1_off(w));
            if (w + W < n1) load_tile_b(qb, rs, v16, w1_off(w + W));
        }
        if (G > 1) gather_granules<W>(X1, S, 2u * (unsigned)t + 1u, c.o_h1, w, lane, ctl + C_ABORT, 7);
        arrive(ctl + C_H1CNT, lane);
        wait_seq(ctl + C_H1CNT, W * (t + 1), ctl + C_ABORT, 4);   // h1 complete in this workgroup's LDS
        ACQUIRE_WG();
        if (wprof) wp[46] = __builtin_amdgcn_s_memtime();
        // ---- model.py:158-160 conv1d_1 (S->S) + relu for the owned output blocks
        if (!split1) {
            // a worker owns whole output blocks m = w + n*W and walks their chunks in order
            int nown = 0;
#pragma unroll
            for (int n = 0; n < NTW; ++n) if (w + n * W < NSJL) nown = n + 1;
            const int ntiles = nown * NCH;
            Tile ta, tb;
            auto tile_off = [&](int i) -> int {
                const int n = i / NCH, ch = i - n * NCH;
                return ((int)L.off_w1 + (((w + n * W) * G + g) * NCH + ch) * kTile) * 4;
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
                    const int jb = (w + n * W) * G + g;
                    if (use_bias) r = r + load_f32_b(rs, v4, ((int)L.off_b1 + jb * 64) * 4);
                    const float h = r > 0.0f ? r : 0.0f;
                    if (G == 1) lds[c.o_h2 + jb * 64 + lane] = h;
                    else granule_store(X2 + jb * 64 + lane, 2u * (unsigned)t + 2u, h);
                }
            }
        } else {
            // few output blocks: tiles (m, ch) round-robin over the workers, chunk values summed in order afterwards
            for (int idx = w; idx < n1; idx += 2 * W) {
                {
                    const float r = dot_ldso(qa, c.o_h1 + (idx % NCH) * 32);
                    __builtin_amdgcn_sched_barrier(0);
                    lds[c.o_cpart1 + idx * 64 + lane] = r;
                    if (idx + 2 * W < n1) load_tile_b(qa, rs, v16, w1_off(idx + 2 * W));
                }
                if (idx + W < n1) {
                    const float r = dot_ldso(qb, c.o_h1 + ((idx + W) % NCH) * 32);
                    __builtin_amdgcn_sched_barrier(0);
                    lds[c.o_cpart1 + (idx + W) * 64 + lane] = r;
                    if (idx + 3 * W < n1) load_tile_b(qb, rs, v16, w1_off(idx + 3 * W));
                }
            }
            arrive(ctl + C_P1CNT, lane);
            if (w < NSJL) {
                const int jb = w * G + g;
                const float b1v = use_bias ? load_f32_b(rs, v4, ((int)L.off_b1 + jb * 64) * 4) : 0.0f;
                wait_seq(ctl + C_P1CNT, W * (t + 1), ctl + C_ABORT, 8);
                ACQUIRE_WG();
                float r = 0.0f;
                for (int ch = 0; ch < NCH; ++ch) {
                    const float cp = lds[c.o_cpart1 + (w * NCH + ch) * 64 + lane];
                    r = (ch == 0) ? cp : r + cp;
                }
                if (use_bias) r = r + b1v;
                const float h = r > 0.0f ? r : 0.0f;
                if (G == 1) lds[c.o_h2 + jb * 64 + lane] = h;
                else granule_store(X2 + jb * 64 + lane, 2u * (unsigned)t + 2u, h);
            }
        }
        }   // !HELP
        if (wprof && HELP) wp[46] = wp[45];
        if (wprof) wp[47] = HELP ? wp[45] : __builtin_amdgcn_s_memtime();
        // prefetches that do not depend on h2 (the sampler's inputs)
        const int n2 = L.NOJ * NCH;
        // HELP == 2: the helper workgroups go on from their h2 slice to conv1d_2's chunk partials of that slice (chunks 2g, 2g+1,
        // 32 output lanes each): what is all-gathered here is the [NCH][32] partial table, and conv1d_2 is not computed at all
        if (split1 && HELP != 2) {
            if (w < n2) load_tile_b(qa, rs, v16, ((int)L.off_w2 + w * kTile) * 4);
            if (w + W < n2) load_tile_b(qb, rs, v16, ((int)L.off_w2 + (w + W) * kTile) * 4);
        }
        float y_help = 0.0f;
        if (HELP == 2) {
            // the sampler wave alone collects the [NCH][32] partial table straight into registers -- lanes 0-31 the first half of
            // the chunks of output (lane & 31), lanes 32-63 the second half -- and adds them up in chunk order (the upper half
            // crosses over with v_permlane32_swap): no LDS staging, no hand-off between the workers
            if (w == 0) {
                const int H = (NCH + 1) >> 1, half = lane >> 5;
                const unsigned epoch = 2u * (unsigned)t + 2u;
                unsigned long long q[8];
                bool ok = false;
#pragma nounroll
                for (int it = 0; it < (1 << 20); ++it) {
                    bool good = true;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int ch = half * H + k;
                        q[k] = ((unsigned long long)epoch) << 32;
                        if (k < H && ch < NCH) q[k] = __hip_atomic_load((gu64*)(X2 + ch * 32 + (lane & 31)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) good = good && ((unsigned)(q[k] >> 32) == epoch);
                    ok = __all(good);
                    if (ok) break;
                    if ((it & 15) == 15 && LDSVI(ctl + C_ABORT)) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!ok && !LDSVI(ctl + C_ABORT)) LDSVI(ctl + C_ABORT) = 9;
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (k < H) { const float v = __uint_as_float((unsigned)q[k]); acc = (k == 0) ? v : acc + v; }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)q[k], (unsigned)q[k], false, false);
                    if (H + k < NCH) acc = acc + __uint_as_float(sw[1]);     // the upper half-wave's chunk H+k of the same output
                }
                y_help = acc;
            }
        } else {
            if (G > 1) gather_granules<W>(X2, S, 2u * (unsigned)t + 2u, c.o_h2, w, lane, ctl + C_ABORT, 9);
            arrive(ctl + C_H2CNT, lane);
            wait_seq(ctl + C_H2CNT, W * (t + 1), ctl + C_ABORT, 5);   // h2 complete
            ACQUIRE_WG();
        }
        if (wprof) wp[48] = __builtin_amdgcn_s_memtime();
        if (HELP != 2) {
            // ---- model.py:161-165 conv1d_2 (S->O): chunk partials, summed in order by the sampler wave
            if (split1) {
                for (int idx = w; idx < n2; idx += 2 * W) {
                    {
                        const float r = dot_ldso(qa, c.o_h2 + (idx % NCH) * 32);
                        __builtin_amdgcn_sched_barrier(0);
                        lds[c.o_cpart + idx * 64 + lane] = r;
                        if (idx + 2 * W < n2) load_tile_b(qa, rs, v16, ((int)L.off_w2 + (idx + 2 * W) * kTile) * 4);
                    }
                    if (idx + W < n2) {
                        const float r = dot_ldso(qb, c.o_h2 + ((idx + W) % NCH) * 32);
                        __builtin_amdgcn_sched_barrier(0);
                        lds[c.o_cpart + (idx + W) * 64 + lane] = r;
                        if (idx + 3 * W < n2) load_tile_b(qb, rs, v16, ((int)L.off_w2 + (idx + 3 * W) * kTile) * 4);
                    }
                }
            } else {
                for (int idx = w; idx < n2; idx += W) {
                    Tile tq;
                    load_tile_b(tq, rs, v16, ((int)L.off_w2 + idx * kTile) * 4);
                    lds[c.o_cpart + idx * 64 + lane] = dot_ldso(tq, c.o_h2 + (idx % NCH) * 32);
                }
            }
        }
        if (wprof) wp[49] = __builtin_amdgcn_s_memtime();
        if (HELP != 2) arrive(ctl + C_CPCNT, lane);
        if (!SCALAR) {
            // one-hot: every worker assembles the logits of its output blocks (chunk partials in order, then bias)
            wait_seq(ctl + C_CPCNT, W * (t + 1), ctl + C_ABORT, 6);
            ACQUIRE_WG();
            for (int ob = w; ob < L.NOJ; ob += W) {
                float y = 0.0f;
                for (int ch = 0; ch < NCH; ++ch) {
                    const float cp = lds[c.o_cpart + (ob * NCH + ch) * 64 + lane];
                    y = (ch == 0) ? cp : y + cp;
                }
                if (use_bias && ob * 64 + lane < L.O) y = y + a.P[L.off_b2 + ob * 64 + lane];
                lds[c.o_cat + ob * 64 + lane] = y;
            }
            arrive(ctl + C_LGCNT, lane);
        }
        if (w == 0) {
            if (HELP != 2) wait_seq(ctl + C_CPCNT, W * (t + 1), ctl + C_ABORT, 6);   // chunk partials complete
            ACQUIRE_WG();
            if (wprof) wp[50] = __builtin_amdgcn_s_memtime();
            if (SCALAR) {
                // raw network output y[lane] (lane < O), then mixture.py:84-114
                float y = y_help;
                if (HELP != 2) {
                    for (int ch = 0; ch < NCH; ++ch) {
                        const float cp = lds[c.o_cpart + ch * 64 + lane];
                        y = (ch == 0) ? cp : y + cp;
                    }
                }
                if (use_bias && lane < L.O) y = y + b2_pre;
                if (INSTR && a.dbg != nullptr && g == 0 && t < a.dbg_steps)
                    a.dbg[((long long)b * a.dbg_steps + t) * ((long long)NL * 64 + L.Opad) + (long long)NL * 64 + lane] = y;
                const int nr = L.nr_mix;
                const float gmb = y - s_lnl;                                 // mixture.py:103 (lanes < nr)
                // argmax over the nr mixture lanes with v_readlane (uniform values, no LDS crossbar round trips)
                int k = 0;
                float best = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gmb), 0));
                for (int i = 1; i < nr; ++i) {
                    const float gi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gmb), i));
                    if (gi > best) { best = gi; k = i; }
                }
                k = __builtin_amdgcn_readfirstlane(k);
                const float mean = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), nr + k));       // mixture.py:105
                float ls = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), 2 * nr + k));            // mixture.py:107
                const float lsmin = (float)-32.23619130191664;
                ls = ls > lsmin ? ls : lsmin;
                const float tq = s_tq;                                       // mixture.py:110-111
                const float e = exp_e(ls);
                const float prod = e * tq;
                float xs = mean + prod;
                xs = xs > -1.0f ? xs : -1.0f;                                // mixture.py:113
                xs = xs < 1.0f ? xs : 1.0f;
                if (lane == 0) {
                    if (g == 0) reinterpret_cast<float*>(a.out)[(long long)b * T + t] = xs;
                    lds[ctl + C_SAMPLE] = xs;
                }
            }
            if (!SCALAR) {
                // ---- one-hot output: model.py:243 float64 softmax -> float32, generate.py:219-222 temperature rescale,
                // generate.py:231 legacy np.random.choice = searchsorted(cumsum(p)/last, u, 'right'): twv_categorical.hpp (AC-5: one wave,
                // nothing sequential over the classes; round 3's left-to-right logaddexp chain was 32 us of a 73 us step).
                // logits were assembled (chunk partials in order + bias) by the workers into lds[o_cat .. o_cat+Q)
                wait_seq(ctl + C_LGCNT, W * (t + 1), ctl + C_ABORT, 11);
                ACQUIRE_WG();
                if (wprof) wp[58] = __builtin_amdgcn_s_memtime();                  // one-hot sampler phases: logits complete
                const int Q = L.Q;
                const int o_lg = c.o_cat;
                float xv[16];                                                        // lane owns classes i = lane + 64k
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int i = lane + 64 * k;
                    xv[k] = (i < Q) ? lds[o_lg + i] : 0.0f;
                }
                if (INSTR && a.dbg != nullptr && g == 0 && t < a.dbg_steps) {
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        if (lane + 64 * k < Q)
                            a.dbg[((long long)b * a.dbg_steps + t) * ((long long)NL * 64 + L.Opad) + (long long)NL * 64 + lane + 64 * k] = xv[k];
                }
                const double uu = reinterpret_cast<const double*>(a.uniforms)[(long long)b * T + t];
                bool bad = false;
                const int idx = categorical_sample<16>(xv, Q, lane, a.temperature, uu, nullptr, nullptr, &bad);
                if (bad && lane == 0) atomicMax(a.status, 31);                     // NaN probabilities: np.random.choice would raise (generate.py:231)
                if (wprof) wp[63] = __builtin_amdgcn_s_memtime();                  // drawn
                if (lane == 0) {
                    if (g == 0) reinterpret_cast<int*>(a.out)[(long long)b * T + t] = idx;
                    LDSI(ctl + C_SAMPLE) = idx;
                }
            }
            publish(ctl + C_SSEQ, t + 1, lane);   // next input sample published
            if (wprof) wp[51] = __builtin_amdgcn_s_memtime();
        }
        if (LDSVI(ctl + C_ABORT)) break;
    }
}

// =============================== HELPER WORKGROUP (one per stream slice, optional) ===============================
// model.py:158-160 conv1d_1 + relu for the output blocks of slice g, off the stream workgroups: the 8 waves keep the slice's
// chunk tiles RESIDENT in registers for the whole launch (2 tiles = 64 VGPRs per wave; the stream workgroups have neither the
// registers nor the LDS for that and re-stream the 128 KB every step), poll the h1 granules of exactly their own chunks
// (one granule per lane), and the block owner publishes the h2 granules.  Same arithmetic, same order (AC-1 chunk partials
// added in chunk order, then bias): the h2 bits do not depend on who computes them.
template <int OFF>
__device__ __forceinline__ float dot_readlane_off(const Tile& t, float xv)
{
    f32x2p s01 = {0.0f, 0.0f}, s23 = {0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < 32; c += 4) {
        const f32x2p x01 = {__int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), OFF + c + 0)),
                            __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), OFF + c + 1))};
        const f32x2p x23 = {__int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), OFF + c + 2)),
                            __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), OFF + c + 3))};
        s01 = pk_fma(f32x2p{t.w[c + 0], t.w[c + 1]}, x01, s01);
        s23 = pk_fma(f32x2p{t.w[c + 2], t.w[c + 3]}, x23, s23);
    }
    return (s01[0] + s01[1]) + (s23[0] + s23[1]);
}
constexpr int kHelperTiles = 16;      // 8 waves x 2 resident tiles
// CONV2: the owner wave also keeps conv1d_2's two chunk tiles of its h2 slice (chunks 2g, 2g+1; needs one output block per
// slice and O <= 32) and publishes those chunk partials instead of h2 (model.py:161-165; the bias joins after the ordered sum)
template <bool CONV2, bool INSTR>
__device__ __forceinline__ void helper_main(const GenArgs& a, int b, int g)
{
    const Layout& L = a.lay;
    const int NCH = L.NCH, S = L.S, G = a.G, T = a.T;
    const int NSJL = L.NSJ / G, n1 = NSJL * NCH;
    const int v = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const bool use_bias = L.use_bias != 0;
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.P), 0, (int)(L.packed_floats * 4), 0x00020000);
    unsigned long long* X1 = a.exch + ((long long)b * 2 + 0) * S;
    unsigned long long* X2 = a.exch + ((long long)b * 2 + 1) * S;
    const int o_part = 0, o_cnt = kHelperTiles * 64, o_abort = o_cnt + 1;      // LDS: chunk partials | arrival counter | abort code
    if (threadIdx.x == 0) { LDSI(o_cnt) = 0; LDSI(o_abort) = 0; }
    __syncthreads();
    const int i0 = 2 * v, i1 = 2 * v + 1;
    const bool has0 = i0 < n1, has1 = i1 < n1;
    if (!has0) return;                                                          // fewer tiles than waves: the rest has nothing to do
    const int nact = (n1 + 1) / 2;
    auto w1_off = [&](int idx) -> int { const int m = idx / NCH, ch = idx - m * NCH; return ((int)L.off_w1 + ((m * G + g) * NCH + ch) * kTile) * 4; };
    Tile ta, tb;
    load_tile_b(ta, rs, lane * 16, w1_off(i0));
    load_tile_b(tb, rs, lane * 16, w1_off(has1 ? i1 : i0));
    const int ch0 = i0 % NCH, ch1 = (has1 ? i1 : i0) % NCH;
    const int gi = lane < 32 ? ch0 * 32 + lane : ch1 * 32 + lane - 32;         // the h1 element this lane fetches
    // CONV2 (one output block per slice): waves 0 and 1 BOTH add up the chunk partials (same bits), so that each of them has the
    // h2 slice in its lanes and they take one conv1d_2 chunk each, side by side; with a single active wave, wave 0 takes both
    const bool two = CONV2 && nact >= 2;
    const bool summer = CONV2 ? (v == 0 || (two && v == 1)) : (v < NSJL);
    const int mb = CONV2 ? 0 : v;                                               // the output block this wave sums
    float b1v = 0.0f;
    if (summer && use_bias) b1v = load_f32_b(rs, lane * 4, ((int)L.off_b1 + (mb * G + g) * 64) * 4);
    Tile t2a, t2b;
    if (CONV2 && summer) {
        load_tile_b(t2a, rs, lane * 16, ((int)L.off_w2 + (2 * g + (v == 1 ? 1 : 0)) * kTile) * 4);
        if (!two) load_tile_b(t2b, rs, lane * 16, ((int)L.off_w2 + (2 * g + 1) * kTile) * 4);
    }
    // h1 arrives once per generation step: after each arrival the wave sleeps through most of the measured step period before it
    // polls again (64 helper workgroups x 8 waves polling flat out are terabytes per second of agent-scope loads on the fabric)
    unsigned long long t_arr = 0, period = 0;
    for (int t = 0; t < T; ++t) {
        const unsigned epoch = 2u * (unsigned)t + 1u;
        unsigned long long q = 0;
        bool ok = false;
        if (period) {
            const unsigned long long target = t_arr + period - (period >> 2);
#pragma nounroll
            for (int it = 0; it < 4096 && __builtin_amdgcn_s_memtime() < target; ++it) __builtin_amdgcn_s_sleep(16);
        }
#pragma nounroll
        for (int it = 0; it < (1 << 22); ++it) {
            q = __hip_atomic_load((gu64*)(X1 + gi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = __all((unsigned)(q >> 32) == epoch);
            if (ok) break;
            if ((it & 15) == 15 && LDSVI(o_abort)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) { if (!LDSVI(o_abort)) LDSVI(o_abort) = 12; break; }
        {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            const unsigned long long d = now - t_arr;
            period = (t_arr != 0 && d < (1ull << 20)) ? d : 0;                // no estimate after the first step or a long gap
            t_arr = now;
        }
        const bool hprof = INSTR && a.prof != nullptr && b == 0 && g == 0 && v == 0 && t < a.prof_steps && lane == 0;
        unsigned long long* hp = a.prof + (long long)t * 80;
        if (hprof) hp[52] = __builtin_amdgcn_s_memtime();      // h1 seen
        const float x = __uint_as_float((unsigned)q);
        lds[o_part + i0 * 64 + lane] = dot_readlane_pipe(ta, x);
        if (has1) lds[o_part + i1 * 64 + lane] = dot_readlane_pipe32(tb, x);
        arrive(o_cnt, lane);
        if (hprof) hp[53] = __builtin_amdgcn_s_memtime();      // own partials stored
        if (summer) {                                                           // owner of output block mb
            if (!wait_seq(o_cnt, nact * (t + 1), o_abort, 13)) break;
            ACQUIRE_WG();
            if (hprof) hp[54] = __builtin_amdgcn_s_memtime();  // all partials in LDS
            float r = 0.0f;
            if (NCH == 16) {
                // all sixteen loads in flight before the first add (the compiler's own schedule waits for every pair of them)
                float cp[16];
#pragma unroll
                for (int ch = 0; ch < 16; ++ch) cp[ch] = lds[o_part + (mb * 16 + ch) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
                r = cp[0];
#pragma unroll
                for (int ch = 1; ch < 16; ++ch) r = r + cp[ch];
            } else {
                for (int ch = 0; ch < NCH; ++ch) {
                    const float cp = lds[o_part + (mb * NCH + ch) * 64 + lane];
                    r = (ch == 0) ? cp : r + cp;
                }
            }
            if (use_bias) r = r + b1v;
            const float h = r > 0.0f ? r : 0.0f;
            if (hprof) hp[55] = __builtin_amdgcn_s_memtime();  // h2 slice
            if (CONV2) {
                const unsigned ep2 = 2u * (unsigned)t + 2u;
                if (two) {
                    const float p = (v == 0) ? dot_readlane_pipe(t2a, h) : dot_readlane_pipe32(t2a, h);
                    if (lane < 32) granule_store(X2 + (2 * g + v) * 32 + lane, ep2, p);
                } else {
                    const float p0 = dot_readlane_pipe(t2a, h), p1 = dot_readlane_pipe32(t2b, h);
                    if (lane < 32) {
                        granule_store(X2 + (2 * g) * 32 + lane, ep2, p0);
                        if (2 * g + 1 < NCH) granule_store(X2 + (2 * g + 1) * 32 + lane, ep2, p1);
                    }
                }
            } else {
                granule_store(X2 + (mb * G + g) * 64 + lane, 2u * (unsigned)t + 2u, h);
            }
            if (hprof) hp[56] = __builtin_amdgcn_s_memtime();  // published
        }
    }
    if (lane == 0 && LDSVI(o_abort)) atomicMax(a.status, LDSI(o_abort));
}

constexpr int kLoaders = 3;

// D = 1 parks an idle wave at index 1 + kLoaders: waves i and i+4 of a workgroup share a SIMD (scripts/ubench/simd_map.hip), so
// with 3 loaders the chain wave (wave 0) then has its SIMD to itself.
template <int W, int NTW, bool SCALAR, int D = 0, bool SPLIT1 = false, bool INSTR = false, int HELP = 0>
__global__ void __launch_bounds__((1 + kLoaders + D + W) * 64) wn_generate_kernel(GenArgs a)
{
    const Layout& L = a.lay;
    if (HELP && (int)blockIdx.x >= a.B * a.G) {      // the second half of the grid: one helper workgroup per (stream, slice)
        const int hb = (int)blockIdx.x - a.B * a.G;
        helper_main<HELP == 2, INSTR>(a, hb / a.G, hb % a.G);
        return;
    }
    const int NL = L.NL, S = L.S, NCH = L.NCH;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int T = a.T;
    Ctx c;
    c.lane = threadIdx.x & 63;
    c.g = blockIdx.x % a.G;      // blocks of one XCD (blockIdx % 8) share g when G == 8: its L2 keeps only that 1/G slice
    c.b = blockIdx.x / a.G;
    c.o_zbuf = 0;                                   // [64][32]   gated outputs z_l of the current step
    c.o_h1 = c.o_zbuf + 64 * 32;                    // [S]        relu(sum of skips)
    c.o_h2 = c.o_h1 + S;                            // [S]        relu(post conv 1)
    c.o_cpart = c.o_h2 + S;                         // [NOJ][NCH][64] chunk partials of the last conv
    c.o_meta = c.o_cpart + L.NOJ * NCH * 64;        // [128]      dil[64] | ring_off[64]
    c.o_ringpos = c.o_meta + 128;                   // [64]       write position of every delay line (chain wave)
    c.o_pos0 = c.o_ringpos + 64;                    // [64]       the same at launch (loader waves, read-only)
    c.o_ready = c.o_pos0 + 64;                      // [16]       per slot: item number staged + 1
    c.o_ctrl = c.o_ready + 16;                      // [16]
    c.o_gc = c.o_ctrl + 16;                         // [NL][64]   gc projections (model.py:71-73), launch-resident
    c.o_ring1 = c.o_gc + NL * 64;                   // [NL][32]   delay lines with d == 1 (LDS-resident)
    c.o_causal = c.o_ring1 + NL * 32;               // [NCA][1024] causal kernel half tiles, launch-resident
    c.o_cpart1 = c.o_causal + L.NCA * 1024;         // [NSJ/G][NCH][64] conv1d_1 chunk partials (only when NSJ/G < W)
    c.o_skl = c.o_cpart1 + ((L.NSJ / a.G < W) ? (L.NSJ / a.G) * NCH * 64 : 0);     // [NSJ/G][NL][64] per-layer skip values (same condition)
    c.o_cat = c.o_skl + ((L.NSJ / a.G < W) ? (L.NSJ / a.G) * NL * 64 : 0);        // one-hot sampler scratch: logits, log-probs, float64 exps/cdf
    c.o_slots = c.o_cat + (L.scalar ? 0 : 4 * L.Opad + 32);                            // [nslot][SlotOff::FLOATS]
    c.stb = a.state + ((long long)c.b * a.G + c.g) * L.state_stride;
    const int* pmeta = reinterpret_cast<const int*>(a.P + L.off_meta);
    c.ring = c.stb + L.st_ring;
    c.GCv = a.cond + (long long)c.b * NL * 64;
    c.LCb = a.cond + (long long)a.B * NL * 64 + (long long)c.b * T * NL * 64;
    int* meta = reinterpret_cast<int*>(c.stb + L.st_meta);
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.P), 0, (int)(L.packed_floats * 4), 0x00020000);

    if (threadIdx.x < 128) LDSI(c.o_meta + threadIdx.x) = pmeta[threadIdx.x];
    if (threadIdx.x < 64) {
        const int p0 = reinterpret_cast<int*>(c.stb + L.st_ringpos)[threadIdx.x];
        LDSI(c.o_ringpos + threadIdx.x) = p0;
        LDSI(c.o_pos0 + threadIdx.x) = p0;
    }
    if (threadIdx.x < 32) LDSI(c.o_ready + threadIdx.x) = 0;   // ready[16] + ctrl[16]
    for (int i = threadIdx.x; i < NL * 64; i += blockDim.x) lds[c.o_gc + i] = L.G > 0 ? c.GCv[i] : 0.0f;
    for (int i = threadIdx.x; i < NL * 32; i += blockDim.x) {
        const int l = i >> 5;
        lds[c.o_ring1 + i] = pmeta[l] == 1 ? c.ring[pmeta[64 + l] + (i & 31)] : 0.0f;
    }
    if (L.scalar)
        for (int i = threadIdx.x; i < L.NCA * 1024; i += blockDim.x) lds[c.o_causal + i] = a.P[L.off_causal + i];
    __syncthreads();

    int hpos = meta[M_HPOS], prev_valid = meta[M_PREV_VALID], qprev = meta[M_QPREV];
    if (wid == 0) chain_main<SCALAR, INSTR>(a, c, hpos, prev_valid, qprev);
    else if (wid <= kLoaders) loader_main<kLoaders, INSTR>(a, c, rs, wid - 1);
    else if (D && wid == kLoaders + 1) {}
    else worker_main<W, NTW, SCALAR, SPLIT1, INSTR, HELP>(a, c, rs, wid - 1 - kLoaders - D);

    // ---------------- persist the per-stream state (model.py:49-64 queues) ----------------
    __syncthreads();
    if (threadIdx.x < 64) {
        const unsigned d = (unsigned)LDSI(c.o_meta + threadIdx.x);
        reinterpret_cast<int*>(c.stb + L.st_ringpos)[threadIdx.x] = (int)(((unsigned)LDSI(c.o_pos0 + threadIdx.x) + (unsigned)T) % (d ? d : 1u));
    }
    if (wid == 0 && c.lane == 0) {
        meta[M_TABS] = meta[M_TABS] + T;
        meta[M_HPOS] = hpos;
        meta[M_PREV_VALID] = prev_valid;
        meta[M_QPREV] = qprev;
        if (LDSI(c.o_ctrl + C_ABORT)) atomicMax(a.status, LDSI(c.o_ctrl + C_ABORT));
    }
    for (int i = threadIdx.x; i < NL * 32; i += blockDim.x) {
        const int l = i >> 5;
        if (pmeta[l] == 1) c.ring[pmeta[64 + l] + (i & 31)] = lds[c.o_ring1 + i];
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
        out[i] = fn == 0 ? exp64_e(x[i]) : (fn == 2 ? exp64_nonpos_e(x[i]) : log64_e(x[i]));
}


// utils/audio.py:14-17 save_wav: wav *= 32767 / max(0.01, max|wav|); astype(int16) -- per utterance (row), on the device
__global__ void __launch_bounds__(256) wn_wav_peak_kernel(const float* wav, long long n, int nchunk, float* part)
{
    // grid (nchunk, rows): partial max |x| of a chunk of row blockIdx.y   (max is exact in any order)
    const float* x = wav + (long long)blockIdx.y * n;
    const long long per = (n + nchunk - 1) / nchunk, i0 = (long long)blockIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
    float m = 0.0f;
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) m = fmaxf(m, fabsf(x[i]));
    __shared__ float sh[256];
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + st]); __syncthreads(); }
    if (threadIdx.x == 0) part[blockIdx.y * nchunk + blockIdx.x] = sh[0];
}
__global__ void wn_wav_int16_kernel(const float* wav, long long n, int nchunk, const float* part, int16_t* out)
{
    const int row = blockIdx.y;
    float m = 0.0f;
    for (int k = 0; k < nchunk; ++k) m = fmaxf(m, part[row * nchunk + k]);
    const double md = (double)m;
    const float scale = (float)(32767.0 / (md > 0.01 ? md : 0.01));          // python float division, then the float32 in-place multiply
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[(long long)row * n + i] = (int16_t)(int)(wav[(long long)row * n + i] * scale);   // astype(int16): truncation toward zero
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
    int workers;   // worker waves per workgroup
    int groups;    // workgroups per stream (0 = auto)
    int helpers;   // 1 = helper workgroups when the launch qualifies (conv1d_1, and conv1d_2's partials if O <= 32; default),
                   // 2 = conv1d_1 only, 0 = never
    int xcd;       // 1 (default) = the XCD-per-stream kernel (twv_wavenet_xcd.hip) whenever model, batch and device qualify; 0 = never
    int xcd_many;  // 0 = the library chooses (the many-streams form from batch 21 on), 1 = the many-streams form at every batch, 2 = never below batch 33
    unsigned long long* prof;
    int prof_steps;
};

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(expr)                                                                                              \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) return fail(TWV_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));         \
    } while (0)

extern "C" const char* twv_last_error(void) { return g_err.c_str(); }
// (twv_version lives in twv_ckpt.hip: the one file that is compiled with the source-hash stamp, so that an edit of one kernel file
// recompiles that file only)

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
    if (d.gc_channels > 0 && d.gc_cardinality < 0) return fail(TWV_E_INVALID, "gc_cardinality must be >= 0 (0 = the caller passes the embedding, model.py:199-207)");
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
    L.off_causal = p; p += L.scalar ? (long long)L.NCA * 1024 : (long long)2 * L.Q * 32;
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
    L.off_xl = 0; L.off_xc = 0;
    if (xcd_model_ok(L)) { L.off_xl = p; p += (long long)kXcdXlFloats * L.NL; L.off_xc = p; p += kXcdXcFloats; }
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
    if (L.packed_floats * 4 > 0x7fffffffLL) return fail(TWV_E_UNSUPPORTED, "packed weights exceed 2 GiB");
    return TWV_OK;
}

static int device_cus()
{
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
    return p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
}
// workgroups per stream: explicit option, else the largest of 8/4/2/1 that divides the output blocks and keeps every
// workgroup of the launch co-resident (one workgroup per CU: the LDS slot ring takes most of the 160 KiB)
// the XCD-per-stream kernel: stream b on XCD b % 8 (8 x 32 CUs; up to four streams per XCD), explicit `groups` keeps the generic kernel
static bool use_xcd(const twv_wavenet* h, int batch)
{
    return h->xcd != 0 && h->groups == 0 && h->lay.off_xl != 0 && batch >= 1 && batch <= xcd_max_streams(h->lay) && device_cus() >= 256;
}
static int resolve_groups(const twv_wavenet* h, int batch)
{
    const int NSJ = h->lay.NSJ;
    if (use_xcd(h, batch)) return 1;
    if (h->groups > 0) return (NSJ % h->groups == 0) ? h->groups : -1;
    const int cus = device_cus();
    for (int g = 8; g >= 1; g >>= 1)
        if (NSJ % g == 0 && (long long)batch * g <= cus) return g;
    return 1;
}
constexpr int kWorkers = 4;
// LDS floats of the generation kernel besides the slot ring
static long long lds_fixed_floats(const Layout& L, int G)
{
    const int nsjl = L.NSJ / G;
    return 64 * 32 + 2LL * L.S + (long long)L.NOJ * L.NCH * 64 + 64 * 4 + 32 + (long long)L.NL * 96 + (long long)L.NCA * 1024 +
           (nsjl < kWorkers ? (long long)nsjl * (L.NCH + L.NL) * 64 : 0) + (L.scalar ? 0 : 4LL * L.Opad + 32);
}
static int resolve_nslot(const Layout& L, int G)
{
    long long ns = (160 * 1024 / 4 - lds_fixed_floats(L, G)) / SlotOff::FLOATS;
    if (ns > 8) ns = 8;
    if (L.NL > 1 && ns > L.NL - 1) ns = L.NL - 1;    // a slot's x[t-d] must have been produced before it is staged
    if (L.NL == 1) ns = 1;
    return (int)ns;
}

extern "C" int twv_wavenet_create(const twv_wavenet_dims* dims, twv_wavenet** out)
{
    if (!dims || !out) return fail(TWV_E_INVALID, "null argument");
    twv_wavenet* h = new twv_wavenet();
    h->dims = *dims;
    h->workers = 4;
    h->helpers = 1;
    h->groups = 0;
    h->xcd = 1;
    h->xcd_many = 0;
    h->prof = nullptr; h->prof_steps = 0;
    const int rc = build_layout(*dims, h);
    if (rc != TWV_OK) { delete h; return rc; }
    if (resolve_nslot(h->lay, 1) < 1) { delete h; return fail(TWV_E_UNSUPPORTED, "model does not fit the 160 KiB LDS budget"); }
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
extern "C" size_t twv_wavenet_state_bytes(const twv_wavenet* h, int batch)
{
    // per (stream, workgroup) delay lines etc., then the all-gather granules [B][2][S] x 8 bytes
    int G = resolve_groups(h, batch);
    if (G < 1) G = 1;
    return (size_t)h->lay.state_stride * 4 * (size_t)batch * G + (size_t)batch * 2 * h->lay.S * 8 + (use_xcd(h, batch) ? xcd_exchange_bytes(batch) : 0);
}
extern "C" size_t twv_wavenet_cond_bytes(const twv_wavenet* h, int batch, int n_steps)
{
    // XCD path: header + gc projections + the rows of the (upsampled) condition; the lc projections are made inside the launch
    if (use_xcd(h, batch)) return ((size_t)XH_WORDS + (size_t)batch * h->lay.NL * 64 + (size_t)batch * (size_t)n_steps * h->lay.L) * 4;
    return ((size_t)batch * h->lay.NL * 64 + (size_t)batch * (size_t)n_steps * h->lay.NL * 64) * 4;
}
extern "C" int twv_wavenet_set_profile_buffer(twv_wavenet* h, void* dev_u64, int steps)
{
    if (!h) return fail(TWV_E_INVALID, "null argument");
    h->prof = (unsigned long long*)dev_u64; h->prof_steps = dev_u64 ? steps : 0;
    return TWV_OK;
}
extern "C" int twv_wavenet_set_option(twv_wavenet* h, const char* name, int value)
{
    if (!h || !name) return fail(TWV_E_INVALID, "null argument");
    if (!strcmp(name, "helpers")) {
        if (value < 0 || value > 2) return fail(TWV_E_INVALID, "helpers must be 0 (off), 1 (auto) or 2 (conv1d_1 only)");
        h->helpers = value;
        return TWV_OK;
    }
    if (!strcmp(name, "workers")) {
        if (value != 4 && value != 3) return fail(TWV_E_INVALID, "workers must be 4 or 3 (3 = idle wave beside the chain wave)");
        h->workers = value;
        return TWV_OK;
    }
    if (!strcmp(name, "xcd")) {      // set BEFORE sizing / resetting the state and conditioning buffers
        if (value != 0 && value != 1) return fail(TWV_E_INVALID, "xcd must be 0 or 1");
        h->xcd = value;
        return TWV_OK;
    }
    if (!strcmp(name, "xcd_many")) { // 0: the library chooses (round 5: the many-streams XCD kernel -- two streams per chain workgroup -- from batch 21 on: at
                                     // batch 24 / 32 it runs at 9.54 / 9.60 us per step against the batch <= 32 kernel's 9.65 / 11.0), 1: the many-streams
                                     // kernel at every batch, 2: the batch <= 32 kernel wherever it can run (tests, A/B)
        if (value < 0 || value > 2) return fail(TWV_E_INVALID, "xcd_many must be 0, 1 or 2");
        h->xcd_many = value;
        return TWV_OK;
    }
    if (!strcmp(name, "groups")) {   // workgroups per stream; set BEFORE sizing / resetting the state buffer
        if (value != 0 && (value < 1 || value > 16 || h->lay.NSJ % value)) return fail(TWV_E_INVALID, "groups must divide skip_channels/64");
        h->groups = value;
        return TWV_OK;
    }
    return fail(TWV_E_INVALID, std::string("unknown option ") + name);
}

void twv_launch_pack_tiles(float* dst, const float* src, const PackTiles& p, hipStream_t st)
{
    const long long total = (long long)p.ngroups * p.njblk * p.nchunk * 32 * p.lanes;
    long long g = (total + 255) / 256; if (g < 1) g = 1; if (g > 8192) g = 8192;
    hipLaunchKernelGGL(wn_pack_tiles_kernel, dim3((unsigned)g), dim3(256), 0, st, dst, src, p);
}
void twv_launch_copy(float* dst, const float* src, long long n, hipStream_t st)
{
    if (n <= 0) return;
    long long g = (n + 255) / 256; if (g > 8192) g = 8192;
    hipLaunchKernelGGL(wn_copy_kernel, dim3((unsigned)g), dim3(256), 0, st, dst, src, n);
}
int twv_fail(int code, const std::string& msg) { return fail(code, msg); }

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
                     int K, int rowlen, int ncols, int halves, int lanes = 64) {
        PackTiles p{dst_off, dst_gs, baseA, baseB, src_gs, ng, njb, nch, K, rowlen, ncols, halves, lanes};
        const long long total = (long long)ng * njb * nch * 32 * lanes;
        hipLaunchKernelGGL(wn_pack_tiles_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, dst, blob, p);
    };
    auto vec = [&](long long dst_off, long long dst_gs, long long baseA, long long baseB, long long src_gs, int ng, int n, int ncols, int halves) {
        PackVec p{dst_off, dst_gs, baseA, baseB, src_gs, ng, n, ncols, halves};
        hipLaunchKernelGGL(wn_pack_vec_kernel, dim3(grid_for((long long)ng * n, 256)), dim3(256), 0, st, dst, blob, p);
    };
    const long long ls = L.layer_stride, cs = L.c_layer_stride, c0 = L.c_layer0, l0 = L.off_layer0;
    if (L.scalar) {
        tiles(L.off_causal, 0, L.c_causal, 0, 0, 1, 1, L.NCA, L.ifw, 32, 32, 0, 32);           // wavenet/conv1d/kernel (ifw,1,R), half tiles
    } else {
        hipLaunchKernelGGL(wn_copy_kernel, dim3(grid_for((long long)2 * L.Q * 32, 256)), dim3(256), 0, st, dst + L.off_causal, blob + L.c_causal, (long long)2 * L.Q * 32);
    }
    // conv_filter|conv_gate kernel (2,R,D): tap 0 rows [0,32), tap 1 rows [32,64)
    tiles(l0 + LayerOff::T0, ls, c0 + L.c_wf, c0 + L.c_wg, cs, L.NL, 1, 1, 32, 32, 32, 1);
    tiles(l0 + LayerOff::T1, ls, c0 + L.c_wf + 32 * 32, c0 + L.c_wg + 32 * 32, cs, L.NL, 1, 1, 32, 32, 32, 1);
    tiles(l0 + LayerOff::WD, ls, c0 + L.c_wd, 0, cs, L.NL, 1, 1, 32, 32, 32, 0, 32);           // dense kernel (1,D,R), half tile
    tiles(l0 + LayerOff::SK, ls, c0 + L.c_ws, 0, cs, L.NL, L.NSJ, 1, 32, L.S, L.S, 0);         // skip kernel (1,D,S)
    if (L.use_bias) {
        vec(l0 + LayerOff::BFG, ls, c0 + L.c_bf, c0 + L.c_bg, cs, L.NL, 64, 32, 1);
        vec(l0 + LayerOff::BD, ls, c0 + L.c_bd, 0, cs, L.NL, 32, 32, 0);
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
    if (L.off_xl) xcd_pack(dst, blob, L, st);
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

// XCD path: cond = [header][gc projections][rows]; the rows are the upsampled condition (mode XLC_UPSAMPLED) or the mel frames
// themselves (XLC_MEL: create_upsample runs row by row inside the generation launch)
__global__ void wn_xcd_hdr_kernel(int* hdr, int mode, int rows)
{
    if (threadIdx.x < XH_WORDS) hdr[threadIdx.x] = threadIdx.x == XH_MAGIC ? kXcdCondMagic : (threadIdx.x == XH_MODE ? mode : (threadIdx.x == XH_ROWS ? rows : 0));
}
static int xcd_condition(const twv_wavenet* h, const void* packed, const float* rowsrc, int mode, const int32_t* gc_ids, int batch,
                         int rows, void* cond, hipStream_t st)
{
    const Layout& L = h->lay;
    const float* P = (const float*)packed;
    float* GCv = (float*)cond + XH_WORDS;
    if (!L.L) mode = XLC_NONE;
    hipLaunchKernelGGL(wn_xcd_hdr_kernel, dim3(1), dim3(64), 0, st, (int*)cond, mode, rows);
    if (L.G) {
        if (!gc_ids) return fail(TWV_E_INVALID, "gc_ids required (generate.py:72-77)");
        hipLaunchKernelGGL(wn_gc_kernel, dim3(batch), dim3(64), 0, st, P, L, gc_ids, GCv);
    } else {
        HIPCHK(hipMemsetAsync(GCv, 0, (size_t)batch * L.NL * 64 * 4, st));
    }
    if (mode != XLC_NONE && rows > 0) {
        if (!rowsrc) return fail(TWV_E_INVALID, "local condition required");
        twv_launch_copy(GCv + (size_t)batch * L.NL * 64, rowsrc, (long long)batch * rows * L.L, st);
    }
    HIPCHK(hipGetLastError());
    return TWV_OK;
}
extern "C" int twv_wavenet_fused_conditioning(const twv_wavenet* h, int batch) { return (h && use_xcd(h, batch) && h->lay.L > 0) ? 1 : 0; }
extern "C" const char* twv_wavenet_kernel_name(const twv_wavenet* h, int batch)
{
    if (!h || batch < 1) return "";
    if (!use_xcd(h, batch)) return "wn_generate_kernel";
    return xcd_uses_many(h->lay, batch, h->xcd_many) ? "wn_xcd_many_kernel" : "wn_xcd_generate_kernel";
}
extern "C" size_t twv_wavenet_cond_bytes_mel(const twv_wavenet* h, int batch, int t_mel)
{
    return ((size_t)XH_WORDS + (size_t)batch * h->lay.NL * 64 + (size_t)batch * (size_t)t_mel * h->lay.L) * 4;
}
extern "C" int twv_wavenet_condition_mel(const twv_wavenet* h, const void* packed, const float* mel, const int32_t* gc_ids,
                                         int batch, int t_mel, void* cond, void* stream)
{
    if (!h || !packed || !cond || !mel || batch < 1 || t_mel < 1) return fail(TWV_E_INVALID, "bad argument");
    if (!twv_wavenet_fused_conditioning(h, batch))
        return fail(TWV_E_UNSUPPORTED, "fused conditioning needs the XCD-per-stream kernel (see twv_wavenet_fused_conditioning); use twv_wavenet_upsample + twv_wavenet_condition");
    return xcd_condition(h, packed, mel, XLC_MEL, gc_ids, batch, t_mel, cond, (hipStream_t)stream);
}

extern "C" int twv_wavenet_condition(const twv_wavenet* h, const void* packed, const float* upsampled, const int32_t* gc_ids,
                                     int batch, int n_steps, void* cond, void* stream)
{
    if (!h || !packed || !cond || batch < 1 || n_steps < 0) return fail(TWV_E_INVALID, "bad argument");
    const Layout& L = h->lay;
    hipStream_t st = (hipStream_t)stream;
    if (use_xcd(h, batch)) return xcd_condition(h, packed, upsampled, XLC_UPSAMPLED, gc_ids, batch, n_steps, cond, st);
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

template <int W, int NTW, bool SCALAR, int D, bool SPLIT1, bool INSTR, int HELP>
static int launch_generate2(const GenArgs& a, size_t shm, hipStream_t st)
{
    auto kern = wn_generate_kernel<W, NTW, SCALAR, D, SPLIT1, INSTR, HELP>;
    if (shm > 32 * 1024) HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    // Plain launch: co-residency of the grid is checked against the CU count by the caller (one workgroup per CU: ~150 KB of LDS each).
    // hipLaunchCooperativeKernel would let the runtime guarantee it, and was measured: same speed on four boxes, but on a fifth the
    // step took 25.45 instead of 21.3 us (and the Tacotron decoder 5 % longer) -- consistent with a different workgroup -> XCD
    // placement, which this kernel's L2 locality (slice g of every stream on XCD g) depends on.
    hipLaunchKernelGGL(kern, dim3(a.B * a.G * (HELP ? 2 : 1)), dim3((1 + kLoaders + D + W) * 64), shm, st, a);
    HIPCHK(hipGetLastError());
    return TWV_OK;
}
// the instrumented build (phase stamps, per-layer dumps) is a separate instantiation: production launches carry none of its branches
template <int W, int NTW, bool SCALAR, int D = 0, bool SPLIT1 = false, int HELP = 0>
static int launch_generate(const GenArgs& a, size_t shm, hipStream_t st)
{
    if (a.prof != nullptr || a.dbg != nullptr) return launch_generate2<W, NTW, SCALAR, D, SPLIT1, true, HELP>(a, shm, st);
    return launch_generate2<W, NTW, SCALAR, D, SPLIT1, false, HELP>(a, shm, st);
}

static int generate_impl(const twv_wavenet* h, const void* packed, void* state, const void* cond,
                         const void* first_input, const void* forced, const void* uniforms, double temperature,
                         int batch, int n_steps, void* out, int32_t* status, float* debug, int debug_steps, void* stream)
{
    if (!h || !packed || !state || !cond || !status) return fail(TWV_E_INVALID, "null argument");
    if (!forced && (!first_input || !uniforms || !out)) return fail(TWV_E_INVALID, "null argument");
    if (batch < 1 || n_steps < 1) return fail(TWV_E_INVALID, "batch and n_steps must be >= 1");
    const Layout& L = h->lay;
    if ((long long)n_steps * L.NL > 2000000000LL) return fail(TWV_E_INVALID, "n_steps too large for one call");
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(status, 0, 16, st));
    GenArgs a;
    a.P = (const float*)packed; a.state = (float*)state; a.cond = (const float*)cond; a.first_input = first_input;
    a.forced = forced;
    a.uniforms = uniforms; a.out = out; a.status = status; a.dbg = debug; a.dbg_steps = debug ? debug_steps : 0;
    a.prof = h->prof; a.prof_steps = h->prof ? h->prof_steps : 0;
    a.B = batch; a.T = n_steps; a.temperature = (float)temperature; a.lay = L;
    if (use_xcd(h, batch)) {
        // stream b on XCD b % 8, every weight register-resident across the XCD's CUs (twv_wavenet_xcd.hip)
        XcdLaunch x;
        x.P = a.P; x.state = a.state; x.cond = a.cond; x.first_input = first_input; x.forced = forced;
        x.uniforms = (const float*)uniforms; x.temperature = (float)temperature; x.out = (float*)out; x.status = status; x.dbg = debug; x.dbg_steps = a.dbg_steps;
        x.prof = a.prof; x.prof_steps = a.prof_steps;
        { const char* e = getenv("TWV_XCD_PROF_STREAM"); x.prof_stream = e ? atoi(e) : 0; if (x.prof_stream < 0 || x.prof_stream >= batch) x.prof_stream = 0; }
        x.B = batch; x.T = n_steps; x.lay = L; x.many = h->xcd_many;
        unsigned char* xb = reinterpret_cast<unsigned char*>((float*)state + (size_t)L.state_stride * (size_t)batch) + (size_t)batch * 2 * L.S * 8;
        HIPCHK(hipMemsetAsync(xb, 0, xcd_exchange_bytes(batch), st));
        x.exch = reinterpret_cast<unsigned long long*>(xb);
        x.roles = reinterpret_cast<int*>(xb + (size_t)batch * XcdExch::WORDS * 8);
        return xcd_launch(x, st);
    }
    const int G = resolve_groups(h, batch);
    if (G < 1) return fail(TWV_E_INVALID, "groups option does not divide skip_channels/64");
    if ((long long)batch * G > device_cus())
        return fail(TWV_E_UNSUPPORTED, "batch * groups exceeds the CU count: the stream workgroups must all be co-resident");
    a.G = G;
    a.exch = reinterpret_cast<unsigned long long*>((float*)state + (size_t)L.state_stride * (size_t)batch * G);
    HIPCHK(hipMemsetAsync(a.exch, 0, (size_t)batch * 2 * L.S * 8, st));
    a.lay.nslot = resolve_nslot(L, G);
    if (a.lay.nslot < 1) return fail(TWV_E_UNSUPPORTED, "model does not fit the 160 KiB LDS budget");
    const size_t shm = (size_t)(lds_fixed_floats(L, G) + (long long)a.lay.nslot * SlotOff::FLOATS) * 4;
    const int nsjl = L.NSJ / G;
    const int ntw = (nsjl + kWorkers - 1) / kWorkers;
    if (!L.scalar && L.Q > 1024) return fail(TWV_E_UNSUPPORTED, "quantization_channels must be <= 1024");
    if (L.scalar && h->workers == 3) {
        if (nsjl < 3) return launch_generate<3, 1, true, 1, true>(a, shm, st);
        if (nsjl == 3) return launch_generate<3, 1, true, 1, false>(a, shm, st);
        return fail(TWV_E_UNSUPPORTED, "workers=3 needs skip_channels/64/groups <= 3");
    }
    const bool sp = nsjl < kWorkers;                 // few output blocks per workgroup: tiles round-robin over the workers
    if (L.scalar) {
        // helper workgroups (conv1d_1 from resident registers): generation only, more than one workgroup per stream, at most
        // kHelperTiles conv1d_1 tiles per slice, and twice the workgroups must still be co-resident
        const bool help = h->helpers != 0 && sp && G > 1 && !forced && nsjl * L.NCH <= kHelperTiles && 2LL * batch * G <= device_cus();
        if (ntw <= 1 && help && h->helpers == 1 && nsjl == 1 && L.NOJ == 1 && L.O <= 32) return launch_generate<kWorkers, 1, true, 0, true, 2>(a, shm, st);
        if (ntw <= 1 && help) return launch_generate<kWorkers, 1, true, 0, true, 1>(a, shm, st);
        if (ntw <= 1) return sp ? launch_generate<kWorkers, 1, true, 0, true>(a, shm, st) : launch_generate<kWorkers, 1, true, 0, false>(a, shm, st);
        if (ntw == 2) return launch_generate<kWorkers, 2, true, 0, false>(a, shm, st);
        if (ntw <= 4) return launch_generate<kWorkers, 4, true, 0, false>(a, shm, st);
    } else {
        if (ntw <= 1) return sp ? launch_generate<kWorkers, 1, false, 0, true>(a, shm, st) : launch_generate<kWorkers, 1, false, 0, false>(a, shm, st);
        if (ntw == 2) return launch_generate<kWorkers, 2, false, 0, false>(a, shm, st);
        if (ntw <= 4) return launch_generate<kWorkers, 4, false, 0, false>(a, shm, st);
    }
    return fail(TWV_E_UNSUPPORTED, "skip_channels too large for the worker count");
}

extern "C" int twv_wavenet_generate(const twv_wavenet* h, const void* packed, void* state, const void* cond,
                                    const void* first_input, const void* uniforms, double temperature,
                                    int batch, int n_steps, void* out, int32_t* status, float* debug, int debug_steps,
                                    void* stream)
{
    return generate_impl(h, packed, state, cond, first_input, nullptr, uniforms, temperature, batch, n_steps, out, status, debug,
                         debug_steps, stream);
}
extern "C" int twv_wavenet_prime(const twv_wavenet* h, const void* packed, void* state, const void* cond, const void* inputs,
                                 int batch, int n_steps, int32_t* status, void* stream)
{
    if (!inputs) return fail(TWV_E_INVALID, "null argument");
    return generate_impl(h, packed, state, cond, nullptr, inputs, nullptr, 1.0, batch, n_steps, nullptr, status, nullptr, 0, stream);
}

extern "C" int twv_wavenet_status(const int32_t* status, void* stream)
{
    int32_t hst[4] = {0, 0, 0, 0};
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    HIPCHK(hipMemcpy(hst, status, sizeof(hst), hipMemcpyDeviceToHost));
    if (hst[0] == 90)
        return fail(TWV_E_BUSY, "the generation kernel's role workgroups did not all become resident within ~50 ms: the device is busy "
                                "(a persistent kernel needs the device to itself, INTEGRATION.md section 3); nothing was generated, the "
                                "state is unchanged -- retry when the other work has drained (status code 90)");
    if (hst[0] == 74)
        return fail(TWV_E_INVALID, "the conditioning buffer was not built for the XCD-per-stream kernel (options changed between "
                                   "twv_wavenet_condition and twv_wavenet_generate?); nothing was generated (status code 74)");
    if (hst[0] == 31)
        return fail(TWV_E_KERNEL, "the class probabilities of a generation step contain NaN (a NaN or infinite logit): np.random.choice at "
                                  "generate.py:231 raises 'ValueError: probabilities contain NaN' there; the samples from that step on are "
                                  "not to be used (status code 31)");
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

extern "C" int twv_wav_to_int16(const float* wav, int rows, int64_t n, int16_t* out, float* scratch, void* stream)
{
    if (!wav || !out || !scratch || rows < 1 || n < 1) return fail(TWV_E_INVALID, "bad argument");
    const int nchunk = 64;
    hipLaunchKernelGGL(wn_wav_peak_kernel, dim3(nchunk, rows), dim3(256), 0, (hipStream_t)stream, wav, (long long)n, nchunk, scratch);
    hipLaunchKernelGGL(wn_wav_int16_kernel, dim3(grid_for(n, 256), rows), dim3(256), 0, (hipStream_t)stream, wav, (long long)n, nchunk, scratch, out);
    HIPCHK(hipGetLastError());
    return TWV_OK;
}
// generate.py:219-231 for rows of logits that are already in HBM: one wave per row (twv_categorical.hpp, AC-5)
__global__ void __launch_bounds__(64) wn_categorical_rows_kernel(const float* logits, int Q, long long rows, float temperature, const double* u,
                                                                  int32_t* out, float* proba)
{
    const int lane = threadIdx.x;
    for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
        float y[16], sp[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) y[k] = (lane + 64 * k < Q) ? logits[r * Q + lane + 64 * k] : 0.0f;
        bool bad = false;
        const int idx = categorical_sample<16>(y, Q, lane, temperature, u[r], sp, nullptr, &bad);
        if (lane == 0) out[r] = bad ? -1 : idx;                     // -1: the row's probabilities contain NaN (np.random.choice raises ValueError)
        if (proba != nullptr) {
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (lane + 64 * k < Q) proba[r * Q + lane + 64 * k] = sp[k];
        }
    }
}
extern "C" int twv_sample_categorical(const float* logits, int64_t rows, int Q, double temperature, const double* uniforms, int32_t* out,
                                      float* proba, void* stream)
{
    if (!logits || !uniforms || !out || rows < 0 || Q < 1 || Q > 1024) return fail(TWV_E_INVALID, "bad argument (1 <= quantization_channels <= 1024)");
    if (rows) {
        const int grid = (int)(rows < 4096 ? rows : 4096);
        hipLaunchKernelGGL(wn_categorical_rows_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, logits, Q, (long long)rows, (float)temperature,
                           uniforms, out, proba);
    }
    HIPCHK(hipGetLastError());
    return TWV_OK;
}
__global__ void wn_occupy_kernel(unsigned long long ticks)
{
    extern __shared__ float hold[];
    if (threadIdx.x == 0) hold[0] = 0.0f;
    const unsigned long long t0 = wall_clock64();                      // 100 MHz chip-wide clock
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
extern "C" int twv_debug_occupy(int blocks, int lds_bytes, double milliseconds, void* stream)
{
    if (blocks < 1 || lds_bytes < 0 || lds_bytes > 160 * 1024 || milliseconds < 0) return fail(TWV_E_INVALID, "bad argument");
    if (lds_bytes > 32 * 1024) HIPCHK(hipFuncSetAttribute((const void*)wn_occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipLaunchKernelGGL(wn_occupy_kernel, dim3(blocks), dim3(64), (size_t)lds_bytes, (hipStream_t)stream, (unsigned long long)(milliseconds * 1e5));
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
    if (!x || !out || n < 0 || fn < 0 || fn > 2) return fail(TWV_E_INVALID, "bad argument");
    if (n) hipLaunchKernelGGL(wn_eval64_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, fn, x, (long long)n, out);
    HIPCHK(hipGetLastError());
    return TWV_OK;
}
