// twv_tacotron.hip -- MI355X (gfx950) Tacotron text -> mel inference path + its C-ABI (include/twv_amd.h).
//
// Replaces, for hccho2/Tacotron-Wavenet-Vocoder-Korean (citations into /root/reference), the graph that
// synthesizer.py:56 builds with Tacotron.initialize(..., rnn_decoder_test_mode=True) and runs in ONE sess.run
// (synthesizer.py:160), default hparams path (deepvoice multi-speaker, bah_mon_norm attention):
//   tacotron/tacotron.py:51-108   embedding, speaker dense layers, encoder prenet      -> tc_embed_kernel + tc_gemm_kernel
//   tacotron/modules.py:25-74     CBHG (conv bank, maxpool, projections, highways)     -> tc_gemm_kernel (implicit conv gather,
//                                                                                         fused bias/act/batch-norm/residual), tc_* elementwise
//   tacotron/modules.py:66-74     bidirectional GRU with sequence lengths              -> tc_gru_seq_kernel (recurrent tiles in registers)
//   tacotron/tacotron.py:130-201, rnn_wrappers.py:282-467, helpers.py:10-41
//                                 200-step decoder loop (tf.while_loop)                -> tc_decoder_kernel: ONE persistent launch, one
//                                                                                         workgroup per utterance
//   tacotron/tacotron.py:204-219  post CBHG + linear projection                        -> the same CBHG kernels + tc_gemm_kernel
// Arithmetic: the contract of DESIGN.md (AC-1 chunked dot products, AC-2 rationals); results are compared bit for bit with the
// CPU checker.  TF-contrib internals (GRUCell, monotonic attention, ...) are restated from memory -- see DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <string>
#include <vector>
#include <algorithm>
#include "../../include/twv_amd.h"
#include "twv_dev.hpp"
#include "twv_dpp.hpp"

// the GEMM options of the pass in flight on this thread: copied from the HANDLE at the top of twv_tacotron_infer (they are per-handle
// options -- one model's A/B toggle must not change another model's launches in the same process)
static thread_local int g_gemm_valu = 0;            // "gemm_valu" option: 1 = the VALU kernel (cross-check of the MFMA one)
static thread_local int g_gemm_group = 1;           // "gemm_group" option: 0 = one launch per GEMM, separate highway kernels (A/B runs, cross-check)
static thread_local int g_hw_stack = 1;             // "highway_stack" option: 0 = one launch per highway layer (the form before round 6: A/B runs, cross-check)
// "gemm_timing" option (measurement aid, bench.py's `tacotron.roofline`): every GEMM launch is bracketed by a pair of HIP events on
// its stream and its useful FLOPs (2 * rows * K * N, unpadded) are counted; twv_tacotron_gemm_stats sums both since the option was set.
// PROCESS-WIDE and single-threaded by design (one bench process, one model): the option set through any handle counts the launches of
// every handle and the events live until the process ends.
struct GemmStat {
    bool on = false;
    double flop = 0.0;
    long long launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    size_t used = 0;
};
static GemmStat g_gemm_stat;
enum { TACT_NONE = 0, TACT_RELU = 1, TACT_TANH = 2, TACT_SIGMOID = 3, TACT_SOFTSIGN = 4 };
__device__ __forceinline__ float tc_act(float v, int act)
{
    switch (act) {
        case TACT_RELU: return v > 0.0f ? v : 0.0f;
        case TACT_TANH: return tanh_e(v);
        case TACT_SIGMOID: return sigmoid_e(v);
        case TACT_SOFTSIGN: return div_(v, fabsf(v) + 1.0f);
        default: return v;
    }
}

// =====================================================================================================
//  AC-1 GEMM: Y[row, col0 + n] = epilogue( cdot_k X'[row, k] * W[k, n] ), rows = (sequence, t) pairs
//  X' is the implicit 'same'-padded conv window: k = tap*Cin + c -> X[row + tap - pl, c] (zero outside the sequence)
// =====================================================================================================
struct GemmArgs {
    const float* X; int ldx;
    int rows, T, Cin, kw, pl;
    const float* Wt; int K, N;            // tiles [nblk][nchunk][8][64][4]
    const float* bias; int act;
    const float* bn_inv; const float* bn_shift;
    const float* add1; int ld1;           // + add1[row, n]
    const float* add2; int ld2;           // + add2[sequence, n]
    float* Y; int ldy, col0;
    const float* bias2; int mode;         // mode 1 (MFMA kernel only): highway pair tiles -- see mm_body
};
constexpr int kGemmRows = 16, kGemmKS = 512;

__global__ void __launch_bounds__(256) tc_gemm_kernel(GemmArgs a)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * kGemmRows;
    const int nblk_total = (a.N + 63) / 64, nchunk = (a.K + 31) / 32;
    for (int nb0 = 0; nb0 < nblk_total; nb0 += 4) {
        const int nb = nb0 + wave;
        const bool active = nb < nblk_total;
        float r[kGemmRows];
#pragma unroll
        for (int i = 0; i < kGemmRows; ++i) r[i] = 0.0f;
        for (int k0 = 0; k0 < a.K; k0 += kGemmKS) {
            __syncthreads();
            for (int i = tid; i < kGemmRows * kGemmKS; i += 256) {
                const int rr = i / kGemmKS, ko = i - rr * kGemmKS, kk = k0 + ko, row = row0 + rr;
                float v = 0.0f;
                if (row < a.rows && kk < a.K) {
                    if (a.kw == 1) v = a.X[(long long)row * a.ldx + kk];
                    else {
                        const int tap = kk / a.Cin, c = kk - tap * a.Cin;
                        const int t = row % a.T, ts = t + tap - a.pl;
                        if (ts >= 0 && ts < a.T) v = a.X[(long long)(row - t + ts) * a.ldx + c];
                    }
                }
                lds[i] = v;
            }
            __syncthreads();
            if (active) {
                const int nch = min(kGemmKS / 32, nchunk - k0 / 32);
                for (int ch = 0; ch < nch; ++ch) {
                    Tile tl;
                    load_tile(tl, a.Wt + ((long long)nb * nchunk + (k0 / 32 + ch)) * kTile, lane);
#pragma unroll
                    for (int rr = 0; rr < kGemmRows; ++rr) {
                        const float acc = dot_ldso(tl, rr * kGemmKS + ch * 32);
                        r[rr] = (k0 == 0 && ch == 0) ? acc : r[rr] + acc;
                    }
                }
            }
        }
        if (active) {
            const int n = nb * 64 + lane;
            if (n < a.N) {
                const float bv = a.bias ? a.bias[n] : 0.0f;
                const float iv = a.bn_inv ? a.bn_inv[n] : 1.0f, sv = a.bn_inv ? a.bn_shift[n] : 0.0f;
#pragma unroll
                for (int rr = 0; rr < kGemmRows; ++rr) {
                    const int row = row0 + rr;
                    if (row < a.rows) {
                        float v = r[rr];
                        if (a.bias) v = v + bv;
                        v = tc_act(v, a.act);
                        if (a.bn_inv) { const float y = v * iv; v = y + sv; }          // x*inv + (beta - mean*inv)
                        if (a.add1) v = v + a.add1[(long long)row * a.ld1 + n];
                        if (a.add2) v = v + a.add2[(long long)(row / a.T) * a.ld2 + n];
                        a.Y[(long long)row * a.ldy + a.col0 + n] = v;
                    }
                }
            }
        }
    }
}

// -----------------------------------------------------------------------------------------------------
//  The same GEMM on the f32 matrix cores.  v_mfma_f32_32x32x2_f32 is bitwise an ascending-k fmaf chain starting from C,
//  subnormals included (scripts/ubench/mfma_f32_order.hip), so AC-1 maps onto it exactly: chain s_j of a 32-term chunk
//  (k = j, j+4, ..., j+28 from +0) is four dependent MFMAs whose two k's are (j+8i, j+8i+4); the four chains are four
//  independent accumulators; chunk = (s0+s1)+(s2+s3) and the running total are VALU adds that overlap the next MFMAs.
//  Workgroup = 4 waves on a 64-row x 128-column output tile, wave (wr, wc) owns 32 rows x 64 columns (two column halves).  No LDS: lane l
//  reads its row's X' (implicit 'same' conv window) straight from global as four float4 per chunk -- lanes l and l+32 together
//  consume the row's whole 128-byte chunk;
//  the B operand of both column halves comes out of ONE v_permlane32_swap of two registers of the usual 64 x 32 weight tile.
// -----------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kMmRows = 64;
// a + b as eight v_pk_add_f32 (two IEEE adds each): left to itself the compiler packed a third of the chunk sums and wrote the rest as
// single adds on register pairs that were just as aligned (88 add instructions per chunk instead of 64)
__device__ __forceinline__ f32x16 mm_add16(const f32x16 a, const f32x16 b)
{
    f32x16 r;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const f32x2p x = {a[2 * p], a[2 * p + 1]}, y = {b[2 * p], b[2 * p + 1]};
        f32x2p z;
        asm("v_pk_add_f32 %0, %1, %2" : "=v"(z) : "v"(x), "v"(y));
        r[2 * p] = z[0]; r[2 * p + 1] = z[1];
    }
    return r;
}

// the lane's four float4 of one 32-term chunk of X': k = 8i + 4*(lane>>5) + {0..3}, i = 0..3 (q[i][j] feeds chain j, pair i)
struct MmA { f32x4 q[4]; };
struct MmRow { unsigned seq; int t; bool ok; };              // this lane's row: float offset of its sequence's start in X, time index, in range
// No branch around a load: X is read through a buffer descriptor of its exact extent, and a lane whose float4 lies outside the matrix
// (row past the end, k past K, conv window outside the sequence) asks for an out-of-range offset and gets zeros.  s_waitcnt vmcnt retires
// in issue order, and behind a BRANCHED load the compiler must assume it was not issued -- its wait for the current chunk then covered
// the next chunk's prefetch as well (vmcnt(0) in front of every chunk's MFMAs; profiles/r03_train_layer_anatomy.txt tells the story).
__device__ __forceinline__ void mm_load_a(MmA& A, const GemmArgs& a, rsrc_t rx, const MmRow& r, int kg /* first k of the lane's first float4 */, int& tap, int& c)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ts = a.kw == 1 ? r.t : r.t + tap - a.pl;
        const bool ok = r.ok && kg + 8 * i < a.K && ts >= 0 && ts < a.T;
        const unsigned off = (r.seq + (unsigned)ts * (unsigned)a.ldx + (unsigned)(a.kw == 1 ? kg + 8 * i : c)) * 4u;
        const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(ok ? off : 0x80000000u), 0, 0);
        A.q[i] = f32x4{__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w)};
        c += 8;
        if (c >= a.Cin) { c -= a.Cin; ++tap; }
    }
}

// mode 1, the HIGHWAY layer in one launch (modules.py:83-89): the weight tiles interleave the two dense kernels -- tile nb holds columns
// [32 nb, 32 nb + 32) of H in lanes 0-31 and the same columns of T in lanes 32-63 -- so a wave's two column halves are H and T of the
// SAME 32 outputs and the epilogue finishes the layer: relu(H + bias) * sigmoid(T + bias2) + x * (1 - sigmoid(..)), x = add1 (the layer's
// input, which is also the A operand: the output goes to another buffer).  Same operations in the same order as the two GEMMs +
// tc_highway_kernel it replaces.
// (MODE is a template parameter and `a` travels by value: with a run-time mode test and a reference into the kernel arguments the
// compiler scheduled the chunk loop with s_waitcnt vmcnt(0) in front of the MFMAs -- the next chunk's prefetch waited for right behind
// its issue -- and every launch ran 35-50 % longer)
// WIN: how the lane's four float4 of a chunk are addressed -- 0: plain product (kw == 1), 1: conv window with Cin % 8 == 0 (the window
// position is the same for every lane: scalar bookkeeping), 2: any other conv window (per-lane bookkeeping, mm_load_a).  A template
// parameter: a run-time test would put the loads behind a branch (see mm_load_a).
template <int MODE, int WIN>
__device__ __forceinline__ void mm_body(const GemmArgs a, const int bx, const int by)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;                  // wave tile: rows wr*32.., columns (2*by + wc)*64..
    const int row0 = bx * kMmRows + wr * 32;
    const int nblk_total = (a.N + 63) / 64, nchunk = (a.K + 31) / 32;
    const int nb = by * 2 + wc;
    if (nb >= nblk_total) return;
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // running totals of the two column halves, from -0: (-0) + x == x for every x, signed zeros included, so "total = first chunk" needs
    // no select (it was sixteen v_cndmask per chunk)
    const f32x16 mzero = {-0.0f, -0.0f, -0.0f, -0.0f, -0.0f, -0.0f, -0.0f, -0.0f, -0.0f, -0.0f, -0.0f, -0.0f, -0.0f, -0.0f, -0.0f, -0.0f};
    f32x16 tot0 = mzero, tot1 = mzero;
    MmRow r;
    {
        const int row = row0 + (lane & 31);
        r.ok = row < a.rows;
        r.t = r.ok ? row % a.T : 0;
        r.seq = (unsigned)(r.ok ? row - r.t : 0) * (unsigned)a.ldx;
    }
    const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X), 0, (int)((((long long)a.rows - 1) * a.ldx + a.Cin) * 4), 0x00020000);
    const int hh = (lane >> 5) * 4;
    int tap = 0, c = hh;                                      // conv window position of the lane's next float4 (Cin >= 8, multiple of 4)
    int tap_s = 0, c_s = 0;                                   // the same for half-wave 0, WIN == 1
    const float* wt = a.Wt + (long long)nb * nchunk * kTile;
    // plain product (kw == 1, the lane's row is one contiguous run of K floats): the chunk's four float4 sit at a fixed stride behind one
    // per-lane offset; only a K that is not a multiple of 32 needs the k < K test (a float4 past K would read the next row)
    const unsigned xrow = r.ok ? (r.seq + (unsigned)r.t * (unsigned)a.ldx + (unsigned)hh) * 4u : 0x80000000u;
    const bool ktail = (a.K & 31) != 0;
    auto fetch = [&](Tile& t_, MmA& A_, int ch) {             // chunk ch's operands (past the last chunk: the last tile again and zeros)
        const int chc = ch < nchunk ? ch : nchunk - 1;
        load_tile(t_, wt + (long long)chc * kTile, lane);
        if constexpr (WIN == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = ch < nchunk && (!ktail || ch * 32 + hh + 8 * i < a.K);
                const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(ok ? xrow + (unsigned)(ch * 32 + 8 * i) * 4u : 0x80000000u), 0, 0);
                A_.q[i] = f32x4{__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w)};
            }
        } else if constexpr (WIN == 1) {
            // k = tap * Cin + c -> X[row + tap - pl][c]: with Cin a multiple of 8 both half-waves (c and c + 4) change taps together, so
            // (tap, c) live in scalar registers and a lane spends an add, a compare and a select per float4 (per-lane bookkeeping: ~20)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int dt = tap_s - a.pl;
                const unsigned ts = (unsigned)(r.t + dt);
                const bool ok = r.ok && ch * 32 + 8 * i < a.K && ts < (unsigned)a.T;
                const unsigned off = xrow + (unsigned)((dt * a.ldx + c_s) * 4);
                const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(ok ? off : 0x80000000u), 0, 0);
                A_.q[i] = f32x4{__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w)};
                c_s += 8;
                if (c_s >= a.Cin) { c_s -= a.Cin; ++tap_s; }
            }
        } else {
            mm_load_a(A_, a, rx, r, ch * 32 + hh, tap, c);
        }
    };
    // one chunk: the two column halves one after the other, on ONE set of four accumulators (64 registers instead of 128): together with
    // the waves-per-SIMD bound of the kernels this keeps a wave under 256 registers -- two waves per SIMD, so one wave's chunk sums and
    // address arithmetic run under the other's MFMAs (inside a wave they do not overlap at all) -- and the accumulators stay in ordinary
    // VGPRs: the v_pk_add_f32 of the chunk sums read them directly instead of through 128 v_accvgpr_read per chunk.
    auto chunk = [&](const Tile& tl, const MmA& A) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x16 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {                     // k pair (j + 8i, j + 8i + 4)
#pragma unroll
                for (int j = 0; j < 4; ++j) {                 // chain
                    const int ka = j + 8 * i;
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tl.w[ka]), __float_as_uint(tl.w[ka + 4]), false, false);
                    const float bh = __uint_as_float(sw[h]);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.q[i][j], bh, i == 0 ? zero : acc[j], 0, 0, 0);
                }
            }
            const f32x16 cs = mm_add16(mm_add16(acc[0], acc[1]), mm_add16(acc[2], acc[3]));
            if (h == 0) tot0 = mm_add16(tot0, cs);
            else tot1 = mm_add16(tot1, cs);
        }
    };
    // Two operand sets in turn: the next chunk's operands travel while this chunk's 32 MFMAs run, and nothing is copied at the end of a
    // trip (`tl = tn; A = An` was 24 v_mov_b64 per chunk).  The prefetch stays in FRONT of the MFMAs: left to itself the scheduler may sink
    // the twelve loads down to their first uses in the next trip (fewer live registers, but then every MFMA group waits for its operand
    // with vmcnt(0): measured 35-50 % longer launches when an unrelated edit of the epilogue tipped its heuristic that way)
    Tile t0_, t1_;
    MmA A0_, A1_;
    fetch(t0_, A0_, 0);
    for (int ch = 0; ch < nchunk; ch += 2) {
        fetch(t1_, A1_, ch + 1);
        __builtin_amdgcn_sched_barrier(0);
        chunk(t0_, A0_);
        fetch(t0_, A0_, ch + 2);                             // (always issued: no memory instruction behind a branch)
        __builtin_amdgcn_sched_barrier(0);                   // (a barrier behind the chunks as well: 46 registers spilled)
        if (ch + 1 < nchunk) chunk(t1_, A1_);
    }
    if constexpr (MODE == 1) {                                // highway pair: tot0 = H, tot1 = T of output column n
        const int n = nb * 32 + (lane & 31);
        if (2 * n < a.N) {
            const float bh = a.bias[n], bt = a.bias2[n];
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int row = row0 + (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5);
                if (row < a.rows) {
                    float hv = tot0[rr] + bh;
                    hv = hv > 0.0f ? hv : 0.0f;
                    const float tv = sigmoid_e(tot1[rr] + bt);
                    const float x = a.add1[(long long)row * a.ld1 + n];
                    const float p0 = hv * tv, p1 = 1.0f - tv;
                    const float p2 = x * p1;
                    a.Y[(long long)row * a.ldy + a.col0 + n] = p0 + p2;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int n = nb * 64 + h * 32 + (lane & 31);
        if (n < a.N) {
            const float bv = a.bias ? a.bias[n] : 0.0f;
            const float iv = a.bn_inv ? a.bn_inv[n] : 1.0f, sv = a.bn_inv ? a.bn_shift[n] : 0.0f;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int row = row0 + (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5);
                if (row < a.rows) {
                    float v = h == 0 ? tot0[rr] : tot1[rr];
                    if (a.bias) v = v + bv;
                    v = tc_act(v, a.act);
                    if (a.bn_inv) { const float y = v * iv; v = y + sv; }          // x*inv + (beta - mean*inv)
                    if (a.add1) v = v + a.add1[(long long)row * a.ld1 + n];
                    if (a.add2) v = v + a.add2[(long long)(row / a.T) * a.ld2 + n];
                    a.Y[(long long)row * a.ldy + a.col0 + n] = v;
                }
            }
        }
    }
}
#define TWV_TWO_WAVES __attribute__((amdgpu_waves_per_eu(2, 2)))
__global__ void __launch_bounds__(256) TWV_TWO_WAVES tc_gemm_mfma_kernel(GemmArgs a)
{
    if (a.kw == 1) mm_body<0, 0>(a, blockIdx.x, blockIdx.y);
    else if ((a.Cin & 7) == 0) mm_body<0, 1>(a, blockIdx.x, blockIdx.y);
    else mm_body<0, 2>(a, blockIdx.x, blockIdx.y);
}
__global__ void __launch_bounds__(256) TWV_TWO_WAVES tc_gemm_mfma_highway_kernel(GemmArgs a)
{
    if (a.kw == 1) mm_body<1, 0>(a, blockIdx.x, blockIdx.y);
    else mm_body<1, 2>(a, blockIdx.x, blockIdx.y);          // (the highway layers are dense: never taken)
}

// The whole HIGHWAY STACK of a CBHG in one launch (round 6; modules.py:40-41, 83-89: four layers of 128 -> 128 | 128): a layer is 2 GFLOP at
// the post-net and 0.2 at the encoder, so one launch per layer (tc_gemm_mfma_highway_kernel) is mostly launch tail and the layer's rows
// make a round trip through memory in between.  Here a workgroup owns 64 rows for ALL layers: the rows live in LDS (two buffers of
// 64 x 132 floats; the pitch keeps sixteen rows' 16-byte reads on distinct banks), wave (wr, wc) computes rows 32 wr .. of the column
// groups wc and wc + 2 of a layer -- the same tiles, the same MFMA chains, chunk sums and epilogue as mm_body<1, 0>, the A operand read
// from LDS instead of memory -- writes the layer's output into the other buffer, and a barrier later the next layer starts.
constexpr int kHwMaxDepth = 8, kHwPitch = 132;
struct HwStackArgs {
    const float* X; int ldx;               // [rows][128] the stack's input
    int rows, depth;
    const float* Wt[kHwMaxDepth];          // highway pair tiles of layer l: [4 column groups][4 chunks][kTile]
    const float* bh[kHwMaxDepth]; const float* bt[kHwMaxDepth];
    float* Y; int ldy;
};
// WR = 2: 64 rows per workgroup as described; WR = 1 (few rows: the encoder's 3232 would fill 51 CUs): 32 rows per workgroup, wave w owns
// column group w alone -- twice the workgroups, half the work in each
template <int WR>
__global__ void __launch_bounds__(256) TWV_TWO_WAVES tc_highway_stack_kernel(HwStackArgs a)
{
    constexpr int kRows = 32 * WR, kGroups = WR;              // rows per workgroup, column groups per wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = WR == 2 ? wave >> 1 : 0, wc = WR == 2 ? wave & 1 : wave;
    const int row0 = blockIdx.x * kRows;
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const f32x16 mzero = {-0.0f, -0.0f, -0.0f, -0.0f, -0.0f, -0.0f, -0.0f, -0.0f, -0.0f, -0.0f, -0.0f, -0.0f, -0.0f, -0.0f, -0.0f, -0.0f};
    // the rows in: thread -> float4 c4 of row r (rows past the end: zeros)
    for (int i = tid; i < kRows * 32; i += 256) {
        const int r = i >> 5, c4 = i & 31;
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (row0 + r < a.rows) v = *reinterpret_cast<const f32x4*>(a.X + (long long)(row0 + r) * a.ldx + c4 * 4);
        LDS4(r * (kHwPitch / 4) + c4) = v;
    }
    __syncthreads();
    int cur = 0, nxt = kRows * kHwPitch;
    const int arow = (wr * 32 + (lane & 31)) * (kHwPitch / 4) + (lane >> 5);        // float4 index of the lane's first operand of chunk 0
    for (int l = 0; l < a.depth; ++l) {
        const bool last = l + 1 == a.depth;
        const float* wt = a.Wt[l];
        Tile t0_, t1_;
        load_tile(t0_, wt + (long long)(wc * 4) * kTile, lane);
#pragma unroll 1
        for (int cgi = 0; cgi < kGroups; ++cgi) {
            const int cg = wc + 2 * cgi;
            f32x16 tot0 = mzero, tot1 = mzero;
            auto chunk = [&](const Tile& tl, int ch) {
                f32x4 q[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) q[i] = LDS4((cur >> 2) + arow + ch * 8 + 2 * i);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x16 acc[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int ka = j + 8 * i;
                            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tl.w[ka]), __float_as_uint(tl.w[ka + 4]), false, false);
                            const float bh_ = __uint_as_float(sw[h]);
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(q[i][j], bh_, i == 0 ? zero : acc[j], 0, 0, 0);
                        }
                    }
                    const f32x16 cs = mm_add16(mm_add16(acc[0], acc[1]), mm_add16(acc[2], acc[3]));
                    if (h == 0) tot0 = mm_add16(tot0, cs);
                    else tot1 = mm_add16(tot1, cs);
                }
            };
            // four chunks on two tile sets in turn; the tile behind the last chunk of column group wc is chunk 0 of group wc + 2
            const float* w0 = wt + (long long)(cg * 4) * kTile;
            const float* wn = wt + (long long)((cgi + 1 < kGroups ? cg + 2 : cg) * 4) * kTile;
            load_tile(t1_, w0 + 1 * kTile, lane);
            __builtin_amdgcn_sched_barrier(0);
            chunk(t0_, 0);
            load_tile(t0_, w0 + 2 * kTile, lane);
            __builtin_amdgcn_sched_barrier(0);
            chunk(t1_, 1);
            load_tile(t1_, w0 + 3 * kTile, lane);
            __builtin_amdgcn_sched_barrier(0);
            chunk(t0_, 2);
            load_tile(t0_, wn, lane);                          // (the second group loads its own chunk 0 again: no load behind a branch)
            __builtin_amdgcn_sched_barrier(0);
            chunk(t1_, 3);
            // epilogue of mm_body mode 1: relu(H + bias) * sigmoid(T + bias2) + x * (1 - sigmoid(..)), x = the layer's input
            const int n = cg * 32 + (lane & 31);
            const float bh = a.bh[l][n], bt = a.bt[l][n];
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int rl = wr * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5);
                float hv = tot0[rr] + bh;
                hv = hv > 0.0f ? hv : 0.0f;
                const float tv = sigmoid_e(tot1[rr] + bt);
                const float x = lds[cur + rl * kHwPitch + n];
                const float p0 = hv * tv, p1 = 1.0f - tv;
                const float p2 = x * p1;
                const float y = p0 + p2;
                if (last) { if (row0 + rl < a.rows) a.Y[(long long)(row0 + rl) * a.ldy + n] = y; }
                else lds[nxt + rl * kHwPitch + n] = y;
            }
        }
        __syncthreads();
        const int t_ = cur; cur = nxt; nxt = t_;
    }
}

// SEVERAL problems in one launch: the conv bank of a CBHG is 16 (encoder) / 8 (post-net) independent GEMMs over the
// same rows (modules.py:30-35), the four input halves of the biGRU kernels and the five speaker dense layers likewise.  One at a time
// each fills a fraction of the chip (3232 encoder rows = 51 row tiles) and pays its own launch; here the workgroups of all of them form
// one grid, deepest contraction first.  The problems travel in the kernel arguments.
constexpr int kGroupMax = 16;
struct GemmGroup {
    int n;
    int start[kGroupMax + 1];             // first linear workgroup of problem p
    int nby[kGroupMax];                   // its column blocks (of 128)
    GemmArgs a[kGroupMax];
};
__global__ void __launch_bounds__(256) TWV_TWO_WAVES tc_gemm_mfma_group_kernel(GemmGroup g)
{
    int p = 0;
    const int wg = blockIdx.x;
#pragma unroll 1
    while (p + 1 < g.n && wg >= g.start[p + 1]) ++p;
    const int local = wg - g.start[p];
    const int by = local % g.nby[p], bx = local / g.nby[p];
    const GemmArgs a = g.a[p];
    if (a.kw == 1) mm_body<0, 0>(a, bx, by);
    else if ((a.Cin & 7) == 0) mm_body<0, 1>(a, bx, by);
    else mm_body<0, 2>(a, bx, by);
}

// Few rows, deep contraction (encoder CBHG: 3232 rows, K up to 6144): the 64-row kernel would fill ~50 CUs and run 192 chunks
// back to back per wave.  Here a workgroup owns ONE 32-row x 64-column tile and its 4 waves take the chunks round-robin; the
// chunk values meet in LDS and are added to the running total strictly in chunk order (AC-1), each wave keeping a quarter of it.
// (Round 6 also tried one 32-column half per workgroup -- twice the workgroups, two per CU: the same 121 us for the encoder's K = 6144
// projection.  Every workgroup streams the whole weight matrix of its columns: 404 workgroups x 1.5 MB + the rows = 0.9 GB through the
// L2s in 121 us; what bounds this launch is that traffic, which only taller tiles -- fewer workgroups -- would cut.)
__global__ void __launch_bounds__(256) tc_gemm_mfma_ck_kernel(GemmArgs a)
{
    constexpr int NH = 2;
    __shared__ float cv[4][NH][16][64];                       // [chunk slot][column half][register][lane]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * 32;
    const int nchunk = (a.K + 31) / 32;
    const int nb = blockIdx.y, hsel = 0;
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float tot[2][4];                                          // registers 4*wave .. 4*wave+3 of both halves
    MmRow r;
    {
        const int row = row0 + (lane & 31);
        r.ok = row < a.rows;
        r.t = r.ok ? row % a.T : 0;
        r.seq = (unsigned)(r.ok ? row - r.t : 0) * (unsigned)a.ldx;
    }
    const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X), 0, (int)((((long long)a.rows - 1) * a.ldx + a.Cin) * 4), 0x00020000);
    const int hh = (lane >> 5) * 4;
    const float* wt = a.Wt + (long long)nb * nchunk * kTile;
    // Round 6: the next round's operands travel while this round's 32 MFMAs run (two operand sets in turn, as in mm_body; before, every
    // round started with its own loads: 48 rounds x a memory round trip for the encoder's K = 6144 projection), and the barriers order LDS
    // only (a __syncthreads() would wait for the prefetch as well)
    auto fetch = [&](Tile& tl, MmA& A, int ch) {              // (past the last chunk: the last one again, never used)
        const int chc = ch < nchunk ? ch : nchunk - 1;
        load_tile(tl, wt + (long long)chc * kTile, lane);
        int kg = chc * 32 + hh, tap = 0, c = kg;
        if (a.kw > 1) { tap = kg / a.Cin; c = kg - tap * a.Cin; }
        mm_load_a(A, a, rx, r, kg, tap, c);
    };
    auto round = [&](const Tile& tl, const MmA& A, int c0) {
        const int ch = c0 + wave;
        if (ch < nchunk) {
            f32x16 acc0[4], acc1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ka = j + 8 * i;
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tl.w[ka]), __float_as_uint(tl.w[ka + 4]), false, false);
                    const float b0 = __uint_as_float(NH == 2 ? sw[0] : (hsel ? sw[1] : sw[0])), b1 = __uint_as_float(sw[1]);
                    const float a0 = A.q[i][j];
                    acc0[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, i == 0 ? zero : acc0[j], 0, 0, 0);
                    if (NH == 2) acc1[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, i == 0 ? zero : acc1[j], 0, 0, 0);
                }
            }
            const f32x16 v0 = (acc0[0] + acc0[1]) + (acc0[2] + acc0[3]);
#pragma unroll
            for (int q = 0; q < 16; ++q) cv[wave][0][q][lane] = v0[q];
            if (NH == 2) {
                const f32x16 v1 = (acc1[0] + acc1[1]) + (acc1[2] + acc1[3]);
#pragma unroll
                for (int q = 0; q < 16; ++q) cv[wave][NH - 1][q][lane] = v1[q];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int nl = min(4, nchunk - c0);
        for (int s = 0; s < nl; ++s)                           // chunk order
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v = cv[s][h][4 * wave + q][lane];
                    tot[h][q] = (c0 == 0 && s == 0) ? v : tot[h][q] + v;
                }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    Tile t0_, t1_;
    MmA A0_, A1_;
    fetch(t0_, A0_, wave);
    for (int c0 = 0; c0 < nchunk; c0 += 8) {
        fetch(t1_, A1_, c0 + 4 + wave);
        __builtin_amdgcn_sched_barrier(0);
        round(t0_, A0_, c0);
        fetch(t0_, A0_, c0 + 8 + wave);
        __builtin_amdgcn_sched_barrier(0);
        if (c0 + 4 < nchunk) round(t1_, A1_, c0 + 4);
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const int n = nb * 64 + (NH == 2 ? h : hsel) * 32 + (lane & 31);
        if (n < a.N) {
            const float bv = a.bias ? a.bias[n] : 0.0f;
            const float iv = a.bn_inv ? a.bn_inv[n] : 1.0f, sv = a.bn_inv ? a.bn_shift[n] : 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rr = 4 * wave + q;
                const int row = row0 + (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5);
                if (row < a.rows) {
                    float v = tot[h][q];
                    if (a.bias) v = v + bv;
                    v = tc_act(v, a.act);
                    if (a.bn_inv) { const float y = v * iv; v = y + sv; }
                    if (a.add1) v = v + a.add1[(long long)row * a.ld1 + n];
                    if (a.add2) v = v + a.add2[(long long)(row / a.T) * a.ld2 + n];
                    a.Y[(long long)row * a.ldy + a.col0 + n] = v;
                }
            }
        }
    }
}

// ---- small elementwise kernels ----------------------------------------------------------------------------------------
// tacotron.py:51-60 embedding lookup with row 0 forced to zeros
__global__ void tc_embed_kernel(const float* table, const int32_t* tokens, int rows, int E, float* out)
{
    const long long total = (long long)rows * E;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / E), c = (int)(i - (long long)row * E);
        const int tok = tokens[row];
        out[i] = tok == 0 ? 0.0f : table[(long long)tok * E + c];
    }
}
__global__ void tc_gather_rows_kernel(const float* table, const int32_t* ids, int rows, int E, float* out)
{
    const long long total = (long long)rows * E;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / E), c = (int)(i - (long long)row * E);
        out[i] = table[(long long)ids[row] * E + c];
    }
}
// modules.py:38 max_pooling1d(2, stride 1, 'same'): out[t] = max(x[t], x[t+1]) inside each sequence
__global__ void tc_maxpool2_scalar_kernel(const float* x, int rows, int T, int Cn, float* out)      // (a channel count that is not a multiple of 4)
{
    const long long total = (long long)rows * Cn;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / Cn);
        const float a = x[i];
        const float b = (row % T) + 1 < T ? x[i + Cn] : a;
        out[i] = a > b ? a : b;
    }
}
// four channels per thread: with Cn a multiple of 4 (hparams: bank x 128 channels) a float4 never straddles rows
__global__ void tc_maxpool2_kernel(const float* x, int rows, int T, int Cn, float* out)
{
    const int c4n = Cn >> 2;
    const long long total = (long long)rows * c4n;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    f32x4* o4 = reinterpret_cast<f32x4*>(out);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / c4n);
        const f32x4 a = x4[i];
        const f32x4 b = (row % T) + 1 < T ? x4[i + c4n] : a;
        f32x4 r;
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] = a[q] > b[q] ? a[q] : b[q];
        o4[i] = r;
    }
}
// modules.py:89 highway: H*T + x*(1-T)
__global__ void tc_highway_kernel(const float* H, const float* Tg, float* x, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float a = H[i] * Tg[i];
        const float b = 1.0f - Tg[i];
        const float c = x[i] * b;
        x[i] = a + c;
    }
}
// memory rows past input_lengths -> 0 ([RECALLED-TF _prepare_memory]); the biGRU already writes zeros there
__global__ void tc_zero_kernel(float* p, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = 0.0f;
}

// =====================================================================================================
//  bidirectional GRU over sequences (modules.py:66-74), units U = 128
//  The input halves of both kernels are hoisted into GEMMs over all time steps (Gx, Cx: the AC-1 running sums after the
//  x chunks); this kernel continues each sum with the h chunks.  Recurrent tiles live in registers for the whole sequence.
// =====================================================================================================
struct GruSeqArgs {
    const float* Gx;      // [rows][2U]  x-part of the gate pre-activations, per direction: + dir*gx_dstride
    const float* Cx;      // [rows][U]
    long long gx_dstride, cx_dstride;
    const float* Wgh[2];  // tiles [4 nblk][4 chunk]  (K = U rows of h, N = 2U)
    const float* Wch[2];  // tiles [2 nblk][4 chunk]
    const float* bg[2]; const float* bc[2];
    const float* init;    // [N][2U] (fw | bw) or nullptr
    const int32_t* lengths;   // [N] or nullptr (= T)
    int T;
    float* out;           // [rows][2U], rows past the length left untouched (pre-zeroed)
};
// workgroup barrier that orders LDS traffic only: outstanding global loads / stores stay in flight across it
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Round 4.  One workgroup walks one (utterance, direction); a step is strictly serial (gates need h, the candidate needs r * h), so
// the step time is the dependent path.  Rounds 1-3 spent it on three things this version removes:
//   * every lane read all 32 operand values of a chunk from LDS itself (24 ds_read_b128 per wave and step: 1536 LDS cycles of a
//     2500-cycle step).  Now a row of 16 lanes reads the chunk once (two 4-byte reads per lane) and v_fmac_f32_dpp row broadcasts
//     feed the fmas (dot32_dpp, twv_dpp.hpp);
//   * the chunk values of an output met through LDS partials: FOUR barriers per step.  Now a wave owns whole outputs and the chunk
//     values meet in registers: TWO barriers (r * h and u visible; h visible);
//   * the x-parts of a step (a fresh row of Gx / Cx every step, out of HBM or the memory-side cache) were loaded when needed.  Now
//     they are fetched kDepth steps ahead into a register ring, and the barriers wait for LDS only (a __syncthreads() is also a fence
//     that drains every outstanding global access).
// Work split, all in half-wave form (lanes 0-31 run chunks 0, 1 and lanes 32-63 chunks 2, 3 of the same 32 outputs; v_permlane32_swap
// brings chunks 2, 3 to the lower half):
//   waves 0-3 "r waves":             r gates 32 w + (lane mod 32): two dots, sigmoid, r * h to LDS -- the critical first half of the
//                                    step, at raised priority (s_setprio) so that their SIMD-mates do not slow their dots
//   waves 4-7 "u + candidate waves": before the first barrier the u gate's two dots for outputs 32 (w - 4) + (lane mod 32) (its sigmoid
//                                    is evaluated next to the candidate's tanh, where two independent chains share the latency); after
//                                    it the candidate's two dots, tanh, and the blend with the u that never left the lane's registers
// The sums keep their order -- ((((x-part + c0) + c1) + c2) + c3) + bias -- so the bits are those of rounds 1-3.
__device__ __forceinline__ void gru_load_column(float (&w)[32], const float* tile, int out_lane)
{
    const f32x4* p = reinterpret_cast<const f32x4*>(tile) + out_lane;       // tile = [kq][64 output lanes][4]
#pragma unroll
    for (int kq = 0; kq < 8; ++kq) { const f32x4 x = p[kq * 64]; w[4 * kq] = x.x; w[4 * kq + 1] = x.y; w[4 * kq + 2] = x.z; w[4 * kq + 3] = x.w; }
}
__global__ void __launch_bounds__(512) tc_gru_seq_kernel(GruSeqArgs a)
{
    constexpr int U = 128;
    constexpr int kDepth = 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const int len = a.lengths ? a.lengths[n] : a.T;
    // LDS: hs[128] | rh[128]
    const int o_h = 0, o_rh = 128;
    if (tid < U) lds[o_h + tid] = a.init ? a.init[(long long)n * 2 * U + dir * U + tid] : 0.0f;
    __syncthreads();
    if (len <= 0) return;
    auto row_of = [&](int s) { const int sc = s < len ? s : len - 1; return (long long)n * a.T + (dir == 0 ? sc : len - 1 - sc); };
    const int hch = (lane >> 5) * 2;                          // half-wave form: lanes 0-31 run chunks 0, 1, lanes 32-63 chunks 2, 3
    if (wave < 4) {
        // ---- r waves: the step's critical first half (r * h feeds the candidate), so they outrank their SIMD-mates
        __builtin_amdgcn_s_setprio(3);
        const int ro = 32 * wave + (lane & 31);               // gate output 0 .. 127 = r
        float w0[32], w1[32];
        const float* tiles = a.Wgh[dir] + ((long long)(ro >> 6) * 4 + hch) * kTile;      // standard tiles [nblk][chunk][kq][64 output lanes][4]
        gru_load_column(w0, tiles, ro & 63); gru_load_column(w1, tiles + kTile, ro & 63);
        const float bgv = a.bg[dir][ro];
        const float* Gx = a.Gx + dir * a.gx_dstride + ro;
        float gxr[kDepth];
#pragma unroll
        for (int j = 0; j < kDepth; ++j) gxr[j] = Gx[row_of(j) * 2 * U];
        const int xo = o_h + hch * 32 + (lane & 15);
        for (int s0 = 0; s0 < len; s0 += kDepth) {
#pragma unroll
            for (int j = 0; j < kDepth; ++j) {
                const int s = s0 + j;
                if (s >= len) break;
                const float gx = gxr[j];
                gxr[j] = Gx[row_of(s + kDepth) * 2 * U];
                float d0, d1;
                dot32_dpp_x2(w0, lds[xo], lds[xo + 16], w1, lds[xo + 32], lds[xo + 48], d0, d1);
                const auto s2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(d0), __float_as_uint(d0), false, false);   // [1]: lanes 32-63's values on both halves
                const auto s3 = __builtin_amdgcn_permlane32_swap(__float_as_uint(d1), __float_as_uint(d1), false, false);
                float g = gx;
                g = g + d0; g = g + d1;                                            // chunks 0, 1 (lower half)
                g = g + __uint_as_float(s2[1]); g = g + __uint_as_float(s3[1]);    // chunks 2, 3
                g = sigmoid_e(g + bgv);
                if (lane < 32) lds[o_rh + ro] = g * lds[o_h + ro];                 // r * h
                lds_barrier();                                             // r * h visible
                lds_barrier();                                             // (the candidate waves' half of the step)
            }
        }
    } else {
        // ---- u + candidate waves: the u gate's dots run in the shadow of the r waves (its sigmoid waits until it is needed and then
        // interleaves with the candidate's tanh); u never leaves the registers of the lane that blends with it
        const int co = 32 * (wave - 4) + (lane & 31);
        float w0[32], w1[32], wu0[32], wu1[32];
        const float* tiles = a.Wch[dir] + ((long long)(co >> 6) * 4 + hch) * kTile;
        gru_load_column(w0, tiles, co & 63); gru_load_column(w1, tiles + kTile, co & 63);
        const float* utiles = a.Wgh[dir] + ((long long)((U + co) >> 6) * 4 + hch) * kTile;
        gru_load_column(wu0, utiles, (U + co) & 63); gru_load_column(wu1, utiles + kTile, (U + co) & 63);
        const float bcv = a.bc[dir][co], buv = a.bg[dir][U + co];
        const float* Cx = a.Cx + dir * a.cx_dstride + co;
        const float* Gx = a.Gx + dir * a.gx_dstride + U + co;
        float cxr[kDepth], uxr[kDepth];
#pragma unroll
        for (int j = 0; j < kDepth; ++j) { cxr[j] = Cx[row_of(j) * U]; uxr[j] = Gx[row_of(j) * 2 * U]; }
        const int xo = o_rh + hch * 32 + (lane & 15), xh = o_h + hch * 32 + (lane & 15);
        for (int s0 = 0; s0 < len; s0 += kDepth) {
#pragma unroll
            for (int j = 0; j < kDepth; ++j) {
                const int s = s0 + j;
                if (s >= len) break;
                const long long row = row_of(s);
                const float cx = cxr[j], ux = uxr[j];
                cxr[j] = Cx[row_of(s + kDepth) * U];
                uxr[j] = Gx[row_of(s + kDepth) * 2 * U];
                float gu;
                {
                    float d0, d1;
                    dot32_dpp_x2(wu0, lds[xh], lds[xh + 16], wu1, lds[xh + 32], lds[xh + 48], d0, d1);
                    const auto s2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(d0), __float_as_uint(d0), false, false);
                    const auto s3 = __builtin_amdgcn_permlane32_swap(__float_as_uint(d1), __float_as_uint(d1), false, false);
                    gu = ux;
                    gu = gu + d0; gu = gu + d1;
                    gu = gu + __uint_as_float(s2[1]); gu = gu + __uint_as_float(s3[1]);
                    gu = gu + buv;
                }
                const float h = lds[o_h + co];
                lds_barrier();                                             // r * h visible
                float e0, e1;
                dot32_dpp_x2(w0, lds[xo], lds[xo + 16], w1, lds[xo + 32], lds[xo + 48], e0, e1);
                const auto s2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(e0), __float_as_uint(e0), false, false);   // [1]: lanes 32-63's values on both halves
                const auto s3 = __builtin_amdgcn_permlane32_swap(__float_as_uint(e1), __float_as_uint(e1), false, false);
                float c = cx;
                c = c + e0; c = c + e1;                                            // chunks 0, 1 (lower half)
                c = c + __uint_as_float(s2[1]); c = c + __uint_as_float(s3[1]);    // chunks 2, 3
                c = tanh_e(c + bcv);
                const float u = sigmoid_e(gu);
                if (lane < 32) {
                    const float t1 = u * h, t2 = 1.0f - u, t3 = t2 * c;
                    const float hn = t1 + t3;
                    a.out[row * 2 * U + dir * U + co] = hn;
                    lds[o_h + co] = hn;
                }
                lds_barrier();                                             // h visible
            }
        }
    }
}

// AC-6: the two cumulative sums of the monotonic attention (exclusive cumsum of log(1 - p), inclusive cumsum of
// previous / cumprod; tf.contrib.seq2seq.monotonic_attention 'parallel' mode, whose reduction order TensorFlow does not specify) run in
// BLOCKS OF 64 time steps: inside a block the scan64 tree in float32 -- row_shr 1, 2, 4, 8 inside each row of 16 lanes, lane 15 -> the
// next row, lane 31 -> the upper half; padding lanes hold +0 -- and the total of the earlier blocks is added in front of the block's
// values.  Lane i holds element i of the block; returns the lane's inclusive or exclusive prefix and advances `carry` past the block.
// (Rounds 1-3 ran one add chain over all T steps, v_readlane by v_readlane: 4.8 us of the 41 us decoder step at T = 101.)
__device__ __forceinline__ float scan64_f32_wave(float v)
{
#define TWV_SHR_(c_, m_, b_) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (c_), (m_), 0xf, (b_)))
    v = v + TWV_SHR_(0x111, 0xf, true);        // row_shr:1
    v = v + TWV_SHR_(0x112, 0xf, true);        // row_shr:2
    v = v + TWV_SHR_(0x114, 0xf, true);        // row_shr:4
    v = v + TWV_SHR_(0x118, 0xf, true);        // row_shr:8
    v = v + TWV_SHR_(0x142, 0xa, false);       // row_bcast:15 -> rows 1, 3
    v = v + TWV_SHR_(0x143, 0xc, false);       // row_bcast:31 -> rows 2, 3
#undef TWV_SHR_
    return v;
}
__device__ __forceinline__ float decg_scan_block(float v, float& carry, bool first, int lane, bool inclusive)
{
    const float sc = scan64_f32_wave(v);
    const float incl = first ? sc : carry + sc;
    const float up = __shfl_up(incl, 1);
    const float excl = lane == 0 ? (first ? 0.0f : carry) : up;
    carry = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(incl), 63));
    return inclusive ? incl : excl;
}

// =====================================================================================================
//  the decoder: ONE persistent launch, one workgroup (8 waves) per utterance, max_iters steps
// =====================================================================================================
struct DecW {   // offsets (floats) into the packed buffer
    long long dp1, dp1b, dp2, dp2b, aWg, abg, aWc, abc, Wq, nv, ab, asb, cW, cb, rWg[4], rbg[4], rWc[4], rbc[4], oW, ob;
};
struct DecArgs {
    const float* P;
    DecW w;
    const float* keys;     // [N][T][A]
    const float* memo;     // [N][T][ENC]   (zero past the length)
    const float* init;     // [N][ninit]: attention rnn init (AS) then dec_layers x DR
    const int32_t* lengths;
    int N, T, M, R, D0, D1, A, AS, ENC, DR, layers, iters;
    int SEc;               // model_type 'simple': width of the speaker embedding concatenated inside the decoder (0 otherwise)
    const float* semb;     // [N][SEc] the utterances' speaker embeddings (SEc > 0)
    float* mel;            // [N][iters*R][M]
    float* align;          // [N][T][iters] or nullptr
    int32_t* status;
    long long packed_bytes;
    unsigned long long* prof;   // optional [iters][16] s_memtime stamps of workgroup 0 (tuning aid)
    int nbias;                  // total bias floats over the decoder stages
};
// LDS map of the decoder (float offsets)
struct DecLds { int cat, vec, h[5], frame, ctx, part, sc, p, cp, lg, al, q; };

// y_part[(nb*nchunk + ch)*64 + lane] = chunk(nb, ch) of W (tiles at wt) applied to the LDS vector at xo; tiles round-robin over
// the 8 waves, two named buffers so that the next tile travels while the current one is used
__device__ __noinline__ void dec_gemv_partials(rsrc_t rs, int wt_bytes, int K, int N, int xo, int o_part, int wave, int lane)
{
    const int nchunk = (K + 31) / 32, ntile = ((N + 63) / 64) * nchunk;
    const int vo = lane * 16;
    Tile t0, t1, t2, t3;                                  // four tiles in flight per wave: the stream is L2-latency bound
    if (wave < ntile) load_tile_b(t0, rs, vo, wt_bytes + wave * (kTile * 4));
    if (wave + 8 < ntile) load_tile_b(t1, rs, vo, wt_bytes + (wave + 8) * (kTile * 4));
    if (wave + 16 < ntile) load_tile_b(t2, rs, vo, wt_bytes + (wave + 16) * (kTile * 4));
    if (wave + 24 < ntile) load_tile_b(t3, rs, vo, wt_bytes + (wave + 24) * (kTile * 4));
#define TWV_DEC_STEP(tt, j)                                                                      \
    if ((j) < ntile) {                                                                           \
        const float r = dot_ldso(tt, xo + ((j) % nchunk) * 32);                                  \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        lds[o_part + (j) * 64 + lane] = r;                                                       \
        if ((j) + 32 < ntile) load_tile_b(tt, rs, vo, wt_bytes + ((j) + 32) * (kTile * 4));      \
    }
    for (int i = wave; i < ntile; i += 32) {
        TWV_DEC_STEP(t0, i)
        TWV_DEC_STEP(t1, i + 8)
        TWV_DEC_STEP(t2, i + 16)
        TWV_DEC_STEP(t3, i + 24)
    }
#undef TWV_DEC_STEP
}
// chunk values of output column j summed in chunk order (AC-1)
__device__ __forceinline__ float dec_combine(int o_part, int K, int j)
{
    const int nchunk = (K + 31) / 32, nb = j >> 6, l = j & 63;
    float v = 0.0f;
    for (int ch = 0; ch < nchunk; ++ch) {
        const float c = lds[o_part + (nb * nchunk + ch) * 64 + l];
        v = ch == 0 ? c : v + c;
    }
    return v;
}
// tf.contrib.rnn.GRUCell on the LDS vector cat = [x (nin) | h (U)] : h <- u*h + (1-u)*tanh([x, r*h].Wc + bc), returns nothing;
// the new state is written to lds[o_hout .. +U).  All 512 threads call it.
__device__ __forceinline__ void dec_gru(rsrc_t rs, const float* P, long long oWg, long long obg, long long oWc, long long obc, int nin, int U,
                                        int o_cat, int o_hout, int o_part, int o_vec, int tid, int wave, int lane)
{
    dec_gemv_partials(rs, (int)((oWg) * 4), nin + U, 2 * U, o_cat, o_part, wave, lane);
    __syncthreads();
    float g = 0.0f;
    if (tid < 2 * U) g = sigmoid_e(dec_combine(o_part, nin + U, tid) + P[obg + tid]);
    __syncthreads();                                     // partials consumed
    if (tid < U) { lds[o_vec + tid] = lds[o_cat + nin + tid]; lds[o_cat + nin + tid] = g * lds[o_cat + nin + tid]; }   // keep h, cat <- [x, r*h]
    else if (tid < 2 * U) lds[o_vec + tid] = g;          // u
    __syncthreads();
    dec_gemv_partials(rs, (int)((oWc) * 4), nin + U, U, o_cat, o_part, wave, lane);
    __syncthreads();
    if (tid < U) {
        const float c = tanh_e(dec_combine(o_part, nin + U, tid) + P[obc + tid]);
        const float u = lds[o_vec + U + tid], h = lds[o_vec + tid];
        const float t1 = u * h, t2 = 1.0f - u, t3 = t2 * c;
        lds[o_hout + tid] = t1 + t3;
    }
    __syncthreads();
}

__global__ void __launch_bounds__(512) tc_decoder_kernel(DecArgs a)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.x;
    const int T = a.T, M = a.M, R = a.R, A = a.A, AS = a.AS, ENC = a.ENC, DR = a.DR, D0 = a.D0, D1 = a.D1;
    const int len = a.lengths[n];
    const float* P = a.P;
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.P), 0, (int)a.packed_bytes, 0x00020000);
    const float* keys = a.keys + (long long)n * T * A;
    const float* memo = a.memo + (long long)n * T * ENC;
    // ---- LDS carve (all multiples of 4 floats)
    int o = 0;
    const int o_cat = o; o += 1024;                       // concatenated GEMV input
    const int o_vec = o; o += 1024;                       // scratch vector (kept h | u, prenet outputs, ...)
    const int o_ha = o; o += AS;                          // attention GRU state
    int o_hr[4];
    for (int i = 0; i < a.layers; ++i) { o_hr[i] = o; o += DR; }
    const int o_frame = o; o += ((M + 31) / 32) * 32;     // padded to whole chunks (pad stays zero)
    const int o_ctx = o; o += ENC;
    const int o_y = o; o += DR;
    const int Tp = ((T + 3) / 4) * 4;
    const int o_al = o; o += Tp;                          // alignments of the previous step
    const int o_p = o; o += Tp;
    const int o_cp = o; o += Tp;
    const int o_q = o; o += Tp;
    const int o_pq = o; o += A;
    const int o_scp = o; o += Tp * 8;                     // score chunk values [t][A/32]
    const int o_part = o;                                 // GEMV chunk partials (largest: (nin+U)/32 chunks x 2U/64 blocks x 64)

    // ---- initial state (tacotron.py:184-195, AttentionWrapper.zero_state, helpers.py:90-92)
    const float* init = a.init + (long long)n * (AS + a.layers * DR);
    for (int i = tid; i < AS; i += 512) lds[o_ha + i] = init[i];
    for (int l = 0; l < a.layers; ++l)
        for (int i = tid; i < DR; i += 512) lds[o_hr[l] + i] = init[AS + l * DR + i];
    for (int i = tid; i < ((M + 31) / 32) * 32; i += 512) lds[o_frame + i] = 0.0f;
    for (int i = tid; i < ENC; i += 512) lds[o_ctx + i] = 0.0f;
    for (int i = tid; i < Tp; i += 512) lds[o_al + i] = i == 0 ? 1.0f : 0.0f;      // one-hot at 0 [RECALLED-TF initial_alignments]
    __syncthreads();
    const int nAch = A / 32;

#define TWV_STAMP(k) if (a.prof && n == 0 && tid == 0) a.prof[it * 16 + (k)] = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < a.iters; ++it) {
        TWV_STAMP(0)
        // ---- rnn_wrappers.py:425 decoder prenet: dense(M -> D0) relu, dense(D0 -> D1) relu
        dec_gemv_partials(rs, (int)((a.w.dp1) * 4), M, D0, o_frame, o_part, wave, lane);
        __syncthreads();
        if (tid < D0) { const float v = dec_combine(o_part, M, tid) + P[a.w.dp1b + tid]; lds[o_vec + tid] = v > 0.0f ? v : 0.0f; }
        __syncthreads();
        dec_gemv_partials(rs, (int)((a.w.dp2) * 4), D0, D1, o_vec, o_part, wave, lane);
        __syncthreads();
        // ---- rnn_wrappers.py:310-312 cell_inputs = [prenet_out | attention]; attention GRU on state ha
        if (tid < D1) { const float v = dec_combine(o_part, D0, tid) + P[a.w.dp2b + tid]; lds[o_cat + tid] = v > 0.0f ? v : 0.0f; }
        for (int i = tid; i < ENC; i += 512) lds[o_cat + D1 + i] = lds[o_ctx + i];
        for (int i = tid; i < AS; i += 512) lds[o_cat + D1 + ENC + i] = lds[o_ha + i];
        __syncthreads();
        TWV_STAMP(1)
        dec_gru(rs, P, a.w.aWg, a.w.abg, a.w.aWc, a.w.abc, D1 + ENC, AS, o_cat, o_ha, o_part, o_vec, tid, wave, lane);
        TWV_STAMP(2)
        // ---- attention [RECALLED-TF BahdanauMonotonicAttention.__call__]: query layer
        dec_gemv_partials(rs, (int)((a.w.Wq) * 4), AS, A, o_ha, o_part, wave, lane);
        __syncthreads();
        if (tid < A) lds[o_pq + tid] = dec_combine(o_part, AS, tid);
        __syncthreads();
        TWV_STAMP(3)
        // score[t] = cdot_j normed_v[j] * tanh((keys[t][j] + pq[j]) + b[j]) + score_bias ; one (t, chunk) per thread
        for (int task = tid; task < T * nAch; task += 512) {
            const int t = task / nAch, ch = task - t * nAch;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            const float* kr = keys + (long long)t * A + ch * 32;
#pragma unroll
            for (int j = 0; j < 32; j += 4) {
                const int jj = ch * 32 + j;
                s0 = fma_(P[a.w.nv + jj + 0], tanh_e((kr[j + 0] + lds[o_pq + jj + 0]) + P[a.w.ab + jj + 0]), s0);
                s1 = fma_(P[a.w.nv + jj + 1], tanh_e((kr[j + 1] + lds[o_pq + jj + 1]) + P[a.w.ab + jj + 1]), s1);
                s2 = fma_(P[a.w.nv + jj + 2], tanh_e((kr[j + 2] + lds[o_pq + jj + 2]) + P[a.w.ab + jj + 2]), s2);
                s3 = fma_(P[a.w.nv + jj + 3], tanh_e((kr[j + 3] + lds[o_pq + jj + 3]) + P[a.w.ab + jj + 3]), s3);
            }
            lds[o_scp + t * 8 + ch] = (s0 + s1) + (s2 + s3);
        }
        __syncthreads();
        TWV_STAMP(4)
        // p = sigmoid(score) (0 past the length: _maybe_mask_score(-inf)); safe_cumprod pieces
        for (int t = tid; t < T; t += 512) {
            float sc = 0.0f;
            for (int ch = 0; ch < nAch; ++ch) { const float c = lds[o_scp + t * 8 + ch]; sc = ch == 0 ? c : sc + c; }
            sc = sc + P[a.w.asb];
            const float pv = t < len ? sigmoid_e(sc) : 0.0f;
            lds[o_p + t] = pv;
            float om = 1.0f - pv;
            const float tiny = 1.17549435e-38f;
            om = om < tiny ? tiny : (om > 1.0f ? 1.0f : om);
            lds[o_q + t] = log_e(om);
        }
        __syncthreads();
        if (wave == 0) {  // exclusive cumsum of the logs (AC-6: blocks of 64, scan64 tree)
            float run = 0.0f;
            for (int base = 0; base < T; base += 64) {
                const int t = base + lane;
                const float r = decg_scan_block(t < T ? lds[o_q + t] : 0.0f, run, base == 0, lane, false);
                if (t < T) lds[o_q + t] = r;
            }
        }
        __syncthreads();
        for (int t = tid; t < T; t += 512) {
            const float cpv = exp_e(lds[o_q + t]);
            lds[o_cp + t] = cpv;
            float den = cpv;
            den = den < 1e-10f ? 1e-10f : (den > 1.0f ? 1.0f : den);
            lds[o_q + t] = div_(lds[o_al + t], den);
        }
        __syncthreads();
        if (wave == 0) {  // inclusive cumsum (AC-6)
            float cs = 0.0f;
            for (int base = 0; base < T; base += 64) {
                const int t = base + lane;
                const float r = decg_scan_block(t < T ? lds[o_q + t] : 0.0f, cs, base == 0, lane, true);
                if (t < T) lds[o_q + t] = r;
            }
        }
        __syncthreads();
        for (int t = tid; t < T; t += 512) {
            const float pc = lds[o_p + t] * lds[o_cp + t];
            const float al = pc * lds[o_q + t];
            lds[o_al + t] = al;
            if (a.align) a.align[((long long)n * T + t) * a.iters + it] = al;      // tacotron.py:223
        }
        for (int t = T + tid; t < Tp; t += 512) lds[o_al + t] = 0.0f;
        __syncthreads();
        TWV_STAMP(5)
        // ---- rnn_wrappers.py:390 context = alignments . values : cdot over t (AC-1 chunks of 32 time steps)
        if (tid < ENC) {
            float v = 0.0f;
            for (int t0 = 0; t0 < T; t0 += 32) {
                float s[4] = {0.f, 0.f, 0.f, 0.f};
                const int t1 = min(T, t0 + 32);
                for (int t = t0; t < t1; ++t) s[(t - t0) & 3] = fma_(memo[(long long)t * ENC + tid], lds[o_al + t], s[(t - t0) & 3]);
                const float c = (s[0] + s[1]) + (s[2] + s[3]);
                v = t0 == 0 ? c : v + c;
            }
            lds[o_ctx + tid] = v;
        }
        __syncthreads();
        TWV_STAMP(6)
        // ---- rnn_wrappers.py:463 concat(output, attention) -> OutputProjectionWrapper(dec_rnn)
        for (int i = tid; i < AS; i += 512) lds[o_cat + i] = lds[o_ha + i];
        for (int i = tid; i < ENC; i += 512) lds[o_cat + AS + i] = lds[o_ctx + i];
        __syncthreads();
        dec_gemv_partials(rs, (int)((a.w.cW) * 4), AS + ENC, DR, o_cat, o_part, wave, lane);
        __syncthreads();
        if (tid < DR) lds[o_y + tid] = dec_combine(o_part, AS + ENC, tid) + P[a.w.cb + tid];
        __syncthreads();
        TWV_STAMP(7)
        // ---- tacotron.py:167 ResidualWrapper(GRUCell(dec_rnn)): y <- y + GRU(y, h_l)
        for (int l = 0; l < a.layers; ++l) {
            for (int i = tid; i < DR; i += 512) { lds[o_cat + i] = lds[o_y + i]; lds[o_cat + DR + i] = lds[o_hr[l] + i]; }
            __syncthreads();
            dec_gru(rs, P, a.w.rWg[l], a.w.rbg[l], a.w.rWc[l], a.w.rbc[l], DR, DR, o_cat, o_hr[l], o_part, o_vec, tid, wave, lane);
            if (tid < DR) lds[o_y + tid] = lds[o_y + tid] + lds[o_hr[l] + tid];
            __syncthreads();
        }
        TWV_STAMP(8)
        // ---- tacotron.py:173 OutputProjectionWrapper(num_mels * r); helpers.py:40 last frame fed back
        dec_gemv_partials(rs, (int)((a.w.oW) * 4), DR, M * R, o_y, o_part, wave, lane);
        __syncthreads();
        if (tid < M * R) {
            const float v = dec_combine(o_part, DR, tid) + P[a.w.ob + tid];
            a.mel[((long long)n * a.iters + it) * M * R + tid] = v;                 // tacotron.py:204 reshape
            if (tid >= M * (R - 1)) lds[o_frame + tid - M * (R - 1)] = v;
        }
        __syncthreads();
        TWV_STAMP(9)
    }
#undef TWV_STAMP
}

// -----------------------------------------------------------------------------------------------------
//  decoder, G workgroups per utterance (the default when N*G workgroups fit the chip)
// -----------------------------------------------------------------------------------------------------
// The 1.6 M decoder weights (6.4 MB) do not fit one XCD's 4 MB L2, so one workgroup per utterance streams them from the
// memory side every step.  Here workgroup (n, g) owns the 64-column output blocks jb = g, g+G, ... of every wide matvec (32-column
// half blocks where a stage has fewer blocks than workgroups: DecgGeo).  Each workgroup keeps a full replica of the utterance's
// recurrent state in LDS; after a split matvec the activated outputs travel as {epoch, value} granules (one 8-byte store each,
// readers poll -- same mechanism as the WaveNet generation kernel), two alternating buffers: a workgroup can only publish exchange
// e+2 after it gathered e+1, which needs every workgroup to have finished reading e.
// Placement (DecGArgs::local):
//   local  (default): the G workgroups of an utterance sit on ONE XCD (XCD x: utterances x, x+8, ...; the XCD is read from the
//                     hardware, the role is a ticket of that XCD) and exchange through its L2 -- plain stores, sc1 loads; the
//                     weights stream through that L2.  The prenet and the query layer are split stages too (split_all).
//   spread:           blockIdx -> (n = id / G, g = id % G) puts slice g of the weights into XCD g's L2 (0.8 MB at G = 8) and
//                     every exchange crosses XCDs (agent-scope stores, a memory-side hop); prenet and query layer run whole in
//                     every workgroup.
// The attention recurrence runs redundantly in every workgroup.
// Arithmetic is unchanged: every output column's chunks are still summed in order inside one workgroup.
constexpr int kExN = 1024;          // granules per exchange buffer
typedef __attribute__((address_space(1))) unsigned long long tgu64;
typedef unsigned u32x2d __attribute__((ext_vector_type(2)));
typedef unsigned u32x4d __attribute__((ext_vector_type(4)));

struct DecGArgs {
    DecArgs d;
    int G;
    unsigned long long* exch;       // [N][2][kExN]
    int kv_lds;                     // 1: key rows / memory columns of the workgroup fit in LDS
    int local;                      // 1: the G workgroups of an utterance share one XCD and exchange through its L2
    int* tickets;                   // [8] role tickets of the local mode, zeroed before the launch
    int* stab;                      // [workgroups][16 stages][16]: every workgroup's copy of its stage table (read with scalar loads)
    int split_all;                  // 1: the prenet and the query layer are split over the workgroups too (three more exchanges, 300 KB
                                    //    less weight traffic per workgroup and step); 0: every workgroup computes them whole
};

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
// Which tiles of a stage a workgroup owns.  A tile is 64 columns x one 32-term chunk (lane = column).  A split stage with fewer
// column blocks than workgroups (N = 256 at G = 8: four blocks) would leave half of the utterance's workgroups idle while the other
// half streams and multiplies twice its share, so such a stage runs in PART-BLOCK form with sh = 1 or 2: workgroup g owns the
// 64 >> sh columns [(64 >> sh) g, ...) and a tile is (64 >> sh) columns x (1 << sh) chunks -- the wave's lanes are 1 << sh groups, group
// h runs chunk (i << sh) + h of the same columns (the same 8 KB per tile: part tiles of the standard image, 8192 bytes apart).
// Chunk values still meet in LDS in chunk order.
struct DecgGeo {
    int nchunk;     // 32-term chunks of the reduction
    int nct;        // tiles per column block (ceil(nchunk >> sh))
    int ntile;      // this workgroup's tiles
    int nmine;      // this workgroup's column blocks
    int gT, lgT;    // block index = (m << lgT) + gT
    int sh;         // 0 = whole 64-column blocks; 1, 2 = half / quarter blocks
};
__device__ __forceinline__ DecgGeo decg_geo(int K, int N, int split, int g, int lg)
{
    DecgGeo e;
    const int G = 1 << lg;
    e.nchunk = (K + 31) >> 5;
    const int nblk = (N + 63) >> 6;
    e.sh = 0;
    if (split && G > 1) while (e.sh < 2 && (nblk << (e.sh + 1)) <= G) ++e.sh;
    if (e.sh) {
        e.gT = g >> e.sh; e.lgT = 0;
        e.nct = (e.nchunk + (1 << e.sh) - 1) >> e.sh;
        e.nmine = (g >> e.sh) < nblk ? 1 : 0;
    } else {
        e.gT = split ? g : 0; e.lgT = split ? lg : 0;
        e.nct = e.nchunk;
        e.nmine = nblk > e.gT ? (nblk - e.gT + (1 << e.lgT) - 1) >> e.lgT : 0;
    }
    e.ntile = e.nmine * e.nct;
    return e;
}
__device__ __forceinline__ int decg_toff(int wt_bytes, const DecgGeo& e, const DecgPos& p)
{
    return wt_bytes + ((((p.m << e.lgT) + e.gT) * e.nchunk + (p.ch << e.sh)) << 13);
}
// per-lane byte offset inside a tile's image: lane group h = lane >> (6 - sh) reads part tile h, columns sub * (64 >> sh) + ...
__device__ __forceinline__ int decg_voff(const DecgGeo& e, int g, int lane)
{
    const int cw = 64 >> e.sh, sub = g & ((1 << e.sh) - 1);
    return ((lane >> (6 - e.sh)) << 13) + ((sub * cw + (lane & (cw - 1))) << 4);      // (sh = 0: lane << 4)
}
__device__ __forceinline__ int decg_geo_pack(const DecgGeo& e) { return e.sh | (e.lgT << 2) | (e.gT << 5) | (e.nmine << 13); }
__device__ __forceinline__ DecgGeo decg_geo_unpack(int K, int nct, int ntile, int geo)
{
    DecgGeo e;
    e.nchunk = (K + 31) >> 5; e.nct = nct; e.ntile = ntile;
    e.sh = geo & 3; e.lgT = (geo >> 2) & 7; e.gT = (geo >> 5) & 255; e.nmine = geo >> 13;
    return e;
}
// local block m, lane l: chunk values summed in order (AC-1); the LDS reads of four chunks are issued together
__device__ __forceinline__ float decg_combine(int o_part, int nchunk, int m, int l)
{
    const int b = o_part + m * nchunk * 64 + l;
    float v = lds[b];
    int ch = 1;
    for (; ch + 7 < nchunk; ch += 8) {
        const float c0 = lds[b + ch * 64], c1 = lds[b + (ch + 1) * 64], c2 = lds[b + (ch + 2) * 64], c3 = lds[b + (ch + 3) * 64];
        const float c4 = lds[b + (ch + 4) * 64], c5 = lds[b + (ch + 5) * 64], c6 = lds[b + (ch + 6) * 64], c7 = lds[b + (ch + 7) * 64];
        __builtin_amdgcn_sched_barrier(0);
        v = v + c0; v = v + c1; v = v + c2; v = v + c3; v = v + c4; v = v + c5; v = v + c6; v = v + c7;
    }
    for (; ch + 3 < nchunk; ch += 4) {
        const float c0 = lds[b + ch * 64], c1 = lds[b + (ch + 1) * 64], c2 = lds[b + (ch + 2) * 64], c3 = lds[b + (ch + 3) * 64];
        v = v + c0; v = v + c1; v = v + c2; v = v + c3;
    }
    for (; ch < nchunk; ++ch) v = v + lds[b + ch * 64];
    return v;
}
// all 512 threads: collect n values of exchange `epoch` into lds[o_dst ..)
__device__ __forceinline__ bool decg_gather(unsigned long long* X, int n, unsigned epoch, int o_dst, int tid, int o_abort)
{
    bool done0 = tid >= n, done1 = tid + 512 >= n;
    for (int it = 0; it < (1 << 20); ++it) {
        if (!done0) {
            const unsigned long long v = __hip_atomic_load((tgu64*)(X + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(v >> 32) == epoch) { lds[o_dst + tid] = __uint_as_float((unsigned)v); done0 = true; }
        }
        if (!done1) {
            const unsigned long long v = __hip_atomic_load((tgu64*)(X + tid + 512), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(v >> 32) == epoch) { lds[o_dst + tid + 512] = __uint_as_float((unsigned)v); done1 = true; }
        }
        if (__all(done0 && done1)) return true;
        if ((it & 63) == 63 && LDSVI(o_abort)) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    LDSVI(o_abort) = 1;
    return false;
}
// the same collection with the consumer inlined: f(j, value) runs in the thread that received element j, as soon as it arrives.  The
// split stages of tc_decoder_g_kernel finish their cell update there (r * h, the GRU blend, the residual add, the next stage's
// concatenation: all elementwise in j) instead of in a separate pass over LDS behind one more barrier.
template <class F>
__device__ __forceinline__ bool decg_gather_apply(unsigned long long* X, int n, unsigned epoch, int tid, int o_abort, F f)
{
    bool done0 = tid >= n, done1 = tid + 512 >= n;
    for (int it = 0; it < (1 << 20); ++it) {
        if (!done0) {
            const unsigned long long v = __hip_atomic_load((tgu64*)(X + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(v >> 32) == epoch) { f(tid, __uint_as_float((unsigned)v)); done0 = true; }
        }
        if (!done1) {
            const unsigned long long v = __hip_atomic_load((tgu64*)(X + tid + 512), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(v >> 32) == epoch) { f(tid + 512, __uint_as_float((unsigned)v)); done1 = true; }
        }
        if (__all(done0 && done1)) return true;
        if ((it & 63) == 63 && LDSVI(o_abort)) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    LDSVI(o_abort) = 1;
    return false;
}
// local: writer and readers sit on ONE XCD -- a plain store leaves the line in that XCD's L2, where the readers' sc1 loads (L1
// bypassed) find it (the recipe of twv_wavenet_xcd.hip); otherwise an agent-scope store writes through to the memory side
__device__ __forceinline__ void decg_store(unsigned long long* p, unsigned epoch, float v, bool local)
{
    const unsigned long long q = ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(v);
    if (local) { *(tgu64*)p = q; asm volatile("" ::: "memory"); }
    else __hip_atomic_store((tgu64*)p, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// running sums over lds[o .. o+T) in index order (one add chain), one wave, staged through registers; eight v_readlane
// are issued ahead of the dependent adds (padding lanes hold 0 and their results are never stored)
__device__ __forceinline__ void decg_scan(int o, int T, int lane, bool inclusive)
{
    float run = 0.0f;
    for (int base = 0; base < T; base += 64) {
        const int n = T - base < 64 ? T - base : 64;
        const float v = (base + lane < T) ? lds[o + base + lane] : 0.0f;
        float res = 0.0f;
        for (int i = 0; i < n; i += 8) {
            float x[8], r[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (i + k) & 63));
            float prev = run;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                r[k] = prev + x[k];
                if (lane == i + k) res = inclusive ? r[k] : prev;
                prev = r[k];
            }
            run = prev;
        }
        if (base + lane < T) lds[o + base + lane] = res;
    }
}

// The decoder step is a table-driven sequence of matvec stages, so the tile-streaming code exists ONCE (inlined in the stage
// loop): twelve inlined copies made the register allocator spill ~1000 VGPRs, and a real call costs ~4000 cycles because the
// callee saves its VGPRs to scratch (scripts/ubench/gemv_call.hip).
// one stage = one 64-byte record: every wave fetches it with three 16-byte LDS reads (thirteen dword reads, each with its own
// wait in front of the v_readfirstlane, were 0.3 us of every stage)
enum { DS_W = 0, DS_BIAS, DS_K, DS_N, DS_X, DS_DST, DS_ACT, DS_SPLIT, DS_POST, DS_P0, DS_P1, DS_P2, DS_BIASG, DS_NCT, DS_NTILE, DS_GEO, DS_STRIDE = 16 };
// DS_NCT, DS_NTILE, DS_GEO: this workgroup's tile geometry of the stage (DecgGeo), worked out once when the table is written
// (for the GRU stages DS_X is also the buffer the cell update works in: [x | h] -> [x | r*h])
typedef int i32x4s __attribute__((ext_vector_type(4)));
#define LDS4I(off4) (((__attribute__((address_space(3))) i32x4s*)lds)[(off4)])
enum { DP_NONE = 0, DP_CAT_ATT, DP_GATES, DP_CAND, DP_QUERY, DP_PROJ, DP_OUT };
enum { DA_NONE = 0, DA_SIGMOID, DA_TANH, DA_RELU };

// DEF: the hparams-default decoder sizes (hparams.py:126-158: num_mels 80, r 5, prenet 256 / 128, attention 256, rnn sizes 256, two
// residual layers) with 8 workgroups per utterance as compile-time constants.  The kernel is short of scalar registers (a table-driven
// loop over run-time sizes keeps ~100 scalars alive: 85-105 SGPR spills, ~40 v_readlane / v_writelane per stage); with the sizes
// folded the LDS carve and most index arithmetic are constants.  Any other shape runs the same code with DEF = false.
// A stage's 64-byte record straight into scalar registers (s_load from the workgroup's copy of the table in global memory, served by
// the scalar cache): the LDS copy cost four ds_read_b128, their wait and sixteen v_readfirstlane at every stage start, and again
// twice per stage for the next stage's tile requests -- 0.3 us of a 2.4 us stage.
typedef int i32x16s __attribute__((ext_vector_type(16)));
__device__ __forceinline__ i32x16s decg_sload16(const int* p)
{
    i32x16s r;
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
    return r;
}
// words 0-3 and 12-15 of a record (what a tile request needs: weights, K, tile geometry)
__device__ __forceinline__ void decg_sload_geo(const int* p, i32x4s& lo, i32x4s& hi)
{
    asm volatile("s_load_dwordx4 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0x30\n\ts_waitcnt lgkmcnt(0)" : "=&s"(lo), "=&s"(hi) : "s"(p) : "memory");
}
template <bool PROF, bool DEF>
__global__ void __launch_bounds__(512) tc_decoder_g_kernel(DecGArgs ga)
{
    const DecArgs& a = ga.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lg = DEF ? 3 : (ga.G == 16 ? 4 : (ga.G == 8 ? 3 : (ga.G == 4 ? 2 : (ga.G == 2 ? 1 : 0)))), G = 1 << lg;
    // workgroup -> (utterance n, slice g):
    //   spread (local = 0): n = id / G, g = id % G.  Workgroups go to the XCDs round-robin, so slice g of the weights lives in XCD
    //                       g's L2 -- and every exchange of the step crosses XCDs (a memory-side hop)
    //   local  (local = 1): XCD x hosts the utterances x, x + 8, ... with all their slices: exchanges stay inside one L2 (plain
    //                       stores, sc1 loads), the weights stream through it (6.4 MB through 4 MB: the memory-side cache feeds
    //                       them, four utterances per fetch).  The XCD is read from the hardware and the role is a ticket of that
    //                       XCD, so the placement does not rest on the dispatcher's order.
    const bool loc = ga.local != 0;
    int n = blockIdx.x >> lg, g = blockIdx.x & (G - 1);
    if (loc) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 15u;
        __shared__ int s_ticket;
        if (tid == 0) s_ticket = xcc < 8u ? atomicAdd(ga.tickets + xcc, 1) : 1 << 20;
        __syncthreads();
        const int slot = __builtin_amdgcn_readfirstlane(s_ticket);
        g = slot & (G - 1);
        n = slot < (1 << 20) ? ((slot >> lg) << 3) + (int)xcc : a.N;
    }
    const int T = a.T, M = DEF ? 80 : a.M, R = DEF ? 5 : a.R, A = DEF ? 256 : a.A, AS = DEF ? 256 : a.AS, ENC = DEF ? 256 : a.ENC;
    const int DR = DEF ? 256 : a.DR, D0 = DEF ? 256 : a.D0, D1 = DEF ? 128 : a.D1, NL = DEF ? 2 : a.layers;
    // model_type 'simple' (tacotron.py:85-90): the utterance's speaker embedding (SEc values) sits between the prenet output and the context in
    // the attention cell's input (rnn_wrappers.py:429-430) and behind [output, attention] in the first projection's input (:458-460)
    const int SEc = DEF ? 0 : a.SEc, DE = D1 + SEc;
    if (n >= a.N) return;
    const int len = a.lengths[n];
    const float* P = a.P;
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.P), 0, (int)a.packed_bytes, 0x00020000);
    const float* keys = a.keys + (long long)n * T * A;
    const float* memo = a.memo + (long long)n * T * ENC;
    unsigned long long* X = ga.exch + (long long)n * 2 * kExN;
    // ---- LDS carve
    int o = 0;
    const int o_cat = o; o += 1024;
    // the attention GRU's input [prenet_out | attention | ha] has a buffer of its own: its three parts are written where they are
    // produced (the prenet's epilogue, the context's gather of the PREVIOUS step, the attention cell's update), not copied together
    // behind a barrier at the start of every step
    const int o_cat2 = o; o += ((DE + ENC + AS + 63) / 64) * 64;
    const int o_vec = o; o += 1024;                       // gathered gates (r | u), prenet hidden
    const int o_cand = o; o += 512;
    const int o_keep = o; o += 512;                       // h before the update
    const int o_ha = o; o += AS;
    const int o_hr0 = o; o += NL * DR;              // residual GRU states, layer l at o_hr0 + l*DR
    const int o_frame = o; o += ((M + 31) / 32) * 32;
    const int o_ctx = o; o += ENC;
    const int o_y = o; o += DR;
    const int o_out = o; o += ((M * R + 63) / 64) * 64;
    // (areas whose size does not depend on the input length first: with the default sizes folded in, their offsets are constants)
    const int o_pq = o; o += A + A / 8;                   // attention tables are skewed by 4 words per 32 (see the score phase)
    const int o_abort = o; o += 4;
    o = (o + 3) & ~3;
    const int o_tab = o; o += 16 * DS_STRIDE;             // stage table (16-byte aligned records)
    const int o_nv = o; o += A + A / 8;                   // normed_v, attention bias
    const int o_ab = o; o += A + A / 8;
    const int o_bias = o; o += DEF ? 256 + 128 + 3 * 256 + 256 + 2 * 3 * 256 + 400 : a.nbias;   // every stage's bias vector, in stage order
    const int Tp = ((T + 3) / 4) * 4;
    const int o_al = o; o += Tp;
    const int o_p = o; o += Tp;
    const int o_cp = o; o += Tp;
    const int o_q = o; o += Tp;
    const int o_scp = o; o += Tp * 8;
    const bool kv = ga.kv_lds != 0;                       // this workgroup's key rows / memory columns held in LDS
    const int nt_all = T > g ? (T - g + G - 1) >> lg : 0;
    const int o_keys = o; o += kv ? ((T + G - 1) >> lg) * (A + A / 8) : 0;
    const int o_memo = o; o += kv ? T * ((ENC >> lg) + 8) : 0;   // rows 8 words apart in the banks: the context's chains k = t mod 4 read four rows at once
    const int o_part = o;

    const float* init = a.init + (long long)n * (AS + NL * DR);
    for (int i = tid; i < AS; i += 512) lds[o_ha + i] = init[i];
    for (int i = tid; i < NL * DR; i += 512) lds[o_hr0 + i] = init[AS + i];
    for (int i = tid; i < ((M + 31) / 32) * 32; i += 512) lds[o_frame + i] = 0.0f;
    for (int i = tid; i < ENC; i += 512) { lds[o_ctx + i] = 0.0f; lds[o_cat2 + DE + i] = 0.0f; }
    for (int i = tid; i < AS; i += 512) lds[o_cat2 + DE + ENC + i] = init[i];
    for (int i = tid; i < SEc; i += 512) { const float e = a.semb[(long long)n * SEc + i]; lds[o_cat2 + D1 + i] = e; lds[o_cat + AS + ENC + i] = e; }
    for (int i = tid; i < Tp; i += 512) lds[o_al + i] = i == 0 ? 1.0f : 0.0f;
    if (tid < 4) LDSI(o_abort + tid) = 0;
    for (int i = tid; i < A; i += 512) { lds[o_nv + i + ((i >> 5) << 2)] = P[a.w.nv + i]; lds[o_ab + i + ((i >> 5) << 2)] = P[a.w.ab + i]; }
    if (kv) {
        // (every 32-word chunk of the attention tables is skewed by 4 words: the score's threads -- one per (t, chunk, chain) -- read
        // addresses 32 and 128 words apart, i.e. all in four of the 32 LDS banks without the skew: the phase was these reads)
        for (int i = tid; i < nt_all * A; i += 512) { const int tl = i / A, j = i - tl * A; lds[o_keys + i + ((i >> 5) << 2)] = keys[(long long)((tl << lg) + g) * A + j]; }
        const int ncol = ENC >> lg;
        for (int i = tid; i < T * ncol; i += 512) { const int t = i / ncol, cl = i - t * ncol; lds[o_memo + t * (ncol + 8) + cl] = memo[(long long)t * ENC + g * ncol + cl]; }
    }
    const int nst = 7 + 2 * NL;
    if (tid == 0) {
        int s = 0, bo = o_bias;
        auto put = [&](long long w, long long b, int K, int N, int x, int dst, int act, int split, int post, int p0, int p1, int p2) {
            const int q = o_tab + s * DS_STRIDE;
            LDSI(q + DS_W) = (int)(w * 4); LDSI(q + DS_BIAS) = b >= 0 ? bo : -1; LDSI(q + DS_BIASG) = (int)b;
            if (b >= 0) bo += N; LDSI(q + DS_K) = K; LDSI(q + DS_N) = N; LDSI(q + DS_X) = x;
            LDSI(q + DS_DST) = dst; LDSI(q + DS_ACT) = act; LDSI(q + DS_SPLIT) = split; LDSI(q + DS_POST) = post;
            LDSI(q + DS_P0) = p0; LDSI(q + DS_P1) = p1; LDSI(q + DS_P2) = p2;
            const DecgGeo e = decg_geo(K, N, split, g, lg);
            LDSI(q + DS_NCT) = e.nct; LDSI(q + DS_NTILE) = e.ntile; LDSI(q + DS_GEO) = decg_geo_pack(e);
            ++s;
        };
        // rnn_wrappers.py:425 decoder prenet (redundant in every workgroup: 28 tiles)
        put(a.w.dp1, a.w.dp1b, M, D0, o_frame, o_vec, DA_RELU, ga.split_all, DP_NONE, 0, 0, 0);
        put(a.w.dp2, a.w.dp2b, D0, D1, o_vec, o_cat2, DA_RELU, ga.split_all, DP_NONE, 0, 0, 0);
        // rnn_wrappers.py:310-312 attention GRU on [prenet_out | attention | ha]
        put(a.w.aWg, a.w.abg, DE + ENC + AS, 2 * AS, o_cat2, o_vec, DA_SIGMOID, 1, DP_GATES, DE + ENC, AS, 0);
        put(a.w.aWc, a.w.abc, DE + ENC + AS, AS, o_cat2, o_cand, DA_TANH, 1, DP_CAND, AS, o_ha, -1);
        // attention query layer (redundant: 32 tiles), then score / recurrence / context
        put(a.w.Wq, -1, AS, A, o_ha, o_pq, DA_NONE, ga.split_all, DP_QUERY, 0, 0, 0);
        // rnn_wrappers.py:463 concat(output, attention) -> OutputProjectionWrapper(dec_rnn)
        put(a.w.cW, a.w.cb, AS + ENC + SEc, DR, o_cat, o_y, DA_NONE, 1, DP_PROJ, 0, 0, 0);
        // tacotron.py:167 ResidualWrapper(GRUCell(dec_rnn)): y <- y + GRU(y, h_l)
        for (int l = 0; l < NL; ++l) {
            put(a.w.rWg[l], a.w.rbg[l], 2 * DR, 2 * DR, o_cat, o_vec, DA_SIGMOID, 1, DP_GATES, DR, DR, 0);
            put(a.w.rWc[l], a.w.rbc[l], 2 * DR, DR, o_cat, o_cand, DA_TANH, 1, DP_CAND, DR, o_hr0 + l * DR, l + 1 < NL ? o_hr0 + (l + 1) * DR : 0);
        }
        // tacotron.py:173 OutputProjectionWrapper(num_mels * r)
        put(a.w.oW, a.w.ob, DR, M * R, o_y, o_out, DA_NONE, 1, DP_OUT, 0, 0, 0);
    }
    __syncthreads();
    for (int st = 0; st < nst; ++st) {                    // bias vectors -> LDS
        const int q = o_tab + st * DS_STRIDE, bl = LDSI(q + DS_BIAS), bg = LDSI(q + DS_BIASG), N = LDSI(q + DS_N);
        if (bl >= 0) for (int i = tid; i < N; i += 512) lds[bl + i] = P[bg + i];
    }
    // the stage table, once more, where scalar loads can reach it
    int* const stab = ga.stab + (long long)blockIdx.x * (16 * DS_STRIDE);
    for (int e = tid; e < 16 * DS_STRIDE; e += 512) stab[e] = LDSI(o_tab + e);
    __threadfence();
    __syncthreads();
    __builtin_amdgcn_s_dcache_inv();
    const int nAch = A / 32;
    unsigned ep = 0;                                      // exchanges completed so far
    bool ok = true;
    Tile t0, t1, t2;
    // request the first three tiles of stage `sn` for this wave, in two parts (see the stage loop): part A = the first tile, part B =
    // the other two
#define DECG_PREFETCH_(sn, A_, B_)                                                                                              \
    {                                                                                                                            \
        i32x4s n0_, n3_;                                                                                                         \
        decg_sload_geo(stab + (sn) * DS_STRIDE, n0_, n3_);                                                                       \
        const int wn_ = n0_.x, Kn_ = n0_.z;                                                                                      \
        const DecgGeo en_ = decg_geo_unpack(Kn_, n3_.y, n3_.z, n3_.w);                                                           \
        const int von_ = decg_voff(en_, g, lane);                                                                                \
        DecgPos q0_{0, 0}, q1_, q2_;                                                                                             \
        decg_adv(q0_, wave, en_.nct);                                                                                            \
        q1_ = q0_; decg_adv(q1_, 8, en_.nct);                                                                                    \
        q2_ = q1_; decg_adv(q2_, 8, en_.nct);                                                                                    \
        if ((A_) && wave < en_.ntile) load_tile_b(t0, rs, von_, decg_toff(wn_, en_, q0_));                                       \
        if ((B_) && wave + 8 < en_.ntile) load_tile_b(t1, rs, von_, decg_toff(wn_, en_, q1_));                                   \
        if ((B_) && wave + 16 < en_.ntile) load_tile_b(t2, rs, von_, decg_toff(wn_, en_, q2_));                                  \
    }
#define DECG_PREFETCH(sn) DECG_PREFETCH_(sn, true, true)
    DECG_PREFETCH(0)
// instrumented build (its own instantiation): workgroup 0, thread 0 stamps s_memtime (calibrated against the launch's event time) into prof[it][64]:
//   4*st + 0 stage start | + 1 this wave's tile dots done | + 2 chunk sums, bias, activation, publish done | + 3 gathered + barrier
//   48 score chunk dots done | 49 p gathered | 50 recurrence done | 51 context partials done | 52 context gathered | 53 end of the step
#define TWV_STAMP(k) if (PROF && a.prof && blockIdx.x == 0 && tid == 0) a.prof[it * 64 + (k)] = __builtin_amdgcn_s_memtime();

    for (int it = 0; it < a.iters && ok; ++it) {
        for (int st = 0; st < nst && ok; ++st) {
            const i32x16s rec = decg_sload16(stab + st * DS_STRIDE);
            const int w_bytes = rec[DS_W], bias = rec[DS_BIAS], K = rec[DS_K], N = rec[DS_N], xo = rec[DS_X], dst = rec[DS_DST];
            const int act = rec[DS_ACT], split = rec[DS_SPLIT], post = rec[DS_POST], sp0 = rec[DS_P0], sp1 = rec[DS_P1], sp2 = rec[DS_P2];
            const DecgGeo e = decg_geo_unpack(K, rec[DS_NCT], rec[DS_NTILE], rec[DS_GEO]);
            const int nchunk = e.nchunk, nmine = e.nmine, ntile = e.ntile;
            TWV_STAMP(4 * st + 0)
            // ---- this workgroup's tiles: wave w takes local tiles w, w+8, ... (three in flight), partials to LDS.  The first three
            // were requested while the previous stage was still combining / exchanging (weights do not depend on data).
            {
                const int vo = decg_voff(e, g, lane);
                const int hi = lane >> (6 - e.sh);                         // part-block form: lane group h runs chunk (i << sh) + h
                const int pl = lane & ((64 >> e.sh) - 1);                  // column inside the partial row
                DecgPos p0{0, 0}, p1, p2;
                decg_adv(p0, wave, e.nct);
                p1 = p0; decg_adv(p1, 8, e.nct);
                p2 = p1; decg_adv(p2, 8, e.nct);
#define DECG_DOT(T_, P_, I_)                                                                                                    \
                    {                                                                                                            \
                        const int c_ = (P_.ch << e.sh) + hi;                      /* this lane's chunk */                        \
                        const int xq = xo + c_ * 32 + (lane & 15);                                                               \
                        const float r = dot32_dpp(T_.w, lds[xq], lds[xq + 16]);                                                  \
                        __builtin_amdgcn_sched_barrier(0);                                                                       \
                        if (c_ < nchunk) lds[o_part + (P_.m * nchunk + c_) * 64 + pl] = r;                                       \
                        decg_adv(P_, 24, e.nct);                                                                                 \
                        if ((I_) + 24 < ntile) load_tile_b(T_, rs, vo, decg_toff(w_bytes, e, P_));                               \
                    }
                for (int i = wave; i < ntile; i += 24) {
                    DECG_DOT(t0, p0, i)
                    if (i + 8 < ntile) DECG_DOT(t1, p1, i + 8)
                    if (i + 16 < ntile) DECG_DOT(t2, p2, i + 16)
                }
#undef DECG_DOT
                TWV_STAMP(4 * st + 1)
            }
            // A CU's vector-memory pipeline is one FIFO of ~64 B/clk: a stage's tiles (up to 160 KB per workgroup) take ~1 us to pass
            // through it, and everything issued behind them -- the publish stores of the epilogue, the polls of the gather -- waits its
            // turn.  So the wave's FIRST tile of the next stage is requested here, where the chunk sums and activations (LDS and VALU
            // only) cover it, and the other two after the publish.  All three here delayed every publish (36.2 -> 40.4 us per step,
            // round 4); all three after the publish left the gather waiting for 1.1 us of tile traffic behind a 0.4 us hop.
            DECG_PREFETCH_(st + 1 < nst ? st + 1 : 0, true, false)
            lds_barrier();                                // partials visible (LDS only: the tile request stays in flight)
            // ---- epilogue: chunk sums in order (AC-1) + bias + activation; split stages publish and all-gather
            {
                const bool xch = split && G > 1;
                unsigned long long* Xb = X;
                if (xch) { ++ep; Xb = X + (ep & 1) * kExN; }
                const int ncolw = 64 >> e.sh;                              // columns per owned block
                for (int qq = tid; qq < nmine * ncolw; qq += 512) {
                    const int m = e.sh ? 0 : qq >> 6;
                    const int j = e.sh ? (e.gT << 6) + (g & ((1 << e.sh) - 1)) * ncolw + qq : (((m << e.lgT) + e.gT) << 6) + (qq & 63);
                    if (j < N) {
                        float v = decg_combine(o_part, nchunk, m, qq & 63);
                        if (bias >= 0) v = v + lds[bias + j];
                        if (act == DA_SIGMOID) v = sigmoid_e(v);
                        else if (act == DA_TANH) v = tanh_e(v);
                        else if (act == DA_RELU) v = v > 0.0f ? v : 0.0f;
                        if (xch) decg_store(Xb + j, ep, v, loc); else lds[dst + (post == DP_QUERY ? j + ((j >> 5) << 2) : j)] = v;
                    }
                }
                TWV_STAMP(4 * st + 2)
                // the rest of the next stage's first tiles: this workgroup's values are already on their way to the others
                DECG_PREFETCH_(st + 1 < nst ? st + 1 : 0, false, true)
                if (xch) {
                    // a split stage's values arrive one per thread: the cell update that follows the matvec is elementwise in the
                    // output index, so the receiving thread does it on arrival (no second pass over LDS, no second barrier)
                    const int p0 = sp0, p1 = sp1, p2 = sp2;
                    decg_gather_apply(Xb, N, ep, tid, o_abort, [&](const int j, const float v) {
                        if (post == DP_GATES) {              // tf.contrib.rnn.GRUCell, gate order r | u: keep h, cat <- [x, r*h]   (p0 = nin, p1 = U)
                            lds[dst + j] = v;
                            if (j < p1) {
                                const float h = lds[xo + p0 + j];
                                lds[o_keep + j] = h;
                                lds[xo + p0 + j] = v * h;
                            }
                        } else if (post == DP_CAND) {        // h <- u*h + (1-u)*c   (p0 = U, p1 = where h lives, p2 = the next layer's h / 0 / -1)
                            const float u = lds[o_vec + p0 + j], h = lds[o_keep + j];
                            const float t1 = u * h, t2 = 1.0f - u, t3 = t2 * v;
                            const float hn = t1 + t3;
                            lds[p1 + j] = hn;
                            if (p2 >= 0) {                   // residual layer: y <- y + h ; next layer's input [y | h_next]
                                const float yn = lds[o_y + j] + hn;
                                lds[o_y + j] = yn;
                                if (p2 > 0) { lds[o_cat + j] = yn; lds[o_cat + DR + j] = lds[p2 + j]; }
                            } else {                         // the attention cell: its new state is the tail of its own next input and
                                lds[xo + DE + ENC + j] = hn; // the head of the concat projection's input (rnn_wrappers.py:463)
                                lds[o_cat + j] = hn;
                            }
                        } else if (post == DP_PROJ) {
                            lds[o_y + j] = v; lds[o_cat + j] = v; lds[o_cat + DR + j] = lds[o_hr0 + j];
                        } else if (post == DP_OUT) {         // tacotron.py:204 reshape; helpers.py:40 last frame fed back
                            if (g == 0) a.mel[((long long)n * a.iters + it) * M * R + j] = v;
                            if (j >= M * (R - 1)) lds[o_frame + j - M * (R - 1)] = v;
                        } else if (post == DP_QUERY) {       // the processed query goes into its skewed table
                            lds[dst + j + ((j >> 5) << 2)] = v;
                        } else {
                            lds[dst + j] = v;
                        }
                    });
                }
                __syncthreads();
                TWV_STAMP(4 * st + 3)
                ok = LDSI(o_abort) == 0;
                if (xch && post != DP_QUERY && post != DP_NONE) {
                    if (post == DP_OUT) { TWV_STAMP(53) }
                    continue;
                }
            }
            // ---- what follows the matvec (stages that were not exchanged: one workgroup per utterance, the redundant stages)
            if (post == DP_GATES) {               // tf.contrib.rnn.GRUCell, gate order r | u: keep h, cat <- [x, r*h]
                const int nin = sp0, U = sp1;
                for (int i = tid; i < U; i += 512) {
                    const float h = lds[xo + nin + i];
                    lds[o_keep + i] = h;
                    lds[xo + nin + i] = lds[o_vec + i] * h;
                }
                __syncthreads();
            } else if (post == DP_CAND) {                // h <- u*h + (1-u)*c
                const int U = sp0, o_h = sp1, o_next = sp2;
                for (int i = tid; i < U; i += 512) {
                    const float c = lds[o_cand + i], u = lds[o_vec + U + i], h = lds[o_keep + i];
                    const float t1 = u * h, t2 = 1.0f - u, t3 = t2 * c;
                    const float hn = t1 + t3;
                    lds[o_h + i] = hn;
                    if (o_next >= 0) {                   // residual layer: y <- y + h ; next layer's input [y | h_next]
                        const float yn = lds[o_y + i] + hn;
                        lds[o_y + i] = yn;
                        if (o_next > 0) { lds[o_cat + i] = yn; lds[o_cat + DR + i] = lds[o_next + i]; }
                    } else {
                        lds[xo + DE + ENC + i] = hn;
                        lds[o_cat + i] = hn;
                    }
                }
                __syncthreads();
            } else if (post == DP_PROJ) {
                for (int i = tid; i < DR; i += 512) { lds[o_cat + i] = lds[o_y + i]; lds[o_cat + DR + i] = lds[o_hr0 + i]; }
                __syncthreads();
            } else if (post == DP_OUT) {                 // tacotron.py:204 reshape; helpers.py:40 last frame fed back
                for (int i = tid; i < M * R; i += 512) {
                    const float v = lds[o_out + i];
                    if (g == 0) a.mel[((long long)n * a.iters + it) * M * R + i] = v;
                    if (i >= M * (R - 1)) lds[o_frame + i - M * (R - 1)] = v;
                }
                __syncthreads();
                TWV_STAMP(53)
            } else if (post == DP_QUERY) {
                // [RECALLED-TF BahdanauMonotonicAttention.__call__] score for the time steps t = g, g+G, ...: one (t, chunk) per thread
                const int nt = T > g ? (T - g + G - 1) >> lg : 0;
                // one thread per (t, chunk, chain k): s_k = fma chain over j = k, k+4, ..., k+28; the four chains of a chunk sit in
                // adjacent lanes and are combined as (s0+s1)+(s2+s3)
                for (int task0 = 0; task0 < nt * nAch * 4; task0 += 512) {
                    const int task = task0 + tid;
                    const bool live = task < nt * nAch * 4;
                    const int k = task & 3, tc = task >> 2;
                    const int tl = live ? tc / nAch : 0, ch = live ? tc - tl * nAch : 0, t = (tl << lg) + g;
                    float sk = 0.f;
                    const float* kr = keys + (long long)t * A + ch * 32 + k;
                    const int jb = ch * 32 + k, js = jb + 4 * ch, kl = o_keys + tl * A + jb + 4 * (tl * nAch + ch);      // js, kl: skewed
                    TWV_STAMP(54)
                    if (kv) {
#pragma unroll
                        for (int j = 0; j < 32; j += 4) sk = fma_(lds[o_nv + js + j], tanh_e((lds[kl + j] + lds[o_pq + js + j]) + lds[o_ab + js + j]), sk);
                    } else {
#pragma unroll
                        for (int j = 0; j < 32; j += 4) sk = fma_(lds[o_nv + js + j], tanh_e((kr[j] + lds[o_pq + js + j]) + lds[o_ab + js + j]), sk);
                    }
                    TWV_STAMP(55)
                    const float s1 = __shfl_xor(sk, 1);
                    const float pr = (k & 1) ? s1 + sk : sk + s1;       // lanes k=0,1 hold s0+s1 ; k=2,3 hold s2+s3 (operand order as written)
                    const float p2 = __shfl_xor(pr, 2);
                    if (live && k == 0) lds[o_scp + tl * 8 + ch] = pr + p2;
                }
                TWV_STAMP(56)
                __syncthreads();
                TWV_STAMP(48)
                {
                    ++ep;
                    unsigned long long* Xb = X + (ep & 1) * kExN;
                    for (int tl = tid; tl < nt; tl += 512) {
                        const int t = (tl << lg) + g;
                        float sc = 0.0f;
                        for (int ch = 0; ch < nAch; ++ch) { const float c = lds[o_scp + tl * 8 + ch]; sc = ch == 0 ? c : sc + c; }
                        sc = sc + P[a.w.asb];
                        const float pv = t < len ? sigmoid_e(sc) : 0.0f;       // _maybe_mask_score(-inf) -> p = 0
                        if (G == 1) lds[o_p + t] = pv; else decg_store(Xb + t, ep, pv, loc);
                    }
                    if (G > 1) decg_gather(Xb, T, ep, o_p, tid, o_abort);
                    __syncthreads();
                    ok = LDSI(o_abort) == 0;
                }
                TWV_STAMP(49)
                // monotonic attention recurrence (redundant in every workgroup), all in wave 0 and in registers:
                // cumprod(1 - p) as exp(exclusive cumsum(log(clip(1 - p)))) [RECALLED-TF safe_cumprod], then
                // alignments = p * cumprod * inclusive cumsum(previous / clip(cumprod, 1e-10, 1))
                // The 64-step blocks of AC-6 run side by side, one wave each (eight blocks per round): a block needs from its
                // predecessors only the running total in front of its own scan, and that is the same chain of adds -- total of block
                // 0, + block 1's own total, ... -- whether the blocks are scanned one after the other or their totals are summed
                // afterwards.  (One wave walking the blocks: 1.02 us at T = 101, most of it the log / exp / division of block 2 waiting
                // behind block 1's.)
                {
                    const int o_tot = o_scp;                               // block totals (the score chunk values are consumed)
                    float run = 0.0f, run2 = 0.0f;                         // totals in front of this round's first block
                    for (int base0 = 0; base0 < T; base0 += 512) {
                        const int base = base0 + wave * 64, t = base + lane;
                        const bool mine = base < T, live = t < T;
                        const int nblk = (T - base0 + 63) >> 6 < 8 ? (T - base0 + 63) >> 6 : 8;
                        const float pv = live ? lds[o_p + t] : 0.0f;
                        float om = 1.0f - pv;
                        const float tiny = 1.17549435e-38f;
                        om = om < tiny ? tiny : (om > 1.0f ? 1.0f : om);
                        const float lq = live ? log_e(om) : 0.0f;
                        const float sc = scan64_f32_wave(lq);
                        if (mine && lane == 63) lds[o_tot + wave] = sc;
                        __syncthreads();
                        float carry = run, all = run;                      // this block's carry; the total after the round's last block
                        for (int b = 0; b < nblk; ++b) {
                            const float tb = lds[o_tot + b];
                            const float nx = (base0 == 0 && b == 0) ? tb : all + tb;
                            if (b < wave) carry = nx;
                            all = nx;
                        }
                        run = all;
                        const bool first = base0 == 0 && wave == 0;
                        const float incl = first ? sc : carry + sc;
                        const float up = __shfl_up(incl, 1);
                        const float ex = lane == 0 ? (first ? 0.0f : carry) : up;
                        const float cpv = exp_e(ex);
                        float den = cpv;
                        den = den < 1e-10f ? 1e-10f : (den > 1.0f ? 1.0f : den);
                        const float q2 = live ? div_(lds[o_al + t], den) : 0.0f;
                        const float sc2 = scan64_f32_wave(q2);
                        __syncthreads();                                   // (the first totals have been read)
                        if (mine && lane == 63) lds[o_tot + wave] = sc2;
                        __syncthreads();
                        float carry2 = run2, all2 = run2;
                        for (int b = 0; b < nblk; ++b) {
                            const float tb = lds[o_tot + b];
                            const float nx = (base0 == 0 && b == 0) ? tb : all2 + tb;
                            if (b < wave) carry2 = nx;
                            all2 = nx;
                        }
                        run2 = all2;
                        const float cs = first ? sc2 : carry2 + sc2;
                        if (live) {
                            const float pc = pv * cpv;
                            const float al = pc * cs;
                            lds[o_al + t] = al;
                            if (a.align && g == 0) a.align[((long long)n * T + t) * a.iters + it] = al;      // tacotron.py:223
                        }
                        __syncthreads();                                   // (the second totals have been read)
                    }
                    for (int t = T + tid; t < Tp; t += 512) lds[o_al + t] = 0.0f;
                }
                __syncthreads();
                TWV_STAMP(50)
                // rnn_wrappers.py:390 context = alignments . values: workgroup g takes the columns [g*ENC/G, (g+1)*ENC/G) -- its XCD's
                // L2 then only ever sees that column slice of the encoder memory -- one thread per (column, 32-step chunk), all-gather
                {
                    const int nch = (T + 31) / 32, ncol = ENC >> lg, c0 = g * ncol;
                    // one thread per (column, 32-step chunk, CHAIN k): the chunk's four interleaved fma chains (AC-1) sit in four adjacent
                    // lanes, eight dependent fmas each instead of thirty-two in one thread (1.8 us of the step when a thread ran all four),
                    // and meet as (s0+s1)+(s2+s3) through two lane exchanges -- the same adds in the same order
                    for (int task0 = 0; task0 < ncol * nch * 4; task0 += 512) {
                        const int task = task0 + tid;
                        const bool live = task < ncol * nch * 4;
                        const int k = task & 3, tc = live ? task >> 2 : 0;
                        const int cl = tc % ncol, ch = tc / ncol;
                        const int ta = ch * 32, tb = T < ta + 32 ? T : ta + 32;
                        float sk = 0.f;
                        if (kv) {
                            const int mo = o_memo + cl;
                            for (int t = ta + k; t < tb; t += 4) sk = fma_(lds[mo + t * (ncol + 8)], lds[o_al + t], sk);
                        } else {
                            const float* mp = memo + c0 + cl;
                            for (int t = ta + k; t < tb; t += 4) sk = fma_(mp[(long long)t * ENC], lds[o_al + t], sk);
                        }
                        const float s1 = __shfl_xor(sk, 1);
                        const float pr = (k & 1) ? s1 + sk : sk + s1;       // lanes k = 0,1 hold s0+s1 ; k = 2,3 hold s2+s3 (operand order as written)
                        const float p2 = __shfl_xor(pr, 2);
                        if (live && k == 0) lds[o_part + ch * ncol + cl] = pr + p2;
                    }
                    __syncthreads();
                    TWV_STAMP(51)
                    ++ep;
                    unsigned long long* Xb = X + (ep & 1) * kExN;
                    // the context goes where it is read: the concat projection's input [ha | context] of this step and the attention
                    // cell's input [prenet | context | ha] of the NEXT step (AttentionWrapper feeds state.attention back, rnn_wrappers.py:310)
                    if (tid < ncol) {
                        float v = 0.0f;
                        for (int ch = 0; ch < nch; ++ch) { const float c = lds[o_part + ch * ncol + tid]; v = ch == 0 ? c : v + c; }
                        if (G == 1) { lds[o_ctx + tid] = v; lds[o_cat + AS + tid] = v; lds[o_cat2 + DE + tid] = v; }
                        else decg_store(Xb + c0 + tid, ep, v, loc);
                    }
                    if (G > 1)
                        decg_gather_apply(Xb, ENC, ep, tid, o_abort, [&](const int j, const float v) {
                            lds[o_ctx + j] = v; lds[o_cat + AS + j] = v; lds[o_cat2 + DE + j] = v;
                        });
                    __syncthreads();
                    ok = LDSI(o_abort) == 0;
                }
                TWV_STAMP(52)
            }
        }
    }
    if (!ok && tid == 0) a.status[0] = 21;                    // exchange watchdog
#undef TWV_STAMP
#undef DECG_PREFETCH
#undef DECG_PREFETCH_
}

// -----------------------------------------------------------------------------------------------------
//  decoder, XCD-local: the utterances of one XCD share 32 workgroups that hold the decoder REGISTER-RESIDENT
// -----------------------------------------------------------------------------------------------------
// The split kernel above puts the 8 workgroups of an utterance on 8 XCDs (slice g of the weights stays in XCD g's L2), so every one
// of its 10+ exchanges per step is a memory-side hop (~0.65 us) and every stage re-streams its tiles from L2.  An XCD has 16 MB of
// vector registers: the whole decoder (6.4 MB) fits.  Here workgroup (x, g) -- XCD x by HW_REG_XCC_ID, slice g = role ticket 0..31 --
// keeps columns [16g, 16g+16) of EVERY decoder matrix in registers for the whole launch and applies them to all (up to 4) utterances
// of its XCD; activations travel as plain-store / sc1-load granules inside the XCD's L2 (~0.27 us, twv_wavenet_xcd.hip).
// A wave task = 16 output columns x 4 chunks of 32 terms (one chunk per 16-lane row), an AC-1 chunk = 32 v_fmac_f32_dpp with the
// operand fed by row_newbcast (twv_dpp.hpp); chunk values are added in chunk order by one wave.  Same arithmetic, same order:
// bit-identical to the other two decoder kernels.
constexpr int kXU = 4;              // utterances per XCD
constexpr int kXSlots = 5;          // register-resident tasks per wave
constexpr int kXStages = 16;
struct XStageTab {
    int nst;
    int K[kXStages], N[kXStages], act[kXStages], post[kXStages], xsel[kXStages], dsel[kXStages], p0[kXStages], p1[kXStages], p2[kXStages];
    long long w[kXStages], b[kXStages];          // packed offsets (floats): standard tiles, bias (-1 = none)
};
struct DecXArgs {
    DecArgs d;
    XStageTab tab;
    unsigned long long* exch;       // [8][2][kXU*512] granules
    int* tickets;                   // [8], zeroed before the launch
    long long xt_off;               // packed offset (floats) of the row tiles [32][8][kXSlots][2048]
    int upx;                        // utterances per XCD (N <= 8*upx)
    int* stab;                      // [workgroups][kXStages][16]: every workgroup's copy of the stage table (read with scalar loads)
    int nap_idle, nap_owner;        // sleeps in front of a gather's first poll, in units of 64 clocks (options "xdec_nap_idle" / "xdec_nap_owner")
    int nap_round, nap_w0;          // sleeps between an idle workgroup's polls; wave 0's nap behind its publish
};
// the tasks of slice g, in (stage, chunk group) order: task i belongs to wave i % 8, register slot i / 8
__host__ __device__ inline int xdec_tasks(const XStageTab& t, int g, int wave, int (&st_of)[kXSlots], int (&grp_of)[kXSlots])
{
    for (int j = 0; j < kXSlots; ++j) { st_of[j] = -1; grp_of[j] = 0; }
    int id = 0, worst = 0;
    for (int st = 0; st < t.nst; ++st) {
        if (16 * g >= t.N[st]) continue;
        const int ngrp = (((t.K[st] + 31) >> 5) + 3) >> 2;
        for (int q = 0; q < ngrp; ++q, ++id) {
            const int slot = id >> 3;
            if (slot > worst) worst = slot;
            if ((id & 7) == wave && slot < kXSlots) { st_of[slot] = st; grp_of[slot] = q; }
        }
    }
    return worst + 1;       // slots needed
}
// row tiles from the standard tiles: lane (row r, n) of task (st, group q) holds W[32*(4q+r) + k][16g + n], k = 0..31
__global__ void tc_xdec_pack_kernel(float* P, XStageTab t, long long xt_off)
{
    const int g = blockIdx.x >> 3, wave = blockIdx.x & 7;
    int st_of[kXSlots], grp_of[kXSlots];
    xdec_tasks(t, g, wave, st_of, grp_of);
    for (int j = 0; j < kXSlots; ++j) {
        float* dst = P + xt_off + (((long long)g * 8 + wave) * kXSlots + j) * kTile;
        const int st = st_of[j];
        for (int e = threadIdx.x; e < kTile; e += blockDim.x) {
            float v = 0.0f;
            if (st >= 0) {
                const int kq = e >> 8, lane = (e >> 2) & 63, c4 = e & 3;
                const int k = 4 * kq + c4, r = lane >> 4, n = lane & 15;
                const int nchunk = (t.K[st] + 31) >> 5, c = 4 * grp_of[j] + r, col = 16 * g + n;
                if (c < nchunk && col < t.N[st])
                    v = P[t.w[st] + ((long long)(col >> 6) * nchunk + c) * kTile + (((k >> 2) * 64 + (col & 63)) << 2) + (k & 3)];
            }
            dst[e] = v;
        }
    }
}
// LDS vectors of an utterance (stage inputs / gather destinations).  The concatenations the cells read are separate buffers, so that a
// gathered value is written straight to every place that reads it (no copy phase between the stages):
//   CATA = [prenet out | context | attention-rnn state]   (rnn_wrappers.py:310-312)        CATB = [attention-rnn state | context] (:463)
//   CATR+l = [y | state of residual GRU l]                (tacotron.py:167)
enum { XV_FRAME = 0, XV_VEC, XV_CATA, XV_CATB, XV_PQ, XV_Y, XV_CATR };
// what the gathering thread does with the value of column `col` (gather kinds)
enum { XG_PLAIN = 0, XG_GATES, XG_CAND_ATT, XG_CAND_RES, XG_OUT, XG_P, XG_CTX, XG_PQ };
// a stage's 64-byte record (one s_load_dwordx16 per stage from the workgroup's copy of the table in global memory)
enum { XS_K = 0, XS_N, XS_ACT, XS_POST, XS_XO, XS_HASB, XS_KIND, XS_GDST, XS_G0, XS_G1, XS_G2, XS_STRIDE = 16 };

// LDS carve of tc_decoder_x_kernel, shared by the kernel and the host's size check (float offsets)
struct XCarve {
    int u_frame, u_vec, u_cata, u_catb, u_catr, u_keep, u_y, u_pq, u_al, u_p, UST;
    int o_nv, o_ab, o_bias, o_scp, o_keys, o_memo, o_part, o_abort, o_tab, total;
    int Tp, ntmax, ncol, nch_t, kpitch, mpitch;
};
__host__ __device__ inline XCarve xdec_carve(int M, int D1, int ENC, int AS, int layers, int DR, int A, int T, int nu, int nst)
{
    XCarve c;
    c.Tp = ((T + 3) / 4) * 4;
    int o = 0;
    c.u_frame = o; o += ((M + 31) / 32) * 32;
    c.u_vec = o; o += 1024;                                // prenet hidden; gates r | u
    c.u_cata = o; o += D1 + ENC + AS;
    c.u_catb = o; o += AS + ENC;
    c.u_catr = o; o += layers * 2 * DR;
    c.u_keep = o; o += 512;                                // h before the update
    c.u_y = o; o += DR;
    c.u_pq = o; o += A + A / 8;                            // processed query, every 32-word chunk skewed by 4 words (score phase)
    c.u_al = o; o += c.Tp;
    c.u_p = o; o += c.Tp;
    c.UST = o;
    o = c.UST * nu;
    c.ntmax = (T + 31) >> 5;                               // time steps of a slice: t = g, g + 32, ...
    c.ncol = ENC >> 5;                                     // context columns of a slice
    c.nch_t = (T + 31) >> 5;
    c.kpitch = A + A / 8 + 16;                             // key rows: chunks skewed by 4 words, consecutive rows 16 banks apart
    c.mpitch = T * c.ncol + 16 * c.nch_t;                  // memory columns of an utterance: 32-step chunks 16 banks apart
    c.o_nv = o; o += A + A / 8;
    c.o_ab = o; o += A + A / 8;
    c.o_bias = o; o += 16 * nst;                           // this slice's 16 bias values per stage
    c.o_scp = o; o += nu * c.ntmax * 8 > 64 ? nu * c.ntmax * 8 : 64;   // score chunk values; then the recurrence's block totals [u][8]
    c.o_keys = o; o += nu * c.ntmax * c.kpitch;
    c.o_memo = o; o += nu * c.mpitch;
    c.o_part = o; o += nu * 16 * 24 > nu * c.nch_t * c.ncol ? nu * 16 * 24 : nu * c.nch_t * c.ncol;
    c.o_abort = o; o += 4;
    c.o_tab = o; o += kXStages * XS_STRIDE;
    c.total = o;
    return c;
}

// chunk values 1 .. NCH-1 of one output added to v in chunk order (AC-1), the depth a compile-time constant: no selects between the
// dependent adds (the run-time form below spends an add AND a select per chunk on the wave every other wave of the stage waits for)
template <int NCH>
__device__ __forceinline__ float xdec_chunk_sum(const int b, float v)
{
#pragma unroll
    for (int c0_ = 1; c0_ < NCH; c0_ += 12) {
        float c[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) if (c0_ + i < NCH) c[i] = lds[b + (c0_ + i) * 16];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 12; ++i) if (c0_ + i < NCH) v = v + c[i];
    }
    return v;
}

// Round 6: the kernel's exchanges and attention block rebuilt with what rounds 4-5 learnt on the split kernel:
//   * a granule exchange is laid out [element][utterance]: the gathering thread fetches the four utterances' values of ITS element with
//     two 16-byte loads (four self-tagged 8-byte halves) -- one polling round trip; round 2 polled the utterances one after the other in
//     branches (a poll is an L2 round trip and polls do not pipeline: 0.9-1.9 us per gather, now ~0.4)
//   * the stage record comes from ONE s_load_dwordx16 (ten LDS reads + v_readfirstlane per stage before)
//   * barriers are LDS-only (s_waitcnt lgkmcnt(0) + s_barrier): the mel / alignment stores of slice 0 no longer drain inside them
//   * the chunk-sum wave requests all chunk values at once; score tables skewed against LDS bank conflicts; the context's four chains
//     of a chunk on four lanes; the attention recurrence's 64-step blocks side by side on separate waves (AC-6, same adds)
// DEF: the hparams-default decoder sizes (hparams.py:126-158) as compile-time constants: the LDS carve and most index arithmetic fold
// (the kernel lives at the scalar-register limit: every run-time size is a spilled SGPR somewhere in the step loop)
// MM: the tasks run on the matrix core (three or four utterances per XCD); with one or two the row-broadcast fmas are shorter.
template <bool PROF, bool DEF, bool MM>
__global__ void __launch_bounds__(512) tc_decoder_x_kernel(DecXArgs xa)
{
#define XSTAMPT(k) if (PROF && a.prof && xcc == 0 && g == 0 && tid == 0 && it == 3) a.prof[st * 16 + (k)] = __builtin_amdgcn_s_memtime();
    const DecArgs& a = xa.d;
    const XStageTab& tb = xa.tab;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = a.T, M = DEF ? 80 : a.M, R = DEF ? 5 : a.R, A = DEF ? 256 : a.A, AS = DEF ? 256 : a.AS, ENC = DEF ? 256 : a.ENC;
    const int DR = DEF ? 256 : a.DR, D1 = DEF ? 128 : a.D1, NLY = DEF ? 2 : a.layers;
    // ---- which XCD, which slice
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 15u;
    __shared__ int s_ticket;
    const int UPX = xa.upx;
    const int n0 = (int)xcc * UPX;
    int nu = a.N - n0;
    nu = nu < 0 ? 0 : (nu > UPX ? UPX : nu);
    if (tid == 0) s_ticket = (xcc < 8u && nu > 0) ? atomicAdd(xa.tickets + xcc, 1) : 1 << 20;
    __syncthreads();
    const int g = __builtin_amdgcn_readfirstlane(s_ticket);
    if (g >= 32) return;
    const float* P = a.P;
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xa.exch + (long long)xcc * 2 * kXU * 512, 0, 2 * kXU * 512 * 8, 0x00020000);
    const int nst = DEF ? 11 : tb.nst;
    const XCarve cv = xdec_carve(M, D1, ENC, AS, NLY, DR, A, T, nu, nst);
    const int Tp = cv.Tp, UST = cv.UST;
    const int u_frame = cv.u_frame, u_vec = cv.u_vec, u_cata = cv.u_cata, u_catb = cv.u_catb, u_catr = cv.u_catr, u_keep = cv.u_keep;
    const int u_y = cv.u_y, u_pq = cv.u_pq, u_al = cv.u_al, u_p = cv.u_p;
    const int nAch = A / 32;
    const int ntmax = cv.ntmax;
    const int nt = T > g ? (T - g + 31) >> 5 : 0;
    const int ncol = cv.ncol, c0 = g * ncol;              // context columns of this slice
    const int nch_t = cv.nch_t, kpitch = cv.kpitch, mpitch = cv.mpitch;
    const int o_nv = cv.o_nv, o_ab = cv.o_ab, o_bias = cv.o_bias, o_scp = cv.o_scp, o_keys = cv.o_keys, o_memo = cv.o_memo;
    const int o_part = cv.o_part, o_abort = cv.o_abort, o_tab = cv.o_tab;

    int len[kXU];
#pragma unroll
    for (int u = 0; u < kXU; ++u) len[u] = u < nu ? a.lengths[n0 + u] : 0;
    for (int u = 0; u < nu; ++u) {
        const int ub = u * UST;
        const float* init = a.init + (long long)(n0 + u) * (AS + NLY * DR);
        for (int i = tid; i < UST; i += 512) lds[ub + i] = 0.0f;
        __syncthreads();
        // initial state (tacotron.py:184-195, AttentionWrapper.zero_state, helpers.py:90-92): go-frame, context and y zero
        for (int i = tid; i < AS; i += 512) { lds[ub + u_cata + D1 + ENC + i] = init[i]; lds[ub + u_catb + i] = init[i]; }
        for (int i = tid; i < NLY * DR; i += 512) { const int l = i / DR, k = i - l * DR; lds[ub + u_catr + l * 2 * DR + DR + k] = init[AS + i]; }
        if (tid == 0) lds[ub + u_al] = 1.0f;                // one-hot at 0 [RECALLED-TF initial_alignments]
        const float* keys = a.keys + (long long)(n0 + u) * T * A;
        const float* memo = a.memo + (long long)(n0 + u) * T * ENC;
        for (int i = tid; i < nt * A; i += 512) { const int tl = i / A, j = i - tl * A; lds[o_keys + (u * ntmax + tl) * kpitch + j + ((j >> 5) << 2)] = keys[(long long)((tl << 5) + g) * A + j]; }
        for (int i = tid; i < T * ncol; i += 512) { const int t = i / ncol, cl = i - t * ncol; lds[o_memo + u * mpitch + t * ncol + 16 * (t >> 5) + cl] = memo[(long long)t * ENC + c0 + cl]; }
    }
    for (int i = tid; i < A; i += 512) { lds[o_nv + i + ((i >> 5) << 2)] = P[a.w.nv + i]; lds[o_ab + i + ((i >> 5) << 2)] = P[a.w.ab + i]; }
    for (int i = tid; i < 16 * nst; i += 512) {
        const int st = i >> 4, col = 16 * g + (i & 15);
        lds[o_bias + i] = (tb.b[st] >= 0 && col < tb.N[st]) ? P[tb.b[st] + col] : 0.0f;
    }
    if (tid < 4) LDSI(o_abort + tid) = 0;
    if (tid < nst) {
        auto vsel = [&](int v) { return v == XV_FRAME ? u_frame : v == XV_VEC ? u_vec : v == XV_CATA ? u_cata : v == XV_CATB ? u_catb : v == XV_PQ ? u_pq : v == XV_Y ? u_y : u_catr + (v - XV_CATR) * 2 * DR; };
        const int q = o_tab + tid * XS_STRIDE;
        const int post = tb.post[tid], xo = vsel(tb.xsel[tid]), dst = vsel(tb.dsel[tid]), tp0 = tb.p0[tid], tp1 = tb.p1[tid], tp2 = tb.p2[tid];
        for (int e = 0; e < XS_STRIDE; ++e) LDSI(q + e) = 0;
        LDSI(q + XS_K) = tb.K[tid]; LDSI(q + XS_N) = tb.N[tid]; LDSI(q + XS_ACT) = tb.act[tid]; LDSI(q + XS_POST) = post;
        LDSI(q + XS_XO) = xo; LDSI(q + XS_HASB) = tb.b[tid] >= 0 ? 1 : 0;
        // gather kind and its parameters.  GATES: (cell input, nin, U); CAND: (U, layer, next layer or 0)
        LDSI(q + XS_KIND) = post == DP_GATES ? XG_GATES : post == DP_CAND ? (tp2 < 0 ? XG_CAND_ATT : XG_CAND_RES) : post == DP_OUT ? XG_OUT : post == DP_QUERY ? XG_PQ : XG_PLAIN;
        LDSI(q + XS_GDST) = post == DP_GATES ? xo : dst;
        LDSI(q + XS_G0) = tp0; LDSI(q + XS_G1) = post == DP_GATES ? tp1 : tp1 - 1; LDSI(q + XS_G2) = tp2 - 1;
    }
    __syncthreads();
    // the stage table once more, where scalar loads can reach it (this workgroup's copy)
    int* const stab = xa.stab + (long long)blockIdx.x * (kXStages * XS_STRIDE);
    for (int e = tid; e < kXStages * XS_STRIDE; e += 512) stab[e] = LDSI(o_tab + e);
    __threadfence();
    const float asb = P[a.w.asb];
    // ---- this wave's tasks and their register-resident row tiles
    int st_of[kXSlots], grp_of[kXSlots];
    xdec_tasks(tb, g, wave, st_of, grp_of);
    Tile wt[kXSlots];
#pragma unroll
    for (int j = 0; j < kXSlots; ++j)
        load_tile(wt[j], P + xa.xt_off + (((long long)g * 8 + wave) * kXSlots + j) * kTile, lane);
    __syncthreads();
    __builtin_amdgcn_s_dcache_inv();

    unsigned ep = 0;
    bool ok = true;
    int it = 0, st = 0;
    const int kAux = (int)(16u | 0x80000000u);
    // Exchange `ep`, element i (< cnt <= 512) of utterance u = granule i * kXU + u: thread i collects its element of EVERY utterance with
    // two 16-byte loads (each half carries its own tag) and puts the values where their readers want them -- `kind` says what a value
    // means (GRU gates / candidates update the cell state on the spot).
    // `owner`: this workgroup published a share of the exchange a moment ago (nothing can arrive before the L2 hop has passed); a
    // workgroup without columns in the stage arrives a whole dots + chunk-sum phase early and would only load the L2 slices the
    // publishers' stores have to pass: it sleeps through most of that phase first.
    auto gather = [&](int tid, int cnt, int kind, int dst, int g0, int g1, int g2, bool owner) {
        const bool need = tid < cnt;
        const int off = need ? (int)(((ep & 1u) * (kXU * 512) + (unsigned)tid * kXU) * 8u) : (int)0x7ffffff0;   // (out of range: zeros)
        u32x4d qa = {0u, 0u, 0u, 0u}, qb = {0u, 0u, 0u, 0u};
        bool fin = false;
        {
            const int naps = !owner ? xa.nap_idle : (wave != 0 ? xa.nap_owner : xa.nap_w0);
            for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(1);
        }
        for (int itp = 0; itp < (1 << 20); ++itp) {
            asm volatile("" ::: "memory");
            qa = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, kAux);
            if (nu > 2) qb = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, kAux);
            const bool got = !need || (qa.y == ep && (nu < 2 || qa.w == ep) && (nu < 3 || qb.y == ep) && (nu < 4 || qb.w == ep));
            if (__all(got)) { fin = true; break; }
            if ((itp & 63) == 63 && LDSVI(o_abort)) return;
            if (!owner) for (int i = 0; i < xa.nap_round; ++i) __builtin_amdgcn_s_sleep(1);
        }
        if (!fin) { LDSVI(o_abort) = 1; return; }
        if (kind != XG_P && kind != XG_CTX) { XSTAMPT(12) }
        if (!need) return;
        const int col = tid;
        // one dispatch on the stage's kind, the utterances unrolled inside each case (a chain of scalar branches per utterance before: most
        // of the 0.3-0.8 us between the arrival and the barrier); a GRU update requests what it reads beside the arriving value for every
        // utterance first (the reads of utterance 1 cannot pass the writes of utterance 0 otherwise: the compiler has to assume they alias)
        float v[kXU];
        v[0] = __uint_as_float(qa.x); v[1] = __uint_as_float(qa.z); v[2] = __uint_as_float(qb.x); v[3] = __uint_as_float(qb.z);
        if (kind == XG_PLAIN) {
#pragma unroll
            for (int u = 0; u < kXU; ++u) if (u < nu) lds[u * UST + dst + col] = v[u];
        } else if (kind == XG_GATES) {            // tf.contrib.rnn.GRUCell, gate order r | u: keep h, cell input <- [x, r*h]
            const int nin = g0, U = g1;           // dst = the cell's concatenated input
            if (col < U) {
                float h[kXU];
#pragma unroll
                for (int u = 0; u < kXU; ++u) h[u] = u < nu ? lds[u * UST + dst + nin + col] : 0.0f;
#pragma unroll
                for (int u = 0; u < kXU; ++u) if (u < nu) { lds[u * UST + u_keep + col] = h[u]; lds[u * UST + dst + nin + col] = v[u] * h[u]; }
            } else {
#pragma unroll
                for (int u = 0; u < kXU; ++u) if (u < nu) lds[u * UST + u_vec + col] = v[u];
            }
        } else if (kind == XG_CAND_ATT || kind == XG_CAND_RES) {       // h <- u*h + (1-u)*c
            const int U = g0;
            const bool res = kind == XG_CAND_RES;
            const int cr = u_catr + (res ? g1 : 0) * 2 * DR;        // residual layer l = g1
            float uu[kXU], h[kXU], y[kXU], hn[kXU];
#pragma unroll
            for (int u = 0; u < kXU; ++u) {
                const bool on = u < nu;
                uu[u] = on ? lds[u * UST + u_vec + U + col] : 0.0f;
                h[u] = on ? lds[u * UST + u_keep + col] : 0.0f;
                y[u] = (on && res) ? lds[u * UST + cr + col] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < kXU; ++u) {
                const float t1 = uu[u] * h[u], t2 = 1.0f - uu[u], t3 = t2 * v[u];
                hn[u] = t1 + t3;
            }
            if (!res) {                            // the attention rnn's state: next step's cell input, this step's query / projection input
#pragma unroll
                for (int u = 0; u < kXU; ++u) if (u < nu) { lds[u * UST + u_cata + D1 + ENC + col] = hn[u]; lds[u * UST + u_catb + col] = hn[u]; }
            } else {                               // y <- y + h (tacotron.py:167); next layer's input or the output projection's
                const int yd = g2 > 0 ? u_catr + g2 * 2 * DR : u_y;
#pragma unroll
                for (int u = 0; u < kXU; ++u) if (u < nu) { lds[u * UST + cr + DR + col] = hn[u]; lds[u * UST + yd + col] = y[u] + hn[u]; }
            }
        } else if (kind == XG_OUT) {               // tacotron.py:204 reshape; helpers.py:40 last frame fed back
            if (col >= M * (R - 1)) {              // (the mel frame itself is written by its publisher)
#pragma unroll
                for (int u = 0; u < kXU; ++u) if (u < nu) lds[u * UST + u_frame + col - M * (R - 1)] = v[u];
            }
        } else if (kind == XG_PQ) {                // the processed query goes into its skewed table
#pragma unroll
            for (int u = 0; u < kXU; ++u) if (u < nu) lds[u * UST + dst + col + ((col >> 5) << 2)] = v[u];
        } else if (kind == XG_P) {
#pragma unroll
            for (int u = 0; u < kXU; ++u) if (u < nu) lds[u * UST + u_p + col] = v[u];
        } else {                                   // XG_CTX
#pragma unroll
            for (int u = 0; u < kXU; ++u) if (u < nu) { lds[u * UST + u_cata + D1 + col] = v[u]; lds[u * UST + u_catb + AS + col] = v[u]; }
        }
    };
    auto publish = [&](int u, int i, float v) {
        __builtin_amdgcn_raw_buffer_store_b64(u32x2d{__float_as_uint(v), ep}, rs, (int)(((ep & 1u) * (kXU * 512) + (unsigned)i * kXU + (unsigned)u) * 8u), 0, 0);
    };

    const int tid_k = tid;
    for (it = 0; it < a.iters && ok; ++it) {
        for (st = 0; st < nst && ok; ++st) {
            // the thread's index as a value the compiler cannot see through, once per stage: everything derived from it (LDS addresses, task
            // decompositions) is recomputed where it is used -- hoisted out of the step loop these values outnumber the registers the tiles
            // leave, and came back from scratch memory (an L2 trip each, 27 of them) in the middle of the stages
            int tid = tid_k;
            asm volatile("" : "+v"(tid));
            const int lane = tid & 63;
            XSTAMPT(0)
            const i32x16s rec = decg_sload16(stab + st * XS_STRIDE);
            const int K = rec[XS_K], N = rec[XS_N], act = rec[XS_ACT], post = rec[XS_POST], xo = rec[XS_XO], has_b = rec[XS_HASB];
            const int nchunk = (K + 31) >> 5;
            const bool mine = 16 * g < N;
            const float bias_v = (wave == 0 && has_b) ? lds[o_bias + st * 16 + (lane & 15)] : 0.0f;      // (requested ahead of the chunk sums)
            // ---- this wave's task of the stage (at most one): 16 columns x 4 chunks for every utterance, partials to LDS.
            // On the matrix core: v_mfma_f32_4x4x1_16b_f32 is sixteen independent 4 x 4 outer products, D[i][j] += A[i] * B[j] -- one fused
            // multiply-add per element, like the 32x32x2 form (scripts/ubench/mfma_f32_order.hip).  Block = four of the task's 64 (chunk,
            // column) lanes, B = the tile register of reduction index k as it stands (lane = (chunk row, column)), A = x_u[k] of the lane's
            // chunk for utterance u = lane mod 4, D register u = utterance u's partial in the tile's own lane layout: ONE instruction per
            // k serves all four utterances (four v_fmac_f32_dpp before).  AC-1: chain j accumulates k = j, j + 4, ... from +0.
#pragma unroll
            for (int j = 0; j < kXSlots; ++j) {
                if (st_of[j] == st) {
                    const int c = 4 * grp_of[j] + (lane >> 4);
                    const bool live = c < nchunk;
                    if (MM) {
                        const int ua = (lane & 3) < nu ? (lane & 3) : 0;
                        const int xb_ = ua * UST + xo + 32 * (live ? c : 0);
                        // the chunk's 32 operands as eight 16-byte LDS reads, four of them in flight: left to itself the compiler reads them one
                        // at a time into the same four registers, eight LDS round trips in a row (0.4 of the task's 0.6 us)
                        f32x4 acc[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                        const int xw = xb_ >> 2;
#define XMF(q, x) { _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) acc[k_] = __builtin_amdgcn_mfma_f32_4x4x1f32((x)[k_], wt[j].w[4 * (q) + k_], acc[k_], 0, 0, 0); }
                        f32x4 x0 = LDS4(xw), x1 = LDS4(xw + 1), x2 = LDS4(xw + 2), x3 = LDS4(xw + 3);
                        __builtin_amdgcn_sched_barrier(0);
                        XMF(0, x0) XMF(1, x1)
                        __builtin_amdgcn_sched_barrier(0);
                        x0 = LDS4(xw + 4); x1 = LDS4(xw + 5);
                        __builtin_amdgcn_sched_barrier(0);
                        XMF(2, x2) XMF(3, x3)
                        __builtin_amdgcn_sched_barrier(0);
                        x2 = LDS4(xw + 6); x3 = LDS4(xw + 7);
                        __builtin_amdgcn_sched_barrier(0);
                        XMF(4, x0) XMF(5, x1)
                        XMF(6, x2) XMF(7, x3)
#undef XMF
#pragma unroll
                        for (int u = 0; u < kXU; ++u) {
                            const float r = (acc[0][u] + acc[1][u]) + (acc[2][u] + acc[3][u]);
                            if (live && u < nu) lds[o_part + (u * 24 + c) * 16 + (lane & 15)] = r;
                        }
                    } else {
                        // one or two utterances: the chunk as 32 v_fmac_f32_dpp per utterance, operand fed by row_newbcast, two utterances interleaved
                        const int xb_ = xo + 32 * c + (lane & 15);
                        const float x00 = live ? lds[xb_] : 0.0f, x01 = live ? lds[xb_ + 16] : 0.0f;
                        const float x10 = (live && nu > 1) ? lds[UST + xb_] : 0.0f, x11 = (live && nu > 1) ? lds[UST + xb_ + 16] : 0.0f;
                        float r0, r1;
                        dot32_dpp_x2(wt[j].w, x00, x01, wt[j].w, x10, x11, r0, r1);
                        if (live) lds[o_part + c * 16 + (lane & 15)] = r0;
                        if (live && nu > 1) lds[o_part + (24 + c) * 16 + (lane & 15)] = r1;
                    }
                }
            }
            XSTAMPT(1)
            lds_barrier();
            XSTAMPT(2)
            // ---- wave 0: chunk values in chunk order (AC-1) + bias + activation -> the slice's 16 values per utterance
            ++ep;
            if (wave == 0 && mine) {
                const int u = lane >> 4, n = lane & 15, col = 16 * g + n;
                if (u < nu && col < N) {
                    const int b = o_part + u * 24 * 16 + n;
                    float v = lds[b];
                    // the depths of the hparams-default stages as straight lines; any other depth: twelve chunk values requested at a time (the
                    // rows exist for 24 chunks whatever the stage's depth), adds in chunk order
                    if (nchunk == 20) v = xdec_chunk_sum<20>(b, v);
                    else if (nchunk == 16) v = xdec_chunk_sum<16>(b, v);
                    else if (nchunk == 8) v = xdec_chunk_sum<8>(b, v);
                    else if (nchunk == 3) v = xdec_chunk_sum<3>(b, v);
                    else
                    for (int c0_ = 1; c0_ < nchunk; c0_ += 12) {
                        float c[12];
#pragma unroll
                        for (int i = 0; i < 12; ++i) c[i] = lds[b + (c0_ + i < 24 ? c0_ + i : 23) * 16];
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < 12; ++i) if (c0_ + i < nchunk) v = v + c[i];
                    }
                    if (has_b) v = v + bias_v;
                    if (act == DA_SIGMOID) v = sigmoid_e(v);
                    else if (act == DA_TANH) v = tanh_e(v);
                    else if (act == DA_RELU) v = v > 0.0f ? v : 0.0f;
                    publish(u, col, v);
                    if (post == DP_OUT) a.mel[((long long)(n0 + u) * a.iters + it) * M * R + col] = v;     // tacotron.py:204 reshape
                }
            }
            XSTAMPT(3)
            gather(tid, N, rec[XS_KIND], rec[XS_GDST], rec[XS_G0], rec[XS_G1], rec[XS_G2], mine);
            XSTAMPT(4)
            lds_barrier();
            XSTAMPT(5)
            ok = LDSI(o_abort) == 0;
            if (!ok) break;
            if (post == DP_QUERY) {
                // [RECALLED-TF BahdanauMonotonicAttention.__call__] score for the time steps t = g, g+32, ... of every utterance:
                // one thread per (u, t, chunk, chain k): s_k = fma chain over j = k, k+4, ..., k+28; the four chains of a chunk sit in
                // adjacent lanes and are combined as (s0+s1)+(s2+s3).  Every 32-word chunk of the tables is skewed by 4 words and
                // consecutive key rows lie 16 banks apart: the 64 lanes of a wave (8 chunks x 4 chains x 2 rows) read 64 distinct banks / words
                const int ntask = nu * nt * nAch * 4;
                for (int task0 = 0; task0 < ntask; task0 += 512) {
                    const int task = task0 + tid;
                    const bool live = task < ntask;
                    const int k = task & 3, tc = task >> 2;
                    const int ch = live ? tc % nAch : 0, ut = live ? tc / nAch : 0;
                    const int u = ut / (nt > 0 ? nt : 1), tl = ut - u * nt;
                    float sk = 0.f;
                    const int js = ch * 36 + k, kl = o_keys + (u * ntmax + tl) * kpitch + js, pq = u * UST + u_pq + js;
                    // (the tanh evaluations two at a time in packed instructions: the same operations on each, and the phase is issue-bound)
                    // every operand requested first (pair by pair the compiler waits for each pair's LDS round trip in turn)
                    f32x2m kv[4], qv[4], bv[4], nv_[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int j = 8 * i;
                        kv[i] = f32x2m{lds[kl + j], lds[kl + j + 4]};
                        qv[i] = f32x2m{lds[pq + j], lds[pq + j + 4]};
                        bv[i] = f32x2m{lds[o_ab + js + j], lds[o_ab + js + j + 4]};
                        nv_[i] = f32x2m{lds[o_nv + js + j], lds[o_nv + js + j + 4]};
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x2m th = tanh_e2((kv[i] + qv[i]) + bv[i]);
                        sk = fma_(nv_[i][0], th[0], sk);
                        sk = fma_(nv_[i][1], th[1], sk);
                    }
                    const float s1 = __shfl_xor(sk, 1);
                    const float pr = (k & 1) ? s1 + sk : sk + s1;       // lanes k=0,1 hold s0+s1 ; k=2,3 hold s2+s3 (operand order as written)
                    const float p2 = __shfl_xor(pr, 2);
                    if (live && k == 0) lds[o_scp + (u * ntmax + tl) * 8 + ch] = pr + p2;
                }
                XSTAMPT(6)
                lds_barrier();
                ++ep;
                for (int i = tid; i < nu * nt; i += 512) {
                    const int u = i / nt, tl = i - u * nt, t = (tl << 5) + g;
                    float sc = 0.0f;
                    for (int ch = 0; ch < nAch; ++ch) { const float c = lds[o_scp + (u * ntmax + tl) * 8 + ch]; sc = ch == 0 ? c : sc + c; }
                    sc = sc + asb;
                    int lu = len[0];
#pragma unroll
                    for (int q = 1; q < kXU; ++q) lu = (u == q) ? len[q] : lu;
                    publish(u, t, t < lu ? sigmoid_e(sc) : 0.0f);         // _maybe_mask_score(-inf) -> p = 0
                }
                XSTAMPT(7)
                gather(tid, T, XG_P, 0, 0, 0, 0, true);
                lds_barrier();
                XSTAMPT(8)
                ok = LDSI(o_abort) == 0;
                if (!ok) break;
                // monotonic attention recurrence (redundant in every workgroup), in registers:
                // cumprod(1 - p) as exp(exclusive cumsum(log(clip(1 - p)))) [RECALLED-TF safe_cumprod], then
                // alignments = p * cumprod * inclusive cumsum(previous / clip(cumprod, 1e-10, 1))
                const int nblk = (T + 63) >> 6;
                if (nu * nblk <= 8) {
                    // the 64-step blocks of AC-6 side by side: wave (u, b) scans block b of utterance u; a block needs from its predecessors
                    // only the running total in front of its own scan -- total of block 0, + block 1's own total, ... : the same chain of
                    // adds whether the blocks are scanned one after the other (the branch below) or their totals are summed afterwards
                    const int wu = wave / nblk, wb = wave - wu * nblk;
                    const bool actv = wave < nu * nblk;
                    const int ub = (actv ? wu : 0) * UST, t = wb * 64 + lane;
                    const bool live = actv && t < T;
                    const int o_tot = o_scp + (actv ? wu : 0) * 8;          // block totals (the score chunk values are consumed)
                    const float pv = live ? lds[ub + u_p + t] : 0.0f;
                    float om = 1.0f - pv;
                    const float tiny = 1.17549435e-38f;
                    om = om < tiny ? tiny : (om > 1.0f ? 1.0f : om);
                    const float lq = live ? log_e(om) : 0.0f;
                    const float sc = scan64_f32_wave(lq);
                    if (actv && lane == 63) lds[o_tot + wb] = sc;             // (the score chunk values were consumed in front of the p gather)
                    lds_barrier();
                    float carry = 0.0f, all = 0.0f;
                    for (int b = 0; b < nblk; ++b) {
                        const float tb_ = lds[o_tot + b];
                        const float nx = b == 0 ? tb_ : all + tb_;
                        if (b < wb) carry = nx;
                        all = nx;
                    }
                    const bool first = wb == 0;
                    const float incl = first ? sc : carry + sc;
                    const float up = __shfl_up(incl, 1);
                    const float ex = lane == 0 ? (first ? 0.0f : carry) : up;
                    const float cpv = exp_e(ex);
                    float den = cpv;
                    den = den < 1e-10f ? 1e-10f : (den > 1.0f ? 1.0f : den);
                    const float q2 = live ? div_(lds[ub + u_al + t], den) : 0.0f;
                    const float sc2 = scan64_f32_wave(q2);
                    if (actv && lane == 63) lds[o_tot + 32 + wb] = sc2;       // (words of their own: the first totals may still be being read)
                    lds_barrier();
                    float carry2 = 0.0f, all2 = 0.0f;
                    for (int b = 0; b < nblk; ++b) {
                        const float tb_ = lds[o_tot + 32 + b];
                        const float nx = b == 0 ? tb_ : all2 + tb_;
                        if (b < wb) carry2 = nx;
                        all2 = nx;
                    }
                    const float cs = first ? sc2 : carry2 + sc2;
                    if (live) {
                        const float pc = pv * cpv;
                        const float al = pc * cs;
                        lds[ub + u_al + t] = al;
                        if (a.align && (t & 31) == g) a.align[((long long)(n0 + wu) * T + t) * a.iters + it] = al;      // tacotron.py:223 (every slice holds all of them: slice t mod 32 writes)
                    }
                    if (actv && wb == nblk - 1) for (int tt = T + lane; tt < Tp; tt += 64) lds[ub + u_al + tt] = 0.0f;
                } else if (wave < nu) {
                    // wave u walks the blocks of utterance u
                    const int ub = wave * UST;
                    float run = 0.0f, run2 = 0.0f;
                    for (int base = 0; base < T; base += 64) {
                        const int t = base + lane;
                        const bool live = t < T;
                        const float pv = live ? lds[ub + u_p + t] : 0.0f;
                        float om = 1.0f - pv;
                        const float tiny = 1.17549435e-38f;
                        om = om < tiny ? tiny : (om > 1.0f ? 1.0f : om);
                        const float lq = live ? log_e(om) : 0.0f;
                        const float ex = decg_scan_block(lq, run, base == 0, lane, false);
                        const float cpv = exp_e(ex);
                        float den = cpv;
                        den = den < 1e-10f ? 1e-10f : (den > 1.0f ? 1.0f : den);
                        const float q2 = live ? div_(lds[ub + u_al + t], den) : 0.0f;
                        const float cs = decg_scan_block(q2, run2, base == 0, lane, true);
                        if (live) {
                            const float pc = pv * cpv;
                            const float al = pc * cs;
                            lds[ub + u_al + t] = al;
                            if (a.align && (t & 31) == g) a.align[((long long)(n0 + wave) * T + t) * a.iters + it] = al;      // tacotron.py:223
                        }
                    }
                    for (int t = T + lane; t < Tp; t += 64) lds[ub + u_al + t] = 0.0f;
                }
                lds_barrier();
                XSTAMPT(9)
                // rnn_wrappers.py:390 context = alignments . values: this slice takes ENC/32 columns; one thread per (utterance, column,
                // 32-step chunk, CHAIN k): the chunk's four interleaved fma chains (AC-1) sit in four adjacent lanes, eight dependent fmas
                // each, and meet as (s0+s1)+(s2+s3) through two lane exchanges -- the same adds in the same order; chunk values added in order
                {
                    const int ntk = nu * ncol * nch_t * 4;
                    for (int task0 = 0; task0 < ntk; task0 += 512) {
                        const int task = task0 + tid;
                        const bool live = task < ntk;
                        const int k = task & 3, tc = live ? task >> 2 : 0;
                        const int cl = tc % ncol, uc = tc / ncol, ch = uc % nch_t, u = uc / nch_t;
                        const int ta = ch * 32, tbb = T < ta + 32 ? T : ta + 32;
                        const int mo = o_memo + u * mpitch + 16 * ch + cl, al = u * UST + u_al;
                        // eight fmas in a straight line, every operand requested first; a term past the end of the input is 0 x 0 added to a chain
                        // that is never -0 (it starts at +0, and x + (-x) = +0): fma(0, 0, sk) == sk exactly
                        float mv[8], av[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int t = ta + k + 4 * i;
                            const bool in = live && t < tbb;
                            mv[i] = in ? lds[mo + t * ncol] : 0.0f;
                            av[i] = in ? lds[al + t] : 0.0f;
                        }
                        float sk = 0.f;
#pragma unroll
                        for (int i = 0; i < 8; ++i) sk = fma_(mv[i], av[i], sk);
                        const float s1 = __shfl_xor(sk, 1);
                        const float pr = (k & 1) ? s1 + sk : sk + s1;       // lanes k = 0,1 hold s0+s1 ; k = 2,3 hold s2+s3 (operand order as written)
                        const float p2 = __shfl_xor(pr, 2);
                        if (live && k == 0) lds[o_part + (u * nch_t + ch) * ncol + cl] = pr + p2;
                    }
                    lds_barrier();
                    ++ep;
                    if (tid < nu * ncol) {
                        const int u = tid / ncol, cl = tid - u * ncol;
                        float v = 0.0f;
                        for (int ch = 0; ch < nch_t; ++ch) { const float c = lds[o_part + (u * nch_t + ch) * ncol + cl]; v = ch == 0 ? c : v + c; }
                        publish(u, c0 + cl, v);
                    }
                    XSTAMPT(10)
                    gather(tid, ENC, XG_CTX, 0, 0, 0, 0, true);
                    lds_barrier();
                    ok = LDSI(o_abort) == 0;
                    if (!ok) break;
                }
            }
            XSTAMPT(11)
        }
    }
    if (!ok && tid == 0) a.status[0] = 22;                    // exchange watchdog
#undef XSTAMPT
}

// =====================================================================================================
//  host side
// =====================================================================================================
struct TMat { long long off; int K, N; };
struct TVec { long long off; int n; };
struct TCbhg {
    TMat W[17]; TVec b[17], inv[17], shift[17];
    TMat pW[2]; TVec pb[2], pinv[2], pshift[2];
    int has_dense; TMat dW; TVec db;
    TMat hH[8], hT[8]; TVec hHb[8], hTb[8];
    TMat hHT[8];                    // highway pair tiles: tile nb = [H columns 32 nb .. | T columns 32 nb ..] (mm_body mode 1)
    TMat gWgx[2], gWgh[2], gWcx[2], gWch[2]; TVec gbg[2], gbc[2];
};
struct twv_tacotron {
    twv_tacotron_dims d;
    unsigned long long* prof = nullptr;
    int xdec_nap_idle = 8, xdec_nap_owner = 12, xdec_nap_round = 1, xdec_nap_w0 = 3;    // tc_decoder_x_kernel: sleeps in front of a gather's first poll (x 64 clocks)
    int dec_groups = 0;             // 0 auto (16, halved until N*G fits the CUs), -1 single-workgroup kernel
    int dec_split_all = -1;         // prenet + query layer split over the workgroups: -1 = when the exchanges are L2-local, 0 / 1
    int dec_local = 1;              // split kernel: 1 = an utterance's workgroups on one XCD (exchanges through its L2), 0 = spread over the XCDs
    int gemm_valu = 0, gemm_group = 1;   // see g_gemm_valu / g_gemm_group
    int hw_stack = 1;                    // see g_hw_stack
    long long blob_floats, packed_floats;
    long long xt_off = 0;           // row tiles of the XCD-local decoder kernel [32 slices][8 waves][kXSlots][2048]
    TMat emb, semb;                 // raw tables (K rows x N)
    TMat dW[8]; TVec db[8]; int ndense, dn[8];
    TMat stab[8];                   // speaker_embedding_size == 1: the five get_embed tables (tacotron.py:69-75), raw rows
    TMat pW1, pW2; TVec pb1, pb2;
    TCbhg enc, post;
    TMat Wm, Wq; TVec av, ag, ab, asb, nv;
    TMat dpW1, dpW2; TVec dpb1, dpb2;
    TMat aWgm, aWcm; TVec abg, abc;
    TMat cW; TVec cb;
    TMat rWg[4], rWc[4]; TVec rbg[4], rbc[4];
    TMat oW; TVec ob;
    TMat lW; TVec lb;
    struct Item { int kind; long long src, dst; int K, N, r0, r1; };   // kind 0 = tiles of rows [r0,r1) of a (K,N) matrix, 1 = raw copy
    std::vector<Item> items;
};

static inline long long tiles_floats(int K, int N) { return (long long)((N + 63) / 64) * ((K + 31) / 32) * kTile; }

// Walks the canonical blob order (tacotron.py tacotron_specs) and lays out the packed buffer.
static void taco_build(twv_tacotron* h)
{
    const twv_tacotron_dims& d = h->d;
    long long src = 0, dst = 0;
    auto mat = [&](int K, int N) { TMat m{dst, K, N}; h->items.push_back({0, src, dst, K, N, 0, K}); src += (long long)K * N; dst += tiles_floats(K, N); return m; };
    auto raw = [&](int K, int N) { TMat m{dst, K, N}; h->items.push_back({1, src, dst, K, N, 0, K}); src += (long long)K * N; dst += ((long long)K * N + 3) / 4 * 4; return m; };
    auto vec = [&](int n) { TVec v{dst, n}; h->items.push_back({1, src, dst, 1, n, 0, 1}); src += n; dst += (n + 3) / 4 * 4; return v; };
    // a (K = nin + U, N) GRU kernel split into its x rows and its h rows (both start on a chunk boundary)
    auto gru_split = [&](int nin, int U, int N, TMat& mx, TMat& mh) {
        mx = TMat{dst, nin, N}; h->items.push_back({0, src, dst, nin + U, N, 0, nin}); dst += tiles_floats(nin, N);
        mh = TMat{dst, U, N}; h->items.push_back({0, src, dst, nin + U, N, nin, nin + U}); dst += tiles_floats(U, N);
        src += (long long)(nin + U) * N;
    };
    auto cbhg = [&](TCbhg& c, int Cin, int bank, int bch, const int32_t* proj, int pw, int depth, int rnn) {
        for (int k = 1; k <= bank; ++k) { c.W[k] = mat(k * Cin, bch); c.b[k] = vec(bch); c.inv[k] = vec(bch); c.shift[k] = vec(bch); }
        int cin = bank * bch;
        for (int i = 0; i < 2; ++i) { c.pW[i] = mat(pw * cin, proj[i]); c.pb[i] = vec(proj[i]); c.pinv[i] = vec(proj[i]); c.pshift[i] = vec(proj[i]); cin = proj[i]; }
        c.has_dense = proj[1] != rnn;
        if (c.has_dense) { c.dW = mat(proj[1], rnn); c.db = vec(rnn); }
        for (int i = 0; i < depth; ++i) {
            const long long sH = src;
            c.hH[i] = mat(rnn, rnn); c.hHb[i] = vec(rnn);
            const long long sT = src;
            c.hT[i] = mat(rnn, rnn); c.hTb[i] = vec(rnn);
            // the same two kernels once more as pair tiles: rnn/32 tiles per chunk, lanes 0-31 from H, lanes 32-63 from T
            c.hHT[i] = TMat{dst, rnn, 2 * rnn};
            h->items.push_back({2, sH, dst, rnn, rnn, (int)(sT - sH), 0});
            dst += (long long)(rnn / 32) * ((rnn + 31) / 32) * kTile;
        }
        for (int dr = 0; dr < 2; ++dr) {
            gru_split(rnn, rnn, 2 * rnn, c.gWgx[dr], c.gWgh[dr]); c.gbg[dr] = vec(2 * rnn);
            gru_split(rnn, rnn, rnn, c.gWcx[dr], c.gWch[dr]); c.gbc[dr] = vec(rnn);
        }
    };
    const int E = d.embedding_size, SE = d.speaker_embedding_size, P0 = d.enc_prenet_sizes[0], P1 = d.enc_prenet_sizes[1],
              RN = d.enc_rnn_size, A = d.attention_size, AS = d.attention_state_size, DR = d.dec_rnn_size, M = d.num_mels,
              R = d.reduction_factor, ENC = 2 * RN;
    h->emb = raw(d.n_symbols, E);
    // tacotron.py:63-104: the speaker tensors exist for num_speakers > 1 only; a single-speaker model (synthesizer.py:375's default)
    // has no speaker embedding, no before_highway and zero initial states
    h->ndense = (d.num_speakers > 1 && !(d.model_simple && SE != 1)) ? 3 + d.dec_layer_num : 0;
    const int dn[8] = {P1, 2 * RN, AS, DR, DR, DR, DR, DR};
    if (d.num_speakers > 1 && d.model_simple && SE != 1) {
        h->semb = raw(d.num_speakers, SE);          // 'simple': the embedding table alone
    } else if (d.num_speakers > 1 && SE == 1) {
        // tacotron.py:69-75: speaker_embedding_size == 1 -> five embedding tables (modules.py:10-12 get_embed), one row per speaker
        for (int i = 0; i < h->ndense; ++i) { h->dn[i] = dn[i]; h->stab[i] = raw(d.num_speakers, dn[i]); }
    } else if (d.num_speakers > 1) {
        h->semb = raw(d.num_speakers, SE);
        for (int i = 0; i < h->ndense; ++i) { h->dn[i] = dn[i]; h->dW[i] = mat(SE, dn[i]); h->db[i] = vec(dn[i]); }
    }
    h->pW1 = mat(E, P0); h->pb1 = vec(P0); h->pW2 = mat(P0, P1); h->pb2 = vec(P1);
    cbhg(h->enc, P1, d.enc_bank_size, d.enc_bank_channel_size, d.enc_proj_sizes, d.enc_proj_width, d.enc_highway_depth, RN);
    h->Wm = mat(ENC, A); h->Wq = mat(AS, A);
    h->av = vec(A); h->ag = vec(1); h->ab = vec(A); h->asb = vec(1);
    h->dpW1 = mat(M, d.dec_prenet_sizes[0]); h->dpb1 = vec(d.dec_prenet_sizes[0]);
    h->dpW2 = mat(d.dec_prenet_sizes[0], d.dec_prenet_sizes[1]); h->dpb2 = vec(d.dec_prenet_sizes[1]);
    // model_type 'simple' (tacotron.py:85-90): only the speaker embedding, concatenated inside the decoder (rnn_wrappers.py:425-432, 455-463)
    const bool simple = d.num_speakers > 1 && d.model_simple && SE != 1;
    const int SEc = simple ? SE : 0;
    const int ain = d.dec_prenet_sizes[1] + SEc + ENC;
    h->aWgm = mat(ain + AS, 2 * AS); h->abg = vec(2 * AS); h->aWcm = mat(ain + AS, AS); h->abc = vec(AS);
    h->cW = mat(AS + ENC + SEc, DR); h->cb = vec(DR);
    for (int i = 0; i < d.dec_layer_num; ++i) { h->rWg[i] = mat(2 * DR, 2 * DR); h->rbg[i] = vec(2 * DR); h->rWc[i] = mat(2 * DR, DR); h->rbc[i] = vec(DR); }
    h->oW = mat(DR, M * R); h->ob = vec(M * R);
    cbhg(h->post, M, d.post_bank_size, d.post_bank_channel_size, d.post_proj_sizes, d.post_proj_width, d.post_highway_depth, d.post_rnn_size);
    h->lW = mat(2 * d.post_rnn_size, d.num_freq); h->lb = vec(d.num_freq);
    h->blob_floats = src;
    h->nv = TVec{dst, A}; dst += (A + 3) / 4 * 4;      // derived: normed_v
    h->xt_off = dst; dst += 32LL * 8 * kXSlots * kTile;
    h->packed_floats = dst;
}

#define HIPCHK(expr)                                                                                              \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) return twv_fail(TWV_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

extern "C" int twv_tacotron_create(const twv_tacotron_dims* dims, twv_tacotron** out)
{
    if (!dims || !out) return twv_fail(TWV_E_INVALID, "null argument");
    const twv_tacotron_dims& d = *dims;
    if (d.enc_rnn_size != 128 || d.post_rnn_size != 128) return twv_fail(TWV_E_UNSUPPORTED, "enc_rnn_size and post_rnn_size must be 128");
    if (d.enc_proj_sizes[1] != d.enc_prenet_sizes[1] || d.enc_proj_sizes[1] != d.enc_rnn_size)
        return twv_fail(TWV_E_UNSUPPORTED, "encoder projection / prenet / rnn sizes must agree (modules.py:47-57)");
    if (d.post_proj_sizes[1] != d.num_mels) return twv_fail(TWV_E_INVALID, "post_proj_sizes[-1] must equal num_mels (modules.py:53)");
    if (d.attention_size % 32 || d.attention_size > 256 || d.attention_state_size != d.dec_rnn_size || d.dec_rnn_size % 64 || d.dec_rnn_size > 256)
        return twv_fail(TWV_E_UNSUPPORTED, "attention_size % 32, attention_state_size == dec_rnn_size <= 256 required");
    if (d.dec_prenet_sizes[1] % 32 || d.dec_prenet_sizes[0] > 512 || d.num_mels * d.reduction_factor > 512 || d.dec_layer_num > 4 || d.dec_layer_num < 1)
        return twv_fail(TWV_E_UNSUPPORTED, "decoder sizes out of range");
    if (d.enc_bank_size > 16 || d.post_bank_size > 16 || d.enc_highway_depth > 8 || d.post_highway_depth > 8) return twv_fail(TWV_E_UNSUPPORTED, "bank / highway depth out of range");
    if (d.num_speakers < 1) return twv_fail(TWV_E_INVALID, "num_speakers must be >= 1");
    if (d.num_speakers > 1 && d.speaker_embedding_size < 1) return twv_fail(TWV_E_INVALID, "speaker_embedding_size must be >= 1");
    if (d.num_speakers > 1 && d.model_simple && d.speaker_embedding_size != 1) {
        // the embedding rides at the tail of the buffer the residual GRU inputs [y | h] share with the first projection's input
        if (2 * d.dec_rnn_size > d.attention_state_size + 2 * d.enc_rnn_size || d.speaker_embedding_size > 64)
            return twv_fail(TWV_E_UNSUPPORTED, "model_type 'simple': needs 2 * dec_rnn_size <= attention_state_size + 2 * enc_rnn_size and speaker_embedding_size <= 64");
    }
    twv_tacotron* h = new twv_tacotron();
    h->d = d;
    taco_build(h);
    *out = h;
    return TWV_OK;
}
extern "C" void twv_tacotron_destroy(twv_tacotron* h) { delete h; }
extern "C" size_t twv_tacotron_blob_floats(const twv_tacotron* h) { return (size_t)h->blob_floats; }
extern "C" int twv_tacotron_set_option(twv_tacotron* h, const char* name, int value)
{
    if (!h || !name) return twv_fail(TWV_E_INVALID, "null argument");
    if (!strcmp(name, "gemm_valu")) { h->gemm_valu = value ? 1 : 0; return TWV_OK; }
    if (!strcmp(name, "highway_stack")) { h->hw_stack = value ? 1 : 0; return TWV_OK; }
    if (!strcmp(name, "gemm_group")) { h->gemm_group = value ? 1 : 0; return TWV_OK; }   // 0: one launch per GEMM, separate highway kernels (A/B runs, cross-check)
    if (!strcmp(name, "gemm_timing")) {   // 1: start counting (resets the sums), 0: stop
        g_gemm_stat.on = value != 0;
        g_gemm_stat.flop = 0.0; g_gemm_stat.launches = 0; g_gemm_stat.used = 0;
        return TWV_OK;
    }
    if (!strcmp(name, "decoder_split_all")) { h->dec_split_all = value < 0 ? -1 : (value ? 1 : 0); return TWV_OK; }
    if (!strcmp(name, "decoder_local")) { h->dec_local = value ? 1 : 0; return TWV_OK; }
    if (!strcmp(name, "xdec_nap_idle")) { h->xdec_nap_idle = value < 0 ? 0 : (value > 200 ? 200 : value); return TWV_OK; }     // tuning aids (round 6)
    if (!strcmp(name, "xdec_nap_owner")) { h->xdec_nap_owner = value < 0 ? 0 : (value > 200 ? 200 : value); return TWV_OK; }
    if (!strcmp(name, "xdec_nap_round")) { h->xdec_nap_round = value < 0 ? 0 : (value > 200 ? 200 : value); return TWV_OK; }
    if (!strcmp(name, "xdec_nap_w0")) { h->xdec_nap_w0 = value < 0 ? 0 : (value > 200 ? 200 : value); return TWV_OK; }
    if (!strcmp(name, "decoder_groups")) {
        if (value != -1 && value != 0 && value != 1 && value != 2 && value != 4 && value != 8 && value != 16 && value != 32)
            return twv_fail(TWV_E_INVALID, "decoder_groups must be -1, 0, 1, 2, 4, 8, 16 or 32 (32 = the XCD-local kernel)");
        h->dec_groups = value;
        return TWV_OK;
    }
    return twv_fail(TWV_E_INVALID, "unknown option");
}
extern "C" int twv_tacotron_set_profile_buffer(twv_tacotron* h, void* dev_u64) { if (!h) return 1; h->prof = (unsigned long long*)dev_u64; return 0; }
extern "C" int twv_tacotron_gemm_stats(twv_tacotron* h, double* flop, double* ms, int64_t* launches)
{
    if (!h || !flop || !ms || !launches) return twv_fail(TWV_E_INVALID, "null argument");
    double total = 0.0;
    for (size_t i = 0; i < g_gemm_stat.used; ++i) {
        if (hipEventSynchronize(g_gemm_stat.ev[i].second) != hipSuccess) return twv_fail(TWV_E_HIP, "hipEventSynchronize");
        float t = 0.0f;
        if (hipEventElapsedTime(&t, g_gemm_stat.ev[i].first, g_gemm_stat.ev[i].second) != hipSuccess) return twv_fail(TWV_E_HIP, "hipEventElapsedTime");
        total += t;
    }
    *flop = g_gemm_stat.flop; *ms = total; *launches = g_gemm_stat.launches;
    return TWV_OK;
}
extern "C" size_t twv_tacotron_packed_bytes(const twv_tacotron* h) { return (size_t)h->packed_floats * 4; }

__global__ void tc_normed_v_kernel(float* P, long long av, long long ag, long long nv, int A)
{
    // normed_v = g * v * rsqrt(sum(v^2)) [RECALLED-TF _bahdanau_score]; the sum as one AC-1 cdot
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float r = 0.0f;
        for (int k0 = 0; k0 < A; k0 += 32) {
            float s[4] = {0.f, 0.f, 0.f, 0.f};
            for (int k = k0; k < min(A, k0 + 32); ++k) s[(k - k0) & 3] = fma_(P[av + k], P[av + k], s[(k - k0) & 3]);
            const float c = (s[0] + s[1]) + (s[2] + s[3]);
            r = k0 == 0 ? c : r + c;
        }
        const float rs = div_(1.0f, __fsqrt_rn(r));
        for (int j = 0; j < A; ++j) { const float gv = P[ag] * P[av + j]; P[nv + j] = gv * rs; }
    }
}

// the decoder step as a stage table (same order and operands as tc_decoder_g_kernel's): used by the XCD-local kernel and its pack
static void taco_xstages(const twv_tacotron* h, XStageTab& t)
{
    const twv_tacotron_dims& d = h->d;
    const int M = d.num_mels, R = d.reduction_factor, D0 = d.dec_prenet_sizes[0], D1 = d.dec_prenet_sizes[1], AS = d.attention_state_size,
              A = d.attention_size, ENC = 2 * d.enc_rnn_size, DR = d.dec_rnn_size;
    int s = 0;
    auto put = [&](const TMat& w, long long b, int K, int N, int x, int dst, int act, int post, int p0, int p1, int p2) {
        t.w[s] = w.off; t.b[s] = b; t.K[s] = K; t.N[s] = N; t.xsel[s] = x; t.dsel[s] = dst; t.act[s] = act; t.post[s] = post;
        t.p0[s] = p0; t.p1[s] = p1; t.p2[s] = p2; ++s;
    };
    put(h->dpW1, h->dpb1.off, M, D0, XV_FRAME, XV_VEC, DA_RELU, DP_NONE, 0, 0, 0);                               // rnn_wrappers.py:425 prenet
    put(h->dpW2, h->dpb2.off, D0, D1, XV_VEC, XV_CATA, DA_RELU, DP_NONE, 0, 0, 0);
    put(h->aWgm, h->abg.off, D1 + ENC + AS, 2 * AS, XV_CATA, XV_VEC, DA_SIGMOID, DP_GATES, D1 + ENC, AS, 0);      // rnn_wrappers.py:310-312 attention GRU
    put(h->aWcm, h->abc.off, D1 + ENC + AS, AS, XV_CATA, XV_VEC, DA_TANH, DP_CAND, AS, 0, -1);
    put(h->Wq, -1, AS, A, XV_CATB, XV_PQ, DA_NONE, DP_QUERY, 0, 0, 0);                                            // attention query layer on the new state
    put(h->cW, h->cb.off, AS + ENC, DR, XV_CATB, XV_CATR, DA_NONE, DP_PROJ, 0, 0, 0);                             // rnn_wrappers.py:463 + OutputProjectionWrapper -> y
    for (int l = 0; l < d.dec_layer_num; ++l) {                                                                  // tacotron.py:167 residual GRUs
        put(h->rWg[l], h->rbg[l].off, 2 * DR, 2 * DR, XV_CATR + l, XV_VEC, DA_SIGMOID, DP_GATES, DR, DR, 0);
        put(h->rWc[l], h->rbc[l].off, 2 * DR, DR, XV_CATR + l, XV_VEC, DA_TANH, DP_CAND, DR, 1 + l, l + 1 < d.dec_layer_num ? 2 + l : 0);
    }
    put(h->oW, h->ob.off, DR, M * R, XV_Y, XV_VEC, DA_NONE, DP_OUT, 0, 0, 0);                                     // tacotron.py:173
    t.nst = s;
}
// does the XCD-local kernel take this model?  (16-column slices over 32 workgroups, register slots per wave, LDS)
static bool taco_xdec_ok(const twv_tacotron* h, const XStageTab& t)
{
    const twv_tacotron_dims& d = h->d;
    if (d.num_speakers > 1 && d.model_simple && d.speaker_embedding_size != 1) return false;      // 'simple': the split kernel carries the embedding concat
    if (t.nst > kXStages || d.attention_size % 32 || (2 * d.enc_rnn_size) % 32 || d.attention_size > 256 || d.dec_layer_num > 4) return false;
    // (the stage inputs are fetched with 16-byte LDS reads: every vector of an utterance's LDS block starts on a multiple of four floats)
    if (d.dec_prenet_sizes[1] % 4 || d.attention_state_size % 4 || d.dec_rnn_size % 4) return false;
    for (int s = 0; s < t.nst; ++s) if (t.N[s] > 512 || t.K[s] > 24 * 32) return false;
    for (int g = 0; g < 32; ++g) {
        int a[kXSlots], b[kXSlots];
        if (xdec_tasks(t, g, 0, a, b) > kXSlots) return false;
    }
    return true;
}

extern "C" int twv_tacotron_pack(const twv_tacotron* h, const float* blob, void* packed, void* stream)
{
    if (!h || !blob || !packed) return twv_fail(TWV_E_INVALID, "null argument");
    hipStream_t st = (hipStream_t)stream;
    float* dst = (float*)packed;
    HIPCHK(hipMemsetAsync(dst, 0, (size_t)h->packed_floats * 4, st));
    for (const auto& it : h->items) {
        if (it.kind == 1) twv_launch_copy(dst + it.dst, blob + it.src, (long long)it.K * it.N, st);
        else if (it.kind == 2) {    // highway pair tiles: group g = 32 output columns; lanes 0-31 <- H[:, 32g + l], lanes 32-63 <- T[:, 32g + l]
            const int nchunk = (it.K + 31) / 32;
            PackTiles p{it.dst, (long long)nchunk * kTile, it.src, it.src + it.r0, 32, it.N / 32, 1, nchunk, it.K, it.N, it.N, 1, 64};
            twv_launch_pack_tiles(dst, blob, p, st);
        }
        else {
            const int K = it.r1 - it.r0;
            PackTiles p{it.dst, 0, it.src + (long long)it.r0 * it.N, 0, 0, 1, (it.N + 63) / 64, (K + 31) / 32, K, it.N, it.N, 0, 64};
            twv_launch_pack_tiles(dst, blob, p, st);
        }
    }
    hipLaunchKernelGGL(tc_normed_v_kernel, dim3(1), dim3(64), 0, st, dst, h->av.off, h->ag.off, h->nv.off, h->d.attention_size);
    {   // row tiles of the XCD-local decoder kernel, from the standard tiles just written
        XStageTab xt;
        taco_xstages(h, xt);
        if (taco_xdec_ok(h, xt)) hipLaunchKernelGGL(tc_xdec_pack_kernel, dim3(32 * 8), dim3(256), 0, st, dst, xt, h->xt_off);
    }
    HIPCHK(hipGetLastError());
    return TWV_OK;
}

// workspace (floats) for N utterances of T tokens
static long long taco_ws_floats(const twv_tacotron* h, int N, int T)
{
    const twv_tacotron_dims& d = h->d;
    const long long rowsE = (long long)N * T, rowsP = (long long)N * d.max_iters * d.reduction_factor;
    const long long rows = rowsE > rowsP ? rowsE : rowsP;
    const int CBe = d.enc_bank_size * d.enc_bank_channel_size, CBp = d.post_bank_size * d.post_bank_channel_size;
    const int CB = CBe > CBp ? CBe : CBp;
    long long f = 0;
    f += rows * CB * 2;                 // bank output + maxpool output
    f += rows * 512 * 4;                // generic row buffers (<= 512 wide): a, b, c, d
    f += rows * 256 * 3;                // Gx (2 dirs x 256), Cx (2 x 128) -> 768 per row
    f += rowsE * 256 * 2;               // encoder output (memory), keys
    f += (long long)N * 4096;           // speaker-dependent vectors
    f += rowsP * 256;                   // post CBHG output
    f += (long long)N * 4096 + 16;      // decoder exchange granules + the local mode's role tickets
    f += 512 * 256;                     // split decoder: stage tables
    f += 8LL * 2 * kXU * 512 * 2 + 64;  // XCD-local decoder: exchange granules per XCD + role tickets
    return f + 1024;
}
extern "C" size_t twv_tacotron_workspace_bytes(const twv_tacotron* h, int batch, int t_in) { return (size_t)taco_ws_floats(h, batch, t_in) * 4; }

static inline int tgrid(long long n) { long long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g)); }

static GemmArgs gemm_args(const float* P, const float* X, int ldx, int rows, int T, int Cin, int kw, const TMat& W,
                          const TVec* bias, int act, const TVec* inv, const TVec* shift, const float* add1, int ld1, const float* add2,
                          int ld2, float* Y, int ldy, int col0)
{
    GemmArgs a;
    a.X = X; a.ldx = ldx; a.rows = rows; a.T = T; a.Cin = Cin; a.kw = kw; a.pl = (kw - 1) / 2;
    a.Wt = P + W.off; a.K = W.K; a.N = W.N;
    a.bias = bias ? P + bias->off : nullptr; a.act = act;
    a.bn_inv = inv ? P + inv->off : nullptr; a.bn_shift = shift ? P + shift->off : nullptr;
    a.add1 = add1; a.ld1 = ld1; a.add2 = add2; a.ld2 = ld2; a.Y = Y; a.ldy = ldy; a.col0 = col0;
    a.bias2 = nullptr; a.mode = 0;
    return a;
}
// "gemm_timing": a pair of events around the launch(es) of one call, useful FLOPs counted per problem
struct GemmTimed {
    hipEvent_t e1 = nullptr;
    hipStream_t st;
    GemmTimed(hipStream_t st_, double flop, int launches) : st(st_)
    {
        if (!g_gemm_stat.on) return;
        if (g_gemm_stat.used == g_gemm_stat.ev.size()) {
            hipEvent_t x0, x1;
            if (hipEventCreate(&x0) == hipSuccess && hipEventCreate(&x1) == hipSuccess) g_gemm_stat.ev.push_back({x0, x1});
        }
        if (g_gemm_stat.used < g_gemm_stat.ev.size()) {
            hipEvent_t e0 = g_gemm_stat.ev[g_gemm_stat.used].first;
            e1 = g_gemm_stat.ev[g_gemm_stat.used].second;
            ++g_gemm_stat.used;
            (void)hipEventRecord(e0, st);
        }
        g_gemm_stat.flop += flop;
        g_gemm_stat.launches += launches;
    }
    ~GemmTimed() { if (e1) (void)hipEventRecord(e1, st); }
};
static void launch_gemm_args(hipStream_t st, const GemmArgs& a)
{
    // mode 1 (highway pair): N counts H and T columns, each useful once
    GemmTimed tm(st, 2.0 * (double)a.rows * (double)a.K * (double)a.N, 1);
    if (g_gemm_valu)
        hipLaunchKernelGGL(tc_gemm_kernel, dim3((a.rows + kGemmRows - 1) / kGemmRows), dim3(256), kGemmRows * kGemmKS * 4, st, a);
    else {
        const int nchunk_ = (a.K + 31) / 32;
        const long long wgs = (long long)((a.rows + kMmRows - 1) / kMmRows) * ((a.N + 127) / 128);
        if (a.mode == 1)
            hipLaunchKernelGGL(tc_gemm_mfma_highway_kernel, dim3((a.rows + kMmRows - 1) / kMmRows, (a.N + 127) / 128), dim3(256), 0, st, a);
        else if (wgs < 192 && nchunk_ >= 8)      // too few tiles for the chip and a deep contraction: chunk-parallel waves
            hipLaunchKernelGGL(tc_gemm_mfma_ck_kernel, dim3((a.rows + 31) / 32, (a.N + 63) / 64), dim3(256), 0, st, a);
        else
            hipLaunchKernelGGL(tc_gemm_mfma_kernel, dim3((a.rows + kMmRows - 1) / kMmRows, (a.N + 127) / 128), dim3(256), 0, st, a);
    }
}
static void launch_gemm(hipStream_t st, const float* P, const float* X, int ldx, int rows, int T, int Cin, int kw, const TMat& W,
                        const TVec* bias, int act, const TVec* inv, const TVec* shift, const float* add1, int ld1, const float* add2,
                        int ld2, float* Y, int ldy, int col0)
{
    launch_gemm_args(st, gemm_args(P, X, ldx, rows, T, Cin, kw, W, bias, act, inv, shift, add1, ld1, add2, ld2, Y, ldy, col0));
}
// independent problems as ONE grid (tc_gemm_mfma_group_kernel), deepest contraction first; with the "gemm_valu" cross-check option, or when
// grouping is switched off ("gemm_group" = 0: A/B runs), one launch per problem as before
static void launch_gemm_group(hipStream_t st, std::vector<GemmArgs> v)
{
    if (v.empty()) return;
    if (g_gemm_valu || !g_gemm_group || v.size() == 1) { for (const auto& a : v) launch_gemm_args(st, a); return; }
    std::stable_sort(v.begin(), v.end(), [](const GemmArgs& x, const GemmArgs& y) { return x.K > y.K; });
    for (size_t i0 = 0; i0 < v.size(); i0 += kGroupMax) {
        GemmGroup g;
        g.n = (int)std::min<size_t>(kGroupMax, v.size() - i0);
        double flop = 0.0;
        int wg = 0;
        for (int p = 0; p < g.n; ++p) {
            const GemmArgs& a = v[i0 + p];
            g.a[p] = a;
            g.start[p] = wg;
            g.nby[p] = (a.N + 127) / 128;
            wg += ((a.rows + kMmRows - 1) / kMmRows) * g.nby[p];
            flop += 2.0 * (double)a.rows * (double)a.K * (double)a.N;
        }
        for (int p = g.n; p <= kGroupMax; ++p) g.start[p] = wg;
        for (int p = g.n; p < kGroupMax; ++p) { g.nby[p] = 1; g.a[p] = v[i0]; }
        GemmTimed tm(st, flop, 1);
        hipLaunchKernelGGL(tc_gemm_mfma_group_kernel, dim3(wg), dim3(256), 0, st, g);
    }
}

// modules.py:25-74 for `rows` = N*T rows
static void run_cbhg(hipStream_t st, const twv_tacotron* h, const float* P, const TCbhg& c, const float* in, int Cin, int N, int T,
                     int bank, int bch, const int32_t* proj, int pw, int depth, const float* before_hw, const float* init,
                     const int32_t* lengths, float* bankbuf, float* poolbuf, float* ra, float* rb, float* rc, float* gx, float* cx, float* out)
{
    const int rows = N * T, CB = bank * bch, rnn = 128;
    {                                   // conv bank -> concatenated channels: `bank` independent GEMMs, one grid
        std::vector<GemmArgs> v;
        for (int k = 1; k <= bank; ++k)
            v.push_back(gemm_args(P, in, Cin, rows, T, Cin, k, c.W[k], &c.b[k], TACT_RELU, &c.inv[k], &c.shift[k], nullptr, 0, nullptr, 0, bankbuf, CB, (k - 1) * bch));
        launch_gemm_group(st, v);
    }
    if (CB % 4 == 0) hipLaunchKernelGGL(tc_maxpool2_kernel, dim3(tgrid((long long)rows * (CB / 4))), dim3(256), 0, st, bankbuf, rows, T, CB, poolbuf);
    else hipLaunchKernelGGL(tc_maxpool2_scalar_kernel, dim3(tgrid((long long)rows * CB)), dim3(256), 0, st, bankbuf, rows, T, CB, poolbuf);
    launch_gemm(st, P, poolbuf, CB, rows, T, CB, pw, c.pW[0], &c.pb[0], TACT_RELU, &c.pinv[0], &c.pshift[0], nullptr, 0, nullptr, 0, ra, proj[0], 0);
    // second projection + residual: (proj + inputs) + before_highway
    launch_gemm(st, P, ra, proj[0], rows, T, proj[0], pw, c.pW[1], &c.pb[1], TACT_NONE, &c.pinv[1], &c.pshift[1], in, Cin, before_hw, before_hw ? rnn : 0, rb, proj[1], 0);
    float* hw = rb;
    if (c.has_dense) { launch_gemm(st, P, rb, proj[1], rows, T, proj[1], 1, c.dW, &c.db, TACT_NONE, nullptr, nullptr, nullptr, 0, nullptr, 0, rc, rnn, 0); hw = rc; }
    float* hH = ra;
    float* hT = (hw == rc) ? rb : rc;
    if (!g_gemm_valu && g_gemm_group && g_hw_stack && depth >= 1 && depth <= kHwMaxDepth) {
        // the whole stack in one launch: rows stay in LDS between the layers
        HwStackArgs ha;
        ha.X = hw; ha.ldx = rnn; ha.rows = rows; ha.depth = depth;
        for (int i = 0; i < depth; ++i) { ha.Wt[i] = P + c.hHT[i].off; ha.bh[i] = P + c.hHb[i].off; ha.bt[i] = P + c.hTb[i].off; }
        ha.Y = hH; ha.ldy = rnn;
        {
            GemmTimed tm(st, 2.0 * (double)rows * (double)rnn * (double)(2 * rnn) * depth, 1);
            if ((rows + 63) / 64 >= 256) {
                const size_t shm = (size_t)2 * 64 * kHwPitch * 4;
                (void)hipFuncSetAttribute((const void*)tc_highway_stack_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
                hipLaunchKernelGGL(tc_highway_stack_kernel<2>, dim3((rows + 63) / 64), dim3(256), shm, st, ha);
            } else {
                const size_t shm = (size_t)2 * 32 * kHwPitch * 4;
                hipLaunchKernelGGL(tc_highway_stack_kernel<1>, dim3((rows + 31) / 32), dim3(256), shm, st, ha);
            }
        }
        hw = hH;
    } else
    for (int i = 0; i < depth; ++i) {
        if (!g_gemm_valu && g_gemm_group) {
            // the whole highway layer in one launch (mm_body, mode 1): H and T columns interleaved in the tiles, output to the other buffer
            GemmArgs a = gemm_args(P, hw, rnn, rows, T, rnn, 1, c.hHT[i], &c.hHb[i], TACT_NONE, nullptr, nullptr, hw, rnn, nullptr, 0, hH, rnn, 0);
            a.bias2 = P + c.hTb[i].off; a.mode = 1;
            launch_gemm_args(st, a);
            float* t = hw; hw = hH; hH = t;
            if (hT == hw) hT = hH;
            continue;
        }
        launch_gemm(st, P, hw, rnn, rows, T, rnn, 1, c.hH[i], &c.hHb[i], TACT_RELU, nullptr, nullptr, nullptr, 0, nullptr, 0, hH, rnn, 0);
        launch_gemm(st, P, hw, rnn, rows, T, rnn, 1, c.hT[i], &c.hTb[i], TACT_SIGMOID, nullptr, nullptr, nullptr, 0, nullptr, 0, hT, rnn, 0);
        hipLaunchKernelGGL(tc_highway_kernel, dim3(tgrid((long long)rows * rnn)), dim3(256), 0, st, hH, hT, hw, (long long)rows * rnn);
    }
    // biGRU: hoisted x halves (four GEMMs over the same rows: one grid), then the recurrent kernel
    {
        std::vector<GemmArgs> v;
        for (int dr = 0; dr < 2; ++dr) {
            v.push_back(gemm_args(P, hw, rnn, rows, T, rnn, 1, c.gWgx[dr], nullptr, TACT_NONE, nullptr, nullptr, nullptr, 0, nullptr, 0, gx + (long long)dr * rows * 2 * rnn, 2 * rnn, 0));
            v.push_back(gemm_args(P, hw, rnn, rows, T, rnn, 1, c.gWcx[dr], nullptr, TACT_NONE, nullptr, nullptr, nullptr, 0, nullptr, 0, cx + (long long)dr * rows * rnn, rnn, 0));
        }
        launch_gemm_group(st, v);
    }
    // (rows past an utterance's length are not written by the recurrent kernel: zeros; without lengths every row is)
    if (lengths) hipLaunchKernelGGL(tc_zero_kernel, dim3(tgrid((long long)rows * 2 * rnn)), dim3(256), 0, st, out, (long long)rows * 2 * rnn);
    GruSeqArgs g;
    g.Gx = gx; g.Cx = cx; g.gx_dstride = (long long)rows * 2 * rnn; g.cx_dstride = (long long)rows * rnn;
    for (int dr = 0; dr < 2; ++dr) { g.Wgh[dr] = P + c.gWgh[dr].off; g.Wch[dr] = P + c.gWch[dr].off; g.bg[dr] = P + c.gbg[dr].off; g.bc[dr] = P + c.gbc[dr].off; }
    g.init = init; g.lengths = lengths; g.T = T; g.out = out;
    hipLaunchKernelGGL(tc_gru_seq_kernel, dim3(N * 2), dim3(512), 384 * 4, st, g);
    (void)h;
}

extern "C" int twv_tacotron_infer(const twv_tacotron* h, const void* packed, const int32_t* tokens, const int32_t* lengths,
                                  const int32_t* speaker_ids, int batch, int t_in, void* workspace, float* mel, float* linear,
                                  float* alignments, int32_t* status, void* stream)
{
    if (!h || !packed || !tokens || !lengths || !workspace || !mel || !status) return twv_fail(TWV_E_INVALID, "null argument");
    if (!speaker_ids && h->d.num_speakers > 1) return twv_fail(TWV_E_INVALID, "speaker_ids is required for a multi-speaker model");
    if (batch < 1 || t_in < 1 || t_in > 1024) return twv_fail(TWV_E_INVALID, "batch >= 1 and 1 <= t_in <= 1024 required");
    g_gemm_valu = h->gemm_valu; g_gemm_group = h->gemm_group; g_hw_stack = h->hw_stack;
    const twv_tacotron_dims& d = h->d;
    hipStream_t st = (hipStream_t)stream;
    const float* P = (const float*)packed;
    const int N = batch, T = t_in, rows = N * T;
    const int E = d.embedding_size, SE = d.speaker_embedding_size, P0 = d.enc_prenet_sizes[0], P1 = d.enc_prenet_sizes[1], RN = 128,
              A = d.attention_size, AS = d.attention_state_size, DR = d.dec_rnn_size, M = d.num_mels, R = d.reduction_factor, ENC = 256;
    const int TO = d.max_iters * R, rowsP = N * TO;
    const long long rmax = rows > rowsP ? rows : rowsP;
    const int CBe = d.enc_bank_size * d.enc_bank_channel_size, CBp = d.post_bank_size * d.post_bank_channel_size, CB = CBe > CBp ? CBe : CBp;
    HIPCHK(hipMemsetAsync(status, 0, 16, st));
    // ---- workspace carve
    float* w = (float*)workspace;
    float* bankbuf = w; w += rmax * CB;
    float* poolbuf = w; w += rmax * CB;
    float* ra = w; w += rmax * 512;
    float* rb = w; w += rmax * 512;
    float* rc = w; w += rmax * 512;
    float* rd = w; w += rmax * 512;
    float* gx = w; w += rmax * 512;
    float* cx = w; w += rmax * 256;
    float* memo = w; w += (long long)rows * 256;
    float* keys = w; w += (long long)rows * 256;
    float* spk = w; w += (long long)N * 4096;
    float* postout = w; w += (long long)rowsP * 256;
    float* exch = w; w += (long long)N * 2 * kExN * 2 + 16; // decoder exchange granules (8 bytes each) + 8 role tickets
    float* stabf = w; w += 512 * 256;                       // the split decoder's per-workgroup stage tables (up to 512 workgroups)
    float* xexch = w; w += 8LL * 2 * kXU * 512 * 2 + 64;    // XCD-local decoder: granules [8][2][kXU*512] + tickets
    // ---- tacotron.py:51-60 embedding, :67-82 speaker embedding + deep_dense (softsign)
    hipLaunchKernelGGL(tc_embed_kernel, dim3(tgrid((long long)rows * E)), dim3(256), 0, st, P + h->emb.off, tokens, rows, E, ra);
    const bool multi = d.num_speakers > 1;
    const bool simple = multi && d.model_simple && SE != 1;     // tacotron.py:85-90: no speaker-dependent states, the embedding goes into the decoder
    // spk layout: [N][SE] at 0, then per dense i a [N][dn_i] block
    float* sv[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // decoder initial states gathered as [N][AS + layers*DR]
    float* dinit = spk + (long long)N * 2048;
    if (simple) {
        hipLaunchKernelGGL(tc_gather_rows_kernel, dim3(tgrid((long long)N * SE)), dim3(256), 0, st, P + h->semb.off, speaker_ids, N, SE, spk);
        HIPCHK(hipMemsetAsync(dinit, 0, (size_t)N * (AS + d.dec_layer_num * DR) * 4, st));
    } else if (multi) {
        { float* q = spk + (long long)N * 64; for (int i = 0; i < h->ndense; ++i) { sv[i] = q; q += (long long)N * h->dn[i]; } }
        if (SE == 1) {
            // tacotron.py:69-75 get_embed: tf.nn.embedding_lookup of five tables by speaker id
            for (int i = 0; i < h->ndense; ++i)
                hipLaunchKernelGGL(tc_gather_rows_kernel, dim3(tgrid((long long)N * h->dn[i])), dim3(256), 0, st, P + h->stab[i].off, speaker_ids, N, h->dn[i], sv[i]);
        } else {
            hipLaunchKernelGGL(tc_gather_rows_kernel, dim3(tgrid((long long)N * SE)), dim3(256), 0, st, P + h->semb.off, speaker_ids, N, SE, spk);
            // (the decoder's initial states -- dense layers 2 .. -- are written where the decoder reads them: [N][AS + layers * DR])
            std::vector<GemmArgs> v;
            const int dstride = AS + d.dec_layer_num * DR;
            for (int i = 0; i < h->ndense; ++i) {
                if (i >= 2 && i < 3 + d.dec_layer_num)
                    v.push_back(gemm_args(P, spk, SE, N, 1, SE, 1, h->dW[i], &h->db[i], TACT_SOFTSIGN, nullptr, nullptr, nullptr, 0, nullptr, 0, dinit, dstride,
                                          i == 2 ? 0 : AS + (i - 3) * DR));
                else
                    v.push_back(gemm_args(P, spk, SE, N, 1, SE, 1, h->dW[i], &h->db[i], TACT_SOFTSIGN, nullptr, nullptr, nullptr, 0, nullptr, 0, sv[i], h->dn[i], 0));
            }
            launch_gemm_group(st, v);
        }
        for (int i = 0; SE == 1 && i < 1 + d.dec_layer_num; ++i) {
            const int wdt = i == 0 ? AS : DR;
            HIPCHK(hipMemcpy2DAsync(dinit + (i == 0 ? 0 : AS + (i - 1) * DR), (size_t)(AS + d.dec_layer_num * DR) * 4, sv[2 + i], (size_t)wdt * 4,
                                    (size_t)wdt * 4, N, hipMemcpyDeviceToDevice, st));
        }
    } else {
        // tacotron.py:97-104: no before_highway (sv[0] stays null: the residual is proj + inputs), zero GRU / attention-cell states
        HIPCHK(hipMemsetAsync(dinit, 0, (size_t)N * (AS + d.dec_layer_num * DR) * 4, st));
    }
    // ---- tacotron.py:108 prenet, :113 encoder CBHG
    launch_gemm(st, P, ra, E, rows, T, E, 1, h->pW1, &h->pb1, TACT_RELU, nullptr, nullptr, nullptr, 0, nullptr, 0, rb, P0, 0);
    launch_gemm(st, P, rb, P0, rows, T, P0, 1, h->pW2, &h->pb2, TACT_RELU, nullptr, nullptr, nullptr, 0, nullptr, 0, rd, P1, 0);
    run_cbhg(st, h, P, h->enc, rd, P1, N, T, d.enc_bank_size, d.enc_bank_channel_size, d.enc_proj_sizes, d.enc_proj_width, d.enc_highway_depth,
             sv[0], sv[1], lengths, bankbuf, poolbuf, ra, rb, rc, gx, cx, memo);
    // memory is already zero past the lengths (pre-zeroed GRU output); keys = memory_layer(memory)
    launch_gemm(st, P, memo, ENC, rows, T, ENC, 1, h->Wm, nullptr, TACT_NONE, nullptr, nullptr, nullptr, 0, nullptr, 0, keys, A, 0);
    // ---- decoder
    DecArgs da;
    da.P = P;
    da.w.dp1 = h->dpW1.off; da.w.dp1b = h->dpb1.off; da.w.dp2 = h->dpW2.off; da.w.dp2b = h->dpb2.off;
    da.w.aWg = h->aWgm.off; da.w.abg = h->abg.off; da.w.aWc = h->aWcm.off; da.w.abc = h->abc.off;
    da.w.Wq = h->Wq.off; da.w.nv = h->nv.off; da.w.ab = h->ab.off; da.w.asb = h->asb.off; da.w.cW = h->cW.off; da.w.cb = h->cb.off;
    for (int i = 0; i < d.dec_layer_num; ++i) { da.w.rWg[i] = h->rWg[i].off; da.w.rbg[i] = h->rbg[i].off; da.w.rWc[i] = h->rWc[i].off; da.w.rbc[i] = h->rbc[i].off; }
    da.w.oW = h->oW.off; da.w.ob = h->ob.off;
    da.keys = keys; da.memo = memo; da.init = dinit; da.lengths = lengths;
    da.N = N; da.T = T; da.M = M; da.R = R; da.D0 = d.dec_prenet_sizes[0]; da.D1 = d.dec_prenet_sizes[1]; da.A = A; da.AS = AS; da.ENC = ENC;
    da.DR = DR; da.layers = d.dec_layer_num; da.iters = d.max_iters; da.mel = mel; da.align = alignments; da.status = status;
    da.SEc = simple ? SE : 0; da.semb = simple ? spk : nullptr;
    da.packed_bytes = (long long)h->packed_floats * 4;
    da.prof = h->prof;
    da.nbias = da.D0 + da.D1 + 3 * AS + DR + d.dec_layer_num * 3 * DR + M * R;
    {
        const int Tp = (T + 3) / 4 * 4;
        const int ain = da.D1 + da.SEc + ENC;
        const int kmax = (ain + AS) > 2 * DR ? (ain + AS) : 2 * DR;
        const long long part = (long long)((kmax + 31) / 32) * ((2 * (AS > DR ? AS : DR) + 63) / 64) * 64;
        const long long part2 = (long long)((DR + 31) / 32) * ((M * R + 63) / 64) * 64;
        const long long part3 = (long long)((T + 31) / 32) * ENC;
        long long pmax = part > part2 ? part : part2;
        pmax = pmax > part3 ? pmax : part3;
        // G workgroups per utterance, all N*G co-resident (they exchange through polled granules)
        int cus = 0, devid = 0;
        HIPCHK(hipGetDevice(&devid));
        HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, devid));
        int G = h->dec_groups > 0 ? h->dec_groups : 16;        // 16 workgroups per utterance up to batch 16, 8 up to 32, 4 up to 64, ...
        while (G > 1 && (long long)N * G > cus) G >>= 1;
        if (2 * (AS > DR ? AS : DR) > kExN || M * R > kExN || T > kExN || ENC > kExN) G = 1;
        while (G > 1 && ENC % G) G >>= 1;
        XStageTab xt;
        taco_xstages(h, xt);
        const int upx = (N + 7) / 8;
        long long xfl = 0;
        {   // LDS of tc_decoder_x_kernel (the kernel's own carve)
            const int nu = upx < N ? upx : N;
            xfl = xdec_carve(M, da.D1, ENC, AS, d.dec_layer_num, DR, A, T, nu < 1 ? 1 : nu, xt.nst).total + 64;
        }
        const bool xok = taco_xdec_ok(h, xt) && cus >= 256 && upx <= kXU && T <= 512 && xfl * 4 <= 160 * 1024 && !simple;
        if (simple && (h->dec_groups == -1 || h->dec_groups == 32))
            return twv_fail(TWV_E_UNSUPPORTED, "model_type 'simple' runs on the split decoder kernel only (decoder_groups 0, 1, 2, 4, 8 or 16)");
        // Which kernel (round 6, scripts/tacotron_bench.py --batch 8 / 16 / 24 / 32): the XCD-resident kernel is the default wherever it fits
        // (up to four utterances per XCD = batch 32); pass times against the split kernel are in profiles/r06_tacotron_decoder_ab.txt.
        // (Until its scratch spills and per-utterance branch chains were removed it tied the split kernel at four utterances per XCD.)
        if (xok && (h->dec_groups == 32 || h->dec_groups == 0)) {
            // XCD-local kernel: every XCD's 32 workgroups hold the decoder in registers and serve that XCD's utterances
            DecXArgs xa;
            xa.d = da; xa.tab = xt; xa.upx = upx; xa.xt_off = h->xt_off; xa.stab = reinterpret_cast<int*>(stabf);
            xa.nap_idle = h->xdec_nap_idle; xa.nap_owner = h->xdec_nap_owner; xa.nap_round = h->xdec_nap_round; xa.nap_w0 = h->xdec_nap_w0;
            xa.exch = reinterpret_cast<unsigned long long*>(xexch);
            xa.tickets = reinterpret_cast<int*>(xexch + 8LL * 2 * kXU * 512 * 2);
            HIPCHK(hipMemsetAsync(xexch, 0, (size_t)(8LL * 2 * kXU * 512 * 2 + 64) * 4, st));
            const size_t shm = (size_t)xfl * 4;
            const bool xdef = M == 80 && R == 5 && A == 256 && AS == 256 && ENC == 256 && DR == 256 && da.D1 == 128 && d.dec_layer_num == 2 && xt.nst == 11;
            const bool mm = upx > 2;
            const void* kfn = da.prof ? (xdef ? (mm ? (const void*)tc_decoder_x_kernel<true, true, true> : (const void*)tc_decoder_x_kernel<true, true, false>)
                                              : (mm ? (const void*)tc_decoder_x_kernel<true, false, true> : (const void*)tc_decoder_x_kernel<true, false, false>))
                                      : (xdef ? (mm ? (const void*)tc_decoder_x_kernel<false, true, true> : (const void*)tc_decoder_x_kernel<false, true, false>)
                                              : (mm ? (const void*)tc_decoder_x_kernel<false, false, true> : (const void*)tc_decoder_x_kernel<false, false, false>));
            HIPCHK(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
            void* kargs[] = {&xa};
            HIPCHK(hipLaunchKernel(kfn, dim3(2 * cus), dim3(512), kargs, shm, st));
        } else if (h->dec_groups == 32) {
            return twv_fail(TWV_E_UNSUPPORTED, "decoder_groups = 32 (XCD-local decoder) needs 256 CUs, batch <= 32, t_in <= 512 and the default decoder sizes");
        } else
        if (h->dec_groups == -1) {       // the single-workgroup kernel (kept as a cross-check of the split one)
            const long long fl = 2048 + AS + d.dec_layer_num * DR + (M + 31) / 32 * 32 + ENC + DR + Tp * 4 + A + Tp * 8 + pmax;
            const size_t shm = (size_t)fl * 4;
            if (shm > 160 * 1024) return twv_fail(TWV_E_UNSUPPORTED, "decoder LDS footprint exceeds 160 KiB (t_in too large)");
            HIPCHK(hipFuncSetAttribute((const void*)tc_decoder_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
            hipLaunchKernelGGL(tc_decoder_kernel, dim3(N), dim3(512), shm, st, da);
        } else {
            DecGArgs ga;
            ga.d = da; ga.G = G; ga.exch = reinterpret_cast<unsigned long long*>(exch);
            const int wgs_local = 8 * G * ((N + 7) / 8);
            // 'local' maps workgroup -> (utterance, slice) by per-XCD tickets and assumes EIGHT XCDs that each receive G * ceil(N / 8)
            // of the launch's workgroups: true of the 256-CU part in SPX mode (the same guard as the WaveNet XCD kernels, use_xcd).  On a
            // part / partition with fewer XCDs or CUs (6-XCD devices, CPX) utterances whose n % 8 names a missing XCD would get no
            // workgroup at all and their outputs would stay unwritten -- there the spread mapping is used.
            ga.local = (h->dec_local && G > 1 && wgs_local <= cus && cus >= 256) ? 1 : 0;
            const int wgs = ga.local ? wgs_local : N * G;
            ga.split_all = h->dec_split_all < 0 ? ga.local : h->dec_split_all;
            ga.stab = reinterpret_cast<int*>(stabf);
            if (wgs > 512) return twv_fail(TWV_E_UNSUPPORTED, "more than 512 decoder workgroups");
            HIPCHK(hipMemsetAsync(exch, 0, (size_t)N * 2 * kExN * 8 + 64, st));
            ga.tickets = reinterpret_cast<int*>(exch + (long long)N * 2 * kExN * 2);
            long long fl = 1024 * 2 + (da.D1 + da.SEc + ENC + AS + 63) / 64 * 64 + 512 * 2 + AS + d.dec_layer_num * DR + (M + 31) / 32 * 32 + ENC + DR + (M * R + 63) / 64 * 64 + Tp * 4 + A +
                           Tp * 8 + 4 + 3 + 16 * 16 + 2 * A + 3 * (A / 8) + da.nbias + pmax;
            const long long kvf = (long long)((T + G - 1) / G) * (A + A / 8) + (long long)T * (ENC / G + 8);
            ga.kv_lds = (fl + kvf) * 4 <= 160 * 1024 ? 1 : 0;
            if (ga.kv_lds) fl += kvf;
            const size_t shm = (size_t)fl * 4;
            if (shm > 160 * 1024) return twv_fail(TWV_E_UNSUPPORTED, "decoder LDS footprint exceeds 160 KiB (t_in too large)");
            // plain launch (N * G <= CU count is enforced above); a cooperative launch was measured and dropped, see twv_wavenet.hip
            // the hparams-default sizes have an instantiation of their own (sizes folded: half the scalar-register spills); the
            // instrumented build (phase stamps) likewise
            const bool def = !simple && G == 8 && M == 80 && R == 5 && A == 256 && AS == 256 && ENC == 256 && DR == 256 && da.D0 == 256 && da.D1 == 128 && d.dec_layer_num == 2;
            const void* kfn = da.prof ? (def ? (const void*)tc_decoder_g_kernel<true, true> : (const void*)tc_decoder_g_kernel<true, false>)
                                      : (def ? (const void*)tc_decoder_g_kernel<false, true> : (const void*)tc_decoder_g_kernel<false, false>);
            HIPCHK(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
            void* kargs[] = {&ga};
            HIPCHK(hipLaunchKernel(kfn, dim3(wgs), dim3(512), kargs, shm, st));
        }
    }
    // ---- tacotron.py:209 post CBHG (no lengths, zero init), :219 linear projection
    if (linear) {
        run_cbhg(st, h, P, h->post, mel, M, N, TO, d.post_bank_size, d.post_bank_channel_size, d.post_proj_sizes, d.post_proj_width,
                 d.post_highway_depth, nullptr, nullptr, nullptr, bankbuf, poolbuf, ra, rb, rc, gx, cx, postout);
        launch_gemm(st, P, postout, 256, rowsP, TO, 256, 1, h->lW, &h->lb, TACT_NONE, nullptr, nullptr, nullptr, 0, nullptr, 0, linear, d.num_freq, 0);
    }
    HIPCHK(hipGetLastError());
    return TWV_OK;
}

// the decoder kernel twv_tacotron_infer launches for this (handle, batch, t_in, options) on the current device -- the rules of the
// launch code above restated for a measurement label (bench.py's `tacotron.roofline.kernel`, `batch_sweep`); static strings
extern "C" const char* twv_tacotron_decoder_kernel_name(const twv_tacotron* h, int batch, int t_in)
{
    if (!h || batch < 1 || t_in < 1) return "";
    const twv_tacotron_dims& d = h->d;
    if (h->dec_groups == -1) return "tc_decoder_kernel";
    int cus = 0, devid = 0;
    if (hipGetDevice(&devid) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, devid) != hipSuccess) return "";
    const bool simple = d.num_speakers > 1 && d.model_simple && d.speaker_embedding_size != 1;
    XStageTab xt;
    taco_xstages(h, xt);
    const int upx = (batch + 7) / 8, nu = upx < batch ? upx : batch;
    const long long xfl = xdec_carve(d.num_mels, d.dec_prenet_sizes[1], 256, d.attention_state_size, d.dec_layer_num, d.dec_rnn_size, d.attention_size, t_in,
                                     nu < 1 ? 1 : nu, xt.nst).total + 64;
    const bool xok = taco_xdec_ok(h, xt) && cus >= 256 && upx <= kXU && t_in <= 512 && xfl * 4 <= 160 * 1024 && !simple;
    return xok && (h->dec_groups == 32 || h->dec_groups == 0) ? "tc_decoder_x_kernel" : "tc_decoder_g_kernel";
}
