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
#include "../../include/twv_amd.h"
#include "twv_dev.hpp"

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
__global__ void tc_maxpool2_kernel(const float* x, int rows, int T, int Cn, float* out)
{
    const long long total = (long long)rows * Cn;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / Cn);
        const float a = x[i];
        const float b = (row % T) + 1 < T ? x[i + Cn] : a;
        out[i] = a > b ? a : b;
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
__global__ void __launch_bounds__(512) tc_gru_seq_kernel(GruSeqArgs a)
{
    constexpr int U = 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const int len = a.lengths ? a.lengths[n] : a.T;
    // LDS: hs[128] | rh[128] | ug[128] | gp[4 nblk][4 chunk][64] | cp[2][4][64]
    const int o_h = 0, o_rh = 128, o_ug = 256, o_gp = 384, o_cp = 384 + 1024;
    Tile tg0, tg1, tc0;
    const int gnb = wave >> 1, gch = (wave & 1) * 2;       // gate tiles (gnb, gch), (gnb, gch+1)
    const int cnb = wave >> 2, cch = wave & 3;             // candidate tile (cnb, cch)
    load_tile(tg0, a.Wgh[dir] + ((long long)gnb * 4 + gch) * kTile, lane);
    load_tile(tg1, a.Wgh[dir] + ((long long)gnb * 4 + gch + 1) * kTile, lane);
    load_tile(tc0, a.Wch[dir] + ((long long)cnb * 4 + cch) * kTile, lane);
    const float bgv = tid < 2 * U ? a.bg[dir][tid] : 0.0f;
    const float bcv = tid < U ? a.bc[dir][tid] : 0.0f;
    if (tid < U) lds[o_h + tid] = a.init ? a.init[(long long)n * 2 * U + dir * U + tid] : 0.0f;
    __syncthreads();
    const float* Gx = a.Gx + dir * a.gx_dstride;
    const float* Cx = a.Cx + dir * a.cx_dstride;
    for (int s = 0; s < len; ++s) {
        const int t = dir == 0 ? s : len - 1 - s;
        const long long row = (long long)n * a.T + t;
        const float gx = tid < 2 * U ? Gx[row * 2 * U + tid] : 0.0f;
        const float cx = tid < U ? Cx[row * U + tid] : 0.0f;
        lds[o_gp + (gnb * 4 + gch) * 64 + lane] = dot_ldso(tg0, o_h + gch * 32);
        lds[o_gp + (gnb * 4 + gch + 1) * 64 + lane] = dot_ldso(tg1, o_h + (gch + 1) * 32);
        __syncthreads();
        if (tid < 2 * U) {
            const int nb = tid >> 6;
            float g = gx;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) g = g + lds[o_gp + (nb * 4 + ch) * 64 + lane];
            g = sigmoid_e(g + bgv);
            if (tid < U) lds[o_rh + tid] = g * lds[o_h + tid];      // r * h
            else lds[o_ug + tid - U] = g;                           // u
        }
        __syncthreads();
        lds[o_cp + (cnb * 4 + cch) * 64 + lane] = dot_ldso(tc0, o_rh + cch * 32);
        __syncthreads();
        if (tid < U) {
            const int nb = tid >> 6;
            float c = cx;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) c = c + lds[o_cp + (nb * 4 + ch) * 64 + lane];
            c = tanh_e(c + bcv);
            const float u = lds[o_ug + tid], h = lds[o_h + tid];
            const float t1 = u * h, t2 = 1.0f - u, t3 = t2 * c;
            const float hn = t1 + t3;
            a.out[row * 2 * U + dir * U + tid] = hn;
            lds[o_h + tid] = hn;
        }
        __syncthreads();
    }
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
    float* mel;            // [N][iters*R][M]
    float* align;          // [N][T][iters] or nullptr
    int32_t* status;
};
// LDS map of the decoder (float offsets)
struct DecLds { int cat, vec, h[5], frame, ctx, part, sc, p, cp, lg, al, q; };

// y_part[(nb*nchunk + ch)*64 + lane] = chunk(nb, ch) of W (tiles at wt) applied to the LDS vector at xo; tiles round-robin over
// the 8 waves, two named buffers so that the next tile travels while the current one is used
__device__ __forceinline__ void dec_gemv_partials(const float* wt, int K, int N, int xo, int o_part, int wave, int lane)
{
    const int nchunk = (K + 31) / 32, ntile = ((N + 63) / 64) * nchunk;
    Tile ta, tb;
    if (wave < ntile) load_tile(ta, wt + (long long)wave * kTile, lane);
    if (wave + 8 < ntile) load_tile(tb, wt + (long long)(wave + 8) * kTile, lane);
    for (int i = wave; i < ntile; i += 16) {
        {
            const float r = dot_ldso(ta, xo + (i % nchunk) * 32);
            __builtin_amdgcn_sched_barrier(0);
            lds[o_part + i * 64 + lane] = r;
            if (i + 16 < ntile) load_tile(ta, wt + (long long)(i + 16) * kTile, lane);
        }
        if (i + 8 < ntile) {
            const float r = dot_ldso(tb, xo + ((i + 8) % nchunk) * 32);
            __builtin_amdgcn_sched_barrier(0);
            lds[o_part + (i + 8) * 64 + lane] = r;
            if (i + 24 < ntile) load_tile(tb, wt + (long long)(i + 24) * kTile, lane);
        }
    }
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
__device__ __forceinline__ void dec_gru(const float* P, long long oWg, long long obg, long long oWc, long long obc, int nin, int U,
                                        int o_cat, int o_hout, int o_part, int o_vec, int tid, int wave, int lane)
{
    dec_gemv_partials(P + oWg, nin + U, 2 * U, o_cat, o_part, wave, lane);
    __syncthreads();
    float g = 0.0f;
    if (tid < 2 * U) g = sigmoid_e(dec_combine(o_part, nin + U, tid) + P[obg + tid]);
    __syncthreads();                                     // partials consumed
    if (tid < U) { lds[o_vec + tid] = lds[o_cat + nin + tid]; lds[o_cat + nin + tid] = g * lds[o_cat + nin + tid]; }   // keep h, cat <- [x, r*h]
    else if (tid < 2 * U) lds[o_vec + tid] = g;          // u
    __syncthreads();
    dec_gemv_partials(P + oWc, nin + U, U, o_cat, o_part, wave, lane);
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

    for (int it = 0; it < a.iters; ++it) {
        // ---- rnn_wrappers.py:425 decoder prenet: dense(M -> D0) relu, dense(D0 -> D1) relu
        dec_gemv_partials(P + a.w.dp1, M, D0, o_frame, o_part, wave, lane);
        __syncthreads();
        if (tid < D0) { const float v = dec_combine(o_part, M, tid) + P[a.w.dp1b + tid]; lds[o_vec + tid] = v > 0.0f ? v : 0.0f; }
        __syncthreads();
        dec_gemv_partials(P + a.w.dp2, D0, D1, o_vec, o_part, wave, lane);
        __syncthreads();
        // ---- rnn_wrappers.py:310-312 cell_inputs = [prenet_out | attention]; attention GRU on state ha
        if (tid < D1) { const float v = dec_combine(o_part, D0, tid) + P[a.w.dp2b + tid]; lds[o_cat + tid] = v > 0.0f ? v : 0.0f; }
        for (int i = tid; i < ENC; i += 512) lds[o_cat + D1 + i] = lds[o_ctx + i];
        for (int i = tid; i < AS; i += 512) lds[o_cat + D1 + ENC + i] = lds[o_ha + i];
        __syncthreads();
        dec_gru(P, a.w.aWg, a.w.abg, a.w.aWc, a.w.abc, D1 + ENC, AS, o_cat, o_ha, o_part, o_vec, tid, wave, lane);
        // ---- attention [RECALLED-TF BahdanauMonotonicAttention.__call__]: query layer
        dec_gemv_partials(P + a.w.Wq, AS, A, o_ha, o_part, wave, lane);
        __syncthreads();
        if (tid < A) lds[o_pq + tid] = dec_combine(o_part, AS, tid);
        __syncthreads();
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
        if (tid == 0) {   // exclusive cumsum of the logs (sequential, T terms)
            float run = 0.0f;
            for (int t = 0; t < T; ++t) { const float l = lds[o_q + t]; lds[o_q + t] = run; run = run + l; }
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
        if (tid == 0) {   // inclusive cumsum (sequential)
            float cs = 0.0f;
            for (int t = 0; t < T; ++t) { cs = cs + lds[o_q + t]; lds[o_q + t] = cs; }
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
        // ---- rnn_wrappers.py:463 concat(output, attention) -> OutputProjectionWrapper(dec_rnn)
        for (int i = tid; i < AS; i += 512) lds[o_cat + i] = lds[o_ha + i];
        for (int i = tid; i < ENC; i += 512) lds[o_cat + AS + i] = lds[o_ctx + i];
        __syncthreads();
        dec_gemv_partials(P + a.w.cW, AS + ENC, DR, o_cat, o_part, wave, lane);
        __syncthreads();
        if (tid < DR) lds[o_y + tid] = dec_combine(o_part, AS + ENC, tid) + P[a.w.cb + tid];
        __syncthreads();
        // ---- tacotron.py:167 ResidualWrapper(GRUCell(dec_rnn)): y <- y + GRU(y, h_l)
        for (int l = 0; l < a.layers; ++l) {
            for (int i = tid; i < DR; i += 512) { lds[o_cat + i] = lds[o_y + i]; lds[o_cat + DR + i] = lds[o_hr[l] + i]; }
            __syncthreads();
            dec_gru(P, a.w.rWg[l], a.w.rbg[l], a.w.rWc[l], a.w.rbc[l], DR, DR, o_cat, o_hr[l], o_part, o_vec, tid, wave, lane);
            if (tid < DR) lds[o_y + tid] = lds[o_y + tid] + lds[o_hr[l] + tid];
            __syncthreads();
        }
        // ---- tacotron.py:173 OutputProjectionWrapper(num_mels * r); helpers.py:40 last frame fed back
        dec_gemv_partials(P + a.w.oW, DR, M * R, o_y, o_part, wave, lane);
        __syncthreads();
        if (tid < M * R) {
            const float v = dec_combine(o_part, DR, tid) + P[a.w.ob + tid];
            a.mel[((long long)n * a.iters + it) * M * R + tid] = v;                 // tacotron.py:204 reshape
            if (tid >= M * (R - 1)) lds[o_frame + tid - M * (R - 1)] = v;
        }
        __syncthreads();
    }
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
    TMat gWgx[2], gWgh[2], gWcx[2], gWch[2]; TVec gbg[2], gbc[2];
};
struct twv_tacotron {
    twv_tacotron_dims d;
    long long blob_floats, packed_floats;
    TMat emb, semb;                 // raw tables (K rows x N)
    TMat dW[8]; TVec db[8]; int ndense, dn[8];
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
        for (int i = 0; i < depth; ++i) { c.hH[i] = mat(rnn, rnn); c.hHb[i] = vec(rnn); c.hT[i] = mat(rnn, rnn); c.hTb[i] = vec(rnn); }
        for (int dr = 0; dr < 2; ++dr) {
            gru_split(rnn, rnn, 2 * rnn, c.gWgx[dr], c.gWgh[dr]); c.gbg[dr] = vec(2 * rnn);
            gru_split(rnn, rnn, rnn, c.gWcx[dr], c.gWch[dr]); c.gbc[dr] = vec(rnn);
        }
    };
    const int E = d.embedding_size, SE = d.speaker_embedding_size, P0 = d.enc_prenet_sizes[0], P1 = d.enc_prenet_sizes[1],
              RN = d.enc_rnn_size, A = d.attention_size, AS = d.attention_state_size, DR = d.dec_rnn_size, M = d.num_mels,
              R = d.reduction_factor, ENC = 2 * RN;
    h->emb = raw(d.n_symbols, E);
    h->semb = raw(d.num_speakers, SE);
    h->ndense = 3 + d.dec_layer_num;
    const int dn[8] = {P1, 2 * RN, AS, DR, DR, DR, DR, DR};
    for (int i = 0; i < h->ndense; ++i) { h->dn[i] = dn[i]; h->dW[i] = mat(SE, dn[i]); h->db[i] = vec(dn[i]); }
    h->pW1 = mat(E, P0); h->pb1 = vec(P0); h->pW2 = mat(P0, P1); h->pb2 = vec(P1);
    cbhg(h->enc, P1, d.enc_bank_size, d.enc_bank_channel_size, d.enc_proj_sizes, d.enc_proj_width, d.enc_highway_depth, RN);
    h->Wm = mat(ENC, A); h->Wq = mat(AS, A);
    h->av = vec(A); h->ag = vec(1); h->ab = vec(A); h->asb = vec(1);
    h->dpW1 = mat(M, d.dec_prenet_sizes[0]); h->dpb1 = vec(d.dec_prenet_sizes[0]);
    h->dpW2 = mat(d.dec_prenet_sizes[0], d.dec_prenet_sizes[1]); h->dpb2 = vec(d.dec_prenet_sizes[1]);
    const int ain = d.dec_prenet_sizes[1] + ENC;
    h->aWgm = mat(ain + AS, 2 * AS); h->abg = vec(2 * AS); h->aWcm = mat(ain + AS, AS); h->abc = vec(AS);
    h->cW = mat(AS + ENC, DR); h->cb = vec(DR);
    for (int i = 0; i < d.dec_layer_num; ++i) { h->rWg[i] = mat(2 * DR, 2 * DR); h->rbg[i] = vec(2 * DR); h->rWc[i] = mat(2 * DR, DR); h->rbc[i] = vec(DR); }
    h->oW = mat(DR, M * R); h->ob = vec(M * R);
    cbhg(h->post, M, d.post_bank_size, d.post_bank_channel_size, d.post_proj_sizes, d.post_proj_width, d.post_highway_depth, d.post_rnn_size);
    h->lW = mat(2 * d.post_rnn_size, d.num_freq); h->lb = vec(d.num_freq);
    h->blob_floats = src;
    h->nv = TVec{dst, A}; dst += (A + 3) / 4 * 4;      // derived: normed_v
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
    if (d.num_speakers < 2 || d.speaker_embedding_size < 2) return twv_fail(TWV_E_UNSUPPORTED, "the deepvoice multi-speaker path needs num_speakers > 1");
    twv_tacotron* h = new twv_tacotron();
    h->d = d;
    taco_build(h);
    *out = h;
    return TWV_OK;
}
extern "C" void twv_tacotron_destroy(twv_tacotron* h) { delete h; }
extern "C" size_t twv_tacotron_blob_floats(const twv_tacotron* h) { return (size_t)h->blob_floats; }
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

extern "C" int twv_tacotron_pack(const twv_tacotron* h, const float* blob, void* packed, void* stream)
{
    if (!h || !blob || !packed) return twv_fail(TWV_E_INVALID, "null argument");
    hipStream_t st = (hipStream_t)stream;
    float* dst = (float*)packed;
    HIPCHK(hipMemsetAsync(dst, 0, (size_t)h->packed_floats * 4, st));
    for (const auto& it : h->items) {
        if (it.kind == 1) twv_launch_copy(dst + it.dst, blob + it.src, (long long)it.K * it.N, st);
        else {
            const int K = it.r1 - it.r0;
            PackTiles p{it.dst, 0, it.src + (long long)it.r0 * it.N, 0, 0, 1, (it.N + 63) / 64, (K + 31) / 32, K, it.N, it.N, 0, 64};
            twv_launch_pack_tiles(dst, blob, p, st);
        }
    }
    hipLaunchKernelGGL(tc_normed_v_kernel, dim3(1), dim3(64), 0, st, dst, h->av.off, h->ag.off, h->nv.off, h->d.attention_size);
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
    return f + 1024;
}
extern "C" size_t twv_tacotron_workspace_bytes(const twv_tacotron* h, int batch, int t_in) { return (size_t)taco_ws_floats(h, batch, t_in) * 4; }

static inline int tgrid(long long n) { long long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g)); }

static void launch_gemm(hipStream_t st, const float* P, const float* X, int ldx, int rows, int T, int Cin, int kw, const TMat& W,
                        const TVec* bias, int act, const TVec* inv, const TVec* shift, const float* add1, int ld1, const float* add2,
                        int ld2, float* Y, int ldy, int col0)
{
    GemmArgs a;
    a.X = X; a.ldx = ldx; a.rows = rows; a.T = T; a.Cin = Cin; a.kw = kw; a.pl = (kw - 1) / 2;
    a.Wt = P + W.off; a.K = W.K; a.N = W.N;
    a.bias = bias ? P + bias->off : nullptr; a.act = act;
    a.bn_inv = inv ? P + inv->off : nullptr; a.bn_shift = shift ? P + shift->off : nullptr;
    a.add1 = add1; a.ld1 = ld1; a.add2 = add2; a.ld2 = ld2; a.Y = Y; a.ldy = ldy; a.col0 = col0;
    hipLaunchKernelGGL(tc_gemm_kernel, dim3((rows + kGemmRows - 1) / kGemmRows), dim3(256), kGemmRows * kGemmKS * 4, st, a);
}

// modules.py:25-74 for `rows` = N*T rows
static void run_cbhg(hipStream_t st, const twv_tacotron* h, const float* P, const TCbhg& c, const float* in, int Cin, int N, int T,
                     int bank, int bch, const int32_t* proj, int pw, int depth, const float* before_hw, const float* init,
                     const int32_t* lengths, float* bankbuf, float* poolbuf, float* ra, float* rb, float* rc, float* gx, float* cx, float* out)
{
    const int rows = N * T, CB = bank * bch, rnn = 128;
    for (int k = 1; k <= bank; ++k)     // conv bank -> concatenated channels
        launch_gemm(st, P, in, Cin, rows, T, Cin, k, c.W[k], &c.b[k], TACT_RELU, &c.inv[k], &c.shift[k], nullptr, 0, nullptr, 0, bankbuf, CB, (k - 1) * bch);
    hipLaunchKernelGGL(tc_maxpool2_kernel, dim3(tgrid((long long)rows * CB)), dim3(256), 0, st, bankbuf, rows, T, CB, poolbuf);
    launch_gemm(st, P, poolbuf, CB, rows, T, CB, pw, c.pW[0], &c.pb[0], TACT_RELU, &c.pinv[0], &c.pshift[0], nullptr, 0, nullptr, 0, ra, proj[0], 0);
    // second projection + residual: (proj + inputs) + before_highway
    launch_gemm(st, P, ra, proj[0], rows, T, proj[0], pw, c.pW[1], &c.pb[1], TACT_NONE, &c.pinv[1], &c.pshift[1], in, Cin, before_hw, before_hw ? rnn : 0, rb, proj[1], 0);
    float* hw = rb;
    if (c.has_dense) { launch_gemm(st, P, rb, proj[1], rows, T, proj[1], 1, c.dW, &c.db, TACT_NONE, nullptr, nullptr, nullptr, 0, nullptr, 0, rc, rnn, 0); hw = rc; }
    float* hH = ra;
    float* hT = (hw == rc) ? rb : rc;
    for (int i = 0; i < depth; ++i) {
        launch_gemm(st, P, hw, rnn, rows, T, rnn, 1, c.hH[i], &c.hHb[i], TACT_RELU, nullptr, nullptr, nullptr, 0, nullptr, 0, hH, rnn, 0);
        launch_gemm(st, P, hw, rnn, rows, T, rnn, 1, c.hT[i], &c.hTb[i], TACT_SIGMOID, nullptr, nullptr, nullptr, 0, nullptr, 0, hT, rnn, 0);
        hipLaunchKernelGGL(tc_highway_kernel, dim3(tgrid((long long)rows * rnn)), dim3(256), 0, st, hH, hT, hw, (long long)rows * rnn);
    }
    // biGRU: hoisted x halves, then the recurrent kernel
    for (int dr = 0; dr < 2; ++dr) {
        launch_gemm(st, P, hw, rnn, rows, T, rnn, 1, c.gWgx[dr], nullptr, TACT_NONE, nullptr, nullptr, nullptr, 0, nullptr, 0, gx + (long long)dr * rows * 2 * rnn, 2 * rnn, 0);
        launch_gemm(st, P, hw, rnn, rows, T, rnn, 1, c.gWcx[dr], nullptr, TACT_NONE, nullptr, nullptr, nullptr, 0, nullptr, 0, cx + (long long)dr * rows * rnn, rnn, 0);
    }
    hipLaunchKernelGGL(tc_zero_kernel, dim3(tgrid((long long)rows * 2 * rnn)), dim3(256), 0, st, out, (long long)rows * 2 * rnn);
    GruSeqArgs g;
    g.Gx = gx; g.Cx = cx; g.gx_dstride = (long long)rows * 2 * rnn; g.cx_dstride = (long long)rows * rnn;
    for (int dr = 0; dr < 2; ++dr) { g.Wgh[dr] = P + c.gWgh[dr].off; g.Wch[dr] = P + c.gWch[dr].off; g.bg[dr] = P + c.gbg[dr].off; g.bc[dr] = P + c.gbc[dr].off; }
    g.init = init; g.lengths = lengths; g.T = T; g.out = out;
    hipLaunchKernelGGL(tc_gru_seq_kernel, dim3(N * 2), dim3(512), (384 + 1024 + 512) * 4, st, g);
    (void)h;
}

extern "C" int twv_tacotron_infer(const twv_tacotron* h, const void* packed, const int32_t* tokens, const int32_t* lengths,
                                  const int32_t* speaker_ids, int batch, int t_in, void* workspace, float* mel, float* linear,
                                  float* alignments, int32_t* status, void* stream)
{
    if (!h || !packed || !tokens || !lengths || !speaker_ids || !workspace || !mel || !status) return twv_fail(TWV_E_INVALID, "null argument");
    if (batch < 1 || t_in < 1 || t_in > 1024) return twv_fail(TWV_E_INVALID, "batch >= 1 and 1 <= t_in <= 1024 required");
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
    // ---- tacotron.py:51-60 embedding, :67-82 speaker embedding + deep_dense (softsign)
    hipLaunchKernelGGL(tc_embed_kernel, dim3(tgrid((long long)rows * E)), dim3(256), 0, st, P + h->emb.off, tokens, rows, E, ra);
    hipLaunchKernelGGL(tc_gather_rows_kernel, dim3(tgrid((long long)N * SE)), dim3(256), 0, st, P + h->semb.off, speaker_ids, N, SE, spk);
    // spk layout: [N][SE] at 0, then per dense i a [N][dn_i] block
    float* sv[8]; { float* q = spk + (long long)N * 64; for (int i = 0; i < h->ndense; ++i) { sv[i] = q; q += (long long)N * h->dn[i]; } }
    for (int i = 0; i < h->ndense; ++i)
        launch_gemm(st, P, spk, SE, N, 1, SE, 1, h->dW[i], &h->db[i], TACT_SOFTSIGN, nullptr, nullptr, nullptr, 0, nullptr, 0, sv[i], h->dn[i], 0);
    // decoder initial states gathered as [N][AS + layers*DR]
    float* dinit = spk + (long long)N * 2048;
    for (int i = 0; i < 1 + d.dec_layer_num; ++i) {
        const int wdt = i == 0 ? AS : DR;
        HIPCHK(hipMemcpy2DAsync(dinit + (i == 0 ? 0 : AS + (i - 1) * DR), (size_t)(AS + d.dec_layer_num * DR) * 4, sv[2 + i], (size_t)wdt * 4,
                                (size_t)wdt * 4, N, hipMemcpyDeviceToDevice, st));
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
    {
        const int Tp = (T + 3) / 4 * 4;
        const int ain = da.D1 + ENC;
        const int kmax = (ain + AS) > 2 * DR ? (ain + AS) : 2 * DR;
        const long long part = (long long)((kmax + 31) / 32) * ((2 * (AS > DR ? AS : DR) + 63) / 64) * 64;
        const long long part2 = (long long)((DR + 31) / 32) * ((M * R + 63) / 64) * 64;
        const long long fl = 2048 + AS + d.dec_layer_num * DR + (M + 31) / 32 * 32 + ENC + DR + Tp * 4 + A + Tp * 8 + (part > part2 ? part : part2);
        const size_t shm = (size_t)fl * 4;
        if (shm > 160 * 1024) return twv_fail(TWV_E_UNSUPPORTED, "decoder LDS footprint exceeds 160 KiB (t_in too large)");
        HIPCHK(hipFuncSetAttribute((const void*)tc_decoder_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        hipLaunchKernelGGL(tc_decoder_kernel, dim3(N), dim3(512), shm, st, da);
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
