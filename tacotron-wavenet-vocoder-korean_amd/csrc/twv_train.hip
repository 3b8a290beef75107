// twv_train.hip -- MI355X (gfx950) teacher-forced WaveNet training step + its C-ABI (include/twv_amd.h).
//
// Replaces, for hccho2/Tacotron-Wavenet-Vocoder-Korean (citations into /root/reference), one `sess.run([loss, optimize])`
// of train_vocoder.py:155-181 (scalar-input MoL model and the one-hot mu-law model):
//   wavenet/model.py:247-312  add_loss: drop last sample, create_upsample, full 'valid' convolution network
//                             (model.py:112-167 with train_mode=True), discretized_mix_logistic_loss (mixture.py:27-81), mean
//   wavenet/model.py:314-346  add_optimizer: exponential-decay LR, Adam (TF defaults), then EMA(0.9999).apply
// The reference runs on ONE device; data parallelism (one process per GPU, gradient all-reduce over RCCL between
// compute_gradients and apply_gradients) is added by the host (train.py) on the flat gradient buffer this file fills.
//
// Layout: every layer's activations are kept RIGHT-ALIGNED at the full length Tn = T-1 (row = (batch, t)), positions before a
// layer's receptive offset are masked.  A 'valid' dilated conv is then a contraction over all rows with a row-shifted operand
// and every weight gradient is ONE reduction over all rows.  The residual stack runs as three fused kernels per layer on the
// f32 matrix cores (tr_layer_fwd / tr_layer_bwd1 / tr_layer_bwd2_kernel: weights as MFMA B operands in registers or LDS,
// 32-row tiles per wave, no pre-activation or conditioning buffer in HBM; with the hparams upsampler the forward and the second
// backward kernel are tr_layer_fwdc / tr_layer_bwd2c_kernel: 1 KB coalesced accesses through LDS patches, no branch around a
// memory instruction); the stacked skip 1x1, conv1d_1/2 and their
// gradients are plain library GEMMs (rocBLAS).  MoL / softmax-CE loss (forward + analytic backward), deterministic
// reductions, the transposed-conv upsampler and the Adam/EMA update are hand-written HIP kernels.  Parameters and gradients
// stay in the canonical checkpoint layout (TF variable order, kernels (K, N) row-major).
// This is a floating-point training step: parity is by tolerance against an independent PyTorch-CPU fp32 autograd model.
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <stdint.h>
#include <string.h>
#include <string>
#include "../../include/twv_amd.h"
#include "twv_dev.hpp"

#define HIPCHK(expr)                                                                                              \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) return twv_fail(TWV_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)
#define BLASCHK(expr)                                                                                             \
    do {                                                                                                          \
        rocblas_status s_ = (expr);                                                                               \
        if (s_ != rocblas_status_success) return twv_fail(TWV_E_HIP, std::string(#expr) + ": rocblas status " + std::to_string((int)s_)); \
    } while (0)

struct TrainLayerOff { long long wf, bf, wg, bg, gcf, gcg, lcf, lcg, wd, bd, ws, bs; };
struct twv_wavenet_trainer {
    twv_wavenet_dims d;
    int B, T, Tn, rf, ow, NL, S, O, L, G, ifw, hop;
    int off[TWV_MAX_LAYERS + 1];          // receptive offset of layer l's INPUT (off[0] = ifw-1), off[NL] = rf-1
    long long c_causal, c_gcemb, c_layer0, c_lstride, c_w1, c_b1, c_w2, c_b2, c_up[4], nparams;
    TrainLayerOff lo;                     // offsets inside a layer block
    rocblas_handle blas;
    long long ws_floats;
    long long l2_off;                     // twv_wavenet_train_l2's scratch inside the workspace (floats)
    const void* ws_clean;                 // the workspace that has been cleared (see twv_wavenet_train_loss_grad)
};

// ---------------------------------------------------------------------------------------------------------------
//  kernels
// ---------------------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4t __attribute__((ext_vector_type(4)));
// 16-byte accesses through a buffer descriptor (out-of-range offset: the load returns zeros, the store is dropped)
__device__ __forceinline__ f32x4t tr_bld4(rsrc_t r, unsigned off)
{
    const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    return f32x4t{__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w)};
}
__device__ __forceinline__ void tr_bst4(rsrc_t r, unsigned off, f32x4t v)
{
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, r, (int)off, 0, 0);
}
#define GRID_STRIDE(i, n) for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)
static inline int tg(long long n) { long long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 32768 ? 32768 : g)); }

// model.py:102-111 one transposed-conv stage: out[b, t*f+a, m] = K[a,0]*in[b,t,m] + K[a,1]*in[b,t,m-1]
__global__ void tr_up_fwd_kernel(const float* K, const float* in, float* out, long long total, int f, int Lc)
{
    GRID_STRIDE(i, total) {
        const int m = (int)(i % Lc);
        const long long ta = i / Lc;
        const int a = (int)(ta % f);
        const long long bt = ta / f;
        const float x0 = in[bt * Lc + m], x1 = m > 0 ? in[bt * Lc + m - 1] : 0.0f;
        out[i] = K[a * 2] * x0 + K[a * 2 + 1] * x1;
    }
}
// the three stages composed: U[frame*hop + phase, m] = sum_j ctab[phase][j] * mel[frame, m - j] (zero for m - j < 0), phase = (a0*f1 + a1)*f2 + a2
__global__ void tr_ctab_kernel(const float* K0, const float* K1, const float* K2, int f0, int f1, int f2, float* ctab)
{
    GRID_STRIDE(i, (long long)f0 * f1 * f2) {
        const int a2 = (int)(i % f2), a1 = (int)((i / f2) % f1), a0 = (int)(i / ((long long)f1 * f2));
        const float p0 = K0[a0 * 2], p1 = K0[a0 * 2 + 1], q0 = K1[a1 * 2], q1 = K1[a1 * 2 + 1], r0 = K2[a2 * 2], r1 = K2[a2 * 2 + 1];
        const float s0 = p0 * q0, s1 = p0 * q1 + p1 * q0, s2 = p1 * q1;       // stage 0 then stage 1
        ctab[i * 4 + 0] = s0 * r0;
        ctab[i * 4 + 1] = s0 * r1 + s1 * r0;
        ctab[i * 4 + 2] = s1 * r1 + s2 * r0;
        ctab[i * 4 + 3] = s2 * r1;
    }
}
// ms[(bf*4 + j)*Lc + m] = mel[bf*Lc + m - j] (0 for m < j): the four bin-shifted copies of every mel frame
__global__ void tr_mel_shift_kernel(const float* mel, float* ms, long long frames, int Lc)
{
    GRID_STRIDE(i, frames * 4 * Lc) {
        const int m = (int)(i % Lc), j = (int)((i / Lc) & 3);
        const long long bf = i / (4LL * Lc);
        ms[i] = m >= j ? mel[bf * Lc + m - j] : 0.0f;
    }
}
// dctab[phase][j] = sum over batch entries and frames of D[(b*T + frame*hop + phase)*4 + j]: one block per phase, thread i takes the
// (b, frame) pairs i, i+256, ... in order, then a fixed-order tree over the 256 partial sums (deterministic)
__global__ __launch_bounds__(256) void tr_dctab_kernel(const float* D, int B, int T, int hop, float* dctab)
{
    const int ph = blockIdx.x, F = T / hop, n = B * F;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = threadIdx.x; i < n; i += 256) {
        const int b = i / F, f = i - b * F;
        const float4 v = *reinterpret_cast<const float4*>(D + ((long long)b * T + (long long)f * hop + ph) * 4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    __shared__ float4 sh[256];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) {
            const float4 o = sh[threadIdx.x + w];
            float4 m = sh[threadIdx.x];
            m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w;
            sh[threadIdx.x] = m;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { dctab[ph * 4 + 0] = sh[0].x; dctab[ph * 4 + 1] = sh[0].y; dctab[ph * 4 + 2] = sh[0].z; dctab[ph * 4 + 3] = sh[0].w; }
}
// gradients of the three upsampling kernels from dctab: ctab[phase] = K0[a0] * K1[a1] * K2[a2] (2-tap polynomials in the bin shift), so
// dK_s[a][k] = sum over the other two stages' phases of sum_m dctab[phase][m + k] * (product of the other two binomials)[m]
__global__ void tr_up_grad_kernel(const float* K0, const float* K1, const float* K2, int f0, int f1, int f2, const float* dctab,
                                  float* g0, float* g1, float* g2)
{
    const int n0 = 2 * f0, n1 = 2 * f1, n2 = 2 * f2;
    GRID_STRIDE(i, (long long)n0 + n1 + n2) {
        int st, a, k;
        if (i < n0) { st = 0; a = (int)i >> 1; k = (int)i & 1; }
        else if (i < n0 + n1) { st = 1; a = (int)(i - n0) >> 1; k = (int)(i - n0) & 1; }
        else { st = 2; a = (int)(i - n0 - n1) >> 1; k = (int)(i - n0 - n1) & 1; }
        const float* KA = st == 0 ? K1 : K0; const int fa = st == 0 ? f1 : f0;       // the two other stages, in stage order
        const float* KB = st == 2 ? K1 : K2; const int fb = st == 2 ? f1 : f2;
        float s = 0.0f;
        for (int x = 0; x < fa; ++x)
            for (int y = 0; y < fb; ++y) {
                const int a0 = st == 0 ? a : x, a1 = st == 0 ? x : (st == 1 ? a : y), a2 = st == 2 ? a : y;
                const int ph = (a0 * f1 + a1) * f2 + a2;
                const float u0 = KA[x * 2], u1 = KA[x * 2 + 1], v0 = KB[y * 2], v1 = KB[y * 2 + 1];
                const float r0 = u0 * v0, r1 = u0 * v1 + u1 * v0, r2 = u1 * v1;
                const float* dc = dctab + ph * 4 + k;
                s += (dc[0] * r0 + dc[1] * r1) + dc[2] * r2;
            }
        (st == 0 ? g0 : (st == 1 ? g1 : g2))[a * 2 + k] = s;
    }
}
// backward of the stage: din[b,t,m] = sum_a K[a,0] dout[tf+a,m] + K[a,1] dout[tf+a,m+1]; dK by atomics of per-thread partials
__global__ void tr_up_bwd_in_kernel(const float* K, const float* dout, float* din, long long total_in, int f, int Lc)
{
    GRID_STRIDE(i, total_in) {
        const int m = (int)(i % Lc);
        const long long bt = i / Lc;
        float acc = 0.0f;
        for (int a = 0; a < f; ++a) {
            const long long o = (bt * f + a) * Lc + m;
            acc += K[a * 2] * dout[o];
            if (m + 1 < Lc) acc += K[a * 2 + 1] * dout[o + 1];
        }
        din[i] = acc;
    }
}
__global__ __launch_bounds__(256) void tr_up_bwd_k_kernel(const float* in, const float* dout, float* part, long long rows_in, int f, int Lc, int nchunk)
{
    // block (chunk, tap a): partial sums over its slab of input rows -> part[chunk][a*2 + j]
    const int a = blockIdx.y;
    const long long per = (rows_in + nchunk - 1) / nchunk;
    const long long r0 = (long long)blockIdx.x * per, r1 = r0 + per < rows_in ? r0 + per : rows_in;
    float s0 = 0.0f, s1 = 0.0f;
    for (long long i = r0 * Lc + threadIdx.x; i < r1 * Lc; i += 256) {
        const int m = (int)(i % Lc);
        const long long bt = i / Lc;
        const float d = dout[(bt * f + a) * Lc + m];
        s0 += d * in[i];
        if (m > 0) s1 += d * in[i - 1];
    }
    __shared__ float r0s[256], r1s[256];
    r0s[threadIdx.x] = s0; r1s[threadIdx.x] = s1;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) { r0s[threadIdx.x] += r0s[threadIdx.x + st]; r1s[threadIdx.x] += r1s[threadIdx.x + st]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[(long long)blockIdx.x * 2 * f + a * 2] = r0s[0]; part[(long long)blockIdx.x * 2 * f + a * 2 + 1] = r1s[0]; }
}

// causal layer input unfolded: xunf[(b,t)][k] = in[b, t-(ifw-1)+k]  (0 before the start)   model.py:41-46
__global__ void tr_unfold_kernel(const float* audio, float* xunf, int B, int T, int Tn, int ifw)
{
    const long long total = (long long)B * Tn * ifw;
    GRID_STRIDE(i, total) {
        const int k = (int)(i % ifw);
        const long long r = i / ifw;
        const int t = (int)(r % Tn), b = (int)(r / Tn);
        const int ts = t - (ifw - 1) + k;
        xunf[i] = ts >= 0 ? audio[(long long)b * T + ts] : 0.0f;
    }
}
// y[r][c] = relu(x[r][c] + bias_sum[c])  (bias_sum = one or several bias vectors added together)
__global__ void tr_bias_relu_kernel(float* x, const float* biases, int nb, long long strideb, const float* bias, int C, long long n)
{
    GRID_STRIDE(i, n) {
        const int c = (int)(i % C);
        float v = x[i];
        if (bias) v += bias[c];
        for (int k = 0; k < nb; ++k) v += biases[(long long)k * strideb + c];
        x[i] = v > 0.0f ? v : 0.0f;
    }
}
__global__ void tr_bias_add_kernel(float* x, const float* bias, int C, long long n) { GRID_STRIDE(i, n) x[i] += bias[i % C]; }
__global__ void tr_relu_bwd_kernel(float* dx, const float* y, long long n) { GRID_STRIDE(i, n) if (!(y[i] > 0.0f)) dx[i] = 0.0f; }
// float4 forms of the two relu passes over the (rows, 512) post-processing activations (C % 4 == 0, n % 4 == 0): HBM-bound
__global__ void tr_bias_relu4_kernel(float4* x, const float4* bias, int C4, long long n4)
{
    GRID_STRIDE(i, n4) {
        float4 v = x[i];
        if (bias) { const float4 b = bias[i % C4]; v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
        v.x = v.x > 0.0f ? v.x : 0.0f; v.y = v.y > 0.0f ? v.y : 0.0f; v.z = v.z > 0.0f ? v.z : 0.0f; v.w = v.w > 0.0f ? v.w : 0.0f;
        x[i] = v;
    }
}
__global__ void tr_relu_bwd4_kernel(float4* dx, const float4* y, long long n4)
{
    GRID_STRIDE(i, n4) {
        const float4 yy = y[i];
        if (!(yy.x > 0.0f) || !(yy.y > 0.0f) || !(yy.z > 0.0f) || !(yy.w > 0.0f)) {
            float4 d = dx[i];
            if (!(yy.x > 0.0f)) d.x = 0.0f;
            if (!(yy.y > 0.0f)) d.y = 0.0f;
            if (!(yy.z > 0.0f)) d.z = 0.0f;
            if (!(yy.w > 0.0f)) d.w = 0.0f;
            dx[i] = d;
        }
    }
}
// out[c] = sum over the nb layers of biases[k * strideb + c], k ascending (the skip biases all land on the same sum, model.py:154)
__global__ void tr_bias_sum_kernel(const float* biases, int nb, long long strideb, float* out, int C)
{
    GRID_STRIDE(c, C) {
        float s = 0.0f;
        for (int k = 0; k < nb; ++k) s += biases[(long long)k * strideb + c];
        out[c] = s;
    }
}
// dst_k[c] = src[c] for k = 1..nb-1 (the skip-bias gradient is the same vector for every layer)
__global__ void tr_bcast_rows_kernel(float* base, long long strideb, int nb, int C)
{
    GRID_STRIDE(i, (long long)(nb - 1) * C) {
        const int k = 1 + (int)(i / C), c = (int)(i % C);
        base[(long long)k * strideb + c] = base[c];
    }
}
__global__ void tr_fill_kernel(float* p, float v, long long n) { GRID_STRIDE(i, n) p[i] = v; }
// column sums, two deterministic stages.  stage 1: block (chunk, col tile, segment) sums its row chunk of segment z
// (segments = consecutive `rows`-row slabs, e.g. one per batch entry) into part[(z*nchunk + chunk)*C + c].
__global__ __launch_bounds__(256) void tr_colsum_partial_kernel(const float* x, long long rows, int C, int ldx, int nchunk, float* part)
{
    const int c = blockIdx.y * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    const long long per = (rows + nchunk - 1) / nchunk;
    const long long r0 = (long long)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
    const float* xs = x + (long long)blockIdx.z * rows * ldx;
    float s0 = 0.0f, s1 = 0.0f;
    if (c < C) {
        long long r = r0 + grp;
        for (; r + 4 < r1; r += 8) { s0 += xs[r * ldx + c]; s1 += xs[(r + 4) * ldx + c]; }
        if (r < r1) s0 += xs[r * ldx + c];
    }
    __shared__ float sh[256];
    sh[threadIdx.x] = s0 + s1;
    __syncthreads();
    if (threadIdx.x < 64 && c < C)
        part[((long long)blockIdx.z * nchunk + blockIdx.x) * C + c] = (sh[threadIdx.x] + sh[threadIdx.x + 64]) + (sh[threadIdx.x + 128] + sh[threadIdx.x + 192]);
}
// relu backward fused with stage 1 of the bias gradient: dx[r][c] = y[r][c] > 0 ? dx[r][c] : 0 is written back and summed per column
// in the same pass (same blocking and partial layout as tr_colsum_partial_kernel, one segment); saves one read of the (rows, C) array
__global__ __launch_bounds__(256) void tr_relu_bwd_colsum_kernel(float* dx, const float* y, long long rows, int C, int nchunk, float* part)
{
    const int c = blockIdx.y * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    const long long per = (rows + nchunk - 1) / nchunk;
    const long long r0 = (long long)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
    float s0 = 0.0f, s1 = 0.0f;
    if (c < C) {
        long long r = r0 + grp;
        for (; r + 4 < r1; r += 8) {
            const long long i0 = r * C + c, i1 = (r + 4) * C + c;
            float d0 = dx[i0], d1 = dx[i1];
            const float y0 = y[i0], y1 = y[i1];
            if (!(y0 > 0.0f)) { d0 = 0.0f; dx[i0] = 0.0f; }
            if (!(y1 > 0.0f)) { d1 = 0.0f; dx[i1] = 0.0f; }
            s0 += d0; s1 += d1;
        }
        if (r < r1) {
            const long long i0 = r * C + c;
            float d0 = dx[i0];
            if (!(y[i0] > 0.0f)) { d0 = 0.0f; dx[i0] = 0.0f; }
            s0 += d0;
        }
    }
    __shared__ float sh[256];
    sh[threadIdx.x] = s0 + s1;
    __syncthreads();
    if (threadIdx.x < 64 && c < C)
        part[(long long)blockIdx.x * C + c] = (sh[threadIdx.x] + sh[threadIdx.x + 64]) + (sh[threadIdx.x + 128] + sh[threadIdx.x + 192]);
}
// stage 2: out[z*ldo + c] = sum over chunks (fixed order: 4 interleaved partial sums, then pairwise); block = 64 outputs x 4 groups
__global__ __launch_bounds__(256) void tr_colsum_final_kernel(const float* part, int nchunk, int C, int nseg, float* out, int ldo)
{
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    const int z = i / C, c = i % C;
    float s = 0.0f;
    if (i < nseg * C)
        for (int k = grp; k < nchunk; k += 4) s += part[((long long)z * nchunk + k) * C + c];
    __shared__ float sh[256];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < 64 && i < nseg * C) out[(long long)z * ldo + c] = (sh[threadIdx.x] + sh[threadIdx.x + 64]) + (sh[threadIdx.x + 128] + sh[threadIdx.x + 192]);
}

// Tall-skinny weight-gradient contraction on the f32 matrix cores:  C[m][n] = sum_k A[k][m] * B[k][n],  K ~ 5e5, M,N <= 96.
// v_mfma_f32_32x32x2_f32: lane l feeds A[k = l>>5][m = l&31] and B[k = l>>5][n = l&31] -- for row-major (k, channel)
// activations that is two coalesced 128-byte rows per operand, no transposition or LDS staging needed.
// Block = 4 waves; wave w owns k rows {2*(4*i + w), +1} of its chunk; per-wave partial tiles go to part[...] and are summed
// in a fixed order by tr_tn_reduce_kernel (deterministic, no atomics).  grid = (chunks, m-blocks, n-blocks).
__global__ __launch_bounds__(256) void tr_tn_partial_kernel(const float* A, int lda, const float* B, int ldb, long long K, int M, int N,
                                                            long long rows_per_chunk, float* part)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = blockIdx.y * 32 + (lane & 31), n = blockIdx.z * 32 + (lane & 31);
    const bool mok = m < M, nok = n < N;
    const long long k0 = (long long)blockIdx.x * rows_per_chunk;
    const long long k1 = k0 + rows_per_chunk < K ? k0 + rows_per_chunk : K;
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float* ap = A + m;
    const float* bp = B + n;
#pragma unroll 8
    for (long long k = k0 + 2 * wave + (lane >> 5); k < k1 + (lane >> 5); k += 8) {
        const bool kok = k < k1;
        const float a = (mok && kok) ? ap[k * lda] : 0.0f;
        const float b = (nok && kok) ? bp[k * ldb] : 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    // C/D map: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).  The four waves' tiles are summed through LDS (fixed order).
    __shared__ float red[4][32][33];
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][lane & 31] = acc[r];
    __syncthreads();
    const int Mp = gridDim.y * 32, Np = gridDim.z * 32;
    float* out = part + ((long long)blockIdx.x * Mp + blockIdx.y * 32) * Np + blockIdx.z * 32;
    for (int i = threadIdx.x; i < 1024; i += 256) {
        const int row = i >> 5, col = i & 31;
        out[(long long)row * Np + col] = (red[0][row][col] + red[1][row][col]) + (red[2][row][col] + red[3][row][col]);
    }
}
// C[m][n] = sum over slabs (fixed order: blockDim.x / 64 interleaved partial sums, then a pairwise tree); block = 64 outputs x 4 or 16 slab
// groups (16: the hundreds of 64 KB-strided partial tiles of a single-tile product were 46 us of dependent loads at 4)
__global__ void tr_tn_reduce_kernel(const float* part, int nslab, int Mp, int Np, int M, int N, float* C, int ldc)
{
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6, ng = blockDim.x >> 6;
    float s = 0.0f;
    const int m = i / N, n = i % N;
    if (i < M * N)
        for (int k = grp; k < nslab; k += ng) s += part[((long long)k * Mp + m) * Np + n];
    __shared__ float sh[1024];
    sh[threadIdx.x] = s;
    for (int w = 1; w < ng; w <<= 1) {                  // ng = 4: (s0 + s1) + (s2 + s3)
        __syncthreads();
        if ((grp & (2 * w - 1)) == 0 && grp + w < ng) sh[threadIdx.x] += sh[threadIdx.x + 64 * w];
    }
    if (threadIdx.x < 64 && i < M * N) C[(long long)m * ldc + n] = sh[threadIdx.x];
}
// Skinny output projection on the f32 matrix cores: Y[r][n] = sum_k X[r][k] * W[k][n] (+ bias[n]), N <= 32, K % 64 == 0 (model.py:161-165
// conv1d_2, 512 -> 30).  rocBLAS spends 0.6 ms on this 9-GFLOP product; it is a single pass over X.  Wave = 32 rows: every lane
// reads 64 contiguous bytes of its row per 32-wide k step (lanes l and l+32 the two halves of one 128-byte line) and feeds sixteen
// v_mfma_f32_32x32x2_f32 with the k pairs (t, 16 + t); W sits in LDS as [k][32] (zero-padded columns).  The next step's four float4
// are requested before this step's MFMAs, through a buffer descriptor (rows past the end: out-of-range offset, zeros) -- with the loads
// behind a branch and no prefetch every step waited out a full memory latency (242 us for 616 MB).
// XRELU: X holds pre-activations, the operand is relu(X[r][k] + xb[k]) (xb nullable): the activation pass over X is never run.
template <bool XRELU>
__global__ __launch_bounds__(256) void tr_skinny_nn_kernel(const float* X, int ldx, const float* W, int ldw, const float* bias, long long rows, int K, int N,
                                                           float* Y, int ldy, const float* xb)
{
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [K][32], XRELU: + [K] (the bias, read through LDS: a global load in the
    for (int i = threadIdx.x; i < K * 32; i += 256) { const int k = i >> 5, n = i & 31; wl[i] = n < N ? W[(long long)k * ldw + n] : 0.0f; }
    if (XRELU)                                          // loop would stand between the prefetch and its s_waitcnt vmcnt)
        for (int i = threadIdx.x; i < K; i += 256) wl[K * 32 + i] = xb ? xb[i] : 0.0f;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5, col = lane & 31;
    const long long ntile = (rows + 31) / 32;
    const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, (int)(((rows - 1) * ldx + K) * 4), 0x00020000);
    const int nstep = K >> 5;
    for (long long tile = (long long)blockIdx.x * 4 + wave; tile < ntile; tile += (long long)gridDim.x * 4) {
        const long long r = tile * 32 + col;            // A operand row of this lane
        const unsigned xo = r < rows ? (unsigned)((r * ldx + half * 16) * 4) : 0x80000000u;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
        auto fetch = [&](f32x4t (&d)[4], int st) {       // step st's 64 bytes of the lane's row (past the last step: zeros, never used)
            const unsigned o = st < nstep ? xo + (unsigned)st * 128u : 0x80000000u;
#pragma unroll
            for (int q = 0; q < 4; ++q) d[q] = tr_bld4(rx, o + q * 16);
        };
        auto step = [&](const f32x4t (&d)[4], int st) {
            const int k0 = st << 5;
            float av[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) av[t] = d[t >> 2][t & 3];
            if (XRELU) {                                 // (rows past the end: their outputs are not stored)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4t bq = *reinterpret_cast<const f32x4t*>(wl + K * 32 + k0 + half * 16 + 4 * q);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const float v = av[4 * q + j] + bq[j]; av[4 * q + j] = v > 0.0f ? v : 0.0f; }
                }
            }
            float bw[16];                                // the step's B operands in one batch of LDS reads (one round trip, not eight)
#pragma unroll
            for (int t = 0; t < 16; ++t) bw[t] = wl[(k0 + half * 16 + t) * 32 + col];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 16; ++t)                 // lanes < 32 carry k = k0 + t, lanes >= 32 carry k = k0 + 16 + t
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bw[t], acc, 0, 0, 0);
        };
        // two register sets in turn (K % 64 == 0): a copy `a = next` at the end of a trip is a wait for the prefetch it was meant to hide
        f32x4t a0[4], a1[4];
        fetch(a0, 0);
        for (int st = 0; st < nstep; st += 2) {
            fetch(a1, st + 1);
            __builtin_amdgcn_sched_barrier(0);
            step(a0, st);
            __builtin_amdgcn_sched_barrier(0);
            fetch(a0, st + 2);
            __builtin_amdgcn_sched_barrier(0);
            step(a1, st + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        const float bv = (bias != nullptr && col < N) ? bias[col] : 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const long long rr = tile * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
            if (rr < rows && col < N) Y[rr * ldy + col] = acc[i] + bv;
        }
    }
}
// training-time weight views: per layer [tap0 (32x64) = wf[0]|wg[0]] [tap1 (32x64)] [lc (L x 64) = lcf|lcg] [gc (G x 64)] and the
// stacked skip kernel (NL*32 x S).  dir 0: canonical -> views, dir 1: views (gradients) -> canonical.
__global__ void tr_views_kernel(float* canon, float* views, float* wsall, int NL, long long c_layer0, long long lstride, long long o_wf, long long o_wg,
                                long long o_lcf, long long o_lcg, long long o_gcf, long long o_gcg, long long o_ws, int L, int G, int S, int dir)
{
    const long long vstride = 64LL * (64 + L + G);
    const long long per = vstride + 32LL * S;
    GRID_STRIDE(i, per * NL) {
        const int l = (int)(i / per);
        long long q = i % per;
        float* cl = canon + c_layer0 + l * lstride;
        float* cp; float* vp;
        if (q < vstride) {
            vp = views + l * vstride + q;
            const int col = (int)(q & 63), half = col >> 5, j = col & 31;
            const long long row = q >> 6;
            if (row < 64) cp = cl + (half ? o_wg : o_wf) + row * 32 + j;                       // (2,32,32): tap*1024 + in*32 + out
            else if (row < 64 + L) cp = cl + (half ? o_lcg : o_lcf) + (row - 64) * 32 + j;
            else cp = cl + (half ? o_gcg : o_gcf) + (row - 64 - L) * 32 + j;
        } else {
            q -= vstride;
            vp = wsall + (long long)l * 32 * S + q;
            cp = cl + o_ws + q;
        }
        if (dir == 0) *vp = *cp; else *cp = *vp;
    }
}
__global__ void tr_gather_emb_kernel(const float* table, const int32_t* ids, float* out, int B, int G)
{
    GRID_STRIDE(i, (long long)B * G) out[i] = table[(long long)ids[i / G] * G + (i % G)];
}
// one thread per table element, batch rows added in order: reproducible (no float atomics)
__global__ void tr_scatter_emb_kernel(const float* demb, const int32_t* ids, float* dtable, int B, int G, int card)
{
    GRID_STRIDE(i, (long long)card * G) {
        const int row = (int)(i / G), g = (int)(i % G);
        float s = 0.0f;
        for (int b = 0; b < B; ++b) if (ids[b] == row) s += demb[(long long)b * G + g];
        dtable[i] += s;
    }
}

// mixture.py:27-81 discretized_mix_logistic_loss(num_class=2**16, reduce=False) + model.py:290 mean, and its gradient.
// one thread per (b, p); y (rows, 3*nr), target = audio[b, p + rf]
__device__ __forceinline__ float softplus_f(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigm_f(float x) { return 1.0f / (1.0f + expf(-x)); }
// one row of the discretized mix-logistic loss and its analytic gradient (mixture.py:27-81): yr = the row's 3 nr outputs, dr = its gradient
// (may alias yr: entry i, nr + i, 2 nr + i is read before it is written); returns the row's loss term.
// NR > 0: the number of mixtures as a compile-time constant (hparams: out_channels 30 -> 10): the five per-mixture arrays stay in registers
// (with a run-time count they lived in 400 bytes of scratch per thread: 0.36 GB read + 0.30 GB written per launch for 72 MB of I/O)
template <int NR>
__device__ __forceinline__ float tr_mol_row(const float* yr, float* dr, float tgt, int nr_, float inv_count)
{
    const int nr = NR > 0 ? NR : nr_;
    constexpr int CAP = NR > 0 ? NR : 32;
    const float lsmin = -32.23619130191664f, h = 1.0f / 65535.0f, logc = logf(65535.0f / 2.0f);
    float lm = -3.0e38f;
#pragma unroll
    for (int i = 0; i < nr; ++i) lm = fmaxf(lm, yr[i]);
    float se = 0.0f;
#pragma unroll
    for (int i = 0; i < nr; ++i) se += expf(yr[i] - lm);
    const float lse_logit = lm + logf(se);
    float a[CAP], dplus[CAP], dmin[CAP], dmid[CAP], dsdirect[CAP];
    float amax = -3.0e38f;
#pragma unroll
    for (int i = 0; i < nr; ++i) {
        const float mu = yr[nr + i], sraw = yr[2 * nr + i];
        const float s = fmaxf(sraw, lsmin);
        const float cen = tgt - mu, inv = expf(-s);
        const float plus = inv * (cen + h), mn = inv * (cen - h), mid = inv * cen;
        const float cp = sigm_f(plus), cm = sigm_f(mn), delta = cp - cm;
        float lp;
        dplus[i] = dmin[i] = dmid[i] = dsdirect[i] = 0.0f;
        if (tgt < -0.999f) { lp = plus - softplus_f(plus); dplus[i] = 1.0f - cp; }
        else if (tgt > 0.999f) { lp = -softplus_f(mn); dmin[i] = -cm; }
        else if (delta > 1e-5f) { lp = logf(fmaxf(delta, 1e-12f)); dplus[i] = cp * (1.0f - cp) / delta; dmin[i] = -cm * (1.0f - cm) / delta; }
        else { lp = mid - s - 2.0f * softplus_f(mid) - logc; dmid[i] = 1.0f - 2.0f * sigm_f(mid); dsdirect[i] = -1.0f; }
        a[i] = lp + (yr[i] - lse_logit);
        amax = fmaxf(amax, a[i]);
    }
    float sa = 0.0f;
#pragma unroll
    for (int i = 0; i < nr; ++i) sa += expf(a[i] - amax);
    const float lse = amax + logf(sa);
#pragma unroll
    for (int i = 0; i < nr; ++i) {
        const float w = expf(a[i] - lse);                      // softmax(a)
        const float sm = expf(yr[i] - lse_logit);              // softmax(logits)
        const float mu = yr[nr + i], sraw = yr[2 * nr + i];
        const float s = fmaxf(sraw, lsmin);
        const float cen = tgt - mu, inv = expf(-s);
        const float plus = inv * (cen + h), mn = inv * (cen - h), mid = inv * cen;
        const float dlp = -w;                                   // dL/dlp_i
        const float gmu = dlp * (-(inv)) * (dplus[i] + dmin[i] + dmid[i]);
        const float gs = dlp * (-(dplus[i] * plus + dmin[i] * mn + dmid[i] * mid) + dsdirect[i]);
        dr[i] = (sm - w) * inv_count;
        dr[nr + i] = gmu * inv_count;
        dr[2 * nr + i] = (sraw > lsmin ? gs : 0.0f) * inv_count;
    }
    return -lse * inv_count;
}
// NR > 0: a workgroup's 256 rows (contiguous: 256 x 3 NR floats) travel through LDS in whole lines -- a thread reading its own 120-byte row
// touched 64 cache lines per load instruction
template <int NR>
__global__ __launch_bounds__(256) void tr_mol_loss_kernel(const float* y, const float* audio, int B, int T, int ow, int rf, int nr, float inv_count,
                                                          float* row_loss, float* dy)
{
    const long long rows = (long long)B * ow;
    if constexpr (NR > 0) {
        __shared__ float sy[256 * 3 * NR];
        for (long long base = (long long)blockIdx.x * 256; base < rows; base += (long long)gridDim.x * 256) {
            const int nrow = rows - base < 256 ? (int)(rows - base) : 256, ne = nrow * 3 * NR;
            __syncthreads();
            for (int e = threadIdx.x; e < ne; e += 256) sy[e] = y[base * 3 * NR + e];
            __syncthreads();
            if ((int)threadIdx.x < nrow) {
                const long long r = base + threadIdx.x;
                const int p = (int)(r % ow), b = (int)(r / ow);
                float* row = sy + threadIdx.x * 3 * NR;
                row_loss[r] = tr_mol_row<NR>(row, row, audio[(long long)b * T + p + rf], NR, inv_count);
            }
            __syncthreads();
            for (int e = threadIdx.x; e < ne; e += 256) dy[base * 3 * NR + e] = sy[e];
        }
    } else {
        GRID_STRIDE(r, rows) {
            const int p = (int)(r % ow), b = (int)(r / ow);
            row_loss[r] = tr_mol_row<0>(y + r * 3 * nr, dy + r * 3 * nr, audio[(long long)b * T + p + rf], nr, inv_count);
        }
    }
}



// ===================================================================================================================
//  Fused residual layer, forward (model.py:66-101 train mode) on the f32 matrix cores.
//  One wave owns 32-row tiles (rows = consecutive t of one batch entry).  Per tile:
//      pre[32 x 64] = X[t-d] W0 + X[t] W1 + U[t-o] Wlc            144-deep contraction, 2 x 72 v_mfma_f32_32x32x2_f32
//      th, sg = tanh / sigmoid(pre + bias + gc[b])  (masked below the layer's receptive offset o), z = th*sg
//      x_next = X[t] + z Wd + bd                                   16 MFMAs; z goes C-layout -> A-layout through 4.6 KB of LDS
//  The layer's weights live in registers for the whole launch as MFMA B operands (lane (n, hh): W[k = 8i + 4hh + j][n]), A
//  operands are float4 row loads (lane (row, hh): k = 8i + 4hh .. +3), the next tile's rows are requested before the
//  current tile's MFMAs.  Writes TH, SG (for the backward pass), x_next and the skip input slice ZC -- no pre-activation
//  or conditioning buffer ever reaches HBM.  Any accumulation order is fine here: training parity is by tolerance.
// ===================================================================================================================
// training-time activations on the transcendental unit (v_exp_f32 / v_rcp_f32, ~1e-6 relative): parity here is by tolerance
// (v_rcp_f32 by name: `__frcp_rn` is the correctly rounded reciprocal -- v_div_scale x 2, v_rcp, four fmas, v_div_fmas, v_div_fixup: ten
// instructions where one was meant, 320 of the forward kernel's ~750 VALU instructions per tile)
__device__ __forceinline__ float tr_sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tr_tanh_fast(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f; }
struct LayerFwdArgs {
    const float* X; const float* U; const float* gcp;        // (B*Tn,32) (B*T,L=80) (B,64)
    const float* W0; const float* W1; const float* Wlc;      // views (32,64) (32,64) (80,64): columns filter | gate
    const float* Wd;                                         // (32,32)
    const float* bf; const float* bg; const float* bd;       // nullable
    float* TH; float* SG; float* XN; float* ZC;              // ZC already offset to this layer's 32 columns
    int B, T, Tn, d, o, ow, ldz, tpb;                        // tpb = tiles per batch entry the launch walks (ceil(Tn / 32) - t_lo / 32)
    int t_lo;                                                // first row the launch walks in every batch entry (a multiple of 32): the rows
                                                             // below it lie in front of the layer's receptive offset (see the host code)
    // FUSED lc projection (see tr_layer_fwd_kernel): Q[((b*F + frame)*4 + j)*64 + column] = (mel shifted by j bins) . Wlc of this layer,
    // ctab[phase*4 + j] = the 4-tap composition of the three upsampling kernels at that phase of the hop
    const float* Q; const float* ctab; int hop, F;
};
constexpr int kLcSteps = 10;                                 // 80 / 8

// the other half-wave's value (lane ^ 32) as ONE v_permlane32_swap_b32 -- __shfl_xor goes through the LDS crossbar (ds_bpermute) and a lone
// wave waits out every one of them (tr_layer_bwd1_kernel: eleven per tile)
__device__ __forceinline__ float tr_xor32(float v)
{
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);   // sw[0]: lanes 0-31's values on both halves, sw[1]: lanes 32-63's
    return __uint_as_float((threadIdx.x & 32) ? sw[0] : sw[1]);
}
__device__ __forceinline__ f32x4t tr_ld4(const float* p, bool ok) { f32x4t z = {0.f, 0.f, 0.f, 0.f}; return ok ? *reinterpret_cast<const f32x4t*>(p) : z; }

// conv1d_2's backward in ONE pass over the (rows, S) activations (model.py:161-165 backward), K = O <= 32, S % 64 == 0:
//   dS[r][c]  = [Y1[r][c] + yb[c] > 0] * sum_k dY[r][k] * W2[c][k]        (input gradient with conv1d_1's relu backward)
//   part[chunk][c]       = the chunk's column sums of dS                  (-> db1)
//   wpart[chunk][c][32]  = the chunk's share of dW2[c][k] = sum_r relu(Y1[r][c] + yb[c]) * dY[r][k]
// Before: a library product (K = 30: a 616 MB store), the relu-backward / column-sum pass over it (another 1.8 GB) and the tall-skinny
// dW2 contraction (616 MB again) = 3.1 GB and 0.79 ms; this moves 1.27 GB.  yb == nullptr: no bias vector (mask = Y1 > 0; Y1 holds
// PRE-activations either way: conv1d_1's own relu pass is never run, see the host code).
// A wave keeps its 64 columns of W2 as MFMA B operands (lane (n, hh): W2[c][k = 16 hh + t]) for the whole launch and walks 32-row tiles.
// Per tile: dY's rows twice (row layout = A operand of dS, C layout = B operand of dW2; 3.8 KB, L1 hits), 32 + 32 v_mfma_f32_32x32x2_f32;
// both results come out in the C layout (lane = column) where the mask, the store, the column sum and the dW2 operand relu(Y1 + yb)
// [row rho + 4 hh][c] need no exchange.  Waves of one column group: gw % CG == cg; two waves per SIMD cover each other's loads.
__global__ __launch_bounds__(256, 2) void tr_conv2_bwd_kernel(const float* dY, int O, const float* W2, const float* Y1, const float* yb,
                                                              long long rows, int S, float* dS, float* part, float* wpart)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n = lane & 31, hh = lane >> 5;
    const int CG = S >> 6, gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    const int cg = gw % CG, c0 = cg * 64;
    float bw[2][16], bias[2], cs[2];
    f32x16 wacc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = c0 + 32 * j + n;
#pragma unroll
        for (int t = 0; t < 16; ++t) { const int k = 16 * hh + t; bw[j][t] = k < O ? W2[(long long)c * O + k] : 0.0f; }
        bias[j] = yb ? yb[c] : 0.0f;
        cs[j] = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) wacc[j][i] = 0.0f;
    }
    const rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dY), 0, (int)(rows * O * 4), 0x00020000);
    const rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Y1), 0, (int)(rows * S * 4), 0x00020000);
    const rsrc_t rds = __builtin_amdgcn_make_buffer_rsrc(dS, 0, (int)(rows * S * 4), 0x00020000);
    const int ntile = (int)((rows + 31) / 32);
    for (int tile = gw / CG; tile < ntile; tile += nw / CG) {
        // every request of the tile first: dY in its two layouts (the padded k get the out-of-range offset; rows past the end lie behind the
        // descriptor's size: loads return 0, stores are dropped), then the 32 activation dwords (two 128-byte row pieces per instruction;
        // per-lane offset of the tile + a scalar row offset), which travel while the first MFMAs run
        float av[16], dyc[16];
        {
            const unsigned va = (unsigned)(((tile * 32 + n) * O + 16 * hh) * 4), vc = (unsigned)(((tile * 32 + 4 * hh) * O + n) * 4);
#pragma unroll
            for (int t = 0; t < 16; ++t)
                av[t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rdy, 16 * hh + t < O ? (int)(va + t * 4) : -1, 0, 0));
#pragma unroll
            for (int i = 0; i < 16; ++i)
                dyc[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rdy, n < O ? (int)vc : -1, ((i & 3) + 8 * (i >> 2)) * O * 4, 0));
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned vb = (unsigned)(((tile * 32 + 4 * hh) * S + c0 + n) * 4);
        float y[2][16];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                y[j][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ry, (int)(vb + j * 128), ((i & 3) + 8 * (i >> 2)) * S * 4, 0));
        __builtin_amdgcn_sched_barrier(0);               // (the scheduler moved the activation loads BEHIND the MFMAs otherwise)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bw[j][t], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);           // (keeps the masks -- and the wait for their loads -- behind the MFMAs)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float pre = y[j][i] + bias[j];
                const bool on = pre > 0.0f;
                const float v = on ? acc[i] : 0.0f;                                       // rows past the end: acc = 0 (their A rows are 0)
                cs[j] += v;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rds, (int)(vb + j * 128), ((i & 3) + 8 * (i >> 2)) * S * 4, 0);
                wacc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(on ? pre : 0.0f, dyc[i], wacc[j], 0, 0, 0);   // rows past the end: dyc = 0
            }
        }
    }
    const long long chunk = gw / CG;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float o = tr_xor32(cs[j]);
        if (hh == 0) part[chunk * S + c0 + 32 * j + n] = cs[j] + o;
        // wacc[j]: lane n = k, register i = column (i & 3) + 8 (i >> 2) + 4 hh of the 32-column tile
#pragma unroll
        for (int i = 0; i < 16; ++i) wpart[(chunk * S + c0 + 32 * j + (i & 3) + 8 * (i >> 2) + 4 * hh) * 32 + n] = wacc[j][i];
    }
}

struct LayerA { f32x4t x0[4], x1[4], u[kLcSteps]; };
template <bool FUSED>
__device__ __forceinline__ void tr_layer_load(LayerA& A, const LayerFwdArgs& a, int tile, int lane)
{
    const int b = tile / a.tpb, t = (tile - b * a.tpb) * 32 + a.t_lo + (lane & 31), hh = (lane >> 5) * 4;
    const bool in = tile < a.B * a.tpb && t < a.Tn;
    const float* xr = a.X + ((long long)b * a.Tn + t) * 32 + hh;
    const float* ur = a.U + ((long long)b * a.T + (t - a.o)) * 80 + hh;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        A.x0[i] = tr_ld4(xr - (long long)a.d * 32 + 8 * i, in && t >= a.d);
        A.x1[i] = tr_ld4(xr + 8 * i, in);
    }
    if (!FUSED) {
#pragma unroll
        for (int i = 0; i < kLcSteps; ++i) A.u[i] = tr_ld4(ur + 8 * i, in && t >= a.o);
    }
}

// Two waves per SIMD (8 per workgroup, <= 256 registers each): while one wave runs its gated-unit epilogue on the VALU the other
// wave's MFMAs keep the matrix pipe busy -- a lone wave issues in order and never overlaps the two (SQ_VALU_MFMA_COEXEC_CYCLES = 0
// with one wave per SIMD).  The B operands therefore live in LDS, one conflict-free ds_read_b64 (filter, gate) per MFMA pair.
constexpr int kFwdSteps = 32 + 4 * kLcSteps;                 // tap0 16 + tap1 16 + lc 40 = 72 MFMA steps per column half
// FUSED: the upsampled local condition is never read.  model.py:102-111's three transposed convolutions (kernel (f, 2): two
// adjacent mel bins) compose to a 4-tap filter over mel bins whose taps depend only on the phase inside the hop, so
//   lc projection(t) = sum_j ctab[phase(t)][j] * Q_j[frame(t)],   Q_j = (mel shifted by j bins) . Wlc      (26 frames per entry)
// -- 8 VALU fmas per output row instead of 80 MFMA k-steps per tile, and 320 bytes per row less HBM traffic (a third of the
// kernel's).  Same function, different float association: the training parity is tolerance-based (tests/test_train_gpu.py).
template <bool FUSED>
__global__ void __launch_bounds__(512) tr_layer_fwd_kernel(LayerFwdArgs a)
{
    __shared__ float bt[(FUSED ? 32 : kFwdSteps) * 128];     // [step][lane][filter, gate]
    __shared__ __attribute__((aligned(16))) float cts[FUSED ? 4 * 512 : 4];   // ctab (hop <= 512)
    if (FUSED) for (int e = threadIdx.x; e < a.hop * 4; e += 512) cts[e] = a.ctab[e];
    __shared__ float bdt[16 * 64];                           // dense: [step][lane]
    __shared__ float zt[8][32 * 36];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 31, hh = lane >> 5;
    for (int e = threadIdx.x; e < (FUSED ? 32 : kFwdSteps) * 64; e += 512) {
        const int s = e >> 6, l = e & 63, nn = l & 31, h2 = l >> 5;
        const float* W; int ij;
        if (s < 16) { W = a.W0; ij = s; } else if (s < 32) { W = a.W1; ij = s - 16; } else { W = a.Wlc; ij = s - 32; }
        const int k = 8 * (ij >> 2) + 4 * h2 + (ij & 3);
        bt[e * 2] = W[k * 64 + nn]; bt[e * 2 + 1] = W[k * 64 + 32 + nn];
    }
    for (int e = threadIdx.x; e < 16 * 64; e += 512) {
        const int s = e >> 6, l = e & 63;
        bdt[e] = a.Wd[(8 * (s >> 2) + 4 * (l >> 5) + (s & 3)) * 32 + (l & 31)];
    }
    __syncthreads();
    typedef float f32x2t __attribute__((ext_vector_type(2)));
    const float vbf = a.bf ? a.bf[n] : 0.0f, vbg = a.bg ? a.bg[n] : 0.0f, vbd = a.bd ? a.bd[n] : 0.0f;
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int ntiles = a.B * a.tpb, nwaves = gridDim.x * 8;
    for (int tile = blockIdx.x * 8 + wave; tile < ntiles; tile += nwaves) {
        LayerA A;
        tr_layer_load<FUSED>(A, a, tile, lane);
        // residual operand and gc projection in the output (C) layout: requested now, consumed after the MFMAs
        const int b = tile / a.tpb, t0 = (tile - b * a.tpb) * 32 + a.t_lo;
        const float gcf = a.gcp ? a.gcp[b * 64 + n] : 0.0f, gcg = a.gcp ? a.gcp[b * 64 + 32 + n] : 0.0f;
        // interior tile: every row is inside the utterance, above the layer's receptive offset and on one side of the skip cut --
        // no per-row predicates, and every access is (per-lane pointer) + (compile-time row offset): the 64 stores and 16 loads
        // of the epilogue then need no address arithmetic at all
        const int cut = a.Tn - a.ow;
        const bool interior = t0 >= a.o && t0 + 32 <= a.Tn && (t0 >= cut || t0 + 32 <= cut);
        const long long lrow = ((long long)b * a.Tn + t0 + 4 * hh) * 32 + n;      // this lane's element of tile row 4*hh
        float xres[16];
        if (interior) {
            const float* xp = a.X + lrow;
#pragma unroll
            for (int r = 0; r < 16; ++r) xres[r] = xp[((r & 3) + 8 * (r >> 2)) * 32];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = t0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                xres[r] = t < a.Tn ? a.X[((long long)b * a.Tn + t) * 32 + n] : 0.0f;
            }
        }
        f32x16 cf = zero, cg = zero;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2t w = *reinterpret_cast<const f32x2t*>(&bt[((4 * i + j) * 64 + lane) * 2]);
                cf = __builtin_amdgcn_mfma_f32_32x32x2f32(A.x0[i][j], w[0], cf, 0, 0, 0);
                cg = __builtin_amdgcn_mfma_f32_32x32x2f32(A.x0[i][j], w[1], cg, 0, 0, 0);
            }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2t w = *reinterpret_cast<const f32x2t*>(&bt[((16 + 4 * i + j) * 64 + lane) * 2]);
                cf = __builtin_amdgcn_mfma_f32_32x32x2f32(A.x1[i][j], w[0], cf, 0, 0, 0);
                cg = __builtin_amdgcn_mfma_f32_32x32x2f32(A.x1[i][j], w[1], cg, 0, 0, 0);
            }
        if (!FUSED) {
#pragma unroll
            for (int i = 0; i < kLcSteps; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x2t w = *reinterpret_cast<const f32x2t*>(&bt[((32 + 4 * i + j) * 64 + lane) * 2]);
                    cf = __builtin_amdgcn_mfma_f32_32x32x2f32(A.u[i][j], w[0], cf, 0, 0, 0);
                    cg = __builtin_amdgcn_mfma_f32_32x32x2f32(A.u[i][j], w[1], cg, 0, 0, 0);
                }
        } else {
            // rows of the tile: U row u = t - o; at most two frames per 32-row tile (hop >= 32)
            int u0 = t0 - a.o; u0 = u0 < 0 ? 0 : u0;
            const int fA = u0 / a.hop, fB = fA + 1 < a.F ? fA + 1 : fA;
            const int edge = (fA + 1) * a.hop;                               // first U row of frame fA + 1
            const float* qa = a.Q + (((long long)b * a.F + fA) * 4) * 64 + n;
            const float* qb = a.Q + (((long long)b * a.F + fB) * 4) * 64 + n;
            float qfa[4], qga[4], qfb[4], qgb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { qfa[j] = qa[j * 64]; qga[j] = qa[j * 64 + 32]; qfb[j] = qb[j * 64]; qgb[j] = qb[j * 64 + 32]; }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int u = t0 - a.o + (r & 3) + 8 * (r >> 2) + 4 * hh;
                u = u < 0 ? 0 : u;                                            // rows below the layer's offset are discarded by the epilogue
                const bool hi = u >= edge;
                int ph = u - (hi ? edge : edge - a.hop);
                ph = ph < a.hop ? ph : a.hop - 1;                             // (only rows past the utterance end, discarded as well)
                const f32x4t c = *reinterpret_cast<const f32x4t*>(&cts[ph * 4]);
                const float f0 = hi ? qfb[0] : qfa[0], f1 = hi ? qfb[1] : qfa[1], f2 = hi ? qfb[2] : qfa[2], f3 = hi ? qfb[3] : qfa[3];
                const float g0 = hi ? qgb[0] : qga[0], g1 = hi ? qgb[1] : qga[1], g2 = hi ? qgb[2] : qga[2], g3 = hi ? qgb[3] : qga[3];
                cf[r] += ((c[0] * f0 + c[1] * f1) + c[2] * f2) + c[3] * f3;
                cg[r] += ((c[0] * g0 + c[1] * g1) + c[2] * g2) + c[3] * g3;
            }
        }
        // ---- gated unit (C layout: lane = channel n, register r = row (r&3) + 8(r>>2) + 4hh)
        if (interior) {
            float* thp = a.TH + lrow; float* sgp = a.SG + lrow;
            float* zrow = &zt[wave][4 * hh * 36 + n];
            const bool skip = t0 >= cut;
            float* zcp = a.ZC + ((long long)b * a.ow + (t0 - cut) + 4 * hh) * a.ldz + n;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = (r & 3) + 8 * (r >> 2);
                const float th = tr_tanh_fast((cf[r] + vbf) + gcf), sg = tr_sigmoid_fast((cg[r] + vbg) + gcg);
                const float z = th * sg;
                thp[ro * 32] = th; sgp[ro * 32] = sg;
                if (skip) zcp[(long long)ro * a.ldz] = z;
                zrow[ro * 36] = z;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (r & 3) + 8 * (r >> 2) + 4 * hh, t = t0 + rl;
                const bool in = t < a.Tn, valid = in && t >= a.o;
                float th = 0.0f, sg = 0.0f;
                if (valid) { th = tr_tanh_fast((cf[r] + vbf) + gcf); sg = tr_sigmoid_fast((cg[r] + vbg) + gcg); }
                const float z = th * sg;
                const long long row = (long long)b * a.Tn + t;
                if (in) {
                    a.TH[row * 32 + n] = th; a.SG[row * 32 + n] = sg;
                    if (t >= cut) a.ZC[((long long)b * a.ow + (t - cut)) * a.ldz + n] = z;
                }
                zt[wave][rl * 36 + n] = z;
            }
        }
        // ---- dense 1x1 + residual: z as A operand (row layout) back from this wave's LDS patch
        f32x16 cd = zero;
        {
            const float* zr = &zt[wave][(lane & 31) * 36 + 4 * hh];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4t q = *reinterpret_cast<const f32x4t*>(zr + 8 * i);
#pragma unroll
                for (int j = 0; j < 4; ++j) cd = __builtin_amdgcn_mfma_f32_32x32x2f32(q[j], bdt[(4 * i + j) * 64 + lane], cd, 0, 0, 0);
            }
        }
        if (interior) {
            float* xnp = a.XN + lrow;
#pragma unroll
            for (int r = 0; r < 16; ++r) xnp[((r & 3) + 8 * (r >> 2)) * 32] = (xres[r] + cd[r]) + vbd;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = t0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (t < a.Tn) a.XN[((long long)b * a.Tn + t) * 32 + n] = (xres[r] + cd[r]) + vbd;
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------------------------
//  The FUSED forward layer with every HBM access a fully coalesced 16-byte-per-lane instruction (round 3).
//  Anatomy of tr_layer_fwd_kernel<true> (ablation builds, profiles/r03_train_layer_anatomy.txt): its time is the SUM of its phases --
//  the CU's waves run them in step -- and the memory phases are priced per INSTRUCTION, not per byte: ~22 cycles of the CU's
//  address path for a dword store of two 128-byte rows, ~13 for such a load, ~88 for a float4 row load that touches 32 lines; 120 of
//  them per 32-row tile = 2.5 k of the 4.25 k cycles a tile costs a CU (MFMA 1.3 k).  Here a tile enters and leaves as 1 KB
//  instructions (lane l: the l-th float4 of eight consecutive 128-byte rows): 8 loads and 12 (16 with the skip slice) stores per
//  tile; the MFMA operand layouts (A: lane = row, C: lane = channel) are reached through three wave-private 32 x 36 LDS patches
//  (144-byte row stride: the b128 row reads and the b32 column accesses are both conflict-free).  The next tile's rows are
//  requested before this tile's MFMAs.  Same arithmetic as tr_layer_fwd_kernel<true>.
// -------------------------------------------------------------------------------------------------------------------
constexpr int kPatch = 32 * 36;
#ifdef TWV_TRPROF
// tuning aid (TWV_EXTRA_HIPCC_FLAGS=-DTWV_TRPROF): s_memtime stamps of block 0 / wave 0's first tiles, read back by twv_debug_trprof
__device__ unsigned long long g_trprof[3][8][16];
#define TRPROF(k_, slot_) do { __builtin_amdgcn_sched_barrier(0); if (blockIdx.x == 0 && threadIdx.x == 0 && tpi < 8) g_trprof[k_][tpi][slot_] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
extern "C" int twv_debug_trprof(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trprof), sizeof(unsigned long long) * 3 * 8 * 16) == hipSuccess ? 0 : 1; }
#else
#define TRPROF(k_, slot_) do { } while (0)
#endif
// Every global access of the kernel goes through a buffer descriptor with the array's exact size: a lane that must not touch memory
// (row outside the utterance, tile past the end) gets the offset 0xFFFFFFFF -- the load returns 0, the store is dropped -- so the
// loop body has NO branch around a memory instruction.  That matters more than the branch itself: s_waitcnt vmcnt counts in issue
// order, and behind a conditional load or store the compiler has to assume it was not issued -- its wait for an older load then
// covers the younger ones too (the first version waited for the rows it had just prefetched, and for every store of the last tile).
struct FwdcBufs { rsrc_t x, th, sg, xn, q, gc; };
// everything a tile reads from HBM / L2: the 32 rows of X at t and at t - d (1 KB per instruction), the lc-projection rows Q of the (at most
// two) frames the tile touches and the gc projection of its batch entry, laid out as the B operands of the five extra k-pairs
struct FwdcIn { f32x4t g0[4], g1[4]; float qf[4], qg[4], gcf, gcg; };
__device__ __forceinline__ void tr_tile_fetch(FwdcIn& in_, const LayerFwdArgs& a, const FwdcBufs& bf, int tile, int ntiles, int lane)
{
    f32x4t (&g0)[4] = in_.g0; f32x4t (&g1)[4] = in_.g1;
    const rsrc_t rx = bf.x;
    tile = tile < ntiles ? tile : ntiles - 1;                                   // (past the end: any valid tile, never used)
    const int b = tile / a.tpb, t0 = (tile - b * a.tpb) * 32 + a.t_lo, cr = lane >> 3, cc = (lane & 7) * 4;
    {
        const int n = lane & 31, hh = lane >> 5;
        int u0 = t0 - a.o; u0 = u0 < 0 ? 0 : u0;
        const int fA = u0 / a.hop, fB = fA + 1 < a.F ? fA + 1 : fA;
        const int qa = (((b * a.F + fA) * 4 + hh) * 64 + n) * 4, qb = (((b * a.F + fB) * 4 + hh) * 64 + n) * 4;   // byte offsets
        in_.qf[0] = load_f32_b(bf.q, qa, 0); in_.qg[0] = load_f32_b(bf.q, qa + 128, 0); in_.qf[1] = load_f32_b(bf.q, qa + 512, 0); in_.qg[1] = load_f32_b(bf.q, qa + 640, 0);
        in_.qf[2] = load_f32_b(bf.q, qb, 0); in_.qg[2] = load_f32_b(bf.q, qb + 128, 0); in_.qf[3] = load_f32_b(bf.q, qb + 512, 0); in_.qg[3] = load_f32_b(bf.q, qb + 640, 0);
        in_.gcf = load_f32_b(bf.gc, (b * 64 + n) * 4, 0); in_.gcg = load_f32_b(bf.gc, (b * 64 + 32 + n) * 4, 0);           // 0 without gc (empty descriptor)
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = t0 + 8 * k + cr;
        const bool in = t < a.Tn;
        const unsigned o1 = (unsigned)(((b * a.Tn + t) * 32 + cc) * 4);
        g1[k] = tr_bld4(rx, in ? o1 : 0xFFFFFFFFu);
        g0[k] = tr_bld4(rx, in && t >= a.d ? o1 - (unsigned)a.d * 128u : 0xFFFFFFFFu);
    }
}
// rows of a wave-private patch (p[row * 36 + channel]) to global rows row0.. of a (rows, ld) array, 1 KB per instruction; rows
// [rlo, rhi) of the tile are written
__device__ __forceinline__ void tr_patch_store(const float* p, rsrc_t r, long long row0, int ld, int rlo, int rhi, int lane)
{
    const int cr = lane >> 3, cc = (lane & 7) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = 8 * k + cr;
        const f32x4t v = *reinterpret_cast<const f32x4t*>(p + row * 36 + cc);
        tr_bst4(r, row >= rlo && row < rhi ? (unsigned)(((row0 + row) * ld + cc) * 4) : 0xFFFFFFFFu, v);
    }
}
constexpr int kFwdcWaves = 8;                               // 2 per SIMD (3 would need <= 168 VGPRs: spills), one workgroup per CU
__global__ void __launch_bounds__(kFwdcWaves * 64) tr_layer_fwdc_kernel(LayerFwdArgs a)
{
    __shared__ float bt[32 * 128];                            // [step][lane][filter, gate]
    __shared__ __attribute__((aligned(16))) float cts[4 * 512];   // ctab (hop <= 512)
    __shared__ float bdt[16 * 64];                            // dense: [step][lane]
    __shared__ __attribute__((aligned(16))) float patch[kFwdcWaves][2][kPatch];
    for (int e = threadIdx.x; e < a.hop * 4; e += kFwdcWaves * 64) cts[e] = a.ctab[e];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 31, hh = lane >> 5;
    for (int e = threadIdx.x; e < 32 * 64; e += kFwdcWaves * 64) {
        const int s = e >> 6, l = e & 63, nn = l & 31, h2 = l >> 5;
        const float* W = s < 16 ? a.W0 : a.W1;
        const int ij = s & 15, k = 8 * (ij >> 2) + 4 * h2 + (ij & 3);
        bt[e * 2] = W[k * 64 + nn]; bt[e * 2 + 1] = W[k * 64 + 32 + nn];
    }
    for (int e = threadIdx.x; e < 16 * 64; e += kFwdcWaves * 64) {
        const int s = e >> 6, l = e & 63;
        bdt[e] = a.Wd[(8 * (s >> 2) + 4 * (l >> 5) + (s & 3)) * 32 + (l & 31)];
    }
    __syncthreads();
    typedef float f32x2t __attribute__((ext_vector_type(2)));
    const float vbf = a.bf ? a.bf[n] : 0.0f, vbg = a.bg ? a.bg[n] : 0.0f, vbd = a.bd ? a.bd[n] : 0.0f;
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int ntiles = a.B * a.tpb, nwaves = gridDim.x * kFwdcWaves;
    float* pa = patch[wave][0]; float* pb = patch[wave][1];
    const int cr = lane >> 3, cc = (lane & 7) * 4, cut = a.Tn - a.ow;
    FwdcBufs bf;
    {
        const int act = (int)((long long)a.B * a.Tn * 32 * 4);
        bf.x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X), 0, act, 0x00020000);
        bf.th = __builtin_amdgcn_make_buffer_rsrc(a.TH, 0, act, 0x00020000);
        bf.sg = __builtin_amdgcn_make_buffer_rsrc(a.SG, 0, act, 0x00020000);
        bf.xn = __builtin_amdgcn_make_buffer_rsrc(a.XN, 0, act, 0x00020000);
        bf.q = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Q), 0, (int)((long long)a.B * a.F * 4 * 64 * 4), 0x00020000);
        bf.gc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gcp ? a.gcp : a.X), 0, a.gcp ? a.B * 64 * 4 : 0, 0x00020000);
    }
    FwdcIn in;
    int tile = blockIdx.x * kFwdcWaves + wave;
    tr_tile_fetch(in, a, bf, tile, ntiles, lane);
    int tpi = 0;
    for (; tile < ntiles; tile += nwaves, ++tpi) {
        const int b = tile / a.tpb, t0 = (tile - b * a.tpb) * 32 + a.t_lo;
        TRPROF(0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            *reinterpret_cast<f32x4t*>(pa + (8 * k + cr) * 36 + cc) = in.g0[k];
            *reinterpret_cast<f32x4t*>(pb + (8 * k + cr) * 36 + cc) = in.g1[k];
        }
        TRPROF(0, 1);
        // The lc projection, the biases and the gc projection ride on the matrix cores as five more k-pairs:
        //   pre += [c(ph) (1 - hi) | c(ph) hi | 1 0] . [Q(frame fA) ; Q(frame fB) ; bias + gc ; 0]        (k = 0..9; at most two frames per tile)
        // B operand of k-pair s: lane (n, hh) carries k = 2s + hh
        int u0 = t0 - a.o; u0 = u0 < 0 ? 0 : u0;
        const int edge = (u0 / a.hop + 1) * a.hop;                              // first U row of the tile's second frame
        f32x16 cf = zero, cg = zero;
        {
            // A operand of the five extra k-pairs: lane (row, hh) carries k = 2s + hh of its row
            int u = t0 + n - a.o;
            u = u < 0 ? 0 : u;                                                  // rows below the layer's offset are masked in the epilogue
            const bool hi = u >= edge;
            int ph = u - (hi ? edge : edge - a.hop);
            ph = ph < a.hop ? ph : a.hop - 1;                                   // (only rows past the utterance end, never stored)
            const f32x4t c = *reinterpret_cast<const f32x4t*>(&cts[ph * 4]);
            const float c0 = hh ? c[1] : c[0], c1 = hh ? c[3] : c[2];
            const float ax[4] = {hi ? 0.0f : c0, hi ? 0.0f : c1, hi ? c0 : 0.0f, hi ? c1 : 0.0f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                cf = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[q], in.qf[q], cf, 0, 0, 0);
                cg = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[q], in.qg[q], cg, 0, 0, 0);
            }
            const float one = hh ? 0.0f : 1.0f;
            cf = __builtin_amdgcn_mfma_f32_32x32x2f32(one, vbf + in.gcf, cf, 0, 0, 0);
            cg = __builtin_amdgcn_mfma_f32_32x32x2f32(one, vbg + in.gcg, cg, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);                                      // (the prefetch below overwrites `in`: after these MFMAs)
        // the NEXT tile's operands, all of them, before anything of this tile is stored: s_waitcnt vmcnt retires in issue order, so a
        // load issued behind a store waits for that store's acknowledgement -- with this order the wait at the top of the next trip
        // leaves this tile's sixteen stores in flight
        tr_tile_fetch(in, a, bf, tile + nwaves, ntiles, lane);
        __builtin_amdgcn_sched_barrier(0);                                      // (and ahead of the tap MFMAs)
        TRPROF(0, 2);
        // A operands (lane = row) and the residual operand (C layout) out of the patches
        f32x4t A0[4], A1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            A0[i] = *reinterpret_cast<const f32x4t*>(pa + n * 36 + 8 * i + 4 * hh);
            A1[i] = *reinterpret_cast<const f32x4t*>(pb + n * 36 + 8 * i + 4 * hh);
        }
        float xres[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) xres[r] = pb[((r & 3) + 8 * (r >> 2) + 4 * hh) * 36 + n];
        TRPROF(0, 3);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2t w = *reinterpret_cast<const f32x2t*>(&bt[((4 * i + j) * 64 + lane) * 2]);
                cf = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[i][j], w[0], cf, 0, 0, 0);
                cg = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[i][j], w[1], cg, 0, 0, 0);
            }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2t w = *reinterpret_cast<const f32x2t*>(&bt[((16 + 4 * i + j) * 64 + lane) * 2]);
                cf = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[i][j], w[0], cf, 0, 0, 0);
                cg = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[i][j], w[1], cg, 0, 0, 0);
            }
        TRPROF(0, 4);
        // ---- gated unit (C layout: lane = channel n, register r = row (r&3) + 8(r>>2) + 4hh)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = (r & 3) + 8 * (r >> 2) + 4 * hh;
            const bool valid = t0 + rl >= a.o;
            const float thv = tr_tanh_fast(cf[r]), sgv = tr_sigmoid_fast(cg[r]);    // evaluated for every row: no branch in the loop body
            const float th = valid ? thv : 0.0f, sg = valid ? sgv : 0.0f;
            pa[rl * 36 + n] = th; pb[rl * 36 + n] = sg; cf[r] = th * sg;
        }
        TRPROF(0, 5);
        const int nrows = a.Tn - t0 < 32 ? a.Tn - t0 : 32;
        const long long grow = (long long)b * a.Tn + t0;
        tr_patch_store(pa, bf.th, grow, 32, 0, nrows, lane);
        tr_patch_store(pb, bf.sg, grow, 32, 0, nrows, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) pa[((r & 3) + 8 * (r >> 2) + 4 * hh) * 36 + n] = cf[r];      // z: the patch TH has just left
        const float* pz = pa;
        // skip input slice: rows t >= cut (model.py:94-96 keeps the last ow); other tiles' stores are dropped by the descriptor
        {
            // (the stacked skip input is the one array that outgrows a 32-bit byte offset -- B * ow * 960 * 4 bytes, 2 GiB from batch 117 at 7800
            // samples: its descriptor is re-based to the tile, 32 rows of ldz floats; in front of the cut every lane is out of range)
            const rsrc_t rzc = __builtin_amdgcn_make_buffer_rsrc(a.ZC + ((long long)b * a.ow + (t0 - cut)) * a.ldz, 0, (int)((31LL * a.ldz + 32) * 4), 0x00020000);
            tr_patch_store(pz, rzc, 0, a.ldz, cut - t0, nrows, lane);
        }
        TRPROF(0, 6);
        // ---- dense 1x1 + residual: z as A operand (row layout) back from the patch
        f32x16 cd = zero;
        {
            const float* zr = pz + n * 36 + 4 * hh;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4t q = *reinterpret_cast<const f32x4t*>(zr + 8 * i);
#pragma unroll
                for (int j = 0; j < 4; ++j) cd = __builtin_amdgcn_mfma_f32_32x32x2f32(q[j], bdt[(4 * i + j) * 64 + lane], cd, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) pb[((r & 3) + 8 * (r >> 2) + 4 * hh) * 36 + n] = (xres[r] + cd[r]) + vbd;
        TRPROF(0, 7);
        tr_patch_store(pb, bf.xn, grow, 32, 0, nrows, lane);
        TRPROF(0, 8);
    }
}

// ===================================================================================================================
//  Fused residual layer, backward.  Two kernels per layer, both on v_mfma_f32_32x32x2_f32 with 32-row tiles per wave:
//  K1 (tr_layer_bwd1_kernel): dZ = dXn Wd^T (+ the skip path's dZC) -> gated-unit backward -> dPRE (stored once, row major)
//      and EVERY weight gradient of the layer.  The MFMA C layout (lane = channel n, register r = row rho(r,hh)) is exactly
//      the B operand of a "rows are the contraction" MFMA step, so dF / dG feed X^T dPRE, U^T dPRE, Z^T dXn straight from
//      registers; the A operands are coalesced scalar loads (lane = channel).  Eleven 32x32 gradient tiles accumulate in
//      registers over all tiles of the wave, are summed over the 4 waves through LDS and leave as one slab per workgroup
//      (fixed-order reduction afterwards: deterministic).  Bias / gc sums leave as per-tile column sums.
//  K2 (tr_layer_bwd2_kernel): dX = dXn + dPRE[t] W1^T + dPRE[t+d] W0^T and dU[t-o] += dPRE[t] Wlc^T with the transposed
//      weights register-resident.
// ===================================================================================================================
struct LayerBwdArgs {
    const float* dXn; const float* dZC;                      // (B*Tn,32), skip share (B*ow rows, ld ldz) already offset to the layer
    const float* TH; const float* SG; const float* X; const float* U;
    const float* W0; const float* W1; const float* Wlc; const float* Wd;
    float* dPRE;                                             // (B*Tn,64)
    float* slabs;                                            // [gridDim.x][11][32][32] gradient tiles
    float* tsum;                                             // [ntiles][96]: column sums of dF | dG | dXn
    float* dX; float* dU;                                    // K2 outputs
    const float* zeros;                                      // >= 1 KB of zeros (masked LDS-DMA lanes read here)
    int B, T, Tn, d, o, ow, ldz, tpb;
    int t_lo, tpbf;                                          // first row walked per batch entry (see LayerFwdArgs); ceil(Tn / 32) = the tile count tsum / PT are indexed by
    // FUSED lc path (see tr_layer_fwd_kernel): frame-rate projections Q of this layer; D[(b*T + u)*4 + j] accumulates, over the layers,
    // sum_columns dPRE[row of U-row u] * Q_j[frame(u)] -- all the upsampling kernels' gradients need (tr_dctab / tr_up_grad kernels)
    const float* Q; float* D; int hop, F;
    // bwd1: the hop's tap table and this layer's per-tile partial sums PT[tile][2 frame slots][4 taps][64 columns] of ctab[phase][j] * dPRE
    const float* ctab; float* PT;
};
enum { GQ_W1F = 0, GQ_W1G, GQ_W0F, GQ_W0G, GQ_LCF0, GQ_LCF1, GQ_LCF2, GQ_LCG0, GQ_LCG1, GQ_LCG2, GQ_WD, GQ_N };

// LDS staging of one tile's operands (floats, per wave): LDS-DMA (global_load_lds b128) fills it for tile i+1 while tile i's MFMAs run
enum { BS_TH = 0, BS_SG = 1024, BS_DXN = 2048, BS_DZC = 3072, BS_X1 = 4096, BS_X0 = 5120, BS_U = 6144, BS_FLOATS = 6144 + 2560 };

template <bool FUSED>
__device__ __forceinline__ void tr_bwd1_stage(const LayerBwdArgs& a, int tile, int lane, int base /* float offset of the wave's region in lds[] */)
{
    const int b = tile / a.tpb, t0 = (tile - b * a.tpb) * 32 + a.t_lo;
    const long long row0 = (long long)b * a.Tn + t0;
    // four arrays whose 32-row tile is 4 KB contiguous: piece p = 1 KB = 8 rows
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const long long off = row0 * 32 + p * 256 + lane * 4;
        __builtin_amdgcn_global_load_lds((gptr_t)(a.TH + off), (lptr_t)(lds + base + BS_TH + p * 256), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(a.SG + off), (lptr_t)(lds + base + BS_SG + p * 256), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(a.dXn + off), (lptr_t)(lds + base + BS_DXN + p * 256), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(a.X + off), (lptr_t)(lds + base + BS_X1 + p * 256), 16, 0, 0);
        // x[t-d]: rows before the start of the buffer come from the zero page
        const long long offd = off - (long long)a.d * 32;
        __builtin_amdgcn_global_load_lds((gptr_t)(offd >= 0 ? a.X + offd : a.zeros + lane * 4), (lptr_t)(lds + base + BS_X0 + p * 256), 16, 0, 0);
        // skip share: row r = p*8 + lane/8 of the tile, 4 columns per lane, only the last `ow` positions exist
        const int r = p * 8 + (lane >> 3), t = t0 + r, pz = t - (a.Tn - a.ow);
        const bool zok = t < a.Tn && pz >= 0;
        __builtin_amdgcn_global_load_lds((gptr_t)(zok ? a.dZC + ((long long)b * a.ow + pz) * a.ldz + (lane & 7) * 4 : a.zeros + lane * 4),
                                         (lptr_t)(lds + base + BS_DZC + p * 256), 16, 0, 0);
    }
    if (!FUSED) {
        // U rows t0-o .. t0-o+31: 10 KB contiguous
        const long long ubase = ((long long)b * a.T + (t0 - a.o)) * 80;
#pragma unroll
        for (int p = 0; p < 10; ++p) {
            const long long off = ubase + p * 256 + lane * 4;
            __builtin_amdgcn_global_load_lds((gptr_t)(off >= 0 ? a.U + off : a.zeros + lane * 4), (lptr_t)(lds + base + BS_U + p * 256), 16, 0, 0);
        }
    }
}

// FUSED: U is not read and W_lc's six gradient tiles are not accumulated here (96 of the 176 MFMAs per tile).  With the lc term written
// as sum_j ctab[phase][j] * Q_j[frame], dW_lc = sum_frames shift_j(mel)^T R_j with R_j[frame] = sum over the frame's rows of
// ctab[phase][j] * dPRE[row]: each tile leaves its share of R for the one or two frames it touches (PT), summed per frame afterwards.
template <bool FUSED>
__global__ void __launch_bounds__(256) tr_layer_bwd1_kernel(LayerBwdArgs a)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, hh = lane >> 5;
    const int base = wave * BS_FLOATS;
    const int cto = 4 * BS_FLOATS;                           // FUSED: ctab behind the four waves' staging areas
    if (FUSED) {
        for (int e = threadIdx.x; e < a.hop * 4; e += 256) lds[cto + e] = a.ctab[e];
        __syncthreads();
    }
    float bwd[16];                                           // Wd^T as B operand: lane (n, hh): Wd[n][k = 8i + 4hh + j]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) bwd[4 * i + j] = a.Wd[n * 32 + 8 * i + 4 * hh + j];
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 g[GQ_N];
#pragma unroll
    for (int q = 0; q < GQ_N; ++q) g[q] = zero;
    const int ntiles = a.B * a.tpb, nwaves = gridDim.x * 4;
    int tile = blockIdx.x * 4 + wave;
    if (tile < ntiles) tr_bwd1_stage<FUSED>(a, tile, lane, base);
    int tpi = 0;
    for (; tile < ntiles; tile += nwaves, ++tpi) {
        const int b = tile / a.tpb, t0 = (tile - b * a.tpb) * 32 + a.t_lo;
        const int gtile = b * a.tpbf + (t0 >> 5);            // the tile's index in tsum / PT (all ceil(Tn / 32) tiles of every entry)
        TRPROF(1, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this tile's operands are in LDS
        TRPROF(1, 1);
        // ---- dZ = dXn Wd^T  (A operand: row layout, float4 from the staged tile)
        f32x16 cz = zero;
        {
            const int t = t0 + (lane & 31);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4t q = *reinterpret_cast<const f32x4t*>(&lds[base + BS_DXN + (lane & 31) * 32 + 8 * i + 4 * hh]);
                if (t >= a.Tn) q = f32x4t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) cz = __builtin_amdgcn_mfma_f32_32x32x2f32(q[j], bwd[4 * i + j], cz, 0, 0, 0);
            }
        }
        // ---- C layout operands from LDS: lane = channel n, register r = row rho.  Interior tiles (all rows inside the utterance, above
        // the receptive offset and the dilation) need no predicates and use per-lane bases + compile-time offsets.
        const bool interior = t0 >= a.o && t0 >= a.d && t0 + 32 <= a.Tn;
        float th_[16], sg_[16], dzc[16], dxc[16], x1[16], x0[16], u0[16], u1[16], u2[16];
        if (interior) {
            const int lb = base + 4 * hh * 32 + n, ub = base + BS_U + 4 * hh * 80 + n;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = (r & 3) + 8 * (r >> 2);
                th_[r] = lds[lb + BS_TH + ro * 32]; sg_[r] = lds[lb + BS_SG + ro * 32]; dxc[r] = lds[lb + BS_DXN + ro * 32];
                dzc[r] = lds[lb + BS_DZC + ro * 32]; x1[r] = lds[lb + BS_X1 + ro * 32]; x0[r] = lds[lb + BS_X0 + ro * 32];
                if (!FUSED) { u0[r] = lds[ub + ro * 80]; u1[r] = lds[ub + ro * 80 + 32]; u2[r] = n < 16 ? lds[ub + ro * 80 + 64] : 0.0f; }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (r & 3) + 8 * (r >> 2) + 4 * hh, t = t0 + rl;
                const bool in = t < a.Tn, valid = in && t >= a.o;
                th_[r] = in ? lds[base + BS_TH + rl * 32 + n] : 0.0f;
                sg_[r] = in ? lds[base + BS_SG + rl * 32 + n] : 0.0f;
                dxc[r] = in ? lds[base + BS_DXN + rl * 32 + n] : 0.0f;
                dzc[r] = lds[base + BS_DZC + rl * 32 + n];
                x1[r] = in ? lds[base + BS_X1 + rl * 32 + n] : 0.0f;
                x0[r] = (in && t >= a.d) ? lds[base + BS_X0 + rl * 32 + n] : 0.0f;
                if (!FUSED) {
                    u0[r] = valid ? lds[base + BS_U + rl * 80 + n] : 0.0f;
                    u1[r] = valid ? lds[base + BS_U + rl * 80 + 32 + n] : 0.0f;
                    u2[r] = (valid && n < 16) ? lds[base + BS_U + rl * 80 + 64 + n] : 0.0f;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every LDS read has returned: the region may be refilled
        TRPROF(1, 2);
        TRPROF(1, 3);
        float dF[16], dG[16], zc[16];
        float sf = 0.0f, sgs = 0.0f, sx = 0.0f;
        if (interior) {
            float* pp = a.dPRE + ((long long)b * a.Tn + t0 + 4 * hh) * 64 + n;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = (r & 3) + 8 * (r >> 2);
                const float dz = cz[r] + dzc[r];
                const float th = th_[r], sg = sg_[r];
                const float df = dz * sg * (1.0f - th * th), dg = dz * th * sg * (1.0f - sg);
                dF[r] = df; dG[r] = dg; zc[r] = th * sg;
                pp[ro * 64] = df; pp[ro * 64 + 32] = dg;
                sf += df; sgs += dg; sx += dxc[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = t0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const bool in = t < a.Tn, valid = in && t >= a.o;
                const float dz = cz[r] + dzc[r];
                const float th = th_[r], sg = sg_[r];
                float df = 0.0f, dg = 0.0f;
                if (valid) { df = dz * sg * (1.0f - th * th); dg = dz * th * sg * (1.0f - sg); }
                dF[r] = df; dG[r] = dg; zc[r] = th * sg;
                if (in) { const long long row = (long long)b * a.Tn + t; a.dPRE[row * 64 + n] = df; a.dPRE[row * 64 + 32 + n] = dg; }
                sf += df; sgs += dg; sx += dxc[r];
            }
        }
        TRPROF(1, 4);
        // per-tile column sums (bias and gc gradients): halves combined, lanes 0..31 write
        {
            const float of = tr_xor32(sf), og = tr_xor32(sgs), ox = tr_xor32(sx);
            if (hh == 0) { a.tsum[(long long)gtile * 96 + n] = sf + of; a.tsum[(long long)gtile * 96 + 32 + n] = sgs + og; a.tsum[(long long)gtile * 96 + 64 + n] = sx + ox; }
        }
        if (FUSED) {
            // this tile's share of R_j[frame][column] = sum_rows ctab[phase(row)][j] * dPRE[row][column]; slot 0 = the frame of the tile's
            // first U row, slot 1 = the next frame (only tiles that straddle a frame edge fill it)
            const int u0 = t0 - a.o, u0c = u0 < 0 ? 0 : u0;
            const int fA = u0c / a.hop, edge = (fA + 1) * a.hop;
            const bool spans = u0 + 31 >= edge;
            float raf[4] = {0.f, 0.f, 0.f, 0.f}, rag[4] = {0.f, 0.f, 0.f, 0.f}, rbf[4] = {0.f, 0.f, 0.f, 0.f}, rbg[4] = {0.f, 0.f, 0.f, 0.f};
            // four rows at a time: their tap vectors are fetched together (one LDS round trip per four rows instead of one per row --
            // a lone wave waits out every one of them; stamps: this phase was 3.6 k of the tile's 14.3 k cycles)
#pragma unroll
            for (int r4 = 0; r4 < 16; r4 += 4) {
                f32x4t c4[4]; bool hi4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = r4 + q;
                    int u = u0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    u = u < 0 ? 0 : u;                                        // rows below the layer's offset carry dF = dG = 0
                    hi4[q] = u >= edge;
                    int ph = u - (hi4[q] ? edge : edge - a.hop);
                    ph = ph < a.hop ? ph : a.hop - 1;
                    c4[q] = *reinterpret_cast<const f32x4t*>(&lds[cto + ph * 4]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = r4 + q;
                    const f32x4t c = c4[q];
                    const bool hi = hi4[q];
                    const float fl = hi ? 0.0f : dF[r], gl = hi ? 0.0f : dG[r];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { raf[j] += c[j] * fl; rag[j] += c[j] * gl; }
                    if (spans) {
                        const float fh = hi ? dF[r] : 0.0f, gh = hi ? dG[r] : 0.0f;
#pragma unroll
                        for (int j = 0; j < 4; ++j) { rbf[j] += c[j] * fh; rbg[j] += c[j] * gh; }
                    }
                }
            }
            float* pt = a.PT + (long long)gtile * 512 + n;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float f0_ = raf[j] + tr_xor32(raf[j]), g0_ = rag[j] + tr_xor32(rag[j]);
                if (hh == 0) { pt[j * 64] = f0_; pt[j * 64 + 32] = g0_; }
            }
            if (spans) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float f1_ = rbf[j] + tr_xor32(rbf[j]), g1_ = rbg[j] + tr_xor32(rbg[j]);
                    if (hh == 0) { pt[256 + j * 64] = f1_; pt[256 + j * 64 + 32] = g1_; }
                }
            }
        }
        TRPROF(1, 5);
        // the next tile's stage, AFTER this tile's stores (dPRE, tile sums, PT): the s_waitcnt vmcnt(0) at the top of the next trip then waits
        // for requests that are the youngest in the queue -- issued in front of the stores (right after the LDS reads, where the region becomes
        // free) it also waited for the acknowledgement of every store behind them.  The 80 MFMAs below cover the flight (120 -> 115 us).
        if (tile + nwaves < ntiles) tr_bwd1_stage<FUSED>(a, tile + nwaves, lane, base);
        // ---- weight gradients: A = [row rho][channel], B = dF / dG / dXn registers
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            g[GQ_W1F] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[r], dF[r], g[GQ_W1F], 0, 0, 0);
            g[GQ_W1G] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[r], dG[r], g[GQ_W1G], 0, 0, 0);
            g[GQ_W0F] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[r], dF[r], g[GQ_W0F], 0, 0, 0);
            g[GQ_W0G] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[r], dG[r], g[GQ_W0G], 0, 0, 0);
            if (!FUSED) {
                g[GQ_LCF0] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0[r], dF[r], g[GQ_LCF0], 0, 0, 0);
                g[GQ_LCG0] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0[r], dG[r], g[GQ_LCG0], 0, 0, 0);
                g[GQ_LCF1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u1[r], dF[r], g[GQ_LCF1], 0, 0, 0);
                g[GQ_LCG1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u1[r], dG[r], g[GQ_LCG1], 0, 0, 0);
                g[GQ_LCF2] = __builtin_amdgcn_mfma_f32_32x32x2f32(u2[r], dF[r], g[GQ_LCF2], 0, 0, 0);
                g[GQ_LCG2] = __builtin_amdgcn_mfma_f32_32x32x2f32(u2[r], dG[r], g[GQ_LCG2], 0, 0, 0);
            }
            g[GQ_WD] = __builtin_amdgcn_mfma_f32_32x32x2f32(zc[r], dxc[r], g[GQ_WD], 0, 0, 0);
        }
        TRPROF(1, 6);
    }
    // ---- the tiles the launch does not walk (rows in front of the layer's offset): their column sums are zero, and the segment sums
    // over all tiles of an entry read them
    {
        const int nlo = a.t_lo >> 5;
        for (int s_ = blockIdx.x * 4 + wave; s_ < a.B * nlo; s_ += nwaves) {
            const int b = s_ / nlo, gt = b * a.tpbf + (s_ - b * nlo);
            a.tsum[(long long)gt * 96 + lane] = 0.0f;
            if (lane < 32) a.tsum[(long long)gt * 96 + 64 + lane] = 0.0f;
        }
    }
    // ---- one slab per workgroup: the four waves' tiles summed through LDS (the staging area is free now) in a fixed order
    float* slab = a.slabs + (long long)blockIdx.x * GQ_N * 1024;
#pragma unroll
    for (int q = 0; q < GQ_N; ++q) {
        if (FUSED && q >= GQ_LCF0 && q <= GQ_LCG2) continue;          // W_lc's gradient comes from the per-frame sums
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) lds[wave * 1056 + ((r & 3) + 8 * (r >> 2) + 4 * hh) * 33 + n] = g[q][r];
        __syncthreads();
        for (int i = threadIdx.x; i < 1024; i += 256) {
            const int rr = i >> 5, cc = i & 31, o = rr * 33 + cc;
            slab[q * 1024 + i] = (lds[o] + lds[1056 + o]) + (lds[2112 + o] + lds[3168 + o]);
        }
    }
}
// R[l][(b*F + f)*4 + j][c] = sum, in tile order, of the per-tile partials that belong to frame f of batch entry b in layer l (offs[l] = the layer's
// receptive offset o: U row u = t - o).  grid (ceil(B*F*256 / 256), 1, NL), one thread per (b, f, j, c).
struct LayerOffs { int o[64]; };
__global__ __launch_bounds__(256) void tr_frame_reduce_kernel(const float* PT, long long pt_lstride, LayerOffs offs, int B, int F, int hop, int tpb, int Tn,
                                                              float* R, long long r_lstride)
{
    const int l = blockIdx.z, o = offs.o[l];
    const float* pt = PT + (long long)l * pt_lstride;
    GRID_STRIDE(i, (long long)B * F * 256) {
        const int c = (int)(i & 63), j = (int)((i >> 6) & 3);
        const int bf = (int)(i >> 8), b = bf / F, f = bf - b * F;
        int ta = (f * hop + o) >> 5, tb = ((f + 1) * hop + o - 1) >> 5;
        tb = tb < tpb ? tb : tpb - 1;
        float s = 0.0f;
        for (int tl = ta; tl <= tb; ++tl) {
            const int u0 = tl * 32 - o, u0c = u0 < 0 ? 0 : u0;
            const int fA = u0c / hop, edge = (fA + 1) * hop;
            const float* p = pt + ((long long)b * tpb + tl) * 512 + j * 64 + c;
            if (fA == f) s += p[0];
            else if (fA + 1 == f && u0 + 31 >= edge) s += p[256];
        }
        R[(long long)l * r_lstride + i] = s;
    }
}
// every gradient tile q of a layer: out_q[(m, n)] = sum over slabs (fixed order); rows >= mrows are dropped (lc block 2 has 16 rows)
// ONE launch for all layers (blockIdx.z = layer): the layers' slabs are kept until the end of the backward pass, layer l's outputs sit
// lstride[q] floats behind layer 0's (the per-layer reductions were ~300 five-microsecond launches per step, 8 % of it)
struct SlabDst { float* out[GQ_N]; int ldo[GQ_N]; int mrows[GQ_N]; long long lstride[GQ_N]; };
__global__ __launch_bounds__(256) void tr_slab_reduce_kernel(const float* slabs, long long slab_lstride, int nslab, SlabDst dst)
{
    const int q = blockIdx.y, l = blockIdx.z;
    if (dst.mrows[q] == 0) return;                           // (fused lc path: six of the eleven tiles are neither written nor wanted)
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    const float* sl = slabs + (long long)l * slab_lstride;
    float s = 0.0f;
    for (int k = grp; k < nslab; k += 4) s += sl[((long long)k * GQ_N + q) * 1024 + i];
    __shared__ float sh[256];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < 64 && (i >> 5) < dst.mrows[q])
        dst.out[q][(long long)l * dst.lstride[q] + (long long)(i >> 5) * dst.ldo[q] + (i & 31)] =
            (sh[threadIdx.x] + sh[threadIdx.x + 64]) + (sh[threadIdx.x + 128] + sh[threadIdx.x + 192]);
}
// conv biases of every layer from its 64 summed gate pre-activation gradients: bf[l] = s[l][0:32], bg[l] = s[l][32:64]
__global__ void tr_split_bias_kernel(const float* s64, float* base, long long lstride, long long o_bf, long long o_bg, int NL)
{
    GRID_STRIDE(i, (long long)NL * 64) {
        const int l = (int)(i >> 6), c = (int)(i & 63);
        base[(long long)l * lstride + (c < 32 ? o_bf + c : o_bg + c - 32)] = s64[i];
    }
}
// demb[b][g] += sum over layers l (ascending), j (ascending) of dGCP[l][b][j] * Wgc_l[g][j]   (Wgc_l = views + l*vstride, row g of (G x 64))
// one wave per (batch entry, embedding column): lane j carries column j of the 64-wide projection through the layers (coalesced 256-byte
// rows), then a butterfly over the lanes (fixed order).  One THREAD per output walked 30 x 64 dependent loads: 73 us for 2.5 MB.
__global__ __launch_bounds__(256) void tr_demb_kernel(const float* dgcp, const float* wgc0, long long vstride, float* demb, int NL, int B, int G)
{
    const int lane = threadIdx.x & 63;
    const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= (long long)B * G) return;
    const int b = (int)(i / G), g = (int)(i % G);
    float s0 = 0.0f, s1 = 0.0f;
    int l = 0;
    for (; l + 1 < NL; l += 2) {
        s0 += dgcp[((long long)l * B + b) * 64 + lane] * wgc0[(long long)l * vstride + (long long)g * 64 + lane];
        s1 += dgcp[((long long)(l + 1) * B + b) * 64 + lane] * wgc0[(long long)(l + 1) * vstride + (long long)g * 64 + lane];
    }
    if (l < NL) s0 += dgcp[((long long)l * B + b) * 64 + lane] * wgc0[(long long)l * vstride + (long long)g * 64 + lane];
    float s = s0 + s1;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) demb[i] += s;
}

// Two waves per SIMD as in the forward kernel: the transposed weights (B operands) live in LDS, [step][lane][8] with
// (W1^T, W0^T, Wlc^T block 0, 1, 2) per lane, fetched as one ds_read_b128 + one ds_read_b32 per MFMA group.
template <bool FUSED>
__global__ void __launch_bounds__(512) tr_layer_bwd2_kernel(LayerBwdArgs a)
{
    __shared__ __attribute__((aligned(16))) float bt2[32 * 64 * 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 31, hh = lane >> 5;
    for (int e = threadIdx.x; e < 32 * 64; e += 512) {
        const int s_ = e >> 6, l = e & 63, nn = l & 31, h2 = l >> 5;
        const int k = 8 * (s_ >> 2) + 4 * h2 + (s_ & 3);
        float* q = &bt2[e * 8];
        q[0] = a.W1[nn * 64 + k]; q[1] = a.W0[nn * 64 + k];
        q[2] = a.Wlc[nn * 64 + k]; q[3] = a.Wlc[(32 + nn) * 64 + k];
        q[4] = nn < 16 ? a.Wlc[(64 + nn) * 64 + k] : 0.0f;
    }
    __syncthreads();
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int ntiles = a.B * a.tpb, nwaves = gridDim.x * 8;
    // (descriptors are the FUSED path's: twv_wavenet_train_create bounds the sizes to 31 bits only for that path; the pointer path takes any size)
    const rsrc_t rpre = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dPRE), 0, FUSED ? (int)((long long)a.B * a.Tn * 64 * 4) : 0, 0x00020000);
    const rsrc_t rdxn = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dXn), 0, FUSED ? (int)((long long)a.B * a.Tn * 32 * 4) : 0, 0x00020000);
    const rsrc_t rdx = __builtin_amdgcn_make_buffer_rsrc(a.dX, 0, FUSED ? (int)((long long)a.B * a.Tn * 32 * 4) : 0, 0x00020000);
    const rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(FUSED ? a.D : a.dX, 0, FUSED ? (int)((long long)a.B * a.T * 4 * 4) : 0, 0x00020000);
    for (int tile = blockIdx.x * 8 + wave; tile < ntiles; tile += nwaves) {
        const int b = tile / a.tpb, t0 = (tile - b * a.tpb) * 32 + a.t_lo;
        const int t = t0 + (lane & 31);
        f32x4t qa[8], qb[8];
        if (FUSED) {
            // no branch around a memory instruction (see tr_layer_fwdc_kernel): out-of-range rows read 0 through the descriptor
            const unsigned p = (unsigned)((((long long)b * a.Tn + t) * 64 + 4 * hh) * 4);
            const unsigned pa_ = t < a.Tn ? p : 0x80000000u, pb_ = t + a.d < a.Tn ? p + (unsigned)a.d * 256u : 0x80000000u;   // (+ 32 i stays out of range)
#pragma unroll
            for (int i = 0; i < 8; ++i) { qa[i] = tr_bld4(rpre, pa_ + 32 * i); qb[i] = tr_bld4(rpre, pb_ + 32 * i); }
        } else {
            const float* p = a.dPRE + ((long long)b * a.Tn + t) * 64 + 4 * hh;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                qa[i] = tr_ld4(p + 8 * i, t < a.Tn);
                qb[i] = tr_ld4(p + (long long)a.d * 64 + 8 * i, t + a.d < a.Tn);
            }
        }
        // operands of the read-modify-writes in the output (C) layout: requested before the MFMAs.  Interior tiles (all rows inside
        // the utterance and above the layer's receptive offset) use per-lane pointers + compile-time row offsets, no predicates.
        const bool interior = !FUSED && t0 >= a.o && t0 + 32 <= a.Tn;
        const long long lrow = ((long long)b * a.Tn + t0 + 4 * hh) * 32 + n;
        float* up = a.dU + ((long long)b * a.T + (t0 - a.o) + 4 * hh) * 80 + n;
        float rx[16], r0[16], r1[16], r2[16];
        if (FUSED) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = (r & 3) + 8 * (r >> 2);
                rx[r] = load_f32_b(rdxn, t0 + ro + 4 * hh < a.Tn ? (int)((lrow + ro * 32) * 4) : -1, 0);
            }
        } else if (interior) {
            const float* xp = a.dXn + lrow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = (r & 3) + 8 * (r >> 2);
                rx[r] = xp[ro * 32];
                if (!FUSED) { r0[r] = up[ro * 80]; r1[r] = up[ro * 80 + 32]; r2[r] = n < 16 ? up[ro * 80 + 64] : 0.0f; }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tt = t0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const bool in = tt < a.Tn, v = in && tt >= a.o;
                rx[r] = in ? a.dXn[((long long)b * a.Tn + tt) * 32 + n] : 0.0f;
                if (!FUSED) {
                    const float* ur = a.dU + ((long long)b * a.T + (tt - a.o)) * 80;
                    r0[r] = v ? ur[n] : 0.0f; r1[r] = v ? ur[32 + n] : 0.0f; r2[r] = (v && n < 16) ? ur[64 + n] : 0.0f;
                }
            }
        }
        if (FUSED) {
            // D[u][j] += dPRE[row] . Q_j[frame(u)] over the 64 columns: this lane holds 32 of them (row = lane & 31, columns 8i + 4hh + 0..3)
            const int u = t - a.o;
            const bool v = t < a.Tn && u >= 0;
            const int f = v ? u / a.hop : 0;
            const float* qr = a.Q + (((long long)b * a.F + f) * 4) * 64 + 4 * hh;
            float dj[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x4t q4 = *reinterpret_cast<const f32x4t*>(qr + j * 64 + 8 * i);
                    dj[j] += (qa[i][0] * q4[0] + qa[i][1] * q4[1]) + (qa[i][2] * q4[2] + qa[i][3] * q4[3]);
                }
#pragma unroll
            for (int j = 0; j < 4; ++j) dj[j] += __shfl_xor(dj[j], 32);
            {
                const unsigned od = hh == 0 && v ? (unsigned)((((long long)b * a.T + u) * 4) * 4) : 0xFFFFFFFFu;
                f32x4t cur = tr_bld4(rd, od);
                cur[0] += dj[0]; cur[1] += dj[1]; cur[2] += dj[2]; cur[3] += dj[3];
                tr_bst4(rd, od, cur);
            }
        }
        f32x16 cx = zero, c0 = zero, c1 = zero, c2 = zero;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* wq = &bt2[((4 * i + j) * 64 + lane) * 8];
                const f32x4t w4 = *reinterpret_cast<const f32x4t*>(wq);
                const float w5 = wq[4];
                cx = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[i][j], w4[0], cx, 0, 0, 0);
                if (!FUSED) {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[i][j], w4[2], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[i][j], w4[3], c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[i][j], w5, c2, 0, 0, 0);
                }
                cx = __builtin_amdgcn_mfma_f32_32x32x2f32(qb[i][j], w4[1], cx, 0, 0, 0);
            }
        if (FUSED) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = (r & 3) + 8 * (r >> 2);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(rx[r] + cx[r]), rdx, t0 + ro + 4 * hh < a.Tn ? (int)((lrow + ro * 32) * 4) : -1, 0, 0);
            }
        } else if (interior) {
            float* dxp = a.dX + lrow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = (r & 3) + 8 * (r >> 2);
                dxp[ro * 32] = rx[r] + cx[r];
                if (!FUSED) {
                    up[ro * 80] = r0[r] + c0[r]; up[ro * 80 + 32] = r1[r] + c1[r];
                    if (n < 16) up[ro * 80 + 64] = r2[r] + c2[r];
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tt = t0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (tt < a.Tn) {
                    a.dX[((long long)b * a.Tn + tt) * 32 + n] = rx[r] + cx[r];
                    if (!FUSED && tt >= a.o) {
                        float* ur = a.dU + ((long long)b * a.T + (tt - a.o)) * 80;
                        ur[n] = r0[r] + c0[r]; ur[32 + n] = r1[r] + c1[r];
                        if (n < 16) ur[64 + n] = r2[r] + c2[r];
                    }
                }
            }
        }
    }
}

// The FUSED backward-2 layer with the dPRE tiles fetched as 1 KB instructions (lane l: the l-th float4 of four consecutive 256-byte rows)
// and turned into MFMA A operands (lane = row) through two wave-private 32 x 68 LDS patches -- the row loads of tr_layer_bwd2_kernel
// touch 32 lines per instruction (16 + 16 of them per tile, ~88 cycles of the CU's address path each; profiles/r03_train_layer_anatomy.txt).
// Same arithmetic as tr_layer_bwd2_kernel<true>.
__global__ void __launch_bounds__(512) tr_layer_bwd2c_kernel(LayerBwdArgs a)
{
    __shared__ __attribute__((aligned(16))) float btc[32 * 64 * 2];          // [step][lane][W1^T, W0^T]
    __shared__ __attribute__((aligned(16))) float pt[8][2][32 * 68];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 31, hh = lane >> 5;
    for (int e = threadIdx.x; e < 32 * 64; e += 512) {
        const int s_ = e >> 6, l = e & 63, nn = l & 31, h2 = l >> 5;
        const int k = 8 * (s_ >> 2) + 4 * h2 + (s_ & 3);
        btc[e * 2] = a.W1[nn * 64 + k]; btc[e * 2 + 1] = a.W0[nn * 64 + k];
    }
    __syncthreads();
    typedef float f32x2t __attribute__((ext_vector_type(2)));
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int ntiles = a.B * a.tpb, nwaves = gridDim.x * 8;
    const rsrc_t rpre = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dPRE), 0, (int)((long long)a.B * a.Tn * 64 * 4), 0x00020000);
    const rsrc_t rdxn = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dXn), 0, (int)((long long)a.B * a.Tn * 32 * 4), 0x00020000);
    const rsrc_t rdx = __builtin_amdgcn_make_buffer_rsrc(a.dX, 0, (int)((long long)a.B * a.Tn * 32 * 4), 0x00020000);
    const rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(a.D, 0, (int)((long long)a.B * a.T * 4 * 4), 0x00020000);
    float* pa = pt[wave][0]; float* pb = pt[wave][1];
    const int fr = lane >> 4, fq = (lane & 15) * 4;                           // coalesced fetch: row within the 4-row group, first column
    for (int tile = blockIdx.x * 8 + wave; tile < ntiles; tile += nwaves) {
        const int b = tile / a.tpb, t0 = (tile - b * a.tpb) * 32 + a.t_lo;
        const int t = t0 + (lane & 31);
        {
            f32x4t ga[8], gb[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int tt = t0 + 4 * k + fr;
                const unsigned o = (unsigned)((((long long)b * a.Tn + tt) * 64 + fq) * 4);
                ga[k] = tr_bld4(rpre, tt < a.Tn ? o : 0x80000000u);
                gb[k] = tr_bld4(rpre, tt + a.d < a.Tn ? o + (unsigned)a.d * 256u : 0x80000000u);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                *reinterpret_cast<f32x4t*>(pa + (4 * k + fr) * 68 + fq) = ga[k];
                *reinterpret_cast<f32x4t*>(pb + (4 * k + fr) * 68 + fq) = gb[k];
            }
        }
        const long long lrow = ((long long)b * a.Tn + t0 + 4 * hh) * 32 + n;
        float rx[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ro = (r & 3) + 8 * (r >> 2);
            rx[r] = load_f32_b(rdxn, t0 + ro + 4 * hh < a.Tn ? (int)((lrow + ro * 32) * 4) : -1, 0);
        }
        f32x4t qa[8], qb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            qa[i] = *reinterpret_cast<const f32x4t*>(pa + n * 68 + 8 * i + 4 * hh);
            qb[i] = *reinterpret_cast<const f32x4t*>(pb + n * 68 + 8 * i + 4 * hh);
        }
        {
            // D[u][j] += dPRE[row] . Q_j[frame(u)] over the 64 columns: this lane holds 32 of them (row = lane & 31, columns 8i + 4hh + 0..3)
            const int u = t - a.o;
            const bool v = t < a.Tn && u >= 0;
            const int f = v ? u / a.hop : 0;
            const float* qr = a.Q + (((long long)b * a.F + f) * 4) * 64 + 4 * hh;
            float dj[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x4t q4 = *reinterpret_cast<const f32x4t*>(qr + j * 64 + 8 * i);
                    dj[j] += (qa[i][0] * q4[0] + qa[i][1] * q4[1]) + (qa[i][2] * q4[2] + qa[i][3] * q4[3]);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) dj[j] += __shfl_xor(dj[j], 32);
            const unsigned od = hh == 0 && v ? (unsigned)((((long long)b * a.T + u) * 4) * 4) : 0xFFFFFFFFu;
            f32x4t cur = tr_bld4(rd, od);
            cur[0] += dj[0]; cur[1] += dj[1]; cur[2] += dj[2]; cur[3] += dj[3];
            tr_bst4(rd, od, cur);
        }
        f32x16 cx = zero;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2t w = *reinterpret_cast<const f32x2t*>(&btc[((4 * i + j) * 64 + lane) * 2]);
                cx = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[i][j], w[0], cx, 0, 0, 0);
                cx = __builtin_amdgcn_mfma_f32_32x32x2f32(qb[i][j], w[1], cx, 0, 0, 0);
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) {                                        // (rows through an LDS patch and 1 KB stores: measured, 76 against 74 us)
            const int ro = (r & 3) + 8 * (r >> 2);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(rx[r] + cx[r]), rdx, t0 + ro + 4 * hh < a.Tn ? (int)((lrow + ro * 32) * 4) : -1, 0, 0);
        }
    }
}

// ---- one-hot (mu-law) input model: model.py:257-271 (mu_law_encode -> one_hot -> causal conv of width 2 over Q channels) ----
// x0[(b,t)][j] = Wc[0][q[b,t-1]][j] + Wc[1][q[b,t]][j]   (a one-hot row times a kernel is a row gather); t = 0 is masked
__global__ void tr_onehot_causal_fwd_kernel(const float* Wc, const int32_t* q, float* x0, int B, int T, int Tn, int Q)
{
    const long long total = (long long)B * Tn * 32;
    GRID_STRIDE(i, total) {
        const int j = (int)(i & 31);
        const long long r = i >> 5;
        const int t = (int)(r % Tn), b = (int)(r / Tn);
        float v = 0.0f;
        if (t >= 1) {
            const int qa = q[(long long)b * T + t - 1], qb = q[(long long)b * T + t];
            v = Wc[(long long)qa * 32 + j] + Wc[((long long)Q + qb) * 32 + j];
        }
        x0[i] = v;
    }
}
// dWc partial tables: block-local LDS accumulation (ds_add_f32), one (2, Q, 32) table per block, reduced afterwards in a fixed order
__global__ __launch_bounds__(256) void tr_onehot_causal_bwd_kernel(const float* dx0, const int32_t* q, float* part, int B, int T, int Tn, int Q, long long rows_per_block)
{
    extern __shared__ float tab[];
    const int n = 2 * Q * 32;
    for (int i = threadIdx.x; i < n; i += 256) tab[i] = 0.0f;
    __syncthreads();
    const long long R = (long long)B * Tn;
    const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
    const int c = threadIdx.x & 31;
    for (long long r = r0 + (threadIdx.x >> 5); r < r1; r += 8) {
        const int t = (int)(r % Tn), b = (int)(r / Tn);
        if (t >= 1) {
            const float g = dx0[r * 32 + c];
            atomicAdd(&tab[(long long)q[(long long)b * T + t - 1] * 32 + c], g);
            atomicAdd(&tab[((long long)Q + q[(long long)b * T + t]) * 32 + c], g);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) part[(long long)blockIdx.x * n + i] = tab[i];
}
// model.py:293-296 softmax_cross_entropy_with_logits_v2(one-hot target) mean; one wave per row of Q logits; dy = (softmax - onehot)/count
__global__ __launch_bounds__(256) void tr_softmax_ce_kernel(const float* y, const int32_t* q, int B, int T, int ow, int rf, int Q, float inv_count,
                                                           float* row_loss, float* dy)
{
    const int lane = threadIdx.x & 63;
    const long long rows = (long long)B * ow;
    for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long long)gridDim.x * 4) {
        const int p = (int)(r % ow), b = (int)(r / ow);
        const int tgt = q[(long long)b * T + p + rf];
        const float* yr = y + r * Q;
        float m = -3.0e38f;
        for (int i = lane; i < Q; i += 64) m = fmaxf(m, yr[i]);
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float se = 0.0f;
        for (int i = lane; i < Q; i += 64) se += expf(yr[i] - m);
        for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o);
        const float lse = m + logf(se);
        for (int i = lane; i < Q; i += 64) dy[r * Q + i] = (expf(yr[i] - lse) - (i == tgt ? 1.0f : 0.0f)) * inv_count;
        if (lane == 0) row_loss[r] = (lse - yr[tgt]) * inv_count;
    }
}
// MoL variant writes per-row losses too; the scalar loss is their deterministic two-stage sum


// model.py:300-312 optional L2 on every non-bias variable: loss += strength * sum(w^2)/2, grad += strength * w
struct L2Layout { long long c_layer0, lstride, nl, bf, bg, bd, bs, D, R, S, c_b1, c_b2, O, ub; };
__device__ __forceinline__ bool tr_is_bias(long long i, const L2Layout& L)
{
    if (!L.ub) return false;
    if (i >= L.c_b1 && i < L.c_b1 + L.S) return true;
    if (i >= L.c_b2 && i < L.c_b2 + L.O) return true;
    if (i < L.c_layer0 || i >= L.c_layer0 + L.lstride * L.nl) return false;
    const long long q = (i - L.c_layer0) % L.lstride;
    return (q >= L.bf && q < L.bf + L.D) || (q >= L.bg && q < L.bg + L.D) || (q >= L.bd && q < L.bd + L.R) || (q >= L.bs && q < L.bs + L.S);
}
__global__ void tr_l2_kernel(const float* p, float* g, float* sq, long long n, float strength, L2Layout L)
{
    GRID_STRIDE(i, n) {
        const bool b = tr_is_bias(i, L);
        const float w = p[i];
        sq[i] = b ? 0.0f : 0.5f * w * w;
        if (!b) g[i] += strength * w;
    }
}
__global__ void tr_axpy1_kernel(float* loss, const float* l2sum, float strength) { if (threadIdx.x == 0 && blockIdx.x == 0) loss[0] += strength * l2sum[0]; }
// tf.clip_by_global_norm(gradients, clip_norm): g <- g * pre * clip / max(||g * pre||, clip)
__global__ void tr_square_kernel(const float* g, float* sq, long long n, float pre) { GRID_STRIDE(i, n) { const float v = g[i] * pre; sq[i] = v * v; } }
__global__ void tr_clip_scale_kernel(float* g, long long n, float pre, float clip, const float* norm2)
{
    const float nrm = sqrtf(norm2[0]);
    const float sc = pre * clip / (nrm > clip ? nrm : clip);
    GRID_STRIDE(i, n) g[i] *= sc;
}

// model.py:314-346 add_optimizer: AdamOptimizer(lr) with TF defaults, then ExponentialMovingAverage(decay).apply
__global__ void tr_adam_ema_kernel(float* p, const float* g, float* m, float* v, float* ema, long long n, float lr_t, float b1, float b2,
                                   float eps, float decay, float gscale)
{
    GRID_STRIDE(i, n) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float pi = p[i] - lr_t * mi / (sqrtf(vi) + eps);
        p[i] = pi;
        ema[i] = ema[i] - (1.0f - decay) * (ema[i] - pi);           // shadow -= (1 - decay) * (shadow - var)
    }
}

// ---------------------------------------------------------------------------------------------------------------
//  host
// ---------------------------------------------------------------------------------------------------------------
// row-major C[M,N] (ldc) = alpha * op(A)[M,K] * op(B)[K,N] + beta * C, op = transpose when t* is set (A stored [K,M] / B stored [N,K])
static rocblas_status gemm_rm(rocblas_handle h, bool tA, bool tB, int M, int N, int K, float alpha, const float* A, int lda,
                              const float* B, int ldb, float beta, float* C, int ldc)
{
    return rocblas_sgemm(h, tB ? rocblas_operation_transpose : rocblas_operation_none, tA ? rocblas_operation_transpose : rocblas_operation_none,
                         N, M, K, &alpha, B, ldb, A, lda, &beta, C, ldc);
}

// row-major C[M,N] (ldc) = A^T B over K rows, A stored [K, lda], B stored [K, ldb] (a weight gradient: tall K, small M x N).
// One rocBLAS call leaves most of the chip idle here (M*N/(256*64) output tiles for 256 CUs) -- the K range is cut into
// `nsplit` equal slabs computed as ONE strided-batched GEMM into `partials` [nsplit][M][N], and the slabs are added in slab
// order (deterministic).  K must be a multiple of nsplit.
__global__ void tr_splitk_reduce_kernel(const float* part, int nsplit, long long mn, int N, float* C, int ldc)
{
    GRID_STRIDE(i, mn) {
        float s = part[i];
        for (int p = 1; p < nsplit; ++p) s += part[(long long)p * mn + i];
        C[(i / N) * ldc + (i % N)] = s;
    }
}
static rocblas_status gemm_tn_splitk(rocblas_handle h, hipStream_t st, int M, int N, long long K, const float* A, int lda, const float* B, int ldb,
                                     float* C, int ldc, int nsplit, float* partials)
{
    const float one = 1.0f, zero = 0.0f;
    if (nsplit <= 1) return gemm_rm(h, true, false, M, N, (int)K, 1.f, A, lda, B, ldb, 0.f, C, ldc);
    const long long kc = K / nsplit;
    rocblas_status rs = rocblas_sgemm_strided_batched(h, rocblas_operation_none, rocblas_operation_transpose, N, M, (int)kc, &one,
                                                      B, ldb, kc * ldb, A, lda, kc * lda, &zero, partials, N, (long long)M * N, nsplit);
    if (rs != rocblas_status_success) return rs;
    const long long mn = (long long)M * N;
    const long long nb = (mn + 255) / 256;
    hipLaunchKernelGGL(tr_splitk_reduce_kernel, dim3((unsigned)(nb > 4096 ? 4096 : nb)), dim3(256), 0, st, partials, nsplit, mn, N, C, ldc);
    return rocblas_status_success;
}

extern "C" int twv_wavenet_train_create(const twv_wavenet_dims* dims, int batch, int n_samples, twv_wavenet_trainer** out)
{
    if (dims && dims->gc_channels > 0 && dims->gc_cardinality < 1)
        return twv_fail(TWV_E_UNSUPPORTED, "training needs global_condition_cardinality (the gc_embedding table is a trained variable; model.py:191-195)");
    if (!dims || !out || batch < 1) return twv_fail(TWV_E_INVALID, "bad argument");
    const twv_wavenet_dims& d = *dims;
    {
        // only the FUSED layer kernels (tr_layer_fwdc_kernel / tr_layer_bwd2c_kernel: three upsampling stages, 32 <= hop <= 512, 80 mel
        // channels -- the same test as fused_lc in twv_wavenet_train_loss_grad) address a layer's activations with 32-bit byte offsets;
        // the pointer kernels of every other model keep 64-bit addresses and take any size
        int hop = 1;
        for (int i = 0; i < d.n_upsample; ++i) hop *= d.upsample_factor[i];
        const bool fused_lc = d.n_upsample == 3 && hop >= 32 && hop <= 512 && d.lc_channels == 80;
        if (fused_lc && (long long)batch * n_samples * 64 * 4 >= (1LL << 31))
            return twv_fail(TWV_E_UNSUPPORTED, "batch x samples too large: the fused layer kernels address a layer's activations with 32-bit byte offsets (batch * samples < 8.3 M)");
    }
    if (d.residual_channels != 32 || d.dilation_channels != 32) return twv_fail(TWV_E_UNSUPPORTED, "residual/dilation channels must be 32");
    if (d.scalar_input && (d.out_channels % 3 || d.out_channels > 96)) return twv_fail(TWV_E_UNSUPPORTED, "out_channels must be 3*nr_mix <= 96");
    if (!d.scalar_input && (d.quantization_channels < 2 || d.quantization_channels > 512)) return twv_fail(TWV_E_UNSUPPORTED, "quantization_channels must be in [2, 512] for training");
    if (d.lc_channels != 80 || !d.gc_channels) return twv_fail(TWV_E_UNSUPPORTED, "the training step expects num_mels = 80 local and global conditioning (train_vocoder.py, hparams.py:30)");
    twv_wavenet_trainer* h = new twv_wavenet_trainer();
    h->d = d; h->B = batch; h->T = n_samples; h->Tn = n_samples - 1; h->NL = d.n_layers; h->S = d.skip_channels; h->O = d.scalar_input ? d.out_channels : d.quantization_channels;
    h->L = d.lc_channels; h->G = d.gc_channels; h->ifw = d.scalar_input ? d.initial_filter_width : 2;   // model.py:36-39
    h->hop = 1;
    for (int i = 0; i < d.n_upsample; ++i) h->hop *= d.upsample_factor[i];
    if (n_samples % h->hop) { delete h; return twv_fail(TWV_E_INVALID, "n_samples must be a multiple of the hop size (datafeeder_wavenet.py:41-47)"); }
    h->off[0] = h->ifw - 1;
    for (int l = 0; l < h->NL; ++l) h->off[l + 1] = h->off[l] + d.dilations[l];
    h->rf = h->off[h->NL] + 1;
    h->ow = n_samples - h->rf;                                        // model.py:135 output_width
    if (h->ow < 1) { delete h; return twv_fail(TWV_E_INVALID, "n_samples must exceed the receptive field"); }
    // canonical blob offsets (same order as weights.tensor_specs / the generation path)
    long long c = 0;
    const int R = 32, D = 32, ub = d.use_biases ? 1 : 0;
    h->c_causal = c; c += d.scalar_input ? (long long)h->ifw * R : 2LL * d.quantization_channels * R;
    h->c_gcemb = c; c += (long long)d.gc_cardinality * h->G;
    h->c_layer0 = c;
    long long q = 0;
    h->lo.wf = q; q += 2 * R * D; h->lo.bf = q; q += ub * D;
    h->lo.wg = q; q += 2 * R * D; h->lo.bg = q; q += ub * D;
    h->lo.gcf = q; q += (long long)h->G * D; h->lo.gcg = q; q += (long long)h->G * D;
    h->lo.lcf = q; q += (long long)h->L * D; h->lo.lcg = q; q += (long long)h->L * D;
    h->lo.wd = q; q += D * R; h->lo.bd = q; q += ub * R;
    h->lo.ws = q; q += (long long)D * h->S; h->lo.bs = q; q += (long long)ub * h->S;
    h->c_lstride = q; c += q * h->NL;
    h->c_w1 = c; c += (long long)h->S * h->S; h->c_b1 = c; c += (long long)ub * h->S;
    h->c_w2 = c; c += (long long)h->S * h->O; h->c_b2 = c; c += (long long)ub * h->O;
    for (int i = 0; i < d.n_upsample; ++i) { h->c_up[i] = c; c += (long long)d.upsample_factor[i] * 2; }
    h->nparams = c;
    h->blas = nullptr;
    h->ws_clean = nullptr;
    // workspace (generous upper bound of what loss_grad carves, each piece rounded up to 64 floats)
    const long long Rr = (long long)batch * h->Tn, RT = (long long)batch * n_samples, RO = (long long)batch * h->ow;
    long long f = 0;
    f += RT * h->L * 2;                         // upsample stages
    f += RT * h->L * 2;                         // dU ping/pong
    f += Rr * h->ifw + RT;                      // xunf / quantized input
    f += d.scalar_input ? 0 : 256LL * 2 * d.quantization_channels * 32;   // one-hot causal gradient partial tables
    f += Rr * 32 * (h->NL + 1);                 // X[l]
    f += Rr * 32 * 3 * h->NL;                   // TH, SG, Z per layer
    f += RO * 32 * h->NL * 2;                   // ZCall, dZCall
    f += Rr * 64 + RT * 64;                     // PRE / dPRE, LCP / dLCP
    f += Rr * 32 * 3;                           // dX ping/pong, dZ
    f += RO * h->S * 3;                         // SK/H1, C1/H2, dC1
    f += RO * h->O * 2;                         // Y, dY
    f += (long long)batch * 64 * (h->NL + 2) + (long long)batch * h->G * 2;
    f += 2 * ((long long)h->NL * (64LL * (64 + h->L + h->G) + 32LL * h->S));   // weight views + their gradients
    f += 512LL * 96 * 64 + 1024LL * 512;        // reduction partials
    f += (256LL * 11 * 1024 + 64 + (long long)batch * ((h->Tn + 31) / 32) * 96 + 64 + (long long)batch * 64 + 64 + 64) * h->NL + 1024 + 64;   // per layer: gradient slabs, per-tile column sums, dGCP
    f += 16LL * ((long long)h->NL * 32 > h->S ? (long long)h->NL * 32 : h->S) * h->S;   // split-K partials of the wide weight gradients
    f += RT * 4 + 64 + 4LL * 512 + 64;          // fused lc backward: per-row dot products, phase-table gradient
    f += ((long long)batch * ((h->Tn + 31) / 32) * 512 + 64 + RT / h->hop * 256 + 64) * h->NL;   // per-tile / per-frame sums of ctab * dPRE (dW_lc)
    f += 4LL * 512 + 64 + RT / h->hop * 4 * h->L + 64 + (RT / h->hop * 4 * 64 + 64) * h->NL;   // fused lc projection: tap table, shifted mel, per-layer frame projections
    f += 2048LL * 64 * 32 + 64;                 // conv1d_2's fused backward: per-wave shares of dW2
    f += 64 * 64;                               // rounding slack
    // twv_wavenet_train_l2's scratch (nparams squares + the partial sums) has a region of its OWN behind everything loss_grad carves: at
    // the start of the workspace it overwrote rows that the fused layer kernels rely on staying zero (the rows in front of a layer's
    // receptive offset are cleared once and never written again) -- harmless only while every gradient is finite
    h->l2_off = f;
    f += (h->nparams + 63) / 64 * 64 + 1024;
    h->ws_floats = f;
    *out = h;
    return TWV_OK;
}
extern "C" void twv_wavenet_train_destroy(twv_wavenet_trainer* h)
{
    if (h && h->blas) rocblas_destroy_handle(h->blas);
    delete h;
}
extern "C" int twv_wavenet_train_reset_workspace(twv_wavenet_trainer* h)
{
    if (!h) return twv_fail(TWV_E_INVALID, "null argument");
    h->ws_clean = nullptr;                                // the next twv_wavenet_train_loss_grad clears whatever workspace it is given
    return TWV_OK;
}
extern "C" size_t twv_wavenet_train_param_floats(const twv_wavenet_trainer* h) { return (size_t)h->nparams; }
extern "C" size_t twv_wavenet_train_workspace_bytes(const twv_wavenet_trainer* h) { return (size_t)h->ws_floats * 4; }
extern "C" int twv_wavenet_train_output_width(const twv_wavenet_trainer* h) { return h->ow; }

extern "C" int twv_wavenet_train_loss_grad(twv_wavenet_trainer* h, const float* params, const float* audio, const float* lc,
                                           const int32_t* gc_ids, void* workspace, float* loss, float* grads, void* stream)
{
    if (!h || !params || !audio || !lc || !gc_ids || !workspace || !loss || !grads) return twv_fail(TWV_E_INVALID, "null argument");
    hipStream_t st = (hipStream_t)stream;
    if (!h->blas) BLASCHK(rocblas_create_handle(&h->blas));
    BLASCHK(rocblas_set_stream(h->blas, st));
    BLASCHK(rocblas_set_pointer_mode(h->blas, rocblas_pointer_mode_host));
    rocblas_handle bl = h->blas;
    const twv_wavenet_dims& d = h->d;
    const int B = h->B, T = h->T, Tn = h->Tn, NL = h->NL, S = h->S, O = h->O, L = h->L, G = h->G, ow = h->ow, rf = h->rf, nr = O / 3;
    const long long Rr = (long long)B * Tn, RT = (long long)B * T, RO = (long long)B * ow;
    const bool ub = d.use_biases != 0;
    const float* P = params;
    float* Gd = grads;
    HIPCHK(hipMemsetAsync(grads, 0, (size_t)h->nparams * 4, st));
    HIPCHK(hipMemsetAsync(loss, 0, 4, st));
    // A workspace seen for the first time is cleared once: the layer kernels do not walk the tiles in front of a layer's receptive offset,
    // and masked rows of the tiles they do walk may read those rows as operands of products with zero (0 x NaN would not be zero).
    // "Seen" is the pointer of the last call: a caller that frees the buffer and is handed the same address again, or lets anything else
    // write into it between two steps, calls twv_wavenet_train_reset_workspace first (the workspace belongs to the trainer between steps).
    if (h->ws_clean != workspace) {
        HIPCHK(hipMemsetAsync(workspace, 0, (size_t)h->ws_floats * 4, st));
        h->ws_clean = workspace;
    }
    // ---- workspace carve
    float* w = (float*)workspace;
    auto take = [&](long long n) { float* p = w; w += (n + 63) / 64 * 64; return p; };
    float* ups[5]; long long upT[5];
    upT[0] = T / h->hop;
    for (int i = 0; i < d.n_upsample; ++i) upT[i + 1] = upT[i] * d.upsample_factor[i];
    ups[0] = const_cast<float*>(lc);
    for (int i = 1; i <= d.n_upsample; ++i) ups[i] = take((long long)B * upT[i] * L);
    float* U = ups[d.n_upsample];
    float* dUa = take(RT * L); float* dUb = take(RT * L);
    float* xunf = take(Rr * h->ifw);
    int32_t* qin = reinterpret_cast<int32_t*>(take(RT));
    float* ohpart = take(d.scalar_input ? 0 : 256LL * 2 * d.quantization_channels * 32);
    float** X = new float*[NL + 1];
    float **TH = new float*[NL], **SG = new float*[NL];
    for (int l = 0; l <= NL; ++l) X[l] = take(Rr * 32);
    for (int l = 0; l < NL; ++l) { TH[l] = take(Rr * 32); SG[l] = take(Rr * 32); }
    const int ZW = NL * 32;                                  // stacked skip input: ZC[(b,p)][l*32 + j]
    float* ZC = take(RO * ZW); float* dZC = take(RO * ZW);
    float* PRE = take(Rr * 64);
    float* dXa = take(Rr * 32); float* dXb = take(Rr * 32);
    float* SK = take(RO * S); float* C1 = take(RO * S); float* dS = take(RO * S);
    float* Y = take(RO * O); float* dY = take(RO * O);
    float* emb = take((long long)B * G); float* demb = take((long long)B * G);
    float* GCP = take((long long)B * 64 * NL); float* dGCP = take((long long)B * 64);
    const long long vstride = 64LL * (64 + L + G);
    float* WV = take(vstride * NL); float* WS = take((long long)ZW * S);       // weight views
    float* GV = take(vstride * NL); float* GS = take((long long)ZW * S);       // gradient views
    float* part = take(512LL * 96 * 64 + 1024LL * 512);
    const long long slab_ls = 256LL * GQ_N * 1024;
    float* slabs = take(slab_ls * NL);                       // every layer's gradient slabs: reduced in one launch after the layer loop
    float* zpage = take(1024);
    float* kpart = take(16LL * (ZW > S ? ZW : S) * S);       // split-K partials (dW1, dW2, stacked dWs)
    float* wpart = take(2048LL * 64 * 32);                   // tr_conv2_bwd_kernel: [2048 / CG chunks][S][32] = 2048 x 64 x 32 whatever S is
    const int F = T / h->hop;                                // mel frames per entry
    const bool fused_lc = d.n_upsample == 3 && h->hop >= 32 && h->hop <= 512 && L == 80;
    float* ctab = take(4LL * 512);
    float* melsh = take((long long)B * F * 4 * L);
    const long long q_ls = (long long)B * F * 4 * 64;
    float* Qall = take((q_ls + 64) * NL);
    float* Dbuf = take(RT * 4);                              // sum over layers of dPRE . Q_j per U row (fused lc backward)
    float* dctab = take(4LL * 512);
    const long long pt_ls = (long long)B * ((Tn + 31) / 32) * 512;
    float* PTall = take((pt_ls + 64) * NL);
    float* Rall = take((q_ls + 64) * NL);
    int nsplit = 1;
    for (int c = 2; c <= 16 && c <= B; ++c) if (B % c == 0) nsplit = c;   // slabs of whole batch entries: RO = B * ow rows
    HIPCHK(hipMemsetAsync(zpage, 0, 4096, st));
    HIPCHK(hipFuncSetAttribute((const void*)tr_layer_bwd1_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * BS_FLOATS * 4));
    HIPCHK(hipFuncSetAttribute((const void*)tr_layer_bwd1_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (4 * BS_FLOATS + 2048) * 4));
    HIPCHK(hipFuncSetAttribute((const void*)tr_skinny_nn_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    const long long tsum_ls = (long long)B * ((Tn + 31) / 32) * 96;     // layers back to back: the column sums below treat (layer, batch entry) as segments
    float* tsum = take(tsum_ls * NL);
    float* dGCPall = take((long long)B * 64 * NL);
    float* bsum64 = take(64LL * NL);
    int rc = TWV_OK;
#define K1(kern, n, ...) hipLaunchKernelGGL(kern, dim3(tg(n)), dim3(256), 0, st, __VA_ARGS__)
#define LP(l) (P + h->c_layer0 + (long long)(l) * h->c_lstride)
#define LG(l) (Gd + h->c_layer0 + (long long)(l) * h->c_lstride)
    // out[seg][c] = column sums of x (nseg slabs of `rows` rows); deterministic two-stage reduction
    auto colsum = [&](const float* x, long long rows, int C, int ldx, int nseg, float* out, int ldo) {
        int nchunk = (int)(rows / 256); const int cap = nseg > 8 ? 32 : 256; nchunk = nchunk < 1 ? 1 : (nchunk > cap ? cap : nchunk);
        hipLaunchKernelGGL(tr_colsum_partial_kernel, dim3(nchunk, (C + 63) / 64, nseg), dim3(256), 0, st, x, rows, C, ldx, nchunk, part);
        hipLaunchKernelGGL(tr_colsum_final_kernel, dim3((nseg * C + 63) / 64), dim3(256), 0, st, part, nchunk, C, nseg, out, ldo);
    };
    // dx <- relu'(y) * dx over (rows, C) and, with biases, out[c] = column sums of the result
    auto relu_bwd_colsum = [&](float* dx, const float* y, long long rows, int C, float* out) {
        int nchunk = (int)(rows / 256); nchunk = nchunk < 1 ? 1 : (nchunk > 256 ? 256 : nchunk);
        hipLaunchKernelGGL(tr_relu_bwd_colsum_kernel, dim3(nchunk, (C + 63) / 64, 1), dim3(256), 0, st, dx, y, rows, C, nchunk, part);
        hipLaunchKernelGGL(tr_colsum_final_kernel, dim3((C + 63) / 64), dim3(256), 0, st, part, nchunk, C, 1, out, C);
    };
    // C[M x N](ldc) = A^T B over K rows (tall-skinny weight gradient); optional second output for columns [N, 2N) (filter|gate split)
    auto wgrad = [&](const float* A, int lda, const float* Bm, int ldb, long long K, int M, int N, float* C, int ldc) {
        const int mb = (M + 31) / 32, nb = (N + 31) / 32;
        const long long cap = (512LL * 96 * 64 + 1024LL * 512) / ((long long)mb * nb * 1024);     // `part` holds chunks x (mb*32) x (nb*32)
        // enough workgroups to fill the chip whatever the tile count: a single-tile product (the causal kernel's gradient: M = N = 32) ran on
        // 256 workgroups of dependent two-row steps: 124 us for 128 MB
        int chunks = (int)(K / 512); chunks = chunks < 1 ? 1 : (chunks > 1024 ? 1024 : chunks);
        if (chunks * mb * nb > 4096) chunks = 4096 / (mb * nb) > 0 ? 4096 / (mb * nb) : 1;
        if (chunks > cap) chunks = (int)cap;
        long long rpc = (K + chunks - 1) / chunks; rpc = (rpc + 7) / 8 * 8;
        chunks = (int)((K + rpc - 1) / rpc);
        hipLaunchKernelGGL(tr_tn_partial_kernel, dim3(chunks, mb, nb), dim3(256), 0, st, A, lda, Bm, ldb, K, M, N, rpc, part);
        hipLaunchKernelGGL(tr_tn_reduce_kernel, dim3((M * N + 63) / 64), dim3(chunks > 64 ? 1024 : 256), 0, st, part, chunks, mb * 32, nb * 32, M, N, C, ldc);
    };
    do {
        // ================= forward =================
        K1(tr_views_kernel, (vstride + 32LL * S) * NL, const_cast<float*>(P), WV, WS, NL, h->c_layer0, h->c_lstride, h->lo.wf, h->lo.wg, h->lo.lcf, h->lo.lcg,
           h->lo.gcf, h->lo.gcg, h->lo.ws, L, G, S, 0);
        for (int i = 0; i < d.n_upsample && !fused_lc; ++i)     // model.py:276 create_upsample (the fused lc path never materialises it)
            K1(tr_up_fwd_kernel, (long long)B * upT[i + 1] * L, P + h->c_up[i], ups[i], ups[i + 1], (long long)B * upT[i + 1] * L, d.upsample_factor[i], L);
        K1(tr_gather_emb_kernel, (long long)B * G, P + h->c_gcemb, gc_ids, emb, B, G);   // model.py:197-198
        if (d.scalar_input) {
            K1(tr_unfold_kernel, Rr * h->ifw, audio, xunf, B, T, Tn, h->ifw);
            if ((rc = gemm_rm(bl, false, false, (int)Rr, 32, h->ifw, 1.f, xunf, h->ifw, P + h->c_causal, 32, 0.f, X[0], 32))) break;   // model.py:131
        } else {
            if ((rc = twv_mu_law_encode(audio, RT, d.quantization_channels, qin, st))) break;                                         // model.py:257
            K1(tr_onehot_causal_fwd_kernel, Rr * 32, P + h->c_causal, qin, X[0], B, T, Tn, d.quantization_channels);
        }
        if (fused_lc) {
            // per step: the hop's tap table from the three upsampling kernels, the bin-shifted mel frames, and every layer's frame-rate
            // lc projections Q_l = melsh (B*F*4 x 80) . Wlc_l (80 x 64) as one strided-batched GEMM
            K1(tr_ctab_kernel, h->hop, P + h->c_up[0], P + h->c_up[1], P + h->c_up[2], d.upsample_factor[0], d.upsample_factor[1], d.upsample_factor[2], ctab);
            K1(tr_mel_shift_kernel, (long long)B * F * 4 * L, lc, melsh, (long long)B * F, L);
            const float one = 1.0f, zero = 0.0f;
            rc |= rocblas_sgemm_strided_batched(bl, rocblas_operation_none, rocblas_operation_none, 64, B * F * 4, L, &one, WV + 64 * 64, 64, vstride,
                                                melsh, L, 0, &zero, Qall, 64, q_ls + 64, NL);
            if (rc) break;
        }
        {   // model.py:71-73 gc projections of every layer: GCP[l] (B x 64) = emb (B x G) . Wgc_l (G x 64)
            const float one = 1.0f, zero = 0.0f;
            rc |= rocblas_sgemm_strided_batched(bl, rocblas_operation_none, rocblas_operation_none, 64, B, G, &one, WV + (64 + L) * 64, 64, vstride,
                                                emb, G, 0, &zero, GCP, 64, (long long)B * 64, NL);
            if (rc) break;
        }
        for (int l = 0; l < NL && !rc; ++l) {
            const int dl = d.dilations[l], o = h->off[l + 1];
            const float* Lp = LP(l);
            const float* Wv = WV + l * vstride;                 // tap0 | tap1 | lc | gc views, 64 columns = filter|gate
            // model.py:71-73 gc projection (tiny), then the fused layer: conv_filter | conv_gate taps, lc projection, gated unit,
            // dense + residual, skip input slice
            float* gcp = GCP + (long long)l * B * 64;                // all layers' projections: one strided-batched GEMM before the loop
            LayerFwdArgs fa;
            fa.X = X[l]; fa.U = U; fa.gcp = gcp; fa.W0 = Wv; fa.W1 = Wv + 32 * 64; fa.Wlc = Wv + 64 * 64; fa.Wd = Lp + h->lo.wd;
            fa.bf = ub ? Lp + h->lo.bf : nullptr; fa.bg = ub ? Lp + h->lo.bg : nullptr; fa.bd = ub ? Lp + h->lo.bd : nullptr;
            fa.TH = TH[l]; fa.SG = SG[l]; fa.XN = X[l + 1]; fa.ZC = ZC + l * 32;
            fa.B = B; fa.T = T; fa.Tn = Tn; fa.d = dl; fa.o = o; fa.ow = ow; fa.ldz = ZW; fa.tpb = (Tn + 31) / 32;
            // Rows t < o of this layer's output do not exist in the reference ('valid' convolutions, model.py:71-96: every layer is d
            // samples shorter than its input); in the right-aligned layout they are masked rows -- and whole 32-row tiles of them are
            // not walked at all (o reaches 3100 of 7800 rows in layer 29: 16 % of the stack's tiles).  Nothing reads them: a reader's
            // valid rows use t and t - d' with t >= o + d'; its masked rows may see them (zeros: the workspace is cleared once).
            fa.t_lo = fused_lc ? (o / 32) * 32 : 0;
            fa.tpb -= fa.t_lo / 32;
            {
                const int ntiles = B * fa.tpb;
                int nwg = (ntiles + 7) / 8; nwg = nwg > 256 ? 256 : nwg;      // one workgroup (8 waves, 2 per SIMD) per CU, each wave walks its tiles
                int nwgc = (ntiles + kFwdcWaves - 1) / kFwdcWaves; nwgc = nwgc > 256 ? 256 : nwgc;
                fa.Q = Qall + l * (q_ls + 64); fa.ctab = ctab; fa.hop = h->hop; fa.F = F;
                if (fused_lc) hipLaunchKernelGGL(tr_layer_fwdc_kernel, dim3(nwgc), dim3(kFwdcWaves * 64), 0, st, fa);
                else hipLaunchKernelGGL(tr_layer_fwd_kernel<false>, dim3(nwg), dim3(512), 0, st, fa);
            }
        }
        if (rc) break;
        // model.py:94-96,150-165: sum over layers of the skip 1x1 convs == ONE GEMM against the stacked skip kernels, then
        // (+ all skip biases) relu -> 1x1 -> relu -> 1x1
        if ((rc = gemm_rm(bl, false, false, (int)RO, S, ZW, 1.f, ZC, ZW, WS, S, 0.f, SK, S))) break;
        if ((S & 3) == 0) {
            float* bsum = part;                                   // S floats; `part` is free between the column sums
            if (ub) K1(tr_bias_sum_kernel, S, LP(0) + h->lo.bs, NL, h->c_lstride, bsum, S);
            K1(tr_bias_relu4_kernel, RO * S / 4, (float4*)SK, ub ? (const float4*)bsum : nullptr, S / 4, RO * S / 4);
        } else {
            K1(tr_bias_relu_kernel, RO * S, SK, ub ? LP(0) + h->lo.bs : nullptr, ub ? NL : 0, h->c_lstride, nullptr, S, RO * S);
        }
        if ((rc = gemm_rm(bl, false, false, (int)RO, S, S, 1.f, SK, S, P + h->c_w1, S, 0.f, C1, S))) break;
        // conv1d_1's activation: with the skinny output projection (MoL head) the relu(. + b1) pass over the (RO, S) array is never run --
        // its readers (the projection forward; conv1d_2's fused backward: mask of dC1 and operand of dW2) apply it to what they load
        const bool skinny = O <= 32 && (S & 63) == 0 && S * 32 * 4 <= 64 * 1024 && RO * S * 4 < (1LL << 31);
        const bool fuse_c2 = skinny && (S & 63) == 0 && 2048 % (S >> 6) == 0;   // (`part`: 2048 x 64 floats)
        const bool c1_raw = fuse_c2;
        const float* b1p = ub ? P + h->c_b1 : nullptr;
        if (!c1_raw) {
            if ((S & 3) == 0 && (h->c_b1 & 3) == 0) K1(tr_bias_relu4_kernel, RO * S / 4, (float4*)C1, ub ? (const float4*)(P + h->c_b1) : nullptr, S / 4, RO * S / 4);
            else K1(tr_bias_relu_kernel, RO * S, C1, nullptr, 0, 0, ub ? P + h->c_b1 : nullptr, S, RO * S);
        }
        if (skinny) {
            if (c1_raw) hipLaunchKernelGGL(tr_skinny_nn_kernel<true>, dim3(1024), dim3(256), (size_t)S * 33 * 4, st, C1, S, P + h->c_w2, O, ub ? P + h->c_b2 : nullptr, RO, S, O, Y, O, b1p);
            else hipLaunchKernelGGL(tr_skinny_nn_kernel<false>, dim3(1024), dim3(256), (size_t)S * 32 * 4, st, C1, S, P + h->c_w2, O, ub ? P + h->c_b2 : nullptr, RO, S, O, Y, O, (const float*)nullptr);
        } else {
            if ((rc = gemm_rm(bl, false, false, (int)RO, O, S, 1.f, C1, S, P + h->c_w2, O, 0.f, Y, O))) break;
            if (ub) K1(tr_bias_add_kernel, RO * O, Y, P + h->c_b2, O, RO * O);
        }
        // model.py:286-290 loss
        // per-row terms, then the same two-stage column sum as the bias gradients: the loss is bit-reproducible run to run
        float* row_loss = dS;                                       // (RO) scratch: dS is not written before the backward pass
        if (d.scalar_input && nr == 10) K1(tr_mol_loss_kernel<10>, RO, Y, audio, B, T, ow, rf, nr, 1.0f / (float)RO, row_loss, dY);
        else if (d.scalar_input) K1(tr_mol_loss_kernel<0>, RO, Y, audio, B, T, ow, rf, nr, 1.0f / (float)RO, row_loss, dY);
        else hipLaunchKernelGGL(tr_softmax_ce_kernel, dim3(tg(RO * 64)), dim3(256), 0, st, Y, qin, B, T, ow, rf, O, 1.0f / (float)RO, row_loss, dY);
        colsum(row_loss, RO, 1, 1, 1, loss, 1);
        // ================= backward =================
        if (!fuse_c2) wgrad(C1, S, dY, O, RO, S, O, Gd + h->c_w2, O);                                              // dW2 = H2^T dY (the tall-skinny MFMA kernel)
        if (ub) colsum(dY, RO, O, O, 1, Gd + h->c_b2, O);
        if (fuse_c2) {                                                                                           // dC1, db1 and dW2: one pass
            const int nwg = 512, nchunk = nwg * 4 / (S >> 6);             // two workgroups per CU (171 registers; at three, 168 + spills: 276 vs 263 us)
            hipLaunchKernelGGL(tr_conv2_bwd_kernel, dim3(nwg), dim3(256), 0, st, dY, O, P + h->c_w2, C1, b1p, RO, S, dS, part, wpart);
            if (ub) hipLaunchKernelGGL(tr_colsum_final_kernel, dim3((S + 63) / 64), dim3(256), 0, st, part, nchunk, S, 1, Gd + h->c_b1, S);
            hipLaunchKernelGGL(tr_tn_reduce_kernel, dim3((S * O + 63) / 64), dim3(1024), 0, st, wpart, nchunk, S, 32, S, O, Gd + h->c_w2, O);
        } else {
            rc |= gemm_rm(bl, false, true, (int)RO, S, O, 1.f, dY, O, P + h->c_w2, O, 0.f, dS, S);               // dH2
            if (ub) relu_bwd_colsum(dS, C1, RO, S, Gd + h->c_b1);                                                // dC1, db1
            else if ((S & 3) == 0) K1(tr_relu_bwd4_kernel, RO * S / 4, (float4*)dS, (const float4*)C1, RO * S / 4);
            else K1(tr_relu_bwd_kernel, RO * S, dS, C1, RO * S);
        }
        rc |= gemm_tn_splitk(bl, st, S, S, RO, SK, S, dS, S, Gd + h->c_w1, S, nsplit, kpart);                     // dW1 = H1^T dC1
        rc |= gemm_rm(bl, false, true, (int)RO, S, S, 1.f, dS, S, P + h->c_w1, S, 0.f, C1, S);                   // dH1 -> C1 buffer
        if (ub) relu_bwd_colsum(C1, SK, RO, S, LG(0) + h->lo.bs);                                                 // dSK, dbs (layer 0's slot)
        else if ((S & 3) == 0) K1(tr_relu_bwd4_kernel, RO * S / 4, (float4*)C1, (const float4*)SK, RO * S / 4);
        else K1(tr_relu_bwd_kernel, RO * S, C1, SK, RO * S);
        float* dSK = C1;
        // all skip convs at once: dWs (stacked) = ZC^T dSK ; dZC = dSK WS^T ; dbs (identical for every layer) = colsum(dSK)
        rc |= gemm_tn_splitk(bl, st, ZW, S, RO, ZC, ZW, dSK, S, GS, S, nsplit, kpart);
        if (ub) {
            if (NL > 1) K1(tr_bcast_rows_kernel, (long long)(NL - 1) * S, LG(0) + h->lo.bs, h->c_lstride, NL, S);
        }
        rc |= gemm_rm(bl, false, true, (int)RO, ZW, S, 1.f, dSK, S, WS, S, 0.f, dZC, ZW);
        if (rc) break;
        float* dXn = dXa; float* dXc = dXb;
        K1(tr_fill_kernel, Rr * 32, dXn, 0.0f, Rr * 32);
        if (fused_lc) K1(tr_fill_kernel, Rr * 32, dXc, 0.0f, Rr * 32);   // (rows in front of a layer's input offset are not written any more)
        if (fused_lc) K1(tr_fill_kernel, RT * 4, Dbuf, 0.0f, RT * 4);
        else K1(tr_fill_kernel, RT * L, dUa, 0.0f, RT * L);
        K1(tr_fill_kernel, (long long)B * G, demb, 0.0f, (long long)B * G);
        for (int l = NL - 1; l >= 0 && !rc; --l) {
            const int dl = d.dilations[l], o = h->off[l + 1];
            const float* Lp = LP(l);
            float* Lg = LG(l);
            const float* Wv = WV + l * vstride;
            float* Gv = GV + l * vstride;
            LayerBwdArgs ba;
            ba.dXn = dXn; ba.dZC = dZC + l * 32; ba.TH = TH[l]; ba.SG = SG[l]; ba.X = X[l]; ba.U = U;
            ba.W0 = Wv; ba.W1 = Wv + 32 * 64; ba.Wlc = Wv + 64 * 64; ba.Wd = Lp + h->lo.wd;
            ba.dPRE = PRE; ba.slabs = slabs + l * slab_ls; ba.tsum = tsum + l * tsum_ls; ba.dX = dXc; ba.dU = dUa;
            ba.B = B; ba.T = T; ba.Tn = Tn; ba.d = dl; ba.o = o; ba.ow = ow; ba.ldz = ZW; ba.tpb = (Tn + 31) / 32;
            // backward: dPRE is zero in front of o, dX in front of o - d (the layer input's offset).  Both kernels walk the tiles from
            // row o - d on: bwd1 leaves zeros in dPRE rows [o - d, o) for bwd2's second tap, and the rows in front of o - d of the dX
            // buffer it fills are the zeros of the start of the pass (no layer above writes them: their offsets are larger)
            ba.tpbf = ba.tpb;
            ba.t_lo = fused_lc ? ((o - dl > 0 ? o - dl : 0) / 32) * 32 : 0;
            ba.tpb -= ba.t_lo / 32;
            const int ntiles = B * ba.tpbf, ntiles_w = B * ba.tpb;     // (bwd1's grid = its slab count stays that of all tiles: one reduction for all layers)
            int nwg = (ntiles + 3) / 4; nwg = nwg > 256 ? 256 : nwg;
            int nwg2 = (ntiles_w + 7) / 8; nwg2 = nwg2 > 256 ? 256 : nwg2;
            ba.zeros = zpage;
            ba.Q = Qall + l * (q_ls + 64); ba.D = Dbuf; ba.hop = h->hop; ba.F = F; ba.ctab = ctab; ba.PT = PTall + l * (pt_ls + 64);
            if (fused_lc) hipLaunchKernelGGL(tr_layer_bwd1_kernel<true>, dim3(nwg), dim3(256), (4 * BS_FLOATS + 2048) * 4, st, ba);
            else hipLaunchKernelGGL(tr_layer_bwd1_kernel<false>, dim3(nwg), dim3(256), 4 * BS_FLOATS * 4, st, ba);
            if (fused_lc) hipLaunchKernelGGL(tr_layer_bwd2c_kernel, dim3(nwg2), dim3(512), 0, st, ba);
            else hipLaunchKernelGGL(tr_layer_bwd2_kernel<false>, dim3(nwg2), dim3(512), 0, st, ba);
            float* tsw = dXn; dXn = dXc; dXc = tsw;
        }
        if (rc) break;
        // ---- all layers at once: gradient tiles -> views / canonical slots; gc and bias column sums; dWgc; demb
        {
            const int tpb = (Tn + 31) / 32, ntiles = B * tpb;
            int nwg = (ntiles + 3) / 4; nwg = nwg > 256 ? 256 : nwg;
            SlabDst sd;
            auto put = [&](int q, int mrows, float* out, int ldo, long long ls) { sd.out[q] = out; sd.ldo[q] = ldo; sd.mrows[q] = mrows; sd.lstride[q] = ls; };
            float* Gv = GV;
            put(GQ_W0F, 32, Gv, 64, vstride); put(GQ_W0G, 32, Gv + 32, 64, vstride);
            put(GQ_W1F, 32, Gv + 32 * 64, 64, vstride); put(GQ_W1G, 32, Gv + 32 * 64 + 32, 64, vstride);
            put(GQ_LCF0, 32, Gv + 64 * 64, 64, vstride); put(GQ_LCG0, 32, Gv + 64 * 64 + 32, 64, vstride);
            put(GQ_LCF1, 32, Gv + 96 * 64, 64, vstride); put(GQ_LCG1, 32, Gv + 96 * 64 + 32, 64, vstride);
            put(GQ_LCF2, L - 64, Gv + 128 * 64, 64, vstride); put(GQ_LCG2, L - 64, Gv + 128 * 64 + 32, 64, vstride);
            put(GQ_WD, 32, LG(0) + h->lo.wd, 32, h->c_lstride);
            if (fused_lc) {
                for (int q = GQ_LCF0; q <= GQ_LCG2; ++q) sd.mrows[q] = 0;       // W_lc's gradient: per-frame sums -> one strided-batched GEMM
                LayerOffs lo_;
                for (int l = 0; l < NL; ++l) lo_.o[l] = h->off[l + 1];
                const long long nthr = (long long)B * F * 256;
                hipLaunchKernelGGL(tr_frame_reduce_kernel, dim3((unsigned)((nthr + 255) / 256), 1, NL), dim3(256), 0, st, PTall, pt_ls + 64, lo_, B, F, h->hop, tpb, Tn,
                                   Rall, q_ls + 64);
                const float one = 1.0f, zero = 0.0f;
                rc |= rocblas_sgemm_strided_batched(bl, rocblas_operation_none, rocblas_operation_transpose, 64, L, B * F * 4, &one, Rall, 64, q_ls + 64,
                                                    melsh, L, 0, &zero, GV + 64 * 64, 64, vstride, NL);
            }
            hipLaunchKernelGGL(tr_slab_reduce_kernel, dim3(16, GQ_N, NL), dim3(256), 0, st, slabs, slab_ls, nwg, sd);
            // gc: dGCP[l][b] = sum_t dPRE[b,t] (from the per-tile sums); conv biases = sum_b dGCP[l][b]; dense bias = column sums of the tiles
            colsum(tsum, tpb, 64, 96, NL * B, dGCPall, 64);                                   // segment = (layer, batch entry)
            if (ub) {
                colsum(tsum + 64, ntiles, 32, 96, NL, LG(0) + h->lo.bd, (int)h->c_lstride);   // segment = layer, outputs c_lstride apart
                colsum(dGCPall, B, 64, 64, NL, bsum64, 64);
                K1(tr_split_bias_kernel, 64LL * NL, bsum64, LG(0), h->c_lstride, h->lo.bf, h->lo.bg, NL);
            }
            // dWgc_l = emb^T dGCP_l (one strided-batched GEMM over the layers) ; demb += sum_l dGCP_l Wgc_l^T
            {
                const float one = 1.0f, zero = 0.0f;
                rc |= rocblas_sgemm_strided_batched(bl, rocblas_operation_none, rocblas_operation_transpose, 64, G, B, &one, dGCPall, 64, (long long)B * 64,
                                                    emb, G, 0, &zero, GV + (64 + L) * 64, 64, vstride, NL);
            }
            hipLaunchKernelGGL(tr_demb_kernel, dim3((unsigned)(((long long)B * G + 3) / 4)), dim3(256), 0, st, dGCPall, WV + (64 + L) * 64, vstride, demb, NL, B, G);
        }
        if (rc) break;
        // causal layer, gc embedding table, upsampler, and the gradient views back into the canonical order
        if (d.scalar_input) wgrad(xunf, h->ifw, dXn, 32, Rr, h->ifw, 32, Gd + h->c_causal, 32);
        else {
            const int Q = d.quantization_channels, nblk = 256;
            const long long rpb = (Rr + nblk - 1) / nblk;
            HIPCHK(hipFuncSetAttribute((const void*)tr_onehot_causal_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * Q * 32 * 4));
            hipLaunchKernelGGL(tr_onehot_causal_bwd_kernel, dim3(nblk), dim3(256), 2 * Q * 32 * 4, st, dXn, qin, ohpart, B, T, Tn, Q, rpb);
            hipLaunchKernelGGL(tr_colsum_final_kernel, dim3((2 * Q * 32 + 63) / 64), dim3(256), 0, st, ohpart, nblk, 2 * Q * 32, 1, Gd + h->c_causal, 2 * Q * 32);
        }
        K1(tr_scatter_emb_kernel, (long long)d.gc_cardinality * G, demb, gc_ids, Gd + h->c_gcemb, B, G, d.gc_cardinality);
        K1(tr_views_kernel, (vstride + 32LL * S) * NL, Gd, GV, GS, NL, h->c_layer0, h->c_lstride, h->lo.wf, h->lo.wg, h->lo.lcf, h->lo.lcg,
           h->lo.gcf, h->lo.gcg, h->lo.ws, L, G, S, 1);
        if (fused_lc) {
            // upsampling kernels: D (per U row) -> dctab (per phase) -> the three (f, 2) kernels
            hipLaunchKernelGGL(tr_dctab_kernel, dim3(h->hop), dim3(256), 0, st, Dbuf, B, T, h->hop, dctab);
            K1(tr_up_grad_kernel, 2LL * (d.upsample_factor[0] + d.upsample_factor[1] + d.upsample_factor[2]), P + h->c_up[0], P + h->c_up[1], P + h->c_up[2],
               d.upsample_factor[0], d.upsample_factor[1], d.upsample_factor[2], dctab, Gd + h->c_up[0], Gd + h->c_up[1], Gd + h->c_up[2]);
        } else {
            float* dcur = dUa; float* dnxt = dUb;
            for (int i = d.n_upsample - 1; i >= 0; --i) {
                const long long rin = (long long)B * upT[i];
                int nch = (int)(rin / 64); nch = nch < 1 ? 1 : (nch > 256 ? 256 : nch);
                hipLaunchKernelGGL(tr_up_bwd_k_kernel, dim3(nch, d.upsample_factor[i]), dim3(256), 0, st, ups[i], dcur, part, rin, d.upsample_factor[i], L, nch);
                hipLaunchKernelGGL(tr_colsum_final_kernel, dim3((2 * d.upsample_factor[i] + 63) / 64), dim3(256), 0, st, part, nch, 2 * d.upsample_factor[i], 1,
                                   Gd + h->c_up[i], 2 * d.upsample_factor[i]);
                if (i > 0) {
                    K1(tr_up_bwd_in_kernel, (long long)B * upT[i] * L, P + h->c_up[i], dcur, dnxt, (long long)B * upT[i] * L, d.upsample_factor[i], L);
                    float* tsw = dcur; dcur = dnxt; dnxt = tsw;
                }
            }
        }
    } while (0);
    delete[] X; delete[] TH; delete[] SG;
    if (rc) return twv_fail(TWV_E_HIP, "rocBLAS call failed with status " + std::to_string(rc));
    HIPCHK(hipGetLastError());
    if ((long long)(w - (float*)workspace) > h->ws_floats) return twv_fail(TWV_E_INVALID, "internal: workspace overrun");
    return TWV_OK;
}


// deterministic two-stage sum of n floats -> out[0]
static void tr_sum(const float* x, long long n, float* part, float* out, hipStream_t st)
{
    int nchunk = (int)(n / 1024); nchunk = nchunk < 1 ? 1 : (nchunk > 256 ? 256 : nchunk);
    hipLaunchKernelGGL(tr_colsum_partial_kernel, dim3(nchunk, 1, 1), dim3(256), 0, st, x, n, 1, 1, nchunk, part);
    hipLaunchKernelGGL(tr_colsum_final_kernel, dim3(1), dim3(256), 0, st, part, nchunk, 1, 1, out, 1);
}
extern "C" int twv_wavenet_train_l2(twv_wavenet_trainer* h, const float* params, double strength, void* workspace, float* loss, float* grads, void* stream)
{
    if (!h || !params || !workspace || !loss || !grads) return twv_fail(TWV_E_INVALID, "null argument");
    hipStream_t st = (hipStream_t)stream;
    float* sq = (float*)workspace + h->l2_off;           // nparams floats, then 256 partials + the sum: a region of its own (see create)
    float* part = sq + (h->nparams + 63) / 64 * 64;
    L2Layout L{h->c_layer0, h->c_lstride, h->NL, h->lo.bf, h->lo.bg, h->lo.bd, h->lo.bs, 32, 32, h->S, h->c_b1, h->c_b2, h->O, h->d.use_biases};
    hipLaunchKernelGGL(tr_l2_kernel, dim3(tg(h->nparams)), dim3(256), 0, st, params, grads, sq, h->nparams, (float)strength, L);
    tr_sum(sq, h->nparams, part, part + 512, st);
    hipLaunchKernelGGL(tr_axpy1_kernel, dim3(1), dim3(64), 0, st, loss, part + 512, (float)strength);
    HIPCHK(hipGetLastError());
    return TWV_OK;
}
extern "C" int twv_clip_by_global_norm(float* grads, int64_t n, double pre_scale, double clip_norm, void* scratch, void* stream)
{
    if (!grads || !scratch || n < 1 || clip_norm <= 0) return twv_fail(TWV_E_INVALID, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    float* sq = (float*)scratch;                         // n floats, then 256 partials + the sum
    float* part = sq + (n + 63) / 64 * 64;
    hipLaunchKernelGGL(tr_square_kernel, dim3(tg(n)), dim3(256), 0, st, grads, sq, (long long)n, (float)pre_scale);
    tr_sum(sq, n, part, part + 512, st);
    hipLaunchKernelGGL(tr_clip_scale_kernel, dim3(tg(n)), dim3(256), 0, st, grads, (long long)n, (float)pre_scale, (float)clip_norm, part + 512);
    HIPCHK(hipGetLastError());
    return TWV_OK;
}

extern "C" int twv_adam_ema_step(float* params, const float* grads, float* m, float* v, float* ema, int64_t n, double lr, double beta1,
                                 double beta2, double eps, int64_t t, double ema_decay, double grad_scale, void* stream)
{
    if (!params || !grads || !m || !v || !ema || n < 0 || t < 1) return twv_fail(TWV_E_INVALID, "bad argument");
    // tf.train.AdamOptimizer: lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t)
    double b1t = 1.0, b2t = 1.0;
    b1t = pow(beta1, (double)t); b2t = pow(beta2, (double)t);      // beta^t in double (TF keeps running float products; the difference is below fp32 round-off of lr_t)
    const float lr_t = (float)(lr * sqrt(1.0 - b2t) / (1.0 - b1t));
    if (n) hipLaunchKernelGGL(tr_adam_ema_kernel, dim3(tg(n)), dim3(256), 0, (hipStream_t)stream, params, grads, m, v, ema, (long long)n, lr_t,
                              (float)beta1, (float)beta2, (float)eps, (float)ema_decay, (float)grad_scale);
    HIPCHK(hipGetLastError());
    return TWV_OK;
}
