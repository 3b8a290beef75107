// Own f32 MFMA GEMM for the training step's wide products (ubench / development harness):
//   C[M][N] = A[M][K] . Bt[N][K]^T   (A row-major with lda, Bt row-major with ldb: "NT" form; an NN product passes W^T)
// Workgroup = 4 waves on a 128 x 128 tile, wave (wr, wc) owns 64 x 64 = 2 x 2 v_mfma_f32_32x32x2_f32 blocks; K advances 16 per step
// through two LDS buffers (row stride 20 floats: the ds_read_b128 of eight consecutive rows hit disjoint bank groups).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
using rsrc_t = __amdgpu_buffer_rsrc_t;
#ifndef BK_
#define BK_ 32
#endif
constexpr int BM = 128, BN = 128, BK = BK_, LDT = BK + 4, NQ = BK / 4, NH = BM * NQ / 256;   // float4 per row, staging quads per thread

__device__ __forceinline__ f32x4 bld4(rsrc_t r, unsigned voff, unsigned soff)
{
    const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return f32x4{__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w)};
}

__global__ void __launch_bounds__(256, 2) sgemm_nt_kernel(const float* A, int lda, const float* Bt, int ldb, float* C, int ldc, int M, int N, int K)
{
#ifdef __HIP_DEVICE_COMPILE__
    __shared__ __attribute__((aligned(16))) float As[2][BM * LDT], Bs[2][BN * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    // global -> LDS staging: thread t moves float4 #t and #t+256 of the 128 x 16 tile (row = q / 4, k quad = q % 4)
    const rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (int)((((long long)M - 1) * lda + K) * 4), 0x00020000);
    const rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bt), 0, (int)((((long long)N - 1) * ldb + K) * 4), 0x00020000);
    unsigned ga[NH], gb[NH]; int ls[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const int q = tid + 256 * h, row = q / NQ, kq = q % NQ;
        ga[h] = m0 + row < M ? (unsigned)(((long long)(m0 + row) * lda + kq * 4) * 4) : 0x80000000u;
        gb[h] = n0 + row < N ? (unsigned)(((long long)(n0 + row) * ldb + kq * 4) * 4) : 0x80000000u;
        ls[h] = row * LDT + kq * 4;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int nk = K / BK;                                   // K % 16 == 0
    f32x4 sa[NH], sb[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) { sa[h] = bld4(ra, ga[h], 0); sb[h] = bld4(rb, gb[h], 0); }
#pragma unroll
    for (int h = 0; h < NH; ++h) { *reinterpret_cast<f32x4*>(&As[0][ls[h]]) = sa[h]; *reinterpret_cast<f32x4*>(&Bs[0][ls[h]]) = sb[h]; }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < NH; ++h) { const unsigned k1 = (unsigned)(nk > 1 ? 1 : 0) * BK * 4; sa[h] = bld4(ra, ga[h], k1); sb[h] = bld4(rb, gb[h], k1); }
    const int arow = (wr * 64 + (lane & 31)) * LDT + 4 * (lane >> 5), brow = (wc * 64 + (lane & 31)) * LDT + 4 * (lane >> 5);
    for (int s = 0; s < nk; ++s) {
        const int cur = s & 1;
        const unsigned ko = (unsigned)(s + 2 < nk ? s + 2 : s) * BK * 4;      // the operands of step s + 2 travel during steps s and s + 1
        f32x4 ta[NH], tb[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) { ta[h] = bld4(ra, ga[h], ko); tb[h] = bld4(rb, gb[h], ko); }
        __builtin_amdgcn_sched_barrier(0);                   // (the scheduler would sink the requests to their use, behind the MFMAs)
        f32x4 fa[BK / 8][2], fb[BK / 8][2];
#pragma unroll
        for (int i = 0; i < BK / 8; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                fa[i][b] = *reinterpret_cast<const f32x4*>(&As[cur][arow + b * 32 * LDT + 8 * i]);
                fb[i][b] = *reinterpret_cast<const f32x4*>(&Bs[cur][brow + b * 32 * LDT + 8 * i]);
            }
#pragma unroll
        for (int i = 0; i < BK / 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rb_ = 0; rb_ < 2; ++rb_)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
                        acc[rb_][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][rb_][j], fb[i][cb][j], acc[rb_][cb], 0, 0, 0);
#pragma unroll
        for (int h = 0; h < NH; ++h) { *reinterpret_cast<f32x4*>(&As[cur ^ 1][ls[h]]) = sa[h]; *reinterpret_cast<f32x4*>(&Bs[cur ^ 1][ls[h]]) = sb[h]; }
#pragma unroll
        for (int h = 0; h < NH; ++h) { sa[h] = ta[h]; sb[h] = tb[h]; }
        __syncthreads();
    }
    // C layout (lane = column (lane & 31), register r = row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) -> rows through a wave-private 32 x 36 LDS patch
    // -> 1 KB store instructions (lane l: the l-th float4 of eight consecutive 128-byte row segments)
    const rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(C, 0, (int)((((long long)M - 1) * ldc + N) * 4), 0x00020000);
    float* patch = &As[0][0] + wave * (32 * 36);             // the k loop's last barrier has passed: the staging buffers are free
    const int cr = lane >> 3, cc = (lane & 7) * 4;
#pragma unroll
    for (int rb_ = 0; rb_ < 2; ++rb_)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 36 + (lane & 31)] = acc[rb_][cb][r];
            const int mb = m0 + wr * 64 + rb_ * 32, nb = n0 + wc * 64 + cb * 32;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = 8 * k + cr;
                const f32x4 v = *reinterpret_cast<const f32x4*>(patch + row * 36 + cc);
                const bool ok = mb + row < M && nb + cc < N;
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, rc,
                                                       (int)(ok ? (unsigned)((((long long)(mb + row)) * ldc + nb + cc) * 4) : 0x80000000u), 0, 0);
            }
        }
#endif
}

// ---- v2: global -> LDS by LDS-DMA (global_load_lds b128: no VGPR round trip, no ds_write), three LDS buffers (the operands of step s + 2
// travel during steps s and s + 1), unpadded 64-byte rows with an XOR swizzle of the k quad (slot p of row r holds quad p ^ ((r >> 1) & 3)):
// the DMA writes linearly, the ds_read_b128 of eight consecutive rows hit eight disjoint 4-bank groups.
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
extern __shared__ __attribute__((aligned(16))) float dyn[];
__global__ void __launch_bounds__(256, 2) sgemm_nt_dma_kernel(const float* A, int lda, const float* Bt, int ldb, float* C, int ldc, int M, int N, int K)
{
#ifdef __HIP_DEVICE_COMPILE__
    constexpr int TB = 128 * 16;                              // floats per operand tile per step (8 KB)
    float* As = dyn; float* Bs = dyn + 3 * TB;                // [3][128][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wave >> 1, wc = wave & 1;
    const int n0 = blockIdx.y * 128;
    for (int mt = blockIdx.x; mt * 128 < M; mt += gridDim.x) {
    const int m0 = mt * 128;
    // DMA: wave w, instruction h (0, 1): slots (w * 2 + h) * 64 + lane of the 512-slot tile; slot s: row s / 4, position s % 4
    const float* pa[2]; const float* pb[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int sl = (wave * 2 + h) * 64 + lane, row = sl >> 2, q = (sl & 3) ^ ((row >> 1) & 3);
        const int ma = m0 + row < M ? m0 + row : M - 1, nb = n0 + row < N ? n0 + row : N - 1;
        pa[h] = A + (long long)ma * lda + 4 * q; pb[h] = Bt + (long long)nb * ldb + 4 * q;
    }
    auto dma = [&](int step, int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            __builtin_amdgcn_global_load_lds((gptr_t)(pa[h] + step * 16), (lptr_t)(As + buf * TB + (wave * 2 + h) * 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(pb[h] + step * 16), (lptr_t)(Bs + buf * TB + (wave * 2 + h) * 256), 16, 0, 0);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int nk = K / 16;
    dma(0, 0);
    dma(nk > 1 ? 1 : 0, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");          // step 0 has landed (this wave's share)
    __syncthreads();
    // fragment addresses (floats): row = 64 w? + 32 b + (lane & 31), quad 2 i + hh at slot position (2 i + hh) ^ swz
    const int hh = lane >> 5;
    const int ra_ = wr * 64 + (lane & 31), rb_ = wc * 64 + (lane & 31);
    const int fa0 = ra_ * 16 + ((hh ^ ((ra_ >> 1) & 3)) << 2), fa1 = ra_ * 16 + (((2 + hh) ^ ((ra_ >> 1) & 3)) << 2);
    const int fb0 = rb_ * 16 + ((hh ^ ((rb_ >> 1) & 3)) << 2), fb1 = rb_ * 16 + (((2 + hh) ^ ((rb_ >> 1) & 3)) << 2);
    int cur = 0;
    for (int s = 0; s < nk; ++s) {
        int nxt2 = cur + 2; nxt2 = nxt2 >= 3 ? nxt2 - 3 : nxt2;
        dma(s + 2 < nk ? s + 2 : s, nxt2);                    // (past the end: a valid step again, never read)
        const float* a_ = As + cur * TB; const float* b_ = Bs + cur * TB;
        f32x4 fa[2][2], fb[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            fa[0][b] = *reinterpret_cast<const f32x4*>(a_ + fa0 + b * 512); fa[1][b] = *reinterpret_cast<const f32x4*>(a_ + fa1 + b * 512);
            fb[0][b] = *reinterpret_cast<const f32x4*>(b_ + fb0 + b * 512); fb[1][b] = *reinterpret_cast<const f32x4*>(b_ + fb1 + b * 512);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y)
                        acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][x][j], fb[i][y][j], acc[x][y], 0, 0, 0);
        // step s + 1 has landed (this wave's share), step s + 2's four requests may still travel.  NOT __syncthreads(): its fence is
        // s_waitcnt vmcnt(0), i.e. a wait for the requests issued a moment ago, on every step
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        cur = cur + 1 == 3 ? 0 : cur + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(C, 0, (int)((((long long)M - 1) * ldc + N) * 4), 0x00020000);
    float* patch = dyn + wave * (32 * 36);
    const int cr = lane >> 3, cc = (lane & 7) * 4;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * hh) * 36 + (lane & 31)] = acc[x][y][r];
            const int mb = m0 + wr * 64 + x * 32, nb = n0 + wc * 64 + y * 32;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = 8 * k + cr;
                const f32x4 v = *reinterpret_cast<const f32x4*>(patch + row * 36 + cc);
                const bool ok = mb + row < M && nb + cc < N;
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, rc,
                                                       (int)(ok ? (unsigned)((((long long)(mb + row)) * ldc + nb + cc) * 4) : 0x80000000u), 0, 0);
            }
        }
    __syncthreads();                                          // the patches alias the staging buffers of the next tile
    }
#endif
}

int main(int argc, char** argv)
{
    const int M = argc > 1 ? atoi(argv[1]) : 300736, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 960;
    float *dA, *dB, *dC;
    (void)hipMalloc(&dA, (size_t)M * K * 4); (void)hipMalloc(&dB, (size_t)N * K * 4); (void)hipMalloc(&dC, (size_t)M * N * 4);
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : hA) v = rnd();
    for (auto& v : hB) v = rnd();
    (void)hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN);
    const int variant = argc > 4 ? atoi(argv[4]) : 2;
    const size_t shm = 6 * 128 * 16 * 4;
    auto launch = [&]() {
        if (variant >= 2) hipLaunchKernelGGL(sgemm_nt_dma_kernel, dim3(variant == 2 ? grid.x : 192, grid.y), dim3(256), shm, 0, dA, K, dB, K, dC, N, M, N, K);
        else hipLaunchKernelGGL(sgemm_nt_kernel, grid, dim3(256), 0, 0, dA, K, dB, K, dC, N, M, N, K);
    };
    launch();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) launch();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    std::vector<float> hC((size_t)64 * N);
    (void)hipMemcpy(hC.data(), dC + (size_t)(M - 64) * N, hC.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < 64; i += 7)
        for (int j = 0; j < N; j += 37) {
            double ref = 0; const size_t m = (size_t)M - 64 + i;
            for (int k = 0; k < K; ++k) ref += (double)hA[m * K + k] * hB[(size_t)j * K + k];
            worst = fmax(worst, fabs(ref - hC[(size_t)i * N + j]));
        }
    printf("variant %d  M %d N %d K %d: %.3f ms -> %.1f TFLOP/s; worst |err| vs f64 on samples %.3g (hipGetLastError %d)\n", variant, M, N, K, ms, 2.0 * M * N * K / ms / 1e9, worst, (int)hipGetLastError());
    return 0;
}
