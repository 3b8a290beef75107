// Round 6 harness of the training step's own f32-MFMA GEMMs (development copy of csrc/twv_gemm.hpp's kernels):
//   NT:  C[M][N] = A[M][K] . Bt[N][K]^T      TN:  C[M][N] = sum_r A[r][M] B[r][N]  (split over row slabs)
// Workgroup = 4 waves on a 256 x 128 tile (wave = 128 x 64 = 4 x 2 blocks of v_mfma_f32_32x32x2_f32), K advances 16 per step through
// three LDS stages filled by LDS-DMA (global_load_lds b128), fragments of the next half step read while the current one multiplies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
using rsrc_t = __amdgpu_buffer_rsrc_t;
extern __shared__ __attribute__((aligned(16))) float dyn[];
#define LDS4(p) (*reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((const __attribute__((address_space(3))) float*)(p)))

#ifndef NSTAGE
#define NSTAGE 3
#endif
constexpr int TM = 256, TN = 128, TK = 16;
constexpr int SA = TM * TK, SB = TN * TK, SS = SA + SB;          // floats per stage

// LDS image of an operand tile [rows][16]: 64-byte rows, the 16-byte quad q of row r at position q ^ ((r >> 1) & 3) (DMA writes linearly:
// the swizzle is applied to the global address a lane fetches); a ds_read_b128 of 32 consecutive rows at one quad touches every bank group
template <int EPI>
__global__ void __launch_bounds__(256, 2) sgemm_nt3_kernel(const float* __restrict__ A, int lda, const float* __restrict__ Bt, int ldb, float* __restrict__ C, int ldc,
                                                          int M, int N, int K, const float* __restrict__ bias)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wave >> 1, wc = wave & 1;
    const int ntn = (N + TN - 1) / TN, ntm = (M + TM - 1) / TM, ntiles = ntm * ntn;
    const int hh = lane >> 5, l31 = lane & 31;
    const int nk = K / TK;
    // XCD-aware tile order: workgroups go to the 8 XCDs round-robin (blockIdx % 8), each XCD has its own L2.  The ntn column tiles of a
    // row tile share the A rows: they run on ONE XCD at the same time (slot % ntn), so a row tile crosses the fabric once, not ntn times.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;      // gridDim.x is a multiple of 8 * ntn
    const int groups = per / ntn;                                                     // row tiles in flight per XCD
    for (int it = 0;; ++it) {
        const int tm = ((it * groups + slot / ntn) << 3) + xcd, tn = slot % ntn;
        if (tm >= ntm) { if (((it * groups) << 3) >= ntm) break; else continue; }
        if (slot >= groups * ntn) break;
        const int m0 = tm * TM, n0 = tn * TN;
        // DMA addresses: A tile = 1024 float4 slots (4 per thread), B tile = 512 (2 per thread); slot s: row s >> 2, position s & 3
        int va[4], vb[2];                                        // byte offsets from the tile's first row (32-bit: a tile spans < 2 GB)
        const int mrows = M - m0 < TM ? M - m0 : TM, nrows = N - n0 < TN ? N - n0 : TN;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int sl = (wave * 4 + h) * 64 + lane, row = sl >> 2, q = (sl & 3) ^ ((row >> 1) & 3);
            va[h] = ((row < mrows ? row : mrows - 1) * lda + 4 * q) * 4;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int sl = (wave * 2 + h) * 64 + lane, row = sl >> 2, q = (sl & 3) ^ ((row >> 1) & 3);
            vb[h] = ((row < nrows ? row : nrows - 1) * ldb + 4 * q) * 4;
        }
        const rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + (long long)m0 * lda), 0, 0x7fffffff, 0x00020000);
        const rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bt + (long long)n0 * ldb), 0, 0x7fffffff, 0x00020000);
        auto dma = [&](int step, int buf) {
#ifdef ABL_NODMA
            if (step >= 0) return;
#endif
            float* as = dyn + buf * SS; float* bs = as + SA;
            const int so = step * (TK * 4);
#pragma unroll
            for (int h = 0; h < 4; ++h) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(as + (wave * 4 + h) * 256), 16, va[h], so, 0, 0);
#pragma unroll
            for (int h = 0; h < 2; ++h) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)(bs + (wave * 2 + h) * 256), 16, vb[h], so, 0, 0);
        };
        f32x16 acc[4][2];
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.0f;
        // fragment addresses (floats inside a stage): A row = wr * 128 + 32 x + l31, quad (2 i + hh) at position ^ swz(row)
        int fa[4][2], fb[2][2];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int row = wr * 128 + 32 * x + l31;
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[x][i] = row * 16 + (((2 * i + hh) ^ ((row >> 1) & 3)) << 2);
        }
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int row = wc * 64 + 32 * y + l31;
#pragma unroll
            for (int i = 0; i < 2; ++i) fb[y][i] = SA + row * 16 + (((2 * i + hh) ^ ((row >> 1) & 3)) << 2);
        }
        dma(0, 0);
#pragma unroll
        for (int i = 1; i < NSTAGE - 1; ++i) dma(nk > i ? i : 0, i);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * (NSTAGE - 2)) : "memory");
        asm volatile("s_barrier" ::: "memory");
        f32x4 ga[2][4], gb[2][2];                               // fragments of two half steps
#pragma unroll
        for (int x = 0; x < 4; ++x) ga[0][x] = LDS4(dyn + fa[x][0]);
#pragma unroll
        for (int y = 0; y < 2; ++y) gb[0][y] = LDS4(dyn + fb[y][0]);
        int cur = 0;
        for (int s = 0; s < nk; ++s) {
            int nb2 = cur + (NSTAGE - 1); nb2 = nb2 >= NSTAGE ? nb2 - NSTAGE : nb2;
            int nb1 = cur + 1; nb1 = nb1 >= NSTAGE ? nb1 - NSTAGE : nb1;
            dma(s + NSTAGE - 1 < nk ? s + NSTAGE - 1 : s, nb2);          // (past the end: a valid step again, never read)
            const float* st = dyn + cur * SS;
            // half step 1 fragments while half step 0 multiplies
#ifdef ABL_NOLDS
            if (s < 0) {
#else
            {
#endif
#pragma unroll
            for (int x = 0; x < 4; ++x) ga[1][x] = LDS4(st + fa[x][1]);
#pragma unroll
            for (int y = 0; y < 2; ++y) gb[1][y] = LDS4(st + fb[y][1]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[0][x][j], gb[0][y][j], acc[x][y], 0, 0, 0);
#ifndef NO_SGB
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
#endif
            // the next step has landed (this wave's share); everybody's after the barrier
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * (NSTAGE - 2)) : "memory");
#ifndef ABL_NOBAR
            asm volatile("s_barrier" ::: "memory");
#endif
            __builtin_amdgcn_sched_barrier(0);
#ifdef ABL_NOLDS
            const float* sn = dyn + nb1 * SS; if (s < 0) {
#else
            const float* sn = dyn + nb1 * SS; {
#endif
#pragma unroll
            for (int x = 0; x < 4; ++x) ga[0][x] = LDS4(sn + fa[x][0]);
#pragma unroll
            for (int y = 0; y < 2; ++y) gb[0][y] = LDS4(sn + fb[y][0]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[1][x][j], gb[1][y][j], acc[x][y], 0, 0, 0);
#ifndef NO_SGB
#pragma unroll
            for (int i = 0; i < 6; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
#endif
            cur = nb1;
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        // epilogue: C layout lane = column (l31), register r = row (r & 3) + 8 (r >> 2) + 4 hh.  One buffer_store_dword per accumulator
        // register, the row in a SCALAR offset, the lane's (4 hh rows + column) in one VGPR computed once per tile, rows past M dropped by the
        // descriptor's size, columns past N by an out-of-range lane offset: no vector arithmetic per element (the first version spent ~25
        // VALU instructions per store on 64-bit addresses and predicates: 4000 per tile, 8 % of a K = 960 tile's matrix time)
        {
            const int mrows_w = M - (m0 + wr * 128);                                            // rows of this wave's 128 that exist
            const long long cbase = ((long long)(m0 + wr * 128)) * ldc + n0 + wc * 64;
            const long long cbytes = mrows_w > 0 ? ((long long)(mrows_w < 128 ? mrows_w : 128) * ldc) * 4 : 0;
            const rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(C + cbase, 0, (int)(cbytes > 0x7fffffffLL ? 0x7fffffff : cbytes), 0x00020000);
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                const int n = n0 + wc * 64 + 32 * y + l31;
                const float bv = (EPI == 1 && bias && n < N) ? bias[n] : 0.0f;
                const int vo = n < N ? (4 * hh * ldc + 32 * y + l31) * 4 : (int)0x7ffffff0;
#pragma unroll
                for (int x = 0; x < 4; ++x) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[x][y][r];
                        if (EPI == 1) { v = v + bv; v = v > 0.0f ? v : 0.0f; }
                        const int so = __builtin_amdgcn_readfirstlane((32 * x + (r & 3) + 8 * (r >> 2)) * ldc * 4);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rc, vo, so, 0);
                    }
                }
            }
        }
    }
}

// ---- TN: C[M][N] (+ split partials) = sum over rows r in [k0, k1) of A[r][m] * B[r][n]   (a weight gradient: tall K, M x N small)
// Workgroup = 256 x 128 output tile over ONE K slab (blockIdx.y); A / B tiles are [16 k][256 m] / [16 k][128 n] in LDS exactly as they lie
// in memory (rows of 1 KB / 512 B, LDS-DMA b128); the MFMA operands come out of ONE ds_read_b128 (A: four consecutive m = the lane's row
// of FOUR blocks) and ONE ds_read_b64 (B: two consecutive n) per k pair: block j of a wave holds the rows 4 i + j, i = 0..31.
__global__ void __launch_bounds__(256, 2) sgemm_tn3_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ Cpart,
                                                          int M, int N, long long K, int nsplit)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wave >> 1, wc = wave & 1;
    const int ntn = (N + TN - 1) / TN;
    const int tile = blockIdx.x, split = blockIdx.y;
    const int m0 = (tile / ntn) * TM, n0 = (tile % ntn) * TN;
    const int hh = lane >> 5, l31 = lane & 31;
    const long long kper = (K + nsplit - 1) / nsplit;
    const long long ks = (long long)split * kper, ke = ks + kper < K ? ks + kper : K;
    const int krows = (int)(ke > ks ? ke - ks : 0);
    const int nk = (krows + TK - 1) / TK;
    // DMA: A stage = 16 rows x 64 float4 (4 per thread), B stage = 16 rows x 32 float4 (2 per thread); rows past the slab / columns past
    // M, N read zeros (out of the descriptor's range)
    int va[4], vb[2];
    const int acols = M - m0 < TM ? M - m0 : TM, bcols = N - n0 < TN ? N - n0 : TN;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int sl = (wave * 4 + h) * 64 + lane, row = sl >> 6, c4 = sl & 63;
        va[h] = 4 * c4 < acols ? (row * lda + 4 * c4) * 4 : (int)0x7ffffff0;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int sl = (wave * 2 + h) * 64 + lane, row = sl >> 5, c4 = sl & 31;
        vb[h] = 4 * c4 < bcols ? (row * ldb + 4 * c4) * 4 : (int)0x7ffffff0;
    }
    const long long abytes = ((long long)(krows - 1) * lda + acols) * 4, bbytes = ((long long)(krows - 1) * ldb + bcols) * 4;
    const rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + ks * lda + m0), 0, (int)(krows > 0 ? abytes : 0), 0x00020000);
    const rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B + ks * ldb + n0), 0, (int)(krows > 0 ? bbytes : 0), 0x00020000);
    auto dma = [&](int step, int buf) {
        float* as = dyn + buf * SS; float* bs = as + SA;
        const int so = step * TK * lda * 4, sob = step * TK * ldb * 4;
#pragma unroll
        for (int h = 0; h < 4; ++h) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(as + (wave * 4 + h) * 256), 16, va[h], so, 0, 0);
#pragma unroll
        for (int h = 0; h < 2; ++h) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)(bs + (wave * 2 + h) * 256), 16, vb[h], sob, 0, 0);
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.0f;
    // fragment addresses (floats inside a stage): A[k][wr * 128 + 4 l31 ..+3], B[k][wc * 64 + 2 l31 ..+1]; k pair p = rows (2 p + hh)
    const int fa = hh * TM + wr * 128 + 4 * l31, fb = SA + hh * TN + wc * 64 + 2 * l31;
    if (nk > 0) dma(0, 0);
    if (nk > 0) dma(nk > 1 ? 1 : 0, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    f32x4 ga[2][4]; f32x2 gb[2][4];                           // four k pairs = half a stage per set
    auto frag = [&](const float* st, int half, int set) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            ga[set][p] = LDS4(st + fa + (8 * half + 2 * p) * TM);
            gb[set][p] = *reinterpret_cast<const __attribute__((address_space(3))) f32x2*>((const __attribute__((address_space(3))) float*)(st + fb + (8 * half + 2 * p) * TN));
        }
    };
    if (nk > 0) frag(dyn, 0, 0);
    int cur = 0;
    for (int s = 0; s < nk; ++s) {
        int nb2 = cur + 2; nb2 = nb2 >= 3 ? nb2 - 3 : nb2;
        int nb1 = cur + 1; nb1 = nb1 >= 3 ? nb1 - 3 : nb1;
        dma(s + 2 < nk ? s + 2 : s, nb2);
        const float* st = dyn + cur * SS;
        frag(st, 1, 1);
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[0][p][x], gb[0][p][y], acc[x][y], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        frag(dyn + nb1 * SS, 0, 0);
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[1][p][x], gb[1][p][y], acc[x][y], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        cur = nb1;
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // epilogue: block x, register r, lane: row m = wr * 128 + 4 ((r & 3) + 8 (r >> 2) + 4 hh) + x, columns n = wc * 64 + 2 l31 + {0, 1}
    {
        float* Cp = Cpart + (long long)split * M * N;
        const int mrows_w = M - (m0 + wr * 128);
        const long long cbytes = mrows_w > 0 ? ((long long)(mrows_w < 128 ? mrows_w : 128) * N) * 4 : 0;
        const rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(Cp + (long long)(m0 + wr * 128) * N + n0 + wc * 64, 0, (int)cbytes, 0x00020000);
        const int ncol = n0 + wc * 64 + 2 * l31;
        const int vo = ncol + 1 < N ? (16 * hh * N + 2 * l31) * 4 : (int)0x7ffffff0;      // (N even: both columns or none)
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int so = __builtin_amdgcn_readfirstlane((4 * ((r & 3) + 8 * (r >> 2)) + x) * N * 4);
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(acc[x][0][r]), __float_as_uint(acc[x][1][r])}, rc, vo, so, 0);
            }
    }
}

int main(int argc, char** argv)
{
    const int M = argc > 1 ? atoi(argv[1]) : 300736, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 960;
    const int wgs = argc > 4 ? atoi(argv[4]) : 512;
    const int tn_mode = argc > 5 ? atoi(argv[5]) : 0;
    if (tn_mode) {
        // TN: A [K][M], B [K][N] -> C [M][N]; here M, N = argv 1, 2 (small), K = argv 3 (rows), wgs = nsplit
        const long long K2 = K; const int nsplit = wgs;
        float *dA2, *dB2, *dP, *dC2;
        (void)hipMalloc(&dA2, (size_t)K2 * M * 4); (void)hipMalloc(&dB2, (size_t)K2 * N * 4); (void)hipMalloc(&dP, (size_t)nsplit * M * N * 4); (void)hipMalloc(&dC2, (size_t)M * N * 4);
        std::vector<float> hA2((size_t)K2 * M), hB2((size_t)K2 * N);
        unsigned s2 = 777; auto rnd2 = [&]() { s2 = s2 * 1664525u + 1013904223u; return ((s2 >> 8) & 0xffff) / 65536.0f - 0.5f; };
        for (auto& v : hA2) v = rnd2();
        for (auto& v : hB2) v = rnd2();
        (void)hipMemcpy(dA2, hA2.data(), hA2.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB2, hB2.data(), hB2.size() * 4, hipMemcpyHostToDevice);
        const size_t shm2 = (size_t)3 * SS * 4;
        (void)hipFuncSetAttribute((const void*)sgemm_tn3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm2);
        const int tiles = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
        auto launch2 = [&]() { hipLaunchKernelGGL(sgemm_tn3_kernel, dim3(tiles, nsplit), dim3(256), shm2, 0, dA2, M, dB2, N, dP, M, N, K2, nsplit); };
        launch2();
        hipEvent_t f0, f1; (void)hipEventCreate(&f0); (void)hipEventCreate(&f1);
        (void)hipEventRecord(f0);
        for (int r = 0; r < 5; ++r) launch2();
        (void)hipEventRecord(f1); (void)hipEventSynchronize(f1);
        float ms2; (void)hipEventElapsedTime(&ms2, f0, f1); ms2 /= 5;
        std::vector<float> hP((size_t)nsplit * M * N);
        (void)hipMemcpy(hP.data(), dP, hP.size() * 4, hipMemcpyDeviceToHost);
        double worst2 = 0;
        for (int i = 0; i < M; i += 61)
            for (int j = 0; j < N; j += 53) {
                double ref = 0; for (long long k = 0; k < K2; ++k) ref += (double)hA2[k * M + i] * hB2[k * N + j];
                double got = 0; for (int sp = 0; sp < nsplit; ++sp) got += hP[((size_t)sp * M + i) * N + j];
                worst2 = fmax(worst2, fabs(ref - got));
            }
        printf("tn3 nsplit %d tiles %d  M %d N %d K %lld: %.3f ms -> %.1f TFLOP/s; worst |err| vs f64 on samples %.3g (hipGetLastError %d)\n", nsplit, tiles, M, N, K2, ms2, 2.0 * M * N * K2 / ms2 / 1e9, worst2, (int)hipGetLastError());
        return 0;
    }
    float *dA, *dB, *dC;
    (void)hipMalloc(&dA, (size_t)M * K * 4); (void)hipMalloc(&dB, (size_t)N * K * 4); (void)hipMalloc(&dC, (size_t)M * N * 4);
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : hA) v = rnd();
    for (auto& v : hB) v = rnd();
    (void)hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    const size_t shm = (size_t)NSTAGE * SS * 4;
    (void)hipFuncSetAttribute((const void*)sgemm_nt3_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    auto launch = [&]() { hipLaunchKernelGGL(sgemm_nt3_kernel<0>, dim3(wgs), dim3(256), shm, 0, dA, K, dB, K, dC, N, M, N, K, (const float*)nullptr); };
    launch();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) launch();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    std::vector<float> hC((size_t)300 * N);
    (void)hipMemcpy(hC.data(), dC + (size_t)(M - 300) * N, hC.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < 300; i += 7)
        for (int j = 0; j < N; j += 37) {
            double ref = 0; const size_t m = (size_t)M - 300 + i;
            for (int k = 0; k < K; ++k) ref += (double)hA[m * K + k] * hB[(size_t)j * K + k];
            worst = fmax(worst, fabs(ref - hC[(size_t)i * N + j]));
        }
    printf("nt3 stages %d wgs %d  M %d N %d K %d: %.3f ms -> %.1f TFLOP/s; worst |err| vs f64 on samples %.3g (hipGetLastError %d)\n", NSTAGE, wgs, M, N, K, ms, 2.0 * M * N * K / ms / 1e9, worst, (int)hipGetLastError());
    return 0;
}
