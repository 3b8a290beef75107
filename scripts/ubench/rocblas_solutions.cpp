// Which Tensile solution does rocBLAS pick for the training step's wide f32 products, and is there a faster one in the library?
// (rocblas_gemm_ex_get_solutions / rocblas_gemm_algo_solution_index, beta API.)  Shapes as rocBLAS sees them (column major).
#define ROCBLAS_BETA_FEATURES_API
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <vector>
#include <algorithm>
struct Shape { const char* name; rocblas_operation ta, tb; int m, n, k, lda, ldb, ldc; };
int main()
{
    const int RO = 300736;
    const Shape sh[] = {
        {"skip fwd   C[RO,512]  = ZC[RO,960] . WS[960,512]", rocblas_operation_none, rocblas_operation_none, 512, RO, 960, 512, 960, 512},
        {"conv1 fwd  C[RO,512]  = H1[RO,512] . W1[512,512]", rocblas_operation_none, rocblas_operation_none, 512, RO, 512, 512, 512, 512},
        {"dH1        C[RO,512]  = dC1[RO,512] . W1^T", rocblas_operation_transpose, rocblas_operation_none, 512, RO, 512, 512, 512, 512},
        {"dZC        C[RO,960]  = dSK[RO,512] . WS^T", rocblas_operation_transpose, rocblas_operation_none, 960, RO, 512, 512, 512, 960},
        // round 5: the forward products with the weight stored transposed ([N][K] row-major), i.e. rocBLAS's (T, N) kernels instead of (N, N)
        {"skip fwd   C[RO,512]  = ZC[RO,960] . (WSt[512,960])^T", rocblas_operation_transpose, rocblas_operation_none, 512, RO, 960, 960, 960, 512},
    };
    rocblas_handle h; rocblas_create_handle(&h);
    float *A, *B, *C;
    (void)hipMalloc(&A, (size_t)1024 * 1024 * 4); (void)hipMalloc(&B, (size_t)RO * 960 * 4); (void)hipMalloc(&C, (size_t)RO * 960 * 4);
    (void)hipMemset(A, 0, (size_t)1024 * 1024 * 4); (void)hipMemset(B, 0, (size_t)RO * 960 * 4);
    const float one = 1.0f, zero = 0.0f;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (const Shape& s : sh) {
        auto run = [&](int idx, int reps) {
            float best = 1e30f;
            for (int r = 0; r < reps; ++r) {
                (void)hipEventRecord(e0);
                rocblas_status st = rocblas_gemm_ex(h, s.ta, s.tb, s.m, s.n, s.k, &one, A, rocblas_datatype_f32_r, s.lda, B, rocblas_datatype_f32_r, s.ldb, &zero, C,
                                                    rocblas_datatype_f32_r, s.ldc, C, rocblas_datatype_f32_r, s.ldc, rocblas_datatype_f32_r,
                                                    idx < 0 ? rocblas_gemm_algo_standard : rocblas_gemm_algo_solution_index, idx < 0 ? 0 : idx, 0);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                if (st != rocblas_status_success) return -1.0f;
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                best = std::min(best, ms);
            }
            return best;
        };
        run(-1, 2);
        const float def = run(-1, 5);
        rocblas_int n = 0;
        rocblas_gemm_ex_get_solutions(h, s.ta, s.tb, s.m, s.n, s.k, &one, A, rocblas_datatype_f32_r, s.lda, B, rocblas_datatype_f32_r, s.ldb, &zero, C, rocblas_datatype_f32_r,
                                      s.ldc, C, rocblas_datatype_f32_r, s.ldc, rocblas_datatype_f32_r, rocblas_gemm_algo_solution_index, 0, nullptr, &n);
        std::vector<rocblas_int> list(n);
        rocblas_gemm_ex_get_solutions(h, s.ta, s.tb, s.m, s.n, s.k, &one, A, rocblas_datatype_f32_r, s.lda, B, rocblas_datatype_f32_r, s.ldb, &zero, C, rocblas_datatype_f32_r,
                                      s.ldc, C, rocblas_datatype_f32_r, s.ldc, rocblas_datatype_f32_r, rocblas_gemm_algo_solution_index, 0, list.data(), &n);
        std::vector<std::pair<float, int>> res;
        for (int i = 0; i < n; ++i) { const float t = run(list[i], 2); if (t > 0) res.push_back({t, list[i]}); }
        std::sort(res.begin(), res.end());
        const double fl = 2.0 * s.m * (double)s.n * s.k;
        printf("%s: default %.3f ms (%.1f TFLOP/s); %d solutions; best", s.name, def, fl / def / 1e9, n);
        for (size_t i = 0; i < res.size() && i < 4; ++i) printf("  #%d %.3f ms (%.1f)", res[i].second, res[i].first, fl / res[i].first / 1e9);
        printf("\n");
    }
    return 0;
}
