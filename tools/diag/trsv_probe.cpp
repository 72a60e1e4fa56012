// How fast is a triangular solve with ONE right-hand side at the size of the Schur complement (6144 rows)?
// rocblas_dtrsv (what rocsolver_dpotrs ends in for nrhs = 1) against rocblas_dtrsm with 1 .. 8 columns.
//   hipcc -O2 --offload-arch=gfx950 -o etol_amd/lib/trsv_probe tools/diag/trsv_probe.cpp -lrocblas
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <vector>
#include <cmath>
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 6144;
    std::vector<double> hL((size_t)n * n, 0.0), hb((size_t)n * 8, 1.0);
    for (int j = 0; j < n; ++j)
        for (int i = j; i < n; ++i) hL[(size_t)j * n + i] = i == j ? 4.0 + (i % 7) : 0.5 / (1.0 + std::abs(i - j));
    double *L, *b;
    hipMalloc(&L, hL.size() * 8); hipMalloc(&b, hb.size() * 8);
    hipMemcpy(L, hL.data(), hL.size() * 8, hipMemcpyHostToDevice);
    rocblas_handle h; rocblas_create_handle(&h);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double one = 1.0;
    auto time = [&](const char* name, auto f) {
        for (int i = 0; i < 3; ++i) { hipMemcpy(b, hb.data(), hb.size() * 8, hipMemcpyHostToDevice); f(); }
        hipDeviceSynchronize();
        float ms = 0, tot = 0;
        for (int i = 0; i < 10; ++i) {
            hipMemcpy(b, hb.data(), hb.size() * 8, hipMemcpyHostToDevice);
            hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); tot += ms;
        }
        printf("{\"probe\": \"trsv\", \"n\": %d, \"what\": \"%s\", \"ms\": %.4f}\n", n, name, tot / 10);
    };
    time("dtrsv lower N", [&] { rocblas_dtrsv(h, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, n, L, n, b, 1); });
    time("dtrsv lower T", [&] { rocblas_dtrsv(h, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit, n, L, n, b, 1); });
    for (int k : {1, 2, 4, 8}) {
        char nm[64]; snprintf(nm, sizeof nm, "dtrsm lower N, %d columns", k);
        time(nm, [&] { rocblas_dtrsm(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, n, k, &one, L, n, b, n); });
        snprintf(nm, sizeof nm, "dtrsm lower T, %d columns", k);
        time(nm, [&] { rocblas_dtrsm(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit, n, k, &one, L, n, b, n); });
    }
    time("dgemv N (same bytes as a full matrix sweep)", [&] { rocblas_dgemv(h, rocblas_operation_none, n, n, &one, L, n, b, 1, &one, b + n, 1); });
    return 0;
}
