// timing probe: rocsolver dsytrf vs dgetrf vs dpotrf on the device (diagnostics, not part of the library)
#include <hip/hip_runtime.h>
#include <rocsolver/rocsolver.h>
#include <cstdio>
#include <vector>
#include <chrono>
#include <random>
int main(int argc, char** argv) {
    rocblas_handle h; rocblas_create_handle(&h);
    for (int a = 1; a < argc; ++a) {
        const int N = atoi(argv[a]);
        std::vector<double> A((size_t)N * N);
        std::mt19937_64 g(1); std::normal_distribution<double> nd;
        for (int i = 0; i < N; ++i) for (int j = 0; j <= i; ++j) { double v = nd(g); A[(size_t)i * N + j] = v; A[(size_t)j * N + i] = v; }
        double *dA; rocblas_int *dp, *di; hipMalloc(&dA, A.size() * 8); hipMalloc(&dp, N * 4); hipMalloc(&di, 4);
        for (int which = 0; which < 3; ++which) {
            for (int rep = 0; rep < 2; ++rep) {
                if (which == 2) { for (int i = 0; i < N; ++i) A[(size_t)i * N + i] += (rep == 0 ? 4.0 * N : 0); }
                hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
                hipDeviceSynchronize();
                auto t0 = std::chrono::steady_clock::now();
                rocblas_status st;
                if (which == 0) st = rocsolver_dgetrf(h, N, N, dA, N, dp, di);
                else if (which == 1) st = rocsolver_dsytrf(h, rocblas_fill_lower, N, dA, N, dp, di);
                else st = rocsolver_dpotrf(h, rocblas_fill_lower, N, dA, N, di);
                hipDeviceSynchronize();
                double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                int info; hipMemcpy(&info, di, 4, hipMemcpyDeviceToHost);
                printf("N=%d %s rep %d: %.1f ms (status %d info %d)\n", N, which == 0 ? "getrf" : which == 1 ? "sytrf" : "potrf", rep, ms, (int)st, info);
                fflush(stdout);
            }
        }
        hipFree(dA); hipFree(dp); hipFree(di);
    }
}
