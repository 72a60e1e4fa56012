// clock_probe.hip -- diagnostic (not part of the library): what clock does the chip hold under the two kernels
// of the evaluation pass, alone and together?  In-kernel clock = delta s_memtime / delta s_memrealtime x 100 MHz
// (MI355X_MICROARCH.md, DVFS give-back item 6).  Stamps go to their own buffer; no result depends on them.
//
//   arm "mfma" : v_mfma_f64_16x16x4_f64 back to back, operands in registers, 12 independent accumulators per wave
//                (the shape of the defect kernel's inner loop: 6 states x {even, odd}), W waves per SIMD
//   arm "store": every CU streams 16-byte stores over a 1 GiB buffer (the node kernel's traffic shape)
//   arm "both" : the two at once on two streams
// Prints per arm: median in-kernel clock over workgroups, cycles per MFMA per SIMD, TFLOP/s, store GB/s.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256, 2) void mfma_loop(int iters, unsigned long long* stamps, double* sink, double seed) {
    d4 acc[12];
    for (int i = 0; i < 12; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = seed + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        a += 1e-9; b -= 1e-9;            // operands change (random-like data, not zeros)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) sink[0] = s;      // keeps the loop alive
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

__global__ __launch_bounds__(256) void store_loop(double2* buf, size_t n2, int passes, unsigned long long* stamps) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (int p = 0; p < passes; ++p)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride)
            buf[i] = make_double2((double)i, (double)p);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

static double median_clock(const std::vector<unsigned long long>& st, int nblk, double* cycles) {
    std::vector<double> c(nblk), cy(nblk);
    for (int i = 0; i < nblk; ++i) { c[i] = (double)st[2 * i] / (double)st[2 * i + 1] * 100e6; cy[i] = (double)st[2 * i]; }
    std::sort(c.begin(), c.end());
    std::sort(cy.begin(), cy.end());
    *cycles = cy[nblk / 2];
    return c[nblk / 2];
}

int main(int argc, char** argv) {
    const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 1;
    const int iters = 40000;                                   // 480k MFMAs per wave
    const int nblk_m = 256 * waves_per_simd;                   // 256-thread blocks = 1 wave per SIMD each
    const int nblk_s = 2048;
    const size_t n2 = (size_t)1 << 26;                         // 1 GiB of double2
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    unsigned long long *d_st_m, *d_st_s; double* d_sink; double2* d_buf;
    CK(hipMalloc(&d_st_m, 16 * nblk_m)); CK(hipMalloc(&d_st_s, 16 * nblk_s)); CK(hipMalloc(&d_sink, 8)); CK(hipMalloc(&d_buf, n2 * 16));
    std::vector<unsigned long long> st_m(2 * nblk_m), st_s(2 * nblk_s);
    hipEvent_t e0, e1, f0, f1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
    // warm the chip up for ~2 s so DVFS has settled
    for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(mfma_loop, dim3(nblk_m), dim3(256), 0, s1, iters, d_st_m, d_sink, 0.5);
    CK(hipDeviceSynchronize());
    for (int arm = 0; arm < 3; ++arm) {
        const bool do_m = arm != 1, do_s = arm != 0;
        const int passes = 80;
        if (do_m) { CK(hipEventRecord(e0, s1)); for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(mfma_loop, dim3(nblk_m), dim3(256), 0, s1, iters, d_st_m, d_sink, 0.5); CK(hipEventRecord(e1, s1)); }
        if (do_s) { CK(hipEventRecord(f0, s2)); for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(store_loop, dim3(nblk_s), dim3(256), 0, s2, d_buf, n2, passes, d_st_s); CK(hipEventRecord(f1, s2)); }
        CK(hipDeviceSynchronize());
        const char* name = arm == 0 ? "mfma" : arm == 1 ? "store" : "both";
        if (do_m) {
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 4;
            CK(hipMemcpy(st_m.data(), d_st_m, 16 * nblk_m, hipMemcpyDeviceToHost));
            double cyc; const double clk = median_clock(st_m, nblk_m, &cyc);
            const double nm = 12.0 * iters;                     // MFMAs per wave
            printf("{\"arm\": \"%s\", \"kernel\": \"mfma_f64_16x16x4\", \"waves_per_simd\": %d, \"ms\": %.3f, \"clock_GHz\": %.3f, "
                   "\"cycles_per_mfma_per_wave\": %.2f, \"cycles_per_mfma_per_simd\": %.2f, \"TFLOPs\": %.2f}\n",
                   name, waves_per_simd, ms, clk / 1e9, cyc / nm, cyc / nm / waves_per_simd,
                   nm * 2048.0 * 4 * nblk_m / (ms * 1e-3) / 1e12);
        }
        if (do_s) {
            float ms; CK(hipEventElapsedTime(&ms, f0, f1)); ms /= 4;
            CK(hipMemcpy(st_s.data(), d_st_s, 16 * nblk_s, hipMemcpyDeviceToHost));
            double cyc; const double clk = median_clock(st_s, nblk_s, &cyc);
            printf("{\"arm\": \"%s\", \"kernel\": \"store16B\", \"ms\": %.3f, \"clock_GHz\": %.3f, \"store_GBs\": %.1f}\n", name, ms,
                   clk / 1e9, (double)n2 * 16 * passes / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
