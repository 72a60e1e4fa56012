// mfma_mix_probe.hip -- diagnostic: what does each ingredient of the defect kernel's inner loop cost the fp64 matrix
// pipe?  One "tile" = 24 x v_mfma_f64_16x16x4_f64 (12 accumulators x 2 k-steps), as in emi_symdefect_ring2_f64_kernel.
//   bit 0: the A operands are produced by v_add_f64 / v_sub from two loaded values (e = xf + xm, o = xf - xm)
//   bit 1: the operands of every tile are read from LDS (14 ds_read_b128 per wave and tile)
//   bit 2: one s_barrier per tile (256-thread workgroups)
//   bit 3: 5 LDS-DMA instructions (global_load_lds_dwordx4, 1 KB each) per wave and tile from an L2-resident buffer,
//          counted vmcnt two tiles behind
//   bit 4: (with bit 3) the B operands (De / Do rows, private to a wave) do not go through LDS: two 16-byte global
//          loads per lane and tile straight into registers (inline asm, one tile ahead), 3 LDS-DMA instructions left
// W = workgroups per CU (1 or 2).  Prints cycles per MFMA per SIMD (64 = the pipe's rate) and TFLOP/s.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256, 2) void tile_loop(int tiles, unsigned long long* stamps, double* sink, const double* gbuf) {
    extern __shared__ __attribute__((aligned(16))) double smem[];      // 3 stages x 20 KB
    constexpr int STAGE = 320 * 8;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r16 = lane & 15, kq = lane >> 4;
    for (int i = tid; i < 3 * STAGE; i += 256) smem[i] = 1.0 + 1e-6 * i;
    __syncthreads();
    d4 acc_a[6], acc_b[6];
    for (int s = 0; s < 6; ++s) { acc_a[s] = d4{0, 0, 0, 0}; acc_b[s] = d4{0, 0, 0, 0}; }
    double2 be = make_double2(1.0 + lane * 1e-3, 0.5), bo = make_double2(0.25, 2.0 - lane * 1e-3);
    double2 xf[6], xm[6];
    for (int s = 0; s < 6; ++s) { xf[s] = make_double2(0.1 * s + lane, 1.0); xm[s] = make_double2(0.3, 0.7 * s); }
    const double* src = gbuf + ((size_t)blockIdx.x * 64 + lane) * 2 + wid * 128;
    auto issue = [&](int stage, int kt) {
#pragma unroll
        for (int t = 0; t < ((MODE & 16) ? 3 : 5); ++t) {
            double* dst = smem + (size_t)stage * STAGE + (size_t)(wid + 4 * t) * 128;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + ((kt * 5 + t) & 63) * 512), (lds_ptr_t)dst, 16, 0, 0);
        }
    };
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4i dn0 = {0, 0, 0, 0}, dn1 = {0, 0, 0, 0};          // B fragments of the NEXT tile, in flight
    const double* dsrc = gbuf + (size_t)(blockIdx.x & 63) * 4096 + (wid * 16 + r16) * 64 + kq * 2;
    auto dload = [&](int kt) {
        const double* g0 = dsrc + (kt & 7) * 8, *g1 = g0 + 2048;
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %3, off" : "=v"(dn0), "=v"(dn1) : "v"(g0), "v"(g1) : "memory");
    };
    if (MODE & 8) { issue(0, 0); if (MODE & 16) dload(0); issue(1, 1); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int kt = 0; kt < tiles; ++kt) {
        if ((MODE & 24) == 24) {
            // outstanding, oldest first: X(kt) | D(kt) | X(kt+1): all but the 3 youngest must have landed
            asm volatile("s_waitcnt vmcnt(3)" : "+v"(dn0), "+v"(dn1) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_memcpy(&be, &dn0, 16);
            __builtin_memcpy(&bo, &dn1, 16);
        } else if (MODE & 8) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        if (MODE & 4) asm volatile("s_barrier" ::: "memory");
        if ((MODE & 24) == 24) dload(kt + 1);
        if (MODE & 8) issue((kt + 2) % 3, kt + 2);
        if (MODE & 2) {
            const double* S = smem + (size_t)(kt % 3) * STAGE;
            const int rb = wid * 16 + r16;
            if (!(MODE & 16)) {
                be = *reinterpret_cast<const double2*>(S + 192 * 8 + rb * 8 + ((kq ^ ((0 - (rb >> 2)) & 3)) << 1));
                bo = *reinterpret_cast<const double2*>(S + 256 * 8 + rb * 8 + ((kq ^ ((0 - (rb >> 2)) & 3)) << 1));
            }
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const int r = s * 16 + r16;
                xf[s] = *reinterpret_cast<const double2*>(S + r * 8 + ((kq ^ ((0 - (r >> 2)) & 3)) << 1));
                xm[s] = *reinterpret_cast<const double2*>(S + (96 + r) * 8 + (((3 - kq) ^ ((0 - (r >> 2)) & 3)) << 1));
            }
        }
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            const double a1 = (MODE & 1) ? xf[s].x + xm[s].y : xf[s].x, a2 = (MODE & 1) ? xf[s].x - xm[s].y : xm[s].y;
            acc_a[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, be.x, acc_a[s], 0, 0, 0);
            acc_b[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, bo.x, acc_b[s], 0, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            const double a1 = (MODE & 1) ? xf[s].y + xm[s].x : xf[s].y, a2 = (MODE & 1) ? xf[s].y - xm[s].x : xm[s].x;
            acc_a[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, be.y, acc_a[s], 0, 0, 0);
            acc_b[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, bo.y, acc_b[s], 0, 0, 0);
        }
        if (!(MODE & 2)) { be.x += 1e-9; bo.y -= 1e-9; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double sum = 0;
    for (int s = 0; s < 6; ++s) sum += acc_a[s][0] + acc_b[s][1] + acc_a[s][2] + acc_b[s][3];
    if (sum == 12345.678) sink[0] = sum;
    if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE> void run(int wpc, const double* gbuf, unsigned long long* d_st, double* d_sink) {
    const int tiles = 20000, nblk = 256 * wpc;
    const size_t lds = 3 * 320 * 8 * 8;
    CK(hipFuncSetAttribute((const void*)tile_loop<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(tile_loop<MODE>, dim3(nblk), dim3(256), lds, 0, tiles, d_st, d_sink, gbuf);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(tile_loop<MODE>, dim3(nblk), dim3(256), lds, 0, tiles, d_st, d_sink, gbuf);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    std::vector<unsigned long long> st(2 * nblk);
    CK(hipMemcpy(st.data(), d_st, 16 * nblk, hipMemcpyDeviceToHost));
    std::vector<double> clk(nblk), cyc(nblk);
    for (int i = 0; i < nblk; ++i) { clk[i] = (double)st[2 * i] / st[2 * i + 1] * 100e6; cyc[i] = (double)st[2 * i]; }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    const double nm = 24.0 * tiles;
    printf("{\"mode\": %d, \"v_add\": %d, \"lds_reads\": %d, \"barrier\": %d, \"lds_dma\": %d, \"b_direct\": %d, \"wg_per_cu\": %d, \"ms\": %.3f, \"clock_GHz\": %.3f, "
           "\"cycles_per_mfma_per_simd\": %.2f, \"TFLOPs\": %.2f}\n", MODE, MODE & 1, (MODE >> 1) & 1, (MODE >> 2) & 1, (MODE >> 3) & 1, (MODE >> 4) & 1, wpc, ms,
           clk[nblk / 2] / 1e9, cyc[nblk / 2] / nm / wpc, nm * 2048.0 * 4 * nblk / (ms * 1e-3) / 1e12);
}

int main() {
    double *gbuf, *d_sink; unsigned long long* d_st;
    CK(hipMalloc(&gbuf, (size_t)8 << 20)); CK(hipMemset(gbuf, 0, (size_t)8 << 20));
    CK(hipMalloc(&d_sink, 8)); CK(hipMalloc(&d_st, 16 * 512));
    for (int wpc = 1; wpc <= 2; ++wpc) {
        run<0>(wpc, gbuf, d_st, d_sink);
        run<1>(wpc, gbuf, d_st, d_sink);
        run<2>(wpc, gbuf, d_st, d_sink);
        run<3>(wpc, gbuf, d_st, d_sink);
        run<7>(wpc, gbuf, d_st, d_sink);
        run<15>(wpc, gbuf, d_st, d_sink);
        run<31>(wpc, gbuf, d_st, d_sink);
    }
    return 0;
}
