// store_probe.hip -- diagnostic: what store rate does the memory system give the node kernel's ACCESS PATTERN,
// with no arithmetic at all?  VALS[B][116][M] doubles, M = 1024 (8 KB per entry row), B = 1024: 973 MB per launch.
//   linear   : grid-stride 16-byte stores over the whole buffer (the 7.4 TB/s reference of clock_probe)
//   node     : the node kernel's shape: grid (M/512, B) x 256 threads, a thread owns 2 adjacent nodes and issues 116
//              16-byte stores, one per entry row (a wave instruction = 1 KB contiguous, rows 8 KB apart)
//   node_nt  : the same with non-temporal stores
//   inst     : one workgroup per instance: 256 threads x 4 nodes, every entry row written as 2 x 4 KB: the
//              workgroup's stores sweep one contiguous 928 KB region
//   node_rd  : node + the kernel's 8 input loads per thread (X, U rows) in front
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int M = 1024, B = 1024, R = 116;

__global__ void k_linear(double2* buf, size_t n2) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) buf[i] = make_double2((double)i, 1.0);
}
template <bool NT, bool RD>
__global__ __launch_bounds__(256) void k_node(double* buf, const double* in) {
    const int b = blockIdx.y, k0 = (blockIdx.x * 256 + threadIdx.x) * 2;
    double acc = 0;
    if (RD) {
#pragma unroll
        for (int v = 0; v < 8; ++v) { const double2 t = *reinterpret_cast<const double2*>(in + ((size_t)b * 8 + v) * M + k0); acc += t.x + t.y; }
    }
    double* p = buf + (size_t)b * R * M + k0;
#pragma unroll 4
    for (int e = 0; e < R; ++e) {
        double2 v = make_double2(acc + e, (double)k0);
        if (NT) { __builtin_nontemporal_store(v.x, p + (size_t)e * M); __builtin_nontemporal_store(v.y, p + (size_t)e * M + 1); }
        else *reinterpret_cast<double2*>(p + (size_t)e * M) = v;
    }
}
__global__ __launch_bounds__(256) void k_inst(double* buf) {
    const int b = blockIdx.x;
    double* p = buf + (size_t)b * R * M + threadIdx.x * 2;
#pragma unroll 4
    for (int e = 0; e < R; ++e) {
        *reinterpret_cast<double2*>(p + (size_t)e * M) = make_double2((double)e, 1.0);
        *reinterpret_cast<double2*>(p + (size_t)e * M + 512) = make_double2((double)e, 2.0);
    }
}
int main() {
    const size_t n = (size_t)B * R * M;
    double *buf, *in;
    CK(hipMalloc(&buf, n * 8)); CK(hipMalloc(&in, (size_t)B * 8 * M * 8)); CK(hipMemset(in, 0, (size_t)B * 8 * M * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[5] = {"linear", "node", "node_nt", "inst", "node_rd"};
    for (int round = 0; round < 3; ++round)
        for (int v = 0; v < 5; ++v) {
            auto launch = [&] {
                switch (v) {
                    case 0: hipLaunchKernelGGL(k_linear, dim3(2048), dim3(256), 0, 0, (double2*)buf, n / 2); break;
                    case 1: hipLaunchKernelGGL((k_node<false, false>), dim3(M / 512, B), dim3(256), 0, 0, buf, in); break;
                    case 2: hipLaunchKernelGGL((k_node<true, false>), dim3(M / 512, B), dim3(256), 0, 0, buf, in); break;
                    case 3: hipLaunchKernelGGL(k_inst, dim3(B), dim3(256), 0, 0, buf); break;
                    case 4: hipLaunchKernelGGL((k_node<false, true>), dim3(M / 512, B), dim3(256), 0, 0, buf, in); break;
                }
            };
            for (int i = 0; i < 5; ++i) launch();
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 40; ++i) launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 40;
            printf("{\"pattern\": \"%s\", \"round\": %d, \"ms\": %.4f, \"store_GBs\": %.1f}\n", names[v], round, ms, n * 8 / (ms * 1e-3) / 1e9);
        }
    return 0;
}
