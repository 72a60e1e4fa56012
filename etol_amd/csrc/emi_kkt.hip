// emi_kkt.hip -- the Newton step of the NLP iteration on the device.
//
// ePSOPT leaves this step to IPOPT (reference src/ePSOPT/ePSOPT.cpp:62-66, 84): a sparse symmetric
// factorisation of the primal-dual KKT matrix per iteration, on one CPU core.  For the global
// pseudospectral transcription that matrix is
//
//        [ Q    J^T ]     Q  = node-block-diagonal (Lagrangian Hessian + barrier terms, nv x nv per node)
//    K = [ J   -dc I ]    J  = D (x) [I_ns 0]  -  h [f_x f_u]_k  : M x M dense coupling per state
//
// with N = (nv + ns) M rows (14336 at 1024 nodes of the 6-state model).  Here it is assembled
// in HBM straight from the node blocks the evaluator kernels produce (one kernel, every entry written
// exactly once, no memset), factorised with rocSOLVER's LU (the matrix is symmetric indefinite;
// partial pivoting is at least as stable as Bunch-Kaufman and the library's getrf is its fastest
// dense factorisation), and solved per right-hand side.  LU gives no inertia: the caller (emi_nlp.cpp)
// applies the curvature test of inertia-free interior-point methods instead.
// Variables the caller marks fixed keep their slot: row and column are replaced by the identity.
#include <hip/hip_runtime.h>
#include <rocsolver/rocsolver.h>

#include <string>

#include "emi_kernels.hpp"

namespace emi {

struct KktWorkspace {
    rocblas_handle handle = nullptr;
    double* K = nullptr;        // [N][N] column-major (symmetric before the factorisation)
    size_t K_elems = 0;
    rocblas_int* ipiv = nullptr;
    rocblas_int* info = nullptr;
    double* Q = nullptr;        // [nh][M]
    double* J = nullptr;        // [ns*nv][M]
    double* rhs = nullptr;      // [N][nrhs] column-major
    size_t rhs_elems = 0;
    unsigned char* fixed = nullptr;   // [nv*M]
    size_t cap_small = 0;       // elements the small buffers were sized for (N)
    int N = 0;
    bool factored = false;
};

namespace {

// One thread per entry (r, c) of K; c is the fast index of the thread grid, K is symmetric so the
// column-major store below is coalesced.
__global__ __launch_bounds__(256) void emi_kkt_assemble_kernel(double* __restrict__ K, const double* __restrict__ Q,
                                                              const double* __restrict__ J, const double* __restrict__ D,
                                                              const unsigned char* __restrict__ fixed, int M, int ns,
                                                              int nv, double dc) {
    const int N = (nv + ns) * M, nz = nv * M;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (c >= N) return;
    double val = 0.0;
    const bool fr = r < nz && fixed[r], fc = c < nz && fixed[c];
    if (fr || fc) {
        val = r == c ? 1.0 : 0.0;
    } else if (r < nz && c < nz) {
        const int v = r / M, k = r - v * M, q = c / M, j = c - q * M;
        if (j == k) {
            const int hi = v > q ? v : q, lo = v > q ? q : v;
            val = Q[(size_t)(hi * (hi + 1) / 2 + lo) * M + k];
        }
    } else if (r >= nz && c >= nz) {
        val = r == c ? -dc : 0.0;
    } else {
        // constraint row R (state i, node k) against variable (v, node j)
        const int R = (r >= nz ? r : c) - nz, V = r >= nz ? c : r;
        const int i = R / M, k = R - i * M, v = V / M, j = V - v * M;
        if (j == k) val = J[(size_t)(i * nv + v) * M + k];       // -h df_i/dz_v (+ D_kk when v == i)
        else if (v == i) val = D[(size_t)k * M + j];
    }
    K[(size_t)r * N + c] = val;    // K symmetric: row-major position == column-major position of the transpose
}

__global__ void emi_kkt_mask_rhs_kernel(double* __restrict__ rhs, const unsigned char* __restrict__ fixed, int nz, int N) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nz && fixed[q]) rhs[(size_t)blockIdx.y * N + q] = 0.0;
}

const char* rb(rocblas_status s) { return rocblas_status_to_string(s); }

}  // namespace

void kkt_destroy(KktWorkspace* w) {
    if (!w) return;
    if (w->handle) (void)rocblas_destroy_handle(w->handle);
    void* bufs[] = {w->K, w->ipiv, w->info, w->Q, w->J, w->rhs, w->fixed};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    delete w;
}

// Returns an EMI_* status; *info = 0 factorised, > 0 exactly singular (zero pivot at that position).
int kkt_factor(KktWorkspace** pw, hipStream_t stream, const double* dD, int M, int ns, int nv, const double* Qblk,
               const double* Jblk, const unsigned char* fixed, double dc, int* info, std::string* err) {
    const int nh = nv * (nv + 1) / 2, N = (nv + ns) * M, nz = nv * M;
    if (!*pw) *pw = new KktWorkspace();
    KktWorkspace* w = *pw;
    w->factored = false;
#define KKT_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) { *err = std::string(#call) + ": " + hipGetErrorString(e_); return EMI_ERR_HIP; } \
    } while (0)
#define KKT_RB(call)                                                                       \
    do {                                                                                   \
        rocblas_status s_ = (call);                                                        \
        if (s_ != rocblas_status_success) { *err = std::string(#call) + ": " + rb(s_); return EMI_ERR_HIP; } \
    } while (0)
    if (!w->handle) KKT_RB(rocblas_create_handle(&w->handle));
    KKT_RB(rocblas_set_stream(w->handle, stream));
    if (w->K_elems < (size_t)N * N) {
        if (w->K) KKT_HIP(hipFree(w->K));
        w->K = nullptr;
        w->K_elems = 0;
        KKT_HIP(hipMalloc(&w->K, (size_t)N * N * sizeof(double)));
        w->K_elems = (size_t)N * N;
    }
    if (w->cap_small < (size_t)N) {
        void** small[] = {(void**)&w->ipiv, (void**)&w->info, (void**)&w->Q, (void**)&w->J, (void**)&w->fixed};
        for (void** b : small)
            if (*b) { KKT_HIP(hipFree(*b)); *b = nullptr; }
        w->cap_small = 0;
        KKT_HIP(hipMalloc(&w->ipiv, (size_t)N * sizeof(rocblas_int)));
        KKT_HIP(hipMalloc(&w->info, sizeof(rocblas_int)));
        KKT_HIP(hipMalloc(&w->Q, (size_t)nh * M * sizeof(double)));
        KKT_HIP(hipMalloc(&w->J, (size_t)ns * nv * M * sizeof(double)));
        KKT_HIP(hipMalloc(&w->fixed, (size_t)nz));
        w->cap_small = (size_t)N;
    }
    w->N = N;
    KKT_HIP(hipMemcpyAsync(w->Q, Qblk, (size_t)nh * M * sizeof(double), hipMemcpyHostToDevice, stream));
    KKT_HIP(hipMemcpyAsync(w->J, Jblk, (size_t)ns * nv * M * sizeof(double), hipMemcpyHostToDevice, stream));
    KKT_HIP(hipMemcpyAsync(w->fixed, fixed, (size_t)nz, hipMemcpyHostToDevice, stream));
    dim3 grid((N + 255) / 256, N), block(256);
    hipLaunchKernelGGL(emi_kkt_assemble_kernel, grid, block, 0, stream, w->K, w->Q, w->J, dD, w->fixed, M, ns, nv, dc);
    KKT_HIP(hipGetLastError());
    KKT_RB(rocsolver_dgetrf(w->handle, N, N, w->K, N, w->ipiv, w->info));
    rocblas_int hinfo = 0;
    KKT_HIP(hipMemcpyAsync(&hinfo, w->info, sizeof hinfo, hipMemcpyDeviceToHost, stream));
    KKT_HIP(hipStreamSynchronize(stream));
    *info = (int)hinfo;
    w->factored = hinfo == 0;
    return EMI_OK;
}

int kkt_solve(KktWorkspace* w, hipStream_t stream, int nz, double* rhs, int nrhs, std::string* err) {
    if (!w || !w->factored) { *err = "emi_kkt_solve: no factorisation (emi_kkt_factor must succeed first)"; return EMI_ERR_STATE; }
    const int N = w->N;
    const size_t elems = (size_t)N * nrhs;
    if (w->rhs_elems < elems) {
        if (w->rhs) KKT_HIP(hipFree(w->rhs));
        w->rhs = nullptr;
        w->rhs_elems = 0;
        KKT_HIP(hipMalloc(&w->rhs, elems * sizeof(double)));
        w->rhs_elems = elems;
    }
    KKT_HIP(hipMemcpyAsync(w->rhs, rhs, elems * sizeof(double), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(emi_kkt_mask_rhs_kernel, dim3((nz + 255) / 256, nrhs), dim3(256), 0, stream, w->rhs, w->fixed, nz, N);
    KKT_HIP(hipGetLastError());
    KKT_RB(rocblas_set_stream(w->handle, stream));
    KKT_RB(rocsolver_dgetrs(w->handle, rocblas_operation_none, N, nrhs, w->K, N, w->ipiv, w->rhs, N));
    KKT_HIP(hipMemcpyAsync(rhs, w->rhs, elems * sizeof(double), hipMemcpyDeviceToHost, stream));
    KKT_HIP(hipStreamSynchronize(stream));
    return EMI_OK;
#undef KKT_HIP
#undef KKT_RB
}

}  // namespace emi
