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
// hands over a quasi-definite matrix (every Q block positive definite), whose inertia is known.
// Variables the caller marks fixed keep their slot: row and column are replaced by the identity.
//
// Method 1 ("schur") uses that structure instead of treating K as a general matrix.  With Q positive
// definite and block diagonal,
//     S = J Q^-1 J^T + dc I            (ns M rows instead of (nv+ns) M, symmetric positive definite)
// and with J = Doff (x) [I 0] + blockdiag(J_k), Doff = D without its diagonal, P_k = Q_k^-1:
//     S_(i,i') = Doff diag(P_k[i][i']) Doff^T                      one M^3 GEMM per state pair i >= i'
//              + Doff o (1 g^T) + (Doff o (1 g'^T))^T + diag(r)    element-wise, from P_k J_k^T and J_k P_k J_k^T
// -> Cholesky (rocSOLVER potrf: 31 ms at 8192 rows where the LU of the full matrix takes 208 ms at
// 14336), and a solve is two GEMMs with D, two triangular solves and node-local 8x8 products.
#include <atomic>
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocsolver/rocsolver.h>

#include <cstdio>
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "emi_kernels.hpp"

namespace emi {

// Diagnostic switches of the factorisation (process-wide, set through emi_set_option "kkt_*"; the defaults are what
// every number in profiles/ was measured with)
struct KktTuning {
    std::atomic<int> chol_outer{768};       // "kkt_chol_outer": columns of an outer panel of the two-level Cholesky
    std::atomic<int> own_cholesky{2};       // "kkt_cholesky": 2 the library's blocked Cholesky in two-level form from 1024 rows (default), 1 one level, 0 rocsolver_dpotrf (+ confirmation on a copy)
    std::atomic<int> own_diag{2};           // "kkt_chol_diag": 2 the diagonal block by one wave with matrix-pipe updates (default), 1 the 256-thread column-by-column kernel
    std::atomic<int> own_panel{2};          // "kkt_chol_panel": 2 own panel kernel on the matrix pipe (default), 1 its scalar form, 0 rocblas_dtrsm
    std::atomic<int> batched_max_nodes{256};// "kkt_batched_max_nodes": largest mesh with the batched Schur-block build
    std::atomic<int> debug{0};              // "kkt_debug": 1 retries and fallbacks on stderr, 2 also the blocks around a failing pivot
    std::atomic<int> potrf_lock{0};         // "kkt_potrf_lock": serialise rocsolver_dpotrf calls of different host threads
    std::atomic<int> sticky_reg{1};         // "kkt_sticky_reg": start the Schur path at the regularisation level that worked last on this mesh
    std::atomic<int> block_trsv{1};         // "kkt_block_trsv": 1 single right-hand sides through the block-inverse triangular solves below (gemv form), 2 the same with the library's own diagonal-block kernel, 0 rocsolver_dpotrs (trsv)
    std::atomic<int> primal_levels{1};      // "kkt_primal_levels": 1 primal regularisation levels behind the dual ones before the LU fallback, 0 the round-2 ladder
    // batched entry points: from this many rows of the Schur complement on, the rocBLAS calls of a batch go out once per scenario
    // (plain dgemm / dsyrk / dtrtri: large problems, where the pointer-array forms of the library run far below the plain ones)
    // instead of as *_batched calls (small problems, where a launch per scenario is what a batch is there to avoid)
    std::atomic<int> batch_gemm_rows{1 << 30};  // "kkt_batch_gemm_rows" (measured: rocblas_dgemm_batched holds up at 6144 rows -- 5.4 against 6.3 ms per scenario at 8 scenarios of 1024 nodes -- while
                                              // rocblas_dsyrk_batched takes 121 ms where dsyrk takes a few: profiles/r04_kkt_times.jsonl)
    std::atomic<int> batch_syrk_rows{3072};   // "kkt_batch_syrk_rows"
    std::atomic<int> batch_trtri_rows{3072};  // "kkt_batch_trtri_rows"
    // "kkt_refine_exp": the device refinement of a Newton step stops once its residual is below 10^-this of the right-hand side (largest
    // entry, at least 1).  10 is IPOPT's residual_ratio_max; rounds 2 - 4 refined to 1e-14, i.e. to round-off.  64 x 1024-node Monte-Carlo
    // set, solves per iteration on the 1024-node rung / solves per second (profiles/r04_montecarlo_refinement_depth.jsonl): 14: 4.51 / 3.32,
    // 12: 4.40 / 3.31, 10: 3.78 / 3.43, 8: 3.02 / 3.91 -- with 129 - 131 iterations per scenario and all 64 solved at every depth
    std::atomic<int> refine_exp{10};
};
static KktTuning g_tune;
bool kkt_set_option(const char* name, int value) {
    if (!strcmp(name, "kkt_chol_outer")) { g_tune.chol_outer = value; return true; }
    if (!strcmp(name, "kkt_cholesky")) { g_tune.own_cholesky = value < 0 ? 0 : (value > 2 ? 2 : value); return true; }
    if (!strcmp(name, "kkt_chol_diag")) { g_tune.own_diag = value; return true; }
    if (!strcmp(name, "kkt_chol_panel")) { g_tune.own_panel = value < 0 ? 0 : (value > 2 ? 2 : value); return true; }
    if (!strcmp(name, "kkt_batched_max_nodes")) { g_tune.batched_max_nodes = value; return true; }
    if (!strcmp(name, "kkt_debug")) { g_tune.debug = value; return true; }
    if (!strcmp(name, "kkt_potrf_lock")) { g_tune.potrf_lock = value != 0; return true; }
    if (!strcmp(name, "kkt_sticky_reg")) { g_tune.sticky_reg = value != 0; return true; }
    if (!strcmp(name, "kkt_block_trsv")) { g_tune.block_trsv = value < 0 ? 0 : (value > 2 ? 2 : value); return true; }
    if (!strcmp(name, "kkt_primal_levels")) { g_tune.primal_levels = value != 0; return true; }
    if (!strcmp(name, "kkt_batch_gemm_rows")) { g_tune.batch_gemm_rows = value; return true; }
    if (!strcmp(name, "kkt_batch_syrk_rows")) { g_tune.batch_syrk_rows = value; return true; }
    if (!strcmp(name, "kkt_batch_trtri_rows")) { g_tune.batch_trtri_rows = value; return true; }
    if (!strcmp(name, "kkt_refine_exp")) { g_tune.refine_exp = value < 6 ? 6 : (value > 16 ? 16 : value); return true; }
    return false;
}

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
    size_t cap_ipiv = 0, cap_Q = 0, cap_J = 0, cap_fixed = 0;     // bytes allocated (contexts are reused across problems)
    int N = 0;
    bool factored = false;
    // method 1 (Schur complement + Cholesky)
    int method_used = 0;        // what the current factorisation is: 0 LU of K, 1 Cholesky of S
    int M = 0, ns = 0, nv = 0;
    double* S = nullptr;        // [md][md] column-major, lower triangle
    size_t S_elems = 0;
    double* Pinv = nullptr;     // [nv*nv][M]  Q_k^-1 (zero rows/columns for fixed variables)
    double* G = nullptr;        // [ns*ns][M]  (P_k J_k^T) state rows
    double* Rk = nullptr;       // [ns*ns][M]  J_k P_k J_k^T
    double* Doff = nullptr;     // [M][M] D without its diagonal (same storage order as D)
    double* W = nullptr;        // [pairs][M][M] scaled copies of Doff, one per state pair (one [M][M] in the unbatched form)
    double** gemm_ptrs = nullptr;   // device: [3][pairs] operand pointers of the batched GEMM (A: Doff, B: W_p, C: S block)
    const double* ptrs_key[3] = {nullptr, nullptr, nullptr};   // (Doff, W, S) the pointer table was built for
    int ptrs_M = 0, ptrs_ns = 0;
    double* T = nullptr;        // [nz][nrhs] work
    double* Cb = nullptr;       // [md][nrhs] work
    size_t T_elems = 0, Cb_elems = 0;
    size_t cap_Pinv = 0, cap_G = 0, cap_Rk = 0, cap_Doff = 0, cap_W = 0;
    int* flag = nullptr;        // node kernel: a block was not positive definite
    // dual-regularisation level the Schur path starts from (0: nominal, 1: x1e3, 2: x1e6): late interior-point iterations
    // on one mesh fail at the same levels again and again, and every failed level costs a build of S and a Cholesky.
    // Kept per mesh shape; after two successes in a row one level lower is tried again.  The LU is never the starting point:
    // a wandering solve (inertia search: many trial matrices per iteration) stuck there with 209 of 585 factorisations
    // at 85 ms each where the Cholesky path, tried, takes 10 ms (513 nodes, profiles/r02_notes.md section 12).
    int reg_level = 0, reg_hits = 0, reg_M = 0, reg_ns = 0, reg_nv = 0;
    double reg_dc_applied = 0.0, reg_dw_applied = 0.0;    // what the current factorisation really holds (kkt_last_regularisation)
    // single-right-hand-side solves with the Cholesky factor of S (blk_potrs): inverses of its 512 x 512 diagonal blocks
    double* Linv = nullptr;        // [nblk][512][512] column-major, zeros above the diagonal
    size_t cap_Linv = 0;
    double* LinvT = nullptr;       // the same blocks transposed (the forward sweep multiplies with op T as well)
    size_t cap_LinvT = 0;
    int linv_n = 0;                // order of the factor the inverses belong to (0: none -- rocsolver_dpotrs is used)
    double* trsv_tmp = nullptr;    // [512] x_j while its block is being multiplied
    size_t cap_trsv_tmp = 0;
    double* trsv_y = nullptr;      // [n] forward-sweep solution of the gemv form of blk_potrs
    size_t cap_trsv_y = 0;
    double* trsm_y = nullptr;      // [n][nrhs] the same for many right-hand sides (blk_potrs_multi)
    size_t cap_trsm_y = 0;
    double* chol_blk = nullptr;    // [64][64] + [64]: factorised diagonal block and reciprocal diagonal of the current block column
    double* chol_copy = nullptr;   // the matrix handed to dpotrf, kept until the factorisation is confirmed (potrf_checked)
    size_t cap_chol_copy = 0;
    // low-rank correction (kkt_lowrank)
    bool lr_active = false;
    int lr_r = 0, lr_cap = 0, lr_n = 0;
    double* lrY = nullptr;      // [N][r]  K~^-1 U
    double* lrC = nullptr;      // [r][r]  Cholesky factor of Delta^-1 - U^T Y
    double* lrT = nullptr;      // [r][<=64] work
    int* lr_node = nullptr;
    double* lr_vec = nullptr;   // [r][nv]
    double* lr_delta = nullptr;
    // refined solves (kkt_solve_refined_batch): right-hand side, solution, previous solution on the device
    double *ref_b = nullptr, *ref_x = nullptr, *ref_p = nullptr;
    size_t cap_ref_b = 0, cap_ref_x = 0, cap_ref_p = 0;
    // Doff is a function of the mesh only: rebuilt when the mesh changes (kkt_mesh_changed), not at every factorisation
    const double* doff_src = nullptr;
    int doff_M = 0;
    // batched entry points (kkt_factor_batch / kkt_solve_batch): scratch of the workspace that LEADS a batch
    void* b_tab = nullptr;      // device: KktDev[n]
    double** b_ptrs = nullptr;  // device: pointer arrays of the batched rocBLAS calls
    int* b_stat = nullptr;      // device: [2 n] info words, then block flags
    char* b_pin = nullptr;      // pinned host staging of all three
    size_t cap_b_tab = 0, cap_b_ptrs = 0, cap_b_stat = 0, cap_b_pin = 0;
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

static __device__ __forceinline__ void emi_kkt_mask_rhs_kernel_body(double* __restrict__ rhs, const unsigned char* __restrict__ fixed, int nz, int N, int c_in) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nz && fixed[q]) rhs[(size_t)c_in * N + q] = 0.0;
}


// ---- method 1 kernels ----------------------------------------------------------------------
#define KKT_NV_MAX 16
// One thread per node: P_k = Q_k^-1 over the free variables (Cholesky), G_k = (P_k J_k^T)[states], R_k = J_k P_k J_k^T.
static __device__ __forceinline__ void emi_kkt_node_inverse_kernel_body(const double* __restrict__ Q, const double* __restrict__ J,
                                                                 const unsigned char* __restrict__ fixed, int M, int ns,
                                                                 int nv, double* __restrict__ Pinv, double* __restrict__ G,
                                                                 double* __restrict__ Rk, int* __restrict__ flag, double dw) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M) return;
    double A[KKT_NV_MAX][KKT_NV_MAX], X[KKT_NV_MAX][KKT_NV_MAX];
    bool fx[KKT_NV_MAX];
    for (int v = 0; v < nv; ++v) fx[v] = fixed[v * M + k] != 0;
    for (int v = 0; v < nv; ++v)
        for (int q = 0; q <= v; ++q)
            A[v][q] = (fx[v] || fx[q]) ? (v == q ? 1.0 : 0.0) : Q[(size_t)(v * (v + 1) / 2 + q) * M + k] + ((v == q && v < ns) ? dw : 0.0);
    bool ok = true;
    if (dw > 0.0) {
        // the verdict "this block is positive definite" is about the matrix the CALLER built, not about the shifted one: a block that
        // is indefinite by less than dw must not pass as quasi-definite because a sticky regularisation level starts with dw > 0
        double L0[KKT_NV_MAX][KKT_NV_MAX];
        for (int i = 0; i < nv; ++i)
            for (int j = 0; j <= i; ++j) {
                double sum = A[i][j] - ((i == j && i < ns && !fx[i]) ? dw : 0.0);
                for (int t = 0; t < j; ++t) sum -= L0[i][t] * L0[j][t];
                if (i == j) {
                    if (!(sum > 0.0)) { ok = false; sum = 1.0; }
                    L0[i][i] = sqrt(sum);
                } else {
                    L0[i][j] = sum / L0[j][j];
                }
            }
    }
    for (int i = 0; i < nv; ++i)
        for (int j = 0; j <= i; ++j) {
            double sum = A[i][j];
            for (int t = 0; t < j; ++t) sum -= A[i][t] * A[j][t];
            if (i == j) {
                if (!(sum > 0.0)) { ok = false; sum = 1.0; }
                A[i][i] = sqrt(sum);
            } else {
                A[i][j] = sum / A[j][j];
            }
        }
    if (!ok) atomicExch(flag, 1);
    // X = A^-1 (A = L L^T): columns of the identity through forward and backward substitution
    for (int c = 0; c < nv; ++c) {
        double y[KKT_NV_MAX];
        for (int i = 0; i < nv; ++i) {
            double sum = i == c ? 1.0 : 0.0;
            for (int t = 0; t < i; ++t) sum -= A[i][t] * y[t];
            y[i] = sum / A[i][i];
        }
        for (int i = nv - 1; i >= 0; --i) {
            double sum = y[i];
            for (int t = i + 1; t < nv; ++t) sum -= A[t][i] * X[t][c];
            X[i][c] = sum / A[i][i];
        }
    }
    for (int v = 0; v < nv; ++v)
        for (int q = 0; q < nv; ++q) {
            const double val = (fx[v] || fx[q]) ? 0.0 : X[v][q];
            X[v][q] = val;
            Pinv[(size_t)(v * nv + q) * M + k] = val;
        }
    // PJ[v][i'] = sum_q P[v][q] J[i'][q]
    double PJ[KKT_NV_MAX][KKT_NV_MAX];
    for (int v = 0; v < nv; ++v)
        for (int ip = 0; ip < ns; ++ip) {
            double sum = 0;
            for (int q = 0; q < nv; ++q) sum += X[v][q] * J[(size_t)(ip * nv + q) * M + k];
            PJ[v][ip] = sum;
        }
    for (int i = 0; i < ns; ++i)
        for (int ip = 0; ip < ns; ++ip) {
            G[(size_t)(i * ns + ip) * M + k] = PJ[i][ip];
            double sum = 0;
            for (int v = 0; v < nv; ++v) sum += J[(size_t)(i * nv + v) * M + k] * PJ[v][ip];
            Rk[(size_t)(i * ns + ip) * M + k] = sum;
        }
}
// The same with the block sizes known at compile time: every loop unrolled, the 8 x 8 arrays in registers (the generic kernel above
// indexes its arrays with run-time bounds and runs out of 6.3 KB of scratch memory per thread: 252 us per launch whatever the mesh,
// 14 % of a factorisation at 256 nodes).  Same operations in the same order: bitwise the generic kernel's results.
template <int NS_, int NV_>
static __device__ __forceinline__ void emi_kkt_node_inverse_fixed_kernel_body(const double* __restrict__ Q, const double* __restrict__ J,
                                                                 const unsigned char* __restrict__ fixed, int M,
                                                                 double* __restrict__ Pinv, double* __restrict__ G,
                                                                 double* __restrict__ Rk, int* __restrict__ flag, double dw) {
    constexpr int ns = NS_, nv = NV_;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M) return;
    double A[NV_][NV_], X[NV_][NV_];
    bool fx[NV_];
    _Pragma("unroll") for (int v = 0; v < nv; ++v) fx[v] = fixed[v * M + k] != 0;
    _Pragma("unroll") for (int v = 0; v < nv; ++v)
        _Pragma("unroll") for (int q = 0; q <= v; ++q)
            A[v][q] = (fx[v] || fx[q]) ? (v == q ? 1.0 : 0.0) : Q[(size_t)(v * (v + 1) / 2 + q) * M + k] + ((v == q && v < ns) ? dw : 0.0);
    bool ok = true;
    if (dw > 0.0) {         // positive definiteness is judged on the caller's block, not on the shifted one (see the generic kernel)
        double L0[NV_][NV_];
        _Pragma("unroll") for (int i = 0; i < nv; ++i)
            _Pragma("unroll") for (int j = 0; j <= i; ++j) {
                double sum = A[i][j] - ((i == j && i < ns && !fx[i]) ? dw : 0.0);
                _Pragma("unroll") for (int t = 0; t < j; ++t) sum -= L0[i][t] * L0[j][t];
                if (i == j) {
                    if (!(sum > 0.0)) { ok = false; sum = 1.0; }
                    L0[i][i] = sqrt(sum);
                } else {
                    L0[i][j] = sum / L0[j][j];
                }
            }
    }
    _Pragma("unroll") for (int i = 0; i < nv; ++i)
        _Pragma("unroll") for (int j = 0; j <= i; ++j) {
            double sum = A[i][j];
            _Pragma("unroll") for (int t = 0; t < j; ++t) sum -= A[i][t] * A[j][t];
            if (i == j) {
                if (!(sum > 0.0)) { ok = false; sum = 1.0; }
                A[i][i] = sqrt(sum);
            } else {
                A[i][j] = sum / A[j][j];
            }
        }
    if (!ok) atomicExch(flag, 1);
    // X = A^-1 (A = L L^T): columns of the identity through forward and backward substitution
    _Pragma("unroll") for (int c = 0; c < nv; ++c) {
        double y[NV_];
        _Pragma("unroll") for (int i = 0; i < nv; ++i) {
            double sum = i == c ? 1.0 : 0.0;
            _Pragma("unroll") for (int t = 0; t < i; ++t) sum -= A[i][t] * y[t];
            y[i] = sum / A[i][i];
        }
        _Pragma("unroll") for (int i = nv - 1; i >= 0; --i) {
            double sum = y[i];
            _Pragma("unroll") for (int t = i + 1; t < nv; ++t) sum -= A[t][i] * X[t][c];
            X[i][c] = sum / A[i][i];
        }
    }
    _Pragma("unroll") for (int v = 0; v < nv; ++v)
        _Pragma("unroll") for (int q = 0; q < nv; ++q) {
            const double val = (fx[v] || fx[q]) ? 0.0 : X[v][q];
            X[v][q] = val;
            Pinv[(size_t)(v * nv + q) * M + k] = val;
        }
    // PJ[v][i'] = sum_q P[v][q] J[i'][q]
    double PJ[NV_][NS_];
    _Pragma("unroll") for (int v = 0; v < nv; ++v)
        _Pragma("unroll") for (int ip = 0; ip < ns; ++ip) {
            double sum = 0;
            _Pragma("unroll") for (int q = 0; q < nv; ++q) sum += X[v][q] * J[(size_t)(ip * nv + q) * M + k];
            PJ[v][ip] = sum;
        }
    _Pragma("unroll") for (int i = 0; i < ns; ++i)
        _Pragma("unroll") for (int ip = 0; ip < ns; ++ip) {
            G[(size_t)(i * ns + ip) * M + k] = PJ[i][ip];
            double sum = 0;
            _Pragma("unroll") for (int v = 0; v < nv; ++v) sum += J[(size_t)(i * nv + v) * M + k] * PJ[v][ip];
            Rk[(size_t)(i * ns + ip) * M + k] = sum;
        }
}

// Doff = D with zero diagonal (same memory order as D: [k][j] row-major)
__global__ void emi_kkt_doff_kernel(const double* __restrict__ D, double* __restrict__ Doff, int M) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)M * M) return;
    const int k = (int)(idx / M), j = (int)(idx - (size_t)k * M);
    Doff[idx] = j == k ? 0.0 : D[idx];
}
// W[k][j] = Doff[k][j] * p[j]
static __device__ __forceinline__ void emi_kkt_scale_kernel_body(const double* __restrict__ Doff, const double* __restrict__ p, double* __restrict__ W,
                                     int M) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)M * M) return;
    const int j = (int)(idx % M);
    W[idx] = Doff[idx] * p[j];
}
// all state pairs at once (blockIdx.y = pair p <-> (i, ip), ip <= i): W_p[k][j] = Doff[k][j] * P[(i, ip)][j]
__global__ void emi_kkt_scale_all_kernel(const double* __restrict__ Doff, const double* __restrict__ Pinv, double* __restrict__ W,
                                         int M, int nv) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)M * M) return;
    int i = 0;
    while ((i + 1) * (i + 2) / 2 <= (int)blockIdx.y) ++i;
    const int ip = (int)blockIdx.y - i * (i + 1) / 2;
    const int j = (int)(idx % M);
    W[(size_t)blockIdx.y * M * M + idx] = Doff[idx] * Pinv[(size_t)(i * nv + ip) * M + j];
}
// element-wise terms of one state-pair block of S (column-major big matrix, rows i*M+k, columns ip*M+kp)
static __device__ __forceinline__ void emi_kkt_sblock_terms_kernel_body(double* __restrict__ S, const double* __restrict__ Doff,
                                            const double* __restrict__ G, const double* __restrict__ Rk, int M, int ns, int i,
                                            int ip, double dc, double rel, int pair_in) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)M * M) return;
    if (i < 0) {                                   // batched launch: blockIdx.y = pair index
        i = 0;
        while ((i + 1) * (i + 2) / 2 <= pair_in) ++i;
        ip = pair_in - i * (i + 1) / 2;
    }
    const int kp = (int)(idx / M), k = (int)(idx - (size_t)kp * M);      // k fast: coalesced along a column of S
    const size_t md = (size_t)ns * M;
    double add = Doff[(size_t)k * M + kp] * G[(size_t)(i * ns + ip) * M + kp] +
                 Doff[(size_t)kp * M + k] * G[(size_t)(ip * ns + i) * M + k];
    if (k == kp) add += Rk[(size_t)(i * ns + ip) * M + k];
    double* dst = S + ((size_t)ip * M + kp) * md + (size_t)i * M + k;
    const double val = *dst + add;
    *dst = (k == kp && i == ip) ? val * (1.0 + rel) + dc : val;
}
// out[(v,k),c] = sum_q P_k[v][q] in[(q,k),c]      (in/out: column stride ld_in / ld_out, nz rows used)
static __device__ __forceinline__ void emi_kkt_apply_p_kernel_body(const double* __restrict__ Pinv, const double* __restrict__ in, size_t ld_in,
                                       double* __restrict__ out, size_t ld_out, int M, int nv, int c_in) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = c_in;
    if (k >= M) return;
    double x[KKT_NV_MAX];
    for (int q = 0; q < nv; ++q) x[q] = in[(size_t)c * ld_in + (size_t)q * M + k];
    for (int v = 0; v < nv; ++v) {
        double sum = 0;
        for (int q = 0; q < nv; ++q) sum += Pinv[(size_t)(v * nv + q) * M + k] * x[q];
        out[(size_t)c * ld_out + (size_t)v * M + k] = sum;
    }
}
// Cb[(i,k),c] += sum_v J_k[i][v] t[(v,k),c] - b[(i,k),c]
static __device__ __forceinline__ void emi_kkt_jnode_minus_b_kernel_body(const double* __restrict__ J, const double* __restrict__ t, size_t ld_t,
                                             const double* __restrict__ b, size_t ld_b, double* __restrict__ Cb, size_t ld_c,
                                             int M, int ns, int nv, int c_in) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = c_in;
    if (k >= M) return;
    for (int i = 0; i < ns; ++i) {
        double sum = 0;
        for (int v = 0; v < nv; ++v) sum += J[(size_t)(i * nv + v) * M + k] * t[(size_t)c * ld_t + (size_t)v * M + k];
        Cb[(size_t)c * ld_c + (size_t)i * M + k] += sum - b[(size_t)c * ld_b + (size_t)i * M + k];
    }
}
// y[(v,k),c] -= sum_i J_k[i][v] lam[(i,k),c]
static __device__ __forceinline__ void emi_kkt_jnode_t_kernel_body(const double* __restrict__ J, const double* __restrict__ lam, size_t ld_l,
                                       double* __restrict__ y, size_t ld_y, int M, int ns, int nv, int c_in) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = c_in;
    if (k >= M) return;
    for (int v = 0; v < nv; ++v) {
        double sum = 0;
        for (int i = 0; i < ns; ++i) sum += J[(size_t)(i * nv + v) * M + k] * lam[(size_t)c * ld_l + (size_t)i * M + k];
        y[(size_t)c * ld_y + (size_t)v * M + k] -= sum;
    }
}
// rhs[(i,k) + nz, c] = lam[(i,k), c]
static __device__ __forceinline__ void emi_kkt_copy_lambda_kernel_body(const double* __restrict__ lam, size_t ld_l, double* __restrict__ rhs, size_t ld_r,
                                           int nz, int md, int c_in) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = c_in;
    if (r < md) rhs[(size_t)c * ld_r + nz + r] = lam[(size_t)c * ld_l + r];
}

// ---- low-rank correction kernels ---------------------------------------------------------------
// U[(v, node_c), c] = vec_c[v]
__global__ void emi_kkt_lr_scatter_kernel(double* __restrict__ U, const int* __restrict__ node, const double* __restrict__ vec,
                                          int r, int N, int M, int nv) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= r) return;
    for (int v = 0; v < nv; ++v) U[(size_t)c * N + (size_t)v * M + node[c]] = vec[(size_t)c * nv + v];
}
// C[a][c] = (a == c) / delta_a - sum_v vec_a[v] Y[(v, node_a), c]      (column-major r x r, lower triangle used)
__global__ void emi_kkt_lr_c_kernel(double* __restrict__ Cm, const double* __restrict__ Y, const int* __restrict__ node,
                                    const double* __restrict__ vec, const double* __restrict__ delta, int r, int N, int M, int nv) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y * blockDim.y + threadIdx.y;
    if (a >= r || c >= r) return;
    double dot = 0;
    const double* y = Y + (size_t)c * N + node[a];
    for (int v = 0; v < nv; ++v) dot += vec[(size_t)a * nv + v] * y[(size_t)v * M];
    Cm[(size_t)c * r + a] = (a == c ? 1.0 / delta[a] : 0.0) - dot;
}
// T[a, c] = sum_v vec_a[v] X[(v, node_a), c]
__global__ void emi_kkt_lr_utx_kernel(double* __restrict__ T, const double* __restrict__ X, const int* __restrict__ node,
                                      const double* __restrict__ vec, int r, int N, int M, int nv) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (a >= r) return;
    double dot = 0;
    const double* x = X + (size_t)c * N + node[a];
    for (int v = 0; v < nv; ++v) dot += vec[(size_t)a * nv + v] * x[(size_t)v * M];
    T[(size_t)c * r + a] = dot;
}

const char* rb(rocblas_status s) { return rocblas_status_to_string(s); }

// ---- blocked Cholesky (lower, column-major, in place) ---------------------------------------------------
// Right-looking with 64-column blocks: the diagonal block is factorised by one workgroup with the rows in registers,
// the panel below it is solved one row per thread against the block read through the scalar cache, the trailing
// matrix is updated by rocBLAS dsyrk.
#define CHOL_NB 64
// A[j0.., j0..] diagonal block of size nb <= 64; *info = first non-positive pivot (1-based, global), if none yet.
// Thread (r, g) = (lane, wave) holds row r, columns g, g+4, ..: per column one LDS hand-over of the finished column
// (double-buffered: one barrier), everything else is register arithmetic.  Lout: the factor once more, column-major
// [64][64] with zeros above the diagonal, then the 64 reciprocals of its diagonal -- what the panel kernel reads.
static __device__ __forceinline__ void emi_chol_diag_kernel_body(double* __restrict__ A, int lda, int j0, int nb, int* __restrict__ info,
                                                            double* __restrict__ Lout) {
    __shared__ double col[2][CHOL_NB];
    const int tid = threadIdx.x, r = tid & 63, g = tid >> 6;
    double* blk = A + (size_t)j0 * lda + j0;
    double a[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const int cc = g + 4 * m;
        a[m] = (r >= cc && r < nb) ? blk[(size_t)cc * lda + r] : (r == cc ? 1.0 : 0.0);
    }
#pragma unroll
    for (int c = 0; c < CHOL_NB; ++c) {
        const int gc = c & 3, mc = c >> 2;
        if (g == gc) col[c & 1][r] = a[mc];
        __syncthreads();
        if (c < nb) {
            double d = col[c & 1][c];
            if (!(d > 0.0)) {
                if (tid == 0 && *info == 0) *info = j0 + c + 1;
                d = 1.0;
            }
            const double inv = rsqrt(d), piv = d * inv;       // (one reciprocal square root instead of a square root and a division on the per-column chain)
            const double lrc = col[c & 1][r] * inv;
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                if (4 * m + 3 <= c) continue;              // no column of this slot lies right of c
                const int cc = g + 4 * m;
                if (cc > c && cc <= r) a[m] -= lrc * (col[c & 1][cc] * inv);
            }
            if (g == gc) a[mc] = r == c ? piv : (r > c ? lrc : 0.0);
        }
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const int cc = g + 4 * m;
        const double v = cc <= r ? a[m] : 0.0;
        if (r < nb && cc <= r) blk[(size_t)cc * lda + r] = v;
        Lout[cc * CHOL_NB + r] = v;
        if (cc == r) Lout[CHOL_NB * CHOL_NB + r] = 1.0 / v;
    }
}
// rows i >= j0 + 64 of the block column j0: x <- x L^-T with L the factorised 64 x 64 diagonal block (Lb, wave-uniform
// addresses: scalar loads, no LDS)
#ifndef EMI_PANEL_THREADS
#define EMI_PANEL_THREADS 64        // build-time experiment (tools/ab_build.sh): rows (threads) per workgroup of the panel kernel
#endif
__global__ __launch_bounds__(EMI_PANEL_THREADS) void emi_chol_panel_kernel(double* __restrict__ A, int lda, int n, int j0,
                                                           const double* __restrict__ Lb) {
    const int row = j0 + CHOL_NB + blockIdx.x * EMI_PANEL_THREADS + threadIdx.x;
    if (row >= n) return;
    double y[CHOL_NB];
    double* x = A + (size_t)j0 * lda + row;
#pragma unroll
    for (int c = 0; c < CHOL_NB; ++c) y[c] = x[(size_t)c * lda];
    // right-looking: once y_c is final it leaves every later column (independent FMAs, no dependent chain per column)
#pragma unroll
    for (int c = 0; c < CHOL_NB; ++c) {
        y[c] *= Lb[CHOL_NB * CHOL_NB + c];
#pragma unroll
        for (int t = c + 1; t < CHOL_NB; ++t) y[t] -= y[c] * Lb[c * CHOL_NB + t];
    }
#pragma unroll
    for (int c = 0; c < CHOL_NB; ++c) x[(size_t)c * lda] = y[c];
}

// The same panel solve on the matrix pipe.  With L in 16 x 16 blocks L_ab and the panel transposed, L Y^T = X^T is a blocked forward
// substitution:  Y_a^T = inv(L_aa) (X_a^T - sum_{b<a} L_ab Y_b^T)  -- ten 16 x 16 x 16 products per 16 panel rows, 40
// v_mfma_f64_16x16x4_f64, against 2080 dependent-latency FMAs per row in the scalar form above (28.7 us per launch at 6144 rows:
// every step waits for its scalar loads of the block).  Layouts of the instruction: A[row = lane % 16][k = lane / 16],
// B[k = lane / 16][col = lane % 16], C[row = lane / 16 + 4 i][col = lane % 16] in register i.  The k-step ks of a product takes
// k = lane / 16 + 4 ks (any assignment is fine as long as A and B agree), so register i of a finished block IS the B operand of
// k-step i of the next product: no shuffles between the ten products.  One wave per workgroup: 64 panel rows as four 16-row
// groups interleaved (independent accumulators); the inverses of the four diagonal 16 x 16 blocks are formed by the wave itself
// (a 16-step substitution per lane) and pass through LDS into the A layout.
typedef double chol_d4 __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ void emi_chol_panel_mfma_kernel_body(double* __restrict__ A, int lda, int n, int j0,
                                                                const double* __restrict__ Lb, int rowgrp) {
    __shared__ double inv_s[4][16][17];
    const int lane = threadIdx.x, r16 = lane & 15, kq = lane >> 4;
    {   // lane (a = kq, j = r16): column j of inv(L_aa); L_aa[r][c] = Lb[(16 a + c) * 64 + 16 a + r], reciprocal diagonal behind the block
        const double* La = Lb + (size_t)(16 * kq) * CHOL_NB + 16 * kq;
        const double* rd = Lb + CHOL_NB * CHOL_NB + 16 * kq;
        double x[16];                                       // right-looking: a finished x[c] leaves every later row at once (no long dependent chain)
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = r == r16 ? 1.0 : 0.0;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            x[c] *= rd[c];
#pragma unroll
            for (int r = c + 1; r < 16; ++r) x[r] -= La[(size_t)c * CHOL_NB + r] * x[c];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) inv_s[kq][r][r16] = x[r];
    }
    __syncthreads();
    // A operands: inv(L_aa) and -L_ab (a > b), element [r16][kq + 4 ks] for k-step ks
    double Ainv[4][4], Aoff[6][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) Ainv[a][ks] = inv_s[a][r16][kq + 4 * ks];
#pragma unroll
    for (int a = 1; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < a; ++b)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                Aoff[a * (a - 1) / 2 + b][ks] = -Lb[(size_t)(16 * b + kq + 4 * ks) * CHOL_NB + 16 * a + r16];
    // the panel: element (panel row, column 16 a + kq + 4 i) in T[g][a][i]
    const int row0 = j0 + CHOL_NB + rowgrp * 64;
    chol_d4 T[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        int row = row0 + 16 * g + r16;
        row = row < n ? row : n - 1;                       // (rows past the matrix: loaded from the last row, not stored)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i) T[g][a][i] = A[(size_t)(j0 + 16 * a + kq + 4 * i) * lda + row];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int b = 0; b < a; ++b)                        // X_a^T -= L_ab Y_b^T
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    T[g][a] = __builtin_amdgcn_mfma_f64_16x16x4f64(Aoff[a * (a - 1) / 2 + b][ks], T[g][b][ks], T[g][a], 0, 0, 0);
        chol_d4 Y[4];                                      // Y_a^T = inv(L_aa) (...)
#pragma unroll
        for (int g = 0; g < 4; ++g) Y[g] = chol_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int g = 0; g < 4; ++g) Y[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ainv[a][ks], T[g][a][ks], Y[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) T[g][a] = Y[g];
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int row = row0 + 16 * g + r16;
        if (row >= n) continue;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i) A[(size_t)(j0 + 16 * a + kq + 4 * i) * lda + row] = T[g][a][i];
    }
}

// The 64 x 64 diagonal block in the same 16 x 16 block form, by ONE wave: blocked right-looking Cholesky whose panel and trailing
// updates are matrix-pipe products (layouts as in emi_chol_panel_mfma_kernel: block (b, c) is held transposed, element (row 16 b +
// lane % 16, column 16 c + lane / 16 + 4 i) in register i, which makes a finished block A and B operand of the next products without
// any shuffle), and whose four 16 x 16 diagonal blocks are factorised inside the wave with lane shuffles (16 columns each instead of
// one 64-column chain with an LDS hand-over and a workgroup barrier per column: emi_chol_diag_kernel, 38 us per launch).
// Full blocks only (nb == 64); the last, partial block of a matrix goes through emi_chol_diag_kernel.  Outputs as that kernel's.
static __device__ __forceinline__ void emi_chol_diag_mfma_kernel_body(double* __restrict__ A, int lda, int j0, int* __restrict__ info,
                                                               double* __restrict__ Lout) {
    __shared__ double Ls[16][17];
    __shared__ double inv_s[16][17];
    const int lane = threadIdx.x, r16 = lane & 15, kq = lane >> 4;
    double* blk = A + (size_t)j0 * lda + j0;
    chol_d4 T[10];                                         // block (b, c), b >= c, at b (b + 1) / 2 + c
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int c = 0; c <= b; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 16 * b + r16, col = 16 * c + kq + 4 * i;
                T[b * (b + 1) / 2 + c][i] = row >= col ? blk[(size_t)col * lda + row] : 0.0;
            }
    int bad = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        chol_d4& D = T[a * (a + 1) / 2 + a];
        double rdiag[16];                                  // reciprocals of the diagonal of L_aa (wave-uniform)
        // 1. L_aa: lane (r = r16, q = kq) holds the columns q + 4 m of its row in D[m].  Four columns at a time (4 s .. 4 s + 3 = register s
        //    of the four lane groups): a column step updates only the rest of its group of four (one shuffle each for the pivot, the lane's
        //    row and the lane's column), and the columns behind the group get ONE rank-4 update, D -= L4 L4^T, which in this layout is a
        //    single instruction: register s is the A operand (row kappa = lane % 16, k = lane / 16) and the B operand (k, col r = lane % 16).
#pragma unroll
        for (int sb = 0; sb < 4; ++sb) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int c = 4 * sb + t;
                const double dsrc = D[sb];
                double d = __shfl(dsrc, c + 16 * t);
                if (!(d > 0.0)) {
                    if (!bad) bad = 16 * a + c + 1;
                    d = 1.0;
                }
                // the next pivot waits for this column's update, and the update needs 1 / d only (L[r][c] L[cc][c] = A[r][c] A[cc][c] / d):
                // a reciprocal with two Newton steps is on that chain, the reciprocal square root the column itself needs is not
                double rd = __builtin_amdgcn_rcp(d);
                rd = __builtin_fma(__builtin_fma(-d, rd, 1.0), rd, rd);
                rd = __builtin_fma(__builtin_fma(-d, rd, 1.0), rd, rd);
                const double inv = rsqrt(d), piv = d * inv;
                rdiag[c] = inv;
                const double ar = __shfl(dsrc, r16 + 16 * t);                  // A[r][c] of this lane's row (unscaled)
                if (t < 3) {
                    const int cc = 4 * sb + kq;                                // this lane's column of the group
                    const double acc = __shfl(dsrc, cc + 16 * t);              // A[cc][c]
                    if (kq > t) D[sb] -= (ar * acc) * rd;
                }
                if (kq == t) D[sb] = r16 == c ? piv : (r16 > c ? ar * inv : 0.0);
            }
            if (sb < 3) {
                const chol_d4 U = __builtin_amdgcn_mfma_f64_16x16x4f64(-D[sb], D[sb], D, 0, 0, 0);
#pragma unroll
                for (int i = sb + 1; i < 4; ++i) D[i] = U[i];
            }
        }
        // 2. inv(L_aa): column j = r16 of it per lane (the four lane groups do the same work), through LDS into the A layout
        __syncthreads();                                    // (the previous block's readers of Ls / inv_s are done)
#pragma unroll
        for (int m = 0; m < 4; ++m) Ls[r16][kq + 4 * m] = D[m];
        __syncthreads();
        {
            double x[16];                                   // (right-looking, as in the panel kernel)
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = r == r16 ? 1.0 : 0.0;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                x[c] *= rdiag[c];
#pragma unroll
                for (int r = c + 1; r < 16; ++r) x[r] -= Ls[r][c] * x[c];
            }
            if (kq == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) inv_s[r][r16] = x[r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (lane == c) Lout[CHOL_NB * CHOL_NB + 16 * a + c] = rdiag[c];
        if (a == 3) break;
        double Ainv[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) Ainv[ks] = inv_s[r16][kq + 4 * ks];
        // 3. the blocks below: L_ba^T = inv(L_aa) A_ba^T
#pragma unroll
        for (int b = a + 1; b < 4; ++b) {
            chol_d4 Y = chol_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) Y = __builtin_amdgcn_mfma_f64_16x16x4f64(Ainv[ks], T[b * (b + 1) / 2 + a][ks], Y, 0, 0, 0);
            T[b * (b + 1) / 2 + a] = Y;
        }
        // 4. trailing blocks: A_bc -= L_ba L_ca^T
#pragma unroll
        for (int b = a + 1; b < 4; ++b)
#pragma unroll
            for (int c = a + 1; c <= b; ++c)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    T[b * (b + 1) / 2 + c] = __builtin_amdgcn_mfma_f64_16x16x4f64(-T[c * (c + 1) / 2 + a][ks], T[b * (b + 1) / 2 + a][ks],
                                                                                 T[b * (b + 1) / 2 + c], 0, 0, 0);
    }
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 16 * b + r16, col = 16 * c + kq + 4 * i;
                const double v = (c <= b && row >= col) ? T[(c <= b ? b * (b + 1) / 2 + c : 0)][i] : 0.0;
                if (row >= col) blk[(size_t)col * lda + row] = v;
                Lout[col * CHOL_NB + row] = v;
            }
    if (bad && lane == 0 && *info == 0) *info = j0 + bad;
}

// ---- single-right-hand-side triangular solves with a Cholesky factor (lower, column-major) -----------------------------------
// rocBLAS trsv takes 0.73 ms per triangular solve at 6144 rows (a 96-step dependency chain inside one launch); a 1024-node
// interior-point solve makes ~580 of them: 36 % of its kernel time (profiles/r03_notes.md).  Here: the diagonal 512 x 512 blocks of L
// are inverted once per factorisation (rocblas_dtrtri_strided_batched), and a solve walks the 12 block columns with two launches
// each: x_j = Linv_j b_j (the kernels below, 64 rows per workgroup), then the right-looking update of everything not yet solved as ONE
// gemv (rocBLAS).  Fixed summation order (atomics are off on the handle): bitwise reproducible.
#define TRSV_NB 512
// y = Linv b (forward, trans = 0: row r of the lower-triangular inverse block against b) or x = Linv^T y (backward, trans = 1: column c
// of the block -- contiguous -- against y); in place on x[0 .. bs).  64 rows per workgroup; the right-hand side goes through LDS.
__global__ __launch_bounds__(64) void emi_trsv_diag_kernel(const double* __restrict__ Linv, int bs, int trans, double* __restrict__ x,
                                                          double* __restrict__ out) {
    __shared__ double bsh[TRSV_NB];
    for (int i = threadIdx.x; i < TRSV_NB; i += 64) bsh[i] = i < bs ? x[i] : 0.0;
    __syncthreads();
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= bs) return;
    double acc = 0.0;
    if (!trans) {
        const double* Li = Linv + r;                       // element (r, c) at Linv[c * NB + r]: coalesced over r
#pragma unroll 8
        for (int c = 0; c <= r; ++c) acc += Li[(size_t)c * TRSV_NB] * bsh[c];
    } else {
        // x[r] = sum_{q >= r} Linv[q][r] y[q]: walk the rows q in steps, lanes on consecutive COLUMNS r -> stride NB between lanes;
        // the block is 2 MB and read once: L2 absorbs the stride
        const double* Lc = Linv + (size_t)r * TRSV_NB;
#pragma unroll 8
        for (int q = r; q < bs; ++q) acc += Lc[q] * bsh[q];
    }
    out[r] = acc;
}
// transposes of the inverted diagonal blocks (512 x 512 each, 32 x 32 tiles through LDS): the forward sweep then multiplies with
// op T like the backward one (rocBLAS gemvt 5 us a call, gemvn 22 us on a 512 x 512 block: eight workgroups)
static __device__ __forceinline__ void emi_trsv_transpose_kernel_body(const double* __restrict__ src, double* __restrict__ dst, int blk_in) {
    __shared__ double tile[32][33];
    const double* S = src + (size_t)blk_in * TRSV_NB * TRSV_NB;
    double* D = dst + (size_t)blk_in * TRSV_NB * TRSV_NB;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) tile[ty + 8 * k][tx] = S[(size_t)(c0 + ty + 8 * k) * TRSV_NB + r0 + tx];      // element (r0 + tx, c0 + ty + 8k)
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) D[(size_t)(r0 + ty + 8 * k) * TRSV_NB + c0 + tx] = tile[tx][ty + 8 * k];      // dst(c, r) = src(r, c)
}
// y[0 .. m) -= A x for a TALL block A (m x nc, column-major, leading dimension lda; nc <= 512): 64 rows per workgroup, its four waves
// take a quarter of the columns each and add up through LDS in wave order (fixed summation order).  The forward sweep's update of
// everything below a block column; rocBLAS gemvn takes 22 us for it at 5632 x 512.
static __device__ __forceinline__ void emi_trsv_update_kernel_body(const double* __restrict__ A, int lda, int m, int nc,
                                                              const double* __restrict__ x, double* __restrict__ y) {
    __shared__ double xs[TRSV_NB];
    __shared__ double part[4][64];
    for (int i = threadIdx.x; i < TRSV_NB; i += 256) xs[i] = i < nc ? x[i] : 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int row = blockIdx.x * 64 + lane;
    const int q = (nc + 3) / 4, c0 = w * q, c1 = min(nc, c0 + q);
    double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};      // eight loads of a lane in flight (the kernel is latency-bound otherwise)
    if (row < m) {
        const double* a = A + row;
        int c = c0;
        for (; c + 7 < c1; c += 8) {
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = a[(size_t)(c + k) * lda];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += v[k] * xs[c + k];
        }
        for (; c < c1; ++c) acc[0] += a[(size_t)c * lda] * xs[c];
    }
    part[w][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (w == 0 && row < m) y[row] -= ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
}
__global__ void emi_trsv_copy_kernel(const double* __restrict__ src, double* __restrict__ dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// ---- kernels: the single-instance entry points (one workspace) and the BATCHED ones (a table of workspaces, one entry per
// scenario: emi_kkt_factor_batch / _solve_batch).  Same bodies: a batched launch is the single launch with blockIdx.y (or .x for the
// one-workgroup kernels) choosing the scenario, so a batch of n scenarios amortises every dependency chain of the factorisation
// (96 diagonal-block steps at 1024 nodes) over n matrices.
struct KktDev {                 // device pointers of one workspace, as the batched kernels see them
    double *Q, *J, *Pinv, *G, *Rk, *S, *W, *chol_blk, *Linv, *LinvT, *T, *Cb, *rhs, *y;
    double *bb, *xx, *xp;       // refined solves: right-hand side, solution, solution before the last correction
    const int* lr_node;         // low-rank correction of the scenario (lr_r = 0: none active)
    const double *lr_vec, *lr_delta;
    int lr_r;
    const double* Doff;
    unsigned char* fixed;
    int *flag, *info;
    double dw, dc;
};

__global__ __launch_bounds__(64) void emi_kkt_node_inverse_kernel(const double* __restrict__ Q, const double* __restrict__ J,
                                                                 const unsigned char* __restrict__ fixed, int M, int ns, int nv, double* __restrict__ Pinv,
                                                                 double* __restrict__ G, double* __restrict__ Rk, int* __restrict__ flag, double dw) {
    emi_kkt_node_inverse_kernel_body(Q, J, fixed, M, ns, nv, Pinv, G, Rk, flag, dw);
}
__global__ __launch_bounds__(64) void emi_kkt_node_inverse_b_kernel(const KktDev* __restrict__ tab, int M, int ns, int nv) {
    const KktDev t = tab[blockIdx.y];
    emi_kkt_node_inverse_kernel_body(t.Q, t.J, t.fixed, M, ns, nv, t.Pinv, t.G, t.Rk, t.flag, t.dw);
}
template <int NS_, int NV_>
__global__ __launch_bounds__(64) void emi_kkt_node_inverse_fixed_kernel(const double* __restrict__ Q, const double* __restrict__ J,
                                                                       const unsigned char* __restrict__ fixed, int M, double* __restrict__ Pinv,
                                                                       double* __restrict__ G, double* __restrict__ Rk, int* __restrict__ flag, double dw) {
    emi_kkt_node_inverse_fixed_kernel_body<NS_, NV_>(Q, J, fixed, M, Pinv, G, Rk, flag, dw);
}
template <int NS_, int NV_>
__global__ __launch_bounds__(64) void emi_kkt_node_inverse_fixed_b_kernel(const KktDev* __restrict__ tab, int M) {
    const KktDev t = tab[blockIdx.y];
    emi_kkt_node_inverse_fixed_kernel_body<NS_, NV_>(t.Q, t.J, t.fixed, M, t.Pinv, t.G, t.Rk, t.flag, t.dw);
}
__global__ void emi_kkt_scale_kernel(const double* __restrict__ Doff, const double* __restrict__ p, double* __restrict__ W, int M) {
    emi_kkt_scale_kernel_body(Doff, p, W, M);
}
__global__ void emi_kkt_scale_b_kernel(const KktDev* __restrict__ tab, int M, int pofs) {      // p = Pinv + pofs: entry (i, ip) of the node inverses
    const KktDev t = tab[blockIdx.y];
    emi_kkt_scale_kernel_body(t.Doff, t.Pinv + pofs, t.W, M);
}
__global__ void emi_kkt_sblock_terms_kernel(double* __restrict__ S, const double* __restrict__ Doff, const double* __restrict__ G,
                                            const double* __restrict__ Rk, int M, int ns, int i, int ip, double dc, double rel) {
    emi_kkt_sblock_terms_kernel_body(S, Doff, G, Rk, M, ns, i, ip, dc, rel, (int)blockIdx.y);
}
__global__ void emi_kkt_sblock_terms_b_kernel(const KktDev* __restrict__ tab, int M, int ns, int i, int ip) {
    const KktDev t = tab[blockIdx.y];
    emi_kkt_sblock_terms_kernel_body(t.S, t.Doff, t.G, t.Rk, M, ns, i, ip, t.dc, 0.0, 0);
}
__global__ void emi_kkt_mask_rhs_kernel(double* __restrict__ rhs, const unsigned char* __restrict__ fixed, int nz, int N) {
    emi_kkt_mask_rhs_kernel_body(rhs, fixed, nz, N, (int)blockIdx.y);
}
__global__ void emi_kkt_mask_rhs_b_kernel(const KktDev* __restrict__ tab, int nz, int N) {
    const KktDev t = tab[blockIdx.y];
    emi_kkt_mask_rhs_kernel_body(t.rhs, t.fixed, nz, N, 0);
}
__global__ void emi_kkt_apply_p_kernel(const double* __restrict__ Pinv, const double* __restrict__ in, size_t ld_in, double* __restrict__ out,
                                       size_t ld_out, int M, int nv) {
    emi_kkt_apply_p_kernel_body(Pinv, in, ld_in, out, ld_out, M, nv, (int)blockIdx.y);
}
__global__ void emi_kkt_apply_p_b_kernel(const KktDev* __restrict__ tab, int M, int nv, int back) {    // back: T <- P rhs, else rhs -> T
    const KktDev t = tab[blockIdx.y];
    (void)back;
    emi_kkt_apply_p_kernel_body(t.Pinv, t.rhs, 0, t.T, 0, M, nv, 0);
}
__global__ void emi_kkt_jnode_minus_b_kernel(const double* __restrict__ J, const double* __restrict__ t, size_t ld_t, const double* __restrict__ b,
                                             size_t ld_b, double* __restrict__ Cb, size_t ld_c, int M, int ns, int nv) {
    emi_kkt_jnode_minus_b_kernel_body(J, t, ld_t, b, ld_b, Cb, ld_c, M, ns, nv, (int)blockIdx.y);
}
__global__ void emi_kkt_jnode_minus_b_b_kernel(const KktDev* __restrict__ tab, int M, int ns, int nv) {
    const KktDev t = tab[blockIdx.y];
    emi_kkt_jnode_minus_b_kernel_body(t.J, t.T, 0, t.rhs + (size_t)nv * M, 0, t.Cb, 0, M, ns, nv, 0);
}
__global__ void emi_kkt_jnode_t_kernel(const double* __restrict__ J, const double* __restrict__ lam, size_t ld_l, double* __restrict__ y, size_t ld_y,
                                       int M, int ns, int nv) {
    emi_kkt_jnode_t_kernel_body(J, lam, ld_l, y, ld_y, M, ns, nv, (int)blockIdx.y);
}
__global__ void emi_kkt_jnode_t_b_kernel(const KktDev* __restrict__ tab, int M, int ns, int nv) {
    const KktDev t = tab[blockIdx.y];
    emi_kkt_jnode_t_kernel_body(t.J, t.Cb, 0, t.rhs, 0, M, ns, nv, 0);
}
__global__ void emi_kkt_copy_lambda_kernel(const double* __restrict__ lam, size_t ld_l, double* __restrict__ rhs, size_t ld_r, int nz, int md) {
    emi_kkt_copy_lambda_kernel_body(lam, ld_l, rhs, ld_r, nz, md, (int)blockIdx.y);
}
// the tail of a batched solve: rhs[0 .. nz) <- T (the primal step), rhs[nz ..) <- Cb (the multipliers)
__global__ void emi_kkt_finish_solve_b_kernel(const KktDev* __restrict__ tab, int nz, int md) {
    const KktDev t = tab[blockIdx.y];
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nz) t.rhs[q] = t.T[q];
    else if (q < nz + md) t.rhs[q] = t.Cb[q - nz];
}
__global__ __launch_bounds__(256) void emi_chol_diag_kernel(double* __restrict__ A, int lda, int j0, int nb, int* __restrict__ info,
                                                            double* __restrict__ Lout) {
    emi_chol_diag_kernel_body(A, lda, j0, nb, info, Lout);
}
__global__ __launch_bounds__(256) void emi_chol_diag_b_kernel(const KktDev* __restrict__ tab, int lda, int j0, int nb) {
    const KktDev t = tab[blockIdx.x];
    emi_chol_diag_kernel_body(t.S, lda, j0, nb, t.info, t.chol_blk);
}
__global__ __launch_bounds__(64) void emi_chol_diag_mfma_kernel(double* __restrict__ A, int lda, int j0, int* __restrict__ info, double* __restrict__ Lout) {
    emi_chol_diag_mfma_kernel_body(A, lda, j0, info, Lout);
}
__global__ __launch_bounds__(64) void emi_chol_diag_mfma_b_kernel(const KktDev* __restrict__ tab, int lda, int j0) {
    const KktDev t = tab[blockIdx.x];
    emi_chol_diag_mfma_kernel_body(t.S, lda, j0, t.info, t.chol_blk);
}
__global__ __launch_bounds__(64) void emi_chol_panel_mfma_kernel(double* __restrict__ A, int lda, int n, int j0, const double* __restrict__ Lb) {
    emi_chol_panel_mfma_kernel_body(A, lda, n, j0, Lb, (int)blockIdx.x);
}
__global__ __launch_bounds__(64) void emi_chol_panel_mfma_b_kernel(const KktDev* __restrict__ tab, int lda, int n, int j0) {
    const KktDev t = tab[blockIdx.y];
    emi_chol_panel_mfma_kernel_body(t.S, lda, n, j0, t.chol_blk, (int)blockIdx.x);
}
__global__ __launch_bounds__(256) void emi_trsv_transpose_kernel(const double* __restrict__ src, double* __restrict__ dst) {
    emi_trsv_transpose_kernel_body(src, dst, (int)blockIdx.z);
}
__global__ __launch_bounds__(256) void emi_trsv_transpose_b_kernel(const KktDev* __restrict__ tab, int nblk) {
    const KktDev t = tab[blockIdx.z / nblk];
    emi_trsv_transpose_kernel_body(t.Linv, t.LinvT, (int)(blockIdx.z % nblk));
}
__global__ __launch_bounds__(256) void emi_trsv_update_kernel(const double* __restrict__ A, int lda, int m, int nc, const double* __restrict__ x,
                                                              double* __restrict__ y) {
    emi_trsv_update_kernel_body(A, lda, m, nc, x, y);
}
// forward sweep of a batched solve, block column j0: everything below it in Cb -= L[rest, block] y_block
__global__ __launch_bounds__(256) void emi_trsv_update_b_kernel(const KktDev* __restrict__ tab, int lda, int j0, int bs, int rest) {
    const KktDev t = tab[blockIdx.y];
    emi_trsv_update_kernel_body(t.S + (size_t)j0 * lda + j0 + bs, lda, rest, bs, t.y + j0, t.Cb + j0 + bs);
}
// status words of a batch gathered in one place (one copy to the host instead of two per scenario)
__global__ void emi_kkt_zero_status_b_kernel(const KktDev* __restrict__ tab, int n) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n) { *tab[b].flag = 0; *tab[b].info = 0; }
}

// error plumbing of the host functions below: they have `std::string* err` in scope and return an EMI_* status
#define KKT_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) { *err = std::string(#call) + ": " + hipGetErrorString(e_); return EMI_ERR_HIP; } \
    } while (0)
#define KKT_ENSURE(ptr, cap, bytes)                                                        \
    do {                                                                                   \
        if ((cap) < (size_t)(bytes)) {                                                     \
            if (ptr) KKT_HIP(hipFree(ptr));                                                \
            (ptr) = nullptr;                                                               \
            (cap) = 0;                                                                     \
            KKT_HIP(hipMalloc((void**)&(ptr), (bytes)));                                   \
            (cap) = (bytes);                                                               \
        }                                                                                  \
    } while (0)
#define KKT_RB(call)                                                                       \
    do {                                                                                   \
        rocblas_status s_ = (call);                                                        \
        if (s_ != rocblas_status_success) { *err = std::string(#call) + ": " + rb(s_); return EMI_ERR_HIP; } \
    } while (0)

// rocsolver_dpotrf (ROCm 7.2) is not reliable while other host threads keep the GPU busy on their own handles and
// streams: about 1 % of the calls (65-node problems, 6 threads) report a non-positive pivot in an odd 64-column block
// of a matrix that factorises when the call is repeated on the same data (tools/race_probe.py,
// profiles/r01_notes.md; dgetrf, dpotrs and the GEMMs showed no such effect).  The solver then took a more
// regularised step than a single-threaded run, and iteration paths differed from run to run.  Two measures:
// the calls are serialised across the process (the stream is drained first, so the lock covers the factorisation
// alone; this alone removes the effect at 65 nodes and leaves 1 in 1000 at 257), and a reported failure is
// confirmed on a kept copy of the matrix before it is believed.
std::mutex g_potrf_mutex;
std::atomic<long> g_potrf_spurious{0};

// A (n x n, lda == n) <- its Cholesky factor; *hinfo = 0, or the position of the first non-positive pivot
int potrf_checked(KktWorkspace* w, hipStream_t stream, rocblas_int n, double* A, rocblas_int* hinfo, std::string* err) {
    const size_t bytes = (size_t)n * n * sizeof(double);
    if (w->cap_chol_copy < bytes) {
        if (w->chol_copy) KKT_HIP(hipFree(w->chol_copy));
        w->chol_copy = nullptr;
        w->cap_chol_copy = 0;
        KKT_HIP(hipMalloc((void**)&w->chol_copy, bytes));
        w->cap_chol_copy = bytes;
    }
    KKT_HIP(hipMemcpyAsync(w->chol_copy, A, bytes, hipMemcpyDeviceToDevice, stream));
    rocblas_int first = 0;
    for (int attempt = 0; attempt < 3; ++attempt) {
        if (attempt > 0) KKT_HIP(hipMemcpyAsync(A, w->chol_copy, bytes, hipMemcpyDeviceToDevice, stream));
        const bool serialise = g_tune.potrf_lock.load() != 0;
        std::unique_lock<std::mutex> lk(g_potrf_mutex, std::defer_lock);
        if (serialise) {
            KKT_HIP(hipStreamSynchronize(stream));
            lk.lock();
        }
        KKT_RB(rocsolver_dpotrf(w->handle, rocblas_fill_lower, n, A, n, w->info));
        KKT_HIP(hipMemcpyAsync(hinfo, w->info, sizeof *hinfo, hipMemcpyDeviceToHost, stream));
        KKT_HIP(hipStreamSynchronize(stream));
        if (serialise) lk.unlock();
        if (*hinfo == 0) {
            if (attempt > 0) {
                ++g_potrf_spurious;
                if (g_tune.debug.load())
                    fprintf(stderr, "emi_kkt: dpotrf reported pivot %d of %d, the same matrix factorised on repeat %d\n", (int)first,
                            (int)n, attempt);
            }
            return EMI_OK;
        }
        if (attempt > 0 && *hinfo == first) return EMI_OK;      // the same verdict twice: the matrix is not positive definite
        first = *hinfo;
    }
    return EMI_OK;
}

// A (n x n, lda == n) <- its lower Cholesky factor with the kernels above; *hinfo as rocsolver_dpotrf reports it
int chol_blocked(KktWorkspace* w, hipStream_t stream, rocblas_int n, double* A, rocblas_int* hinfo, std::string* err) {
    KKT_HIP(hipMemsetAsync(w->info, 0, sizeof(rocblas_int), stream));
    if (!w->chol_blk) KKT_HIP(hipMalloc((void**)&w->chol_blk, (CHOL_NB * CHOL_NB + CHOL_NB) * sizeof(double)));
    const double one = 1.0, mone = -1.0;
    for (int j0 = 0; j0 < n; j0 += CHOL_NB) {
        const int nb = std::min(CHOL_NB, (int)n - j0), rest = (int)n - j0 - nb;
        if (nb == CHOL_NB && g_tune.own_diag.load() == 2)
            hipLaunchKernelGGL(emi_chol_diag_mfma_kernel, dim3(1), dim3(64), 0, stream, A, (int)n, j0, (int*)w->info, w->chol_blk);
        else
            hipLaunchKernelGGL(emi_chol_diag_kernel, dim3(1), dim3(256), 0, stream, A, (int)n, j0, nb, (int*)w->info, w->chol_blk);
        if (rest > 0) {
            double* P = A + (size_t)j0 * n + j0 + nb;
            // own kernel by default; rocblas_dtrsm (EMI_CHOL_PANEL=0) is 5 % faster on a single 1024-node solve and 20-40 %
            // slower on eight concurrent 129-node solves (profiles/r01_notes.md)
            const int own_panel = g_tune.own_panel.load();
            if (own_panel == 2) {
                hipLaunchKernelGGL(emi_chol_panel_mfma_kernel, dim3((rest + 63) / 64), dim3(64), 0, stream, A, (int)n, (int)n, j0,
                                   (const double*)w->chol_blk);
                KKT_HIP(hipGetLastError());
            } else if (own_panel) {
                hipLaunchKernelGGL(emi_chol_panel_kernel, dim3((rest + EMI_PANEL_THREADS - 1) / EMI_PANEL_THREADS), dim3(EMI_PANEL_THREADS), 0, stream, A, (int)n, (int)n, j0,
                                   (const double*)w->chol_blk);
                KKT_HIP(hipGetLastError());
            } else {
                KKT_RB(rocblas_dtrsm(w->handle, rocblas_side_right, rocblas_fill_lower, rocblas_operation_transpose,
                                     rocblas_diagonal_non_unit, rest, nb, &one, A + (size_t)j0 * n + j0, n, P, n));
            }
            double* A22 = A + (size_t)(j0 + nb) * n + j0 + nb;
            KKT_RB(rocblas_dsyrk(w->handle, rocblas_fill_lower, rocblas_operation_none, rest, nb, &mone, P, n, &one, A22, n));
        }
    }
    KKT_HIP(hipGetLastError());
    KKT_HIP(hipMemcpyAsync(hinfo, w->info, sizeof *hinfo, hipMemcpyDeviceToHost, stream));
    KKT_HIP(hipStreamSynchronize(stream));
    return EMI_OK;
}

// Two-level form of chol_blocked: the 64-column steps update only the rest of their 512-column OUTER panel (one small dgemm
// each), and the trailing matrix beyond the outer panel gets one rank-512 dsyrk per outer step instead of eight rank-64 ones
// (same flops, 12 large updates instead of 96 small ones at 6144 rows).  The inner dgemm writes whole rectangles, so the strict
// upper triangle inside an outer panel's diagonal block is scratch afterwards -- every consumer of the factor reads its lower
// triangle only (dtrtri / dtrsm / dpotrs with fill_lower, the gemv updates of blk_potrs on blocks below the diagonal).
int chol_blocked2(KktWorkspace* w, hipStream_t stream, rocblas_int n, double* A, rocblas_int* hinfo, std::string* err) {
    const int NB2 = std::max(128, (g_tune.chol_outer.load() / CHOL_NB) * CHOL_NB);
    KKT_HIP(hipMemsetAsync(w->info, 0, sizeof(rocblas_int), stream));
    if (!w->chol_blk) KKT_HIP(hipMalloc((void**)&w->chol_blk, (CHOL_NB * CHOL_NB + CHOL_NB) * sizeof(double)));
    const double one = 1.0, mone = -1.0;
    for (int J0 = 0; J0 < n; J0 += NB2) {
        const int Jend = std::min((int)n, J0 + NB2);
        for (int j0 = J0; j0 < Jend; j0 += CHOL_NB) {
            const int nb = std::min(CHOL_NB, (int)n - j0), rest = (int)n - j0 - nb;
            if (nb == CHOL_NB && g_tune.own_diag.load() == 2)
            hipLaunchKernelGGL(emi_chol_diag_mfma_kernel, dim3(1), dim3(64), 0, stream, A, (int)n, j0, (int*)w->info, w->chol_blk);
        else
            hipLaunchKernelGGL(emi_chol_diag_kernel, dim3(1), dim3(256), 0, stream, A, (int)n, j0, nb, (int*)w->info, w->chol_blk);
            if (rest <= 0) continue;
            if (g_tune.own_panel.load() == 2)
                hipLaunchKernelGGL(emi_chol_panel_mfma_kernel, dim3((rest + 63) / 64), dim3(64), 0, stream, A, (int)n, (int)n, j0,
                                   (const double*)w->chol_blk);
            else
                hipLaunchKernelGGL(emi_chol_panel_kernel, dim3((rest + EMI_PANEL_THREADS - 1) / EMI_PANEL_THREADS), dim3(EMI_PANEL_THREADS), 0, stream, A, (int)n, (int)n, j0,
                               (const double*)w->chol_blk);
            const int wc = Jend - (j0 + nb);            // columns of the outer panel right of this step
            if (wc > 0) {
                double* P = A + (size_t)j0 * n + j0 + nb;                 // rest x nb, the step's panel below its diagonal block
                double* A22 = A + (size_t)(j0 + nb) * n + j0 + nb;        // rest x wc
                KKT_RB(rocblas_dgemm(w->handle, rocblas_operation_none, rocblas_operation_transpose, rest, wc, nb, &mone, P, n, P, n, &one,
                                     A22, n));
            }
        }
        const int rest2 = (int)n - Jend, W = Jend - J0;
        if (rest2 > 0) {
            double* P2 = A + (size_t)J0 * n + Jend;                       // rest2 x W
            double* A33 = A + (size_t)Jend * n + Jend;
            KKT_RB(rocblas_dsyrk(w->handle, rocblas_fill_lower, rocblas_operation_none, rest2, W, &mone, P2, n, &one, A33, n));
        }
    }
    KKT_HIP(hipGetLastError());
    KKT_HIP(hipMemcpyAsync(hinfo, w->info, sizeof *hinfo, hipMemcpyDeviceToHost, stream));
    KKT_HIP(hipStreamSynchronize(stream));
    return EMI_OK;
}

// which Cholesky ("kkt_cholesky"): 2 (default) the two-level form from 1024 rows (below: the one-level one), 1 one level always,
// 0 rocsolver_dpotrf with the confirmation on a copy.  Factorisation of the 1024 / 512 / 256-node Schur complement (S build
// included; tools/scratch/chol_time.py, one box): one level 16.85 / 7.38 / 3.36 ms, outer panels of 256 / 512 / 768 / 1024
// columns 14.68 / 13.96 / 13.36 / 13.28 ms at 1024 nodes, 5.79 and 2.64 ms with 768 at 512 and 256 nodes.
int cholesky(KktWorkspace* w, hipStream_t stream, rocblas_int n, double* A, rocblas_int* hinfo, std::string* err) {
    const int own = g_tune.own_cholesky.load();
    if (own == 2 && n >= 2 * 512) return chol_blocked2(w, stream, n, A, hinfo, err);
    return own ? chol_blocked(w, stream, n, A, hinfo, err) : potrf_checked(w, stream, n, A, hinfo, err);
}


// inverses of the diagonal TRSV_NB blocks of the factor L (n x n, lda == n) -> w->Linv; call after a successful Cholesky
int blk_invert(KktWorkspace* w, hipStream_t stream, rocblas_int n, const double* L, std::string* err) {
    w->linv_n = 0;
    if (!g_tune.block_trsv.load() || n < 2 * TRSV_NB) return EMI_OK;
    const int nblk = (n + TRSV_NB - 1) / TRSV_NB, full = n / TRSV_NB, tail = n - full * TRSV_NB;
    KKT_ENSURE(w->Linv, w->cap_Linv, (size_t)nblk * TRSV_NB * TRSV_NB * sizeof(double));
    KKT_HIP(hipMemsetAsync(w->Linv, 0, (size_t)nblk * TRSV_NB * TRSV_NB * sizeof(double), stream));
    KKT_RB(rocblas_dtrtri_strided_batched(w->handle, rocblas_fill_lower, rocblas_diagonal_non_unit, TRSV_NB, L, n,
                                          (rocblas_stride)TRSV_NB * (n + 1), w->Linv, TRSV_NB, (rocblas_stride)TRSV_NB * TRSV_NB, full));
    if (tail > 0)
        KKT_RB(rocblas_dtrtri(w->handle, rocblas_fill_lower, rocblas_diagonal_non_unit, tail, L + (size_t)full * TRSV_NB * (n + 1), n,
                              w->Linv + (size_t)full * TRSV_NB * TRSV_NB, TRSV_NB));
    KKT_ENSURE(w->LinvT, w->cap_LinvT, (size_t)nblk * TRSV_NB * TRSV_NB * sizeof(double));
    hipLaunchKernelGGL(emi_trsv_transpose_kernel, dim3(TRSV_NB / 32, TRSV_NB / 32, nblk), dim3(256), 0, stream, (const double*)w->Linv, w->LinvT);
    KKT_HIP(hipGetLastError());
    w->linv_n = n;
    return EMI_OK;
}

// x <- (L L^T)^-1 x for ONE right-hand side through the block inverses (w->linv_n == n).  Two rocBLAS gemv per block column and sweep:
// the inverted diagonal block (stored dense, zeros above its diagonal) against the block's part of the vector, then the update of
// everything not yet solved.  The forward sweep leaves its solution in a second vector y, the backward sweep reads y and writes x, so
// no step works in place.  (The library's own one-row-per-thread product for the diagonal blocks took 28 us per launch plus a copy
// kernel: 0.67 of the 1.23 ms of a solve at 6144 rows; "kkt_block_trsv" 2 keeps that form.)
int blk_potrs(KktWorkspace* w, hipStream_t stream, rocblas_int n, const double* L, double* x, std::string* err) {
    const int nblk = (n + TRSV_NB - 1) / TRSV_NB;
    const double one = 1.0, mone = -1.0, zero = 0.0;
    if (g_tune.block_trsv.load() == 2) {
        KKT_ENSURE(w->trsv_tmp, w->cap_trsv_tmp, (size_t)TRSV_NB * sizeof(double));
        for (int j = 0; j < nblk; ++j) {            // forward: L y = b
            const int j0 = j * TRSV_NB, bs = std::min(TRSV_NB, (int)n - j0), rest = (int)n - j0 - bs;
            hipLaunchKernelGGL(emi_trsv_diag_kernel, dim3((bs + 63) / 64), dim3(64), 0, stream, (const double*)w->Linv + (size_t)j * TRSV_NB * TRSV_NB,
                               bs, 0, x + j0, w->trsv_tmp);
            hipLaunchKernelGGL(emi_trsv_copy_kernel, dim3((bs + 255) / 256), dim3(256), 0, stream, (const double*)w->trsv_tmp, x + j0, bs);
            if (rest > 0)           // b_rest -= L[rest rows, block j] y_j
                KKT_RB(rocblas_dgemv(w->handle, rocblas_operation_none, rest, bs, &mone, L + (size_t)j0 * n + j0 + bs, n, x + j0, 1, &one,
                                     x + j0 + bs, 1));
        }
        for (int j = nblk - 1; j >= 0; --j) {       // backward: L^T x = y
            const int j0 = j * TRSV_NB, bs = std::min(TRSV_NB, (int)n - j0);
            hipLaunchKernelGGL(emi_trsv_diag_kernel, dim3((bs + 63) / 64), dim3(64), 0, stream, (const double*)w->Linv + (size_t)j * TRSV_NB * TRSV_NB,
                               bs, 1, x + j0, w->trsv_tmp);
            hipLaunchKernelGGL(emi_trsv_copy_kernel, dim3((bs + 255) / 256), dim3(256), 0, stream, (const double*)w->trsv_tmp, x + j0, bs);
            if (j0 > 0)             // y_(0 .. j0) -= L[block row j, 0 .. j0)^T x_j
                KKT_RB(rocblas_dgemv(w->handle, rocblas_operation_transpose, bs, j0, &mone, L + j0, n, x + j0, 1, &one, x, 1));
        }
        KKT_HIP(hipGetLastError());
        return EMI_OK;
    }
    KKT_ENSURE(w->trsv_y, w->cap_trsv_y, (size_t)n * sizeof(double));
    double* y = w->trsv_y;
    for (int j = 0; j < nblk; ++j) {                // forward: L y = b (b in x, consumed block by block)
        const int j0 = j * TRSV_NB, bs = std::min(TRSV_NB, (int)n - j0), rest = (int)n - j0 - bs;
        // y_j = Linv_j b_j as a product with the TRANSPOSE of the transposed copy (gemvt: 5 us, gemvn 22 us on 512 x 512)
        KKT_RB(rocblas_dgemv(w->handle, rocblas_operation_transpose, bs, bs, &one, w->LinvT + (size_t)j * TRSV_NB * TRSV_NB, TRSV_NB, x + j0, 1,
                             &zero, y + j0, 1));
        if (rest > 0)               // b_rest -= L[rest rows, block j] y_j
            hipLaunchKernelGGL(emi_trsv_update_kernel, dim3((rest + 63) / 64), dim3(256), 0, stream, L + (size_t)j0 * n + j0 + bs, (int)n, rest, bs,
                               (const double*)(y + j0), x + j0 + bs);
    }
    for (int j = nblk - 1; j >= 0; --j) {           // backward: L^T x = y (y consumed block by block)
        const int j0 = j * TRSV_NB, bs = std::min(TRSV_NB, (int)n - j0);
        KKT_RB(rocblas_dgemv(w->handle, rocblas_operation_transpose, bs, bs, &one, w->Linv + (size_t)j * TRSV_NB * TRSV_NB, TRSV_NB, y + j0, 1, &zero,
                             x + j0, 1));
        if (j0 > 0)                 // y_(0 .. j0) -= L[block row j, 0 .. j0)^T x_j
            KKT_RB(rocblas_dgemv(w->handle, rocblas_operation_transpose, bs, j0, &mone, L + j0, n, x + j0, 1, &one, y, 1));
    }
    KKT_HIP(hipGetLastError());
    return EMI_OK;
}

// X <- (L L^T)^-1 X for MANY right-hand sides (X: n x nrhs, column-major, ldx) through the same block inverses: per block column one
// GEMM with the inverted diagonal block and one with the rows below (forward) / the columns left of it (backward) -- 4 nblk large GEMMs,
// n^2 nrhs flops per sweep like a triangular solve, none of its dependency chains.  The r columns of the low-rank correction
// (r ~ 800 at 1024 nodes) went through rocsolver_dpotrs before (blocked substitution kernels, 225 launches of 56 - 76 us per solve of
// a 1024-node problem: profiles/r04_one_solve_kernel_stats_before_batching.csv).
int blk_potrs_multi(KktWorkspace* w, hipStream_t stream, rocblas_int n, const double* L, double* X, rocblas_int ldx, rocblas_int nrhs,
                    std::string* err) {
    const int nblk = (n + TRSV_NB - 1) / TRSV_NB;
    const double one = 1.0, mone = -1.0, zero = 0.0;
    KKT_ENSURE(w->trsm_y, w->cap_trsm_y, (size_t)n * nrhs * sizeof(double));
    double* Y = w->trsm_y;                              // n x nrhs, leading dimension n
    for (int j = 0; j < nblk; ++j) {                    // forward: L Y = B (B in X, consumed block row by block row)
        const int j0 = j * TRSV_NB, bs = std::min(TRSV_NB, (int)n - j0), rest = (int)n - j0 - bs;
        KKT_RB(rocblas_dgemm(w->handle, rocblas_operation_none, rocblas_operation_none, bs, nrhs, bs, &one, w->Linv + (size_t)j * TRSV_NB * TRSV_NB,
                             TRSV_NB, X + j0, ldx, &zero, Y + j0, n));
        if (rest > 0)
            KKT_RB(rocblas_dgemm(w->handle, rocblas_operation_none, rocblas_operation_none, rest, nrhs, bs, &mone, L + (size_t)j0 * n + j0 + bs, n,
                                 Y + j0, n, &one, X + j0 + bs, ldx));
    }
    for (int j = nblk - 1; j >= 0; --j) {               // backward: L^T X = Y (Y consumed block row by block row)
        const int j0 = j * TRSV_NB, bs = std::min(TRSV_NB, (int)n - j0);
        KKT_RB(rocblas_dgemm(w->handle, rocblas_operation_transpose, rocblas_operation_none, bs, nrhs, bs, &one, w->Linv + (size_t)j * TRSV_NB * TRSV_NB,
                             TRSV_NB, Y + j0, n, &zero, X + j0, ldx));
        if (j0 > 0)
            KKT_RB(rocblas_dgemm(w->handle, rocblas_operation_transpose, rocblas_operation_none, j0, nrhs, bs, &mone, L + j0, n, X + j0, ldx, &one, Y,
                                 n));
    }
    return EMI_OK;
}

}  // namespace

void kkt_destroy(KktWorkspace* w) {
    if (!w) return;
    if (w->handle) (void)rocblas_destroy_handle(w->handle);
    void* bufs[] = {w->K, w->ipiv, w->info, w->Q, w->J, w->rhs, w->fixed, w->S, w->Pinv, w->G, w->Rk, w->Doff, w->W, w->gemm_ptrs, w->T,
                    w->Cb, w->flag, w->chol_blk, w->chol_copy, w->Linv, w->LinvT, w->trsv_tmp, w->trsv_y, w->trsm_y, w->lrY, w->lrC, w->lrT, w->lr_node, w->lr_vec, w->lr_delta};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (w->ref_b) (void)hipFree(w->ref_b);
    if (w->ref_x) (void)hipFree(w->ref_x);
    if (w->ref_p) (void)hipFree(w->ref_p);
    if (w->b_tab) (void)hipFree(w->b_tab);
    if (w->b_ptrs) (void)hipFree(w->b_ptrs);
    if (w->b_stat) (void)hipFree(w->b_stat);
    if (w->b_pin) (void)hipHostFree(w->b_pin);
    delete w;
}

bool kkt_is_schur(const KktWorkspace* w) { return w && w->factored && w->method_used == 1; }

void kkt_mesh_changed(KktWorkspace* w) {
    if (!w) return;
    w->doff_M = 0;
    w->doff_src = nullptr;
}

// Returns an EMI_* status; *info = 0 factorised, > 0 exactly singular (zero pivot at that position).
int kkt_factor(KktWorkspace** pw, hipStream_t stream, const double* dD, int M, int ns, int nv, const double* Qblk,
               const double* Jblk, const unsigned char* fixed, double dc, int method, int* info, std::string* err) {
    const int nh = nv * (nv + 1) / 2, N = (nv + ns) * M, nz = nv * M;
    if (!*pw) *pw = new KktWorkspace();
    KktWorkspace* w = *pw;
    w->factored = false;
    w->lr_active = false;
    w->linv_n = 0;
    w->reg_dc_applied = dc;
    w->reg_dw_applied = 0.0;
    if (!w->handle) {
        KKT_RB(rocblas_create_handle(&w->handle));
        // split-K kernels that accumulate with atomics make the factorisation, and with it the iteration path of
        // the NLP solver, differ from run to run: a solver has to be reproducible
        KKT_RB(rocblas_set_atomics_mode(w->handle, rocblas_atomics_not_allowed));
    }
    KKT_RB(rocblas_set_stream(w->handle, stream));
    if (!w->info) KKT_HIP(hipMalloc(&w->info, sizeof(rocblas_int)));
    KKT_ENSURE(w->ipiv, w->cap_ipiv, (size_t)N * sizeof(rocblas_int));
    KKT_ENSURE(w->Q, w->cap_Q, (size_t)nh * M * sizeof(double));
    KKT_ENSURE(w->J, w->cap_J, (size_t)ns * nv * M * sizeof(double));
    KKT_ENSURE(w->fixed, w->cap_fixed, (size_t)nz);
    w->N = N;
    KKT_HIP(hipMemcpyAsync(w->Q, Qblk, (size_t)nh * M * sizeof(double), hipMemcpyHostToDevice, stream));
    KKT_HIP(hipMemcpyAsync(w->J, Jblk, (size_t)ns * nv * M * sizeof(double), hipMemcpyHostToDevice, stream));
    KKT_HIP(hipMemcpyAsync(w->fixed, fixed, (size_t)nz, hipMemcpyHostToDevice, stream));
    w->M = M; w->ns = ns; w->nv = nv;
    if (method == 1 && nv <= KKT_NV_MAX) {
        const size_t md = (size_t)ns * M;
        if (w->S_elems < md * md) {
            if (w->S) KKT_HIP(hipFree(w->S));
            w->S = nullptr;
            w->S_elems = 0;
            KKT_HIP(hipMalloc(&w->S, md * md * sizeof(double)));
            w->S_elems = md * md;
        }
        KKT_ENSURE(w->Pinv, w->cap_Pinv, (size_t)nv * nv * M * sizeof(double));
        KKT_ENSURE(w->G, w->cap_G, (size_t)ns * ns * M * sizeof(double));
        KKT_ENSURE(w->Rk, w->cap_Rk, (size_t)ns * ns * M * sizeof(double));
        if (w->cap_Doff < (size_t)M * M * sizeof(double)) w->doff_M = 0;
        KKT_ENSURE(w->Doff, w->cap_Doff, (size_t)M * M * sizeof(double));
        // one batched GEMM for all state pairs where the build is launch-bound (measured: +5-10 % Monte-Carlo throughput
        // at 65 nodes, +3 % at 129; no gain per factorisation at 1024 nodes, where the unbatched form is kept)
        const int batched_max_m = g_tune.batched_max_nodes.load();
        const bool batched = M <= batched_max_m;
        const int npairs = ns * (ns + 1) / 2;
        KKT_ENSURE(w->W, w->cap_W, (size_t)(batched ? npairs : 1) * M * M * sizeof(double));
        if (!w->flag) KKT_HIP(hipMalloc(&w->flag, sizeof(int)));
        // S = J Q^-1 J^T squares the conditioning of J; late interior-point iterations (barrier terms of 1e10 in Q)
        // leave it numerically semidefinite.  A dual regularisation of IPOPT's size (its delta_c is 1e-8 mu^1/4),
        // raised x1000 on a failed Cholesky, keeps the factorisation alive; the caller's iterative refinement
        // works against the matrix with the nominal dc.
        KKT_HIP(hipMemsetAsync(w->flag, 0, sizeof(int), stream));
        const unsigned nb2 = (unsigned)(((size_t)M * M + 255) / 256);
        if (w->doff_M != M || w->doff_src != dD) {
            hipLaunchKernelGGL(emi_kkt_doff_kernel, dim3(nb2), dim3(256), 0, stream, dD, w->Doff, M);
            w->doff_M = M;
            w->doff_src = dD;
        }
        // Doff is stored [k][j] row-major, i.e. as the column-major matrix Dc = Doff^T:  Doff diag(p) Doff^T = Dc^T (diag(p) Dc)
        const double one = 1.0, zero = 0.0;
        rocblas_int hinfo = 0;
        int hflag = 0;
        // (Tried: raising the diagonal relatively, S_ii (1 + 1e-12 .. 1e-6), on the retries.  At the clustered end nodes of
        // a 1000-node mesh the entries of S reach 1e16 and an absolute 1e-3 no longer rescues the Cholesky -- the relative
        // shift does, but the factor of such an S is too inaccurate for the refinement to repair: the 1024-node solve went
        // from 12 to 174 iterations.  Those few matrices belong to the LU below.)
        // Regularisation ladder of the Schur path.  Level l factorises S with the dual regularisation dc_l and the node blocks
        // with a primal regularisation dw_l on their free diagonal (IPOPT's delta_c / delta_w, here applied ONLY to make the
        // factorisation exist: the caller's iterative refinement works against the nominal matrix and takes the difference
        // out again).  Round 2 had the dual levels only and sent what failed all three to an LU of the whole assembled KKT
        // matrix (rocSOLVER getrf, one launch per column: 43 000 launches and ~210 ms at 1024 nodes of the 6-state model, two
        // or three times per solve).  Those matrices come from late iterations (mu <= 1e-7): state variables away from their
        // bounds carry ~1e-9 of curvature, P = Q^-1 ~ 1e9, and S = J P J^T ~ 1e19 at the clustered end nodes loses its positive
        // definiteness to rounding whatever the dual shift.  A primal shift bounds P instead (profiles/r03_notes.md) -- on the
        // STATE variables only: they are pinned by the defect equations, so the shift barely moves the step, whereas the
        // controls' own curvature (h w_k L_uu ~ 1e-5 .. 1e-2) would drown in it (shifting every variable: 24 factorisations on
        // the last mesh of scenario 0 instead of 13).
        static const double LV_DC[5] = {1.0, 1e3, 1e3, 1e3, 1e6}, LV_DW[5] = {0.0, 0.0, 1e-7, 1e-5, 1e-3};
        constexpr int NLV = 5;
        const double dc_base = dc > 1e-9 ? dc : 1e-9;
        double dc_schur = dc_base;
        if (w->reg_M != M || w->reg_ns != ns || w->reg_nv != nv || !g_tune.sticky_reg.load()) {
            w->reg_level = w->reg_hits = 0;
            w->reg_M = M; w->reg_ns = ns; w->reg_nv = nv;
        } else if (w->reg_level > 0 && w->reg_hits >= 4) {
            --w->reg_level;
            w->reg_hits = 0;
        }
        const int max_lv = g_tune.primal_levels.load() ? NLV : 2;      // "kkt_primal_levels" 0: the round-2 ladder (dual levels, then the LU)
        const int first_attempt = std::min(w->reg_level, max_lv - 1);
        if (batched && (w->ptrs_key[0] != w->Doff || w->ptrs_key[1] != w->W || w->ptrs_key[2] != w->S || w->ptrs_M != M ||
                        w->ptrs_ns != ns)) {
            std::vector<double*> hp(3 * (size_t)npairs);
            for (int i = 0, p = 0; i < ns; ++i)
                for (int ip = 0; ip <= i; ++ip, ++p) {
                    hp[p] = w->Doff;
                    hp[npairs + p] = w->W + (size_t)p * M * M;
                    hp[2 * npairs + p] = w->S + ((size_t)ip * M) * md + (size_t)i * M;
                }
            if (!w->gemm_ptrs) KKT_HIP(hipMalloc((void**)&w->gemm_ptrs, 3 * 136 * sizeof(double*)));   // ns <= 16
            KKT_HIP(hipMemcpyAsync(w->gemm_ptrs, hp.data(), hp.size() * sizeof(double*), hipMemcpyHostToDevice, stream));
            KKT_HIP(hipStreamSynchronize(stream));
            w->ptrs_key[0] = w->Doff; w->ptrs_key[1] = w->W; w->ptrs_key[2] = w->S;
            w->ptrs_M = M; w->ptrs_ns = ns;
        }
        int attempt = first_attempt;
        double dw_done = -1.0;
        for (; attempt < max_lv; ++attempt) {
            dc_schur = dc_base * LV_DC[attempt];
            if (LV_DW[attempt] != dw_done) {        // node blocks (re)inverted with this level's primal shift
                dw_done = LV_DW[attempt];
                KKT_HIP(hipMemsetAsync(w->flag, 0, sizeof(int), stream));
                if (ns == 6 && nv == 8)
                    hipLaunchKernelGGL((emi_kkt_node_inverse_fixed_kernel<6, 8>), dim3((M + 63) / 64), dim3(64), 0, stream, w->Q, w->J, w->fixed, M,
                                   w->Pinv, w->G, w->Rk, w->flag, dw_done);
                else if (ns == 2 && nv == 4)
                    hipLaunchKernelGGL((emi_kkt_node_inverse_fixed_kernel<2, 4>), dim3((M + 63) / 64), dim3(64), 0, stream, w->Q, w->J, w->fixed, M,
                                   w->Pinv, w->G, w->Rk, w->flag, dw_done);
                else
                    hipLaunchKernelGGL(emi_kkt_node_inverse_kernel, dim3((M + 63) / 64), dim3(64), 0, stream, w->Q, w->J, w->fixed, M, ns, nv,
                                   w->Pinv, w->G, w->Rk, w->flag, dw_done);
                KKT_HIP(hipGetLastError());
            }
            if (batched) {
                // three launches for all ns (ns + 1) / 2 state pairs: small meshes are launch-bound here
                hipLaunchKernelGGL(emi_kkt_scale_all_kernel, dim3(nb2, npairs), dim3(256), 0, stream, w->Doff, w->Pinv, w->W, M, nv);
                KKT_RB(rocblas_dgemm_batched(w->handle, rocblas_operation_transpose, rocblas_operation_none, M, M, M, &one,
                                             (const double* const*)w->gemm_ptrs, M, (const double* const*)(w->gemm_ptrs + npairs), M,
                                             &zero, w->gemm_ptrs + 2 * npairs, (rocblas_int)md, npairs));
                hipLaunchKernelGGL(emi_kkt_sblock_terms_kernel, dim3(nb2, npairs), dim3(256), 0, stream, w->S, w->Doff, w->G, w->Rk,
                                   M, ns, -1, -1, dc_schur, 0.0);
            } else
            for (int i = 0; i < ns; ++i)
                for (int ip = 0; ip <= i; ++ip) {
                    hipLaunchKernelGGL(emi_kkt_scale_kernel, dim3(nb2), dim3(256), 0, stream, w->Doff,
                                       w->Pinv + (size_t)(i * nv + ip) * M, w->W, M);
                    double* Sblk = w->S + ((size_t)ip * M) * md + (size_t)i * M;
                    KKT_RB(rocblas_dgemm(w->handle, rocblas_operation_transpose, rocblas_operation_none, M, M, M, &one, w->Doff, M,
                                         w->W, M, &zero, Sblk, (rocblas_int)md));
                    hipLaunchKernelGGL(emi_kkt_sblock_terms_kernel, dim3(nb2), dim3(256), 0, stream, w->S, w->Doff, w->G, w->Rk, M,
                                       ns, i, ip, dc_schur, 0.0);
                }
            KKT_HIP(hipGetLastError());
            if (int st = cholesky(w, stream, (rocblas_int)md, w->S, &hinfo, err)) return st;
            KKT_HIP(hipMemcpyAsync(&hflag, w->flag, sizeof hflag, hipMemcpyDeviceToHost, stream));
            KKT_HIP(hipStreamSynchronize(stream));
            if (hinfo == 0 || hflag != 0) break;      // factorised, or hopeless (a Q block is not positive definite)
            if (g_tune.debug.load())
                fprintf(stderr, "emi_kkt_factor: S not positive definite at %d with dual regularisation %.1e, primal %.1e (M %d), retrying\n",
                        (int)hinfo, dc_schur, dw_done, M);
        }
        if (hinfo == 0 && hflag == 0) {
            if (int st = blk_invert(w, stream, (rocblas_int)md, w->S, err)) return st;
            if (attempt == first_attempt) ++w->reg_hits; else { w->reg_level = attempt; w->reg_hits = 0; }
            w->reg_dc_applied = dc_schur;
            w->reg_dw_applied = dw_done;
            *info = 0;
            w->factored = true;
            w->method_used = 1;
            return EMI_OK;
        }
        if (hflag == 0) {                           // S not positive definite at any level
            w->reg_level = max_lv - 1;                  // (the next factorisation starts at the last level, not at the LU)
            w->reg_hits = 0;
        }
        // a block was not positive definite or S is not: not the quasi-definite case -- general path below
        if (g_tune.debug.load() >= 2 && hinfo > 0) {
            // diagnosis: the node blocks around the failing pivot (diagonals of Q as uploaded and of P = Q^-1)
            const int kf = ((int)hinfo - 1) % M, i_f = ((int)hinfo - 1) / M;
            std::vector<double> hq((size_t)nh * M), hp((size_t)nv * nv * M);
            std::vector<unsigned char> hf((size_t)nz);
            (void)hipMemcpy(hq.data(), w->Q, hq.size() * sizeof(double), hipMemcpyDeviceToHost);
            (void)hipMemcpy(hp.data(), w->Pinv, hp.size() * sizeof(double), hipMemcpyDeviceToHost);
            (void)hipMemcpy(hf.data(), w->fixed, hf.size(), hipMemcpyDeviceToHost);
            fprintf(stderr, "  failing pivot: state row %d, node %d of %d\n", i_f, kf, M);
            for (int k = std::max(0, kf - 2); k <= std::min(M - 1, kf + 2); ++k) {
                fprintf(stderr, "  node %4d  Qdiag", k);
                for (int v = 0; v < nv; ++v) fprintf(stderr, " %9.2e%s", hq[(size_t)(v * (v + 1) / 2 + v) * M + k], hf[(size_t)v * M + k] ? "f" : "");
                fprintf(stderr, "   Pdiag");
                for (int v = 0; v < nv; ++v) fprintf(stderr, " %9.2e", hp[(size_t)(v * nv + v) * M + k]);
                fprintf(stderr, "\n");
            }
        }
        if (g_tune.debug.load())
            fprintf(stderr, "emi_kkt_factor: Schur path gave up (block flag %d, potrf info %d, M %d, dc %.3g) -> LU\n", hflag,
                    (int)hinfo, M, dc);
    }
    w->method_used = 0;
    if (w->K_elems < (size_t)N * N) {
        if (w->K) KKT_HIP(hipFree(w->K));
        w->K = nullptr;
        w->K_elems = 0;
        KKT_HIP(hipMalloc(&w->K, (size_t)N * N * sizeof(double)));
        w->K_elems = (size_t)N * N;
    }
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

// The regularisation the current factorisation holds: the matrix factorised is [[Q + dw I_x, J^T], [J, -dc I]] with dw on the
// free STATE variables' diagonal only.  dc is at least the caller's; both exceed the nominal values when the Schur path climbed its ladder.
void kkt_last_regularisation(const KktWorkspace* w, double* dc, double* dw) {
    if (dc) *dc = w ? w->reg_dc_applied : 0.0;
    if (dw) *dw = w ? w->reg_dw_applied : 0.0;
}

// X [N][nrhs] on the device, in place: X <- K~^-1 X with the current factorisation
static int solve_dev(KktWorkspace* w, hipStream_t stream, int nz, double* X, int nrhs, std::string* err) {
    const int N = w->N;
    hipLaunchKernelGGL(emi_kkt_mask_rhs_kernel, dim3((nz + 255) / 256, nrhs), dim3(256), 0, stream, X, w->fixed, nz, N);
    KKT_HIP(hipGetLastError());
    KKT_RB(rocblas_set_stream(w->handle, stream));
    if (w->method_used == 1) {
        const int M = w->M, ns = w->ns, nv = w->nv, md = ns * M;
        if (w->T_elems < (size_t)nz * nrhs) {
            if (w->T) KKT_HIP(hipFree(w->T));
            w->T = nullptr; w->T_elems = 0;
            KKT_HIP(hipMalloc(&w->T, (size_t)nz * nrhs * sizeof(double)));
            w->T_elems = (size_t)nz * nrhs;
        }
        if (w->Cb_elems < (size_t)md * nrhs) {
            if (w->Cb) KKT_HIP(hipFree(w->Cb));
            w->Cb = nullptr; w->Cb_elems = 0;
            KKT_HIP(hipMalloc(&w->Cb, (size_t)md * nrhs * sizeof(double)));
            w->Cb_elems = (size_t)md * nrhs;
        }
        const double one = 1.0, zero = 0.0, mone = -1.0;
        dim3 gk((M + 127) / 128, nrhs), bk(128);
        // t = P a
        hipLaunchKernelGGL(emi_kkt_apply_p_kernel, gk, bk, 0, stream, w->Pinv, X, (size_t)N, w->T, (size_t)nz, M, nv);
        // Cb = Doff t_states            (Doff = Dc^T in column-major terms).  One right-hand side: an M x ns panel.  Many (the r columns of
        // the low-rank correction, r ~ 800 at 1024 nodes): batched over the STATES instead -- state i of every right-hand side is the
        // M x nrhs matrix at T + i M with leading dimension nz, so the product is ns large GEMMs (M x nrhs x M) rather than nrhs GEMMs
        // with ns columns each (which ran at a few percent of the matrix pipe)
        if (nrhs >= 8)
            KKT_RB(rocblas_dgemm_strided_batched(w->handle, rocblas_operation_transpose, rocblas_operation_none, M, nrhs, M, &one,
                                                 w->Doff, M, 0, w->T, nz, (rocblas_stride)M, &zero, w->Cb, md, (rocblas_stride)M, ns));
        else
            KKT_RB(rocblas_dgemm_strided_batched(w->handle, rocblas_operation_transpose, rocblas_operation_none, M, ns, M, &one,
                                                 w->Doff, M, 0, w->T, M, (rocblas_stride)nz, &zero, w->Cb, M, (rocblas_stride)md,
                                                 nrhs));
        // Cb += J_node t - b
        hipLaunchKernelGGL(emi_kkt_jnode_minus_b_kernel, gk, bk, 0, stream, w->J, w->T, (size_t)nz, X + nz, (size_t)N, w->Cb,
                           (size_t)md, M, ns, nv);
        KKT_HIP(hipGetLastError());
        // lambda = S^-1 Cb
        if (nrhs == 1 && w->linv_n == md) { if (int st = blk_potrs(w, stream, md, w->S, w->Cb, err)) return st; }
        else if (nrhs >= 16 && w->linv_n == md) { if (int st = blk_potrs_multi(w, stream, md, w->S, w->Cb, md, nrhs, err)) return st; }
        else KKT_RB(rocsolver_dpotrs(w->handle, rocblas_fill_lower, md, nrhs, w->S, md, w->Cb, md));
        // y = a - J^T lambda  (in place in the primal part of X), then x = P y
        if (nrhs >= 8)
            KKT_RB(rocblas_dgemm_strided_batched(w->handle, rocblas_operation_none, rocblas_operation_none, M, nrhs, M, &mone, w->Doff,
                                                 M, 0, w->Cb, md, (rocblas_stride)M, &one, X, N, (rocblas_stride)M, ns));
        else
            KKT_RB(rocblas_dgemm_strided_batched(w->handle, rocblas_operation_none, rocblas_operation_none, M, ns, M, &mone, w->Doff,
                                                 M, 0, w->Cb, M, (rocblas_stride)md, &one, X, M, (rocblas_stride)N, nrhs));
        hipLaunchKernelGGL(emi_kkt_jnode_t_kernel, gk, bk, 0, stream, w->J, w->Cb, (size_t)md, X, (size_t)N, M, ns, nv);
        hipLaunchKernelGGL(emi_kkt_apply_p_kernel, gk, bk, 0, stream, w->Pinv, X, (size_t)N, w->T, (size_t)nz, M, nv);
        KKT_HIP(hipGetLastError());
        KKT_HIP(hipMemcpy2DAsync(X, (size_t)N * sizeof(double), w->T, (size_t)nz * sizeof(double), (size_t)nz * sizeof(double),
                                 nrhs, hipMemcpyDeviceToDevice, stream));
        hipLaunchKernelGGL(emi_kkt_copy_lambda_kernel, dim3((md + 255) / 256, nrhs), dim3(256), 0, stream, w->Cb, (size_t)md, X,
                           (size_t)N, nz, md);
        KKT_HIP(hipGetLastError());
    } else {
        KKT_RB(rocsolver_dgetrs(w->handle, rocblas_operation_none, N, nrhs, w->K, N, w->ipiv, X, N));
    }
    return EMI_OK;
}

// Low-rank correction K = K~ - U Delta U^T (U = [u; 0], one column per reflected eigenpair of a node block):
// Y = K~^-1 U and the Cholesky factor of C = Delta^-1 - U^T Y stay on the device; *exact = 1 iff C is positive
// definite, i.e. iff the unmodified K has the inertia of K~ (emi_nlp.cpp has the argument).  While exact,
// kkt_solve returns solutions of K (Woodbury), otherwise of K~.
int kkt_lowrank(KktWorkspace* w, hipStream_t stream, int nz, int r, const int* node, const double* vec, const double* delta,
                int* exact, std::string* err) {
    if (!w || !w->factored) { *err = "emi_kkt_lowrank: no factorisation"; return EMI_ERR_STATE; }
    w->lr_active = false;
    w->lr_r = 0;
    *exact = r == 0 ? 1 : 0;
    if (r == 0) return EMI_OK;
    const int N = w->N, nv = w->nv, M = w->M;
    if (w->lr_cap < r) {
        void** bufs[] = {(void**)&w->lrY, (void**)&w->lrC, (void**)&w->lr_node, (void**)&w->lr_vec, (void**)&w->lr_delta,
                         (void**)&w->lrT};
        for (void** b : bufs)
            if (*b) { KKT_HIP(hipFree(*b)); *b = nullptr; }
        w->lr_cap = 0;
        const int cap = r + r / 4 + 16;
        KKT_HIP(hipMalloc(&w->lrY, (size_t)N * cap * sizeof(double)));
        KKT_HIP(hipMalloc(&w->lrC, (size_t)cap * cap * sizeof(double)));
        KKT_HIP(hipMalloc(&w->lr_node, (size_t)cap * sizeof(int)));
        KKT_HIP(hipMalloc(&w->lr_vec, (size_t)cap * KKT_NV_MAX * sizeof(double)));
        KKT_HIP(hipMalloc(&w->lr_delta, (size_t)cap * sizeof(double)));
        KKT_HIP(hipMalloc(&w->lrT, (size_t)cap * 64 * sizeof(double)));
        w->lr_cap = cap;
        w->lr_n = N;
    } else if (w->lr_n < N) {
        if (w->lrY) KKT_HIP(hipFree(w->lrY));
        w->lrY = nullptr;
        KKT_HIP(hipMalloc(&w->lrY, (size_t)N * w->lr_cap * sizeof(double)));
        w->lr_n = N;
    }
    KKT_HIP(hipMemcpyAsync(w->lr_node, node, (size_t)r * sizeof(int), hipMemcpyHostToDevice, stream));
    KKT_HIP(hipMemcpyAsync(w->lr_vec, vec, (size_t)r * nv * sizeof(double), hipMemcpyHostToDevice, stream));
    KKT_HIP(hipMemcpyAsync(w->lr_delta, delta, (size_t)r * sizeof(double), hipMemcpyHostToDevice, stream));
    KKT_HIP(hipMemsetAsync(w->lrY, 0, (size_t)N * r * sizeof(double), stream));
    hipLaunchKernelGGL(emi_kkt_lr_scatter_kernel, dim3((r + 63) / 64), dim3(64), 0, stream, w->lrY, w->lr_node, w->lr_vec, r, N, M,
                       nv);
    KKT_HIP(hipGetLastError());
    int st = solve_dev(w, stream, nz, w->lrY, r, err);
    if (st) return st;
    hipLaunchKernelGGL(emi_kkt_lr_c_kernel, dim3((r + 15) / 16, (r + 15) / 16), dim3(16, 16), 0, stream, w->lrC, w->lrY, w->lr_node,
                       w->lr_vec, w->lr_delta, r, N, M, nv);
    KKT_HIP(hipGetLastError());
    rocblas_int hinfo = 0;
    if (int st2 = cholesky(w, stream, r, w->lrC, &hinfo, err)) return st2;
    if (hinfo == 0) {
        w->lr_active = true;
        w->lr_r = r;
        *exact = 1;
    }
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
    int st = solve_dev(w, stream, nz, w->rhs, nrhs, err);
    if (st) return st;
    if (w->lr_active) {
        // x <- x + Y C^-1 (U^T x)
        const int r = w->lr_r;
        if (nrhs > 64) { *err = "emi_kkt_solve: at most 64 right-hand sides while a low-rank correction is active"; return EMI_ERR_ARG; }
        hipLaunchKernelGGL(emi_kkt_lr_utx_kernel, dim3((r + 63) / 64, nrhs), dim3(64), 0, stream, w->lrT, w->rhs, w->lr_node,
                           w->lr_vec, r, N, w->M, w->nv);
        KKT_HIP(hipGetLastError());
        KKT_RB(rocsolver_dpotrs(w->handle, rocblas_fill_lower, r, nrhs, w->lrC, r, w->lrT, r));
        const double one = 1.0;
        KKT_RB(rocblas_dgemm(w->handle, rocblas_operation_none, rocblas_operation_none, N, nrhs, r, &one, w->lrY, N, w->lrT, r,
                             &one, w->rhs, N));
    }
    KKT_HIP(hipMemcpyAsync(rhs, w->rhs, elems * sizeof(double), hipMemcpyDeviceToHost, stream));
    KKT_HIP(hipStreamSynchronize(stream));
    return EMI_OK;
}

// ==================================================================================================================================
// Batched entry points: the Newton steps of n scenarios on ONE mesh at once.
//
// A Monte-Carlo run solves hundreds of independent scenarios of the same transcription (BASELINE configs[3]).
// Factorised one by one -- each from its own host thread -- their ~50 000 small launches per solve take turns on the device: the
// chain of 96 one-workgroup diagonal-block kernels of a 1024-node factorisation keeps one CU of 256 busy, and eight host threads
// reached 2.3 x the throughput of one (profiles/r03_notes.md section 7).  Here every launch of the factorisation carries the whole
// batch: the diagonal-block kernel runs n workgroups, the panel kernel n x rest / 64, every GEMM is a batched GEMM, so the dependency
// chain of ONE factorisation is paid once per batch.  The scenarios keep their own workspaces (factors, node inverses, low-rank
// correction); the batch is a table of their device pointers (KktDev) plus pointer arrays for rocBLAS' *_batched calls, built on the
// host in pinned memory and uploaded once per attempt.
//
// Restrictions (the caller falls back to the single entry points otherwise): Schur method, one common (M, ns, nv), every scenario's
// context holding the same differentiation matrix (all of them come from emi_lgl for the same node count).
// ==================================================================================================================================
namespace {

int batch_scratch(KktWorkspace* L, size_t tab_bytes, size_t ptr_bytes, size_t stat_bytes, std::string* err) {
    KKT_ENSURE(L->b_tab, L->cap_b_tab, tab_bytes);
    KKT_ENSURE(L->b_ptrs, L->cap_b_ptrs, ptr_bytes);
    KKT_ENSURE(L->b_stat, L->cap_b_stat, stat_bytes);
    const size_t pin = tab_bytes + ptr_bytes + stat_bytes;
    if (L->cap_b_pin < pin) {
        if (L->b_pin) KKT_HIP(hipHostFree(L->b_pin));
        L->b_pin = nullptr;
        L->cap_b_pin = 0;
        KKT_HIP(hipHostMalloc((void**)&L->b_pin, pin + pin / 2, hipHostMallocDefault));
        L->cap_b_pin = pin + pin / 2;
    }
    return EMI_OK;
}

// handle + every buffer a Schur factorisation of (M, ns, nv) needs, Doff for the mesh
int ws_prepare(KktWorkspace* w, hipStream_t stream, const double* dD, int M, int ns, int nv, std::string* err) {
    const int nh = nv * (nv + 1) / 2, N = (nv + ns) * M, nz = nv * M;
    const size_t md = (size_t)ns * M;
    if (!w->handle) {
        KKT_RB(rocblas_create_handle(&w->handle));
        KKT_RB(rocblas_set_atomics_mode(w->handle, rocblas_atomics_not_allowed));
    }
    if (!w->info) KKT_HIP(hipMalloc(&w->info, sizeof(rocblas_int)));
    if (!w->flag) KKT_HIP(hipMalloc(&w->flag, sizeof(int)));
    if (!w->chol_blk) KKT_HIP(hipMalloc((void**)&w->chol_blk, (CHOL_NB * CHOL_NB + CHOL_NB) * sizeof(double)));
    KKT_ENSURE(w->ipiv, w->cap_ipiv, (size_t)N * sizeof(rocblas_int));
    KKT_ENSURE(w->Q, w->cap_Q, (size_t)nh * M * sizeof(double));
    KKT_ENSURE(w->J, w->cap_J, (size_t)ns * nv * M * sizeof(double));
    KKT_ENSURE(w->fixed, w->cap_fixed, (size_t)nz);
    if (w->S_elems < md * md) {
        if (w->S) KKT_HIP(hipFree(w->S));
        w->S = nullptr;
        w->S_elems = 0;
        KKT_HIP(hipMalloc(&w->S, md * md * sizeof(double)));
        w->S_elems = md * md;
    }
    KKT_ENSURE(w->Pinv, w->cap_Pinv, (size_t)nv * nv * M * sizeof(double));
    KKT_ENSURE(w->G, w->cap_G, (size_t)ns * ns * M * sizeof(double));
    KKT_ENSURE(w->Rk, w->cap_Rk, (size_t)ns * ns * M * sizeof(double));
    if (w->cap_Doff < (size_t)M * M * sizeof(double)) w->doff_M = 0;
    KKT_ENSURE(w->Doff, w->cap_Doff, (size_t)M * M * sizeof(double));
    KKT_ENSURE(w->W, w->cap_W, (size_t)M * M * sizeof(double));
    if (w->doff_M != M || w->doff_src != dD) {
        hipLaunchKernelGGL(emi_kkt_doff_kernel, dim3((unsigned)(((size_t)M * M + 255) / 256)), dim3(256), 0, stream, dD, w->Doff, M);
        KKT_HIP(hipGetLastError());
        w->doff_M = M;
        w->doff_src = dD;
    }
    w->N = N; w->M = M; w->ns = ns; w->nv = nv;
    return EMI_OK;
}

// Blocked Cholesky of na matrices at once (two-level form of chol_blocked2; lower, column-major, lda = n): per 64-column step one
// diagonal-block launch of na workgroups, one panel launch, one batched dgemm for the rest of the outer panel; per outer panel one
// batched dsyrk.  ptrs: device pointer arrays laid out by the caller as  [step][0 = panel, 1 = trailing][na]  for the steps, then
// [outer][0 = panel, 1 = trailing][na]  for the outer updates (host copy hp: the same layout, filled here).
int chol_batched(KktWorkspace* L, hipStream_t stream, const KktDev* d_tab, int na, double* const* Sptr, int n, double** hp, double** dp,
                 size_t* used, std::string* err) {
    const int NB2 = std::max(128, (g_tune.chol_outer.load() / CHOL_NB) * CHOL_NB);
    const double one = 1.0, mone = -1.0;
    size_t q = 0;
    // pass 1: every pointer array on the host (the launches below read them from dp, which the caller uploads AFTER this function
    // has filled hp?  No: the arrays must be on the device before the first launch that uses them, so they are filled first ...)
    struct Step { int j0, nb, rest, wc; size_t at; };
    struct Outer { int J0, Jend, rest2; size_t at; };
    std::vector<Step> steps;
    std::vector<Outer> outers;
    for (int J0 = 0; J0 < n; J0 += NB2) {
        const int Jend = std::min(n, J0 + NB2);
        for (int j0 = J0; j0 < Jend; j0 += CHOL_NB) {
            const int nb = std::min(CHOL_NB, n - j0), rest = n - j0 - nb, wc = Jend - (j0 + nb);
            Step st{j0, nb, rest, wc, q};
            if (rest > 0 && wc > 0) {
                for (int a = 0; a < na; ++a) hp[q + a] = Sptr[a] + (size_t)j0 * n + j0 + nb;                 // panel below the diagonal block
                for (int a = 0; a < na; ++a) hp[q + na + a] = Sptr[a] + (size_t)(j0 + nb) * n + j0 + nb;     // rest of the outer panel
                q += 2 * (size_t)na;
            }
            steps.push_back(st);
        }
        const int rest2 = n - Jend;
        Outer o{J0, Jend, rest2, q};
        if (rest2 > 0) {
            for (int a = 0; a < na; ++a) hp[q + a] = Sptr[a] + (size_t)J0 * n + Jend;
            for (int a = 0; a < na; ++a) hp[q + na + a] = Sptr[a] + (size_t)Jend * n + Jend;
            q += 2 * (size_t)na;
        }
        outers.push_back(o);
    }
    *used = q;
    KKT_HIP(hipMemcpyAsync(dp, hp, q * sizeof(double*), hipMemcpyHostToDevice, stream));
    // ... then the launches
    size_t si = 0;
    for (const Outer& o : outers) {
        for (; si < steps.size() && steps[si].j0 < o.Jend; ++si) {
            const Step& st = steps[si];
            if (st.nb == CHOL_NB)
                hipLaunchKernelGGL(emi_chol_diag_mfma_b_kernel, dim3(na), dim3(64), 0, stream, d_tab, n, st.j0);
            else
                hipLaunchKernelGGL(emi_chol_diag_b_kernel, dim3(na), dim3(256), 0, stream, d_tab, n, st.j0, st.nb);
            if (st.rest <= 0) continue;
            hipLaunchKernelGGL(emi_chol_panel_mfma_b_kernel, dim3((st.rest + 63) / 64, na), dim3(64), 0, stream, d_tab, n, n, st.j0);
            if (st.wc > 0) {
                if (n < g_tune.batch_gemm_rows.load())
                    KKT_RB(rocblas_dgemm_batched(L->handle, rocblas_operation_none, rocblas_operation_transpose, st.rest, st.wc, st.nb, &mone,
                                                 (const double* const*)(dp + st.at), n, (const double* const*)(dp + st.at), n, &one,
                                                 dp + st.at + na, n, na));
                else
                    for (int a = 0; a < na; ++a)
                        KKT_RB(rocblas_dgemm(L->handle, rocblas_operation_none, rocblas_operation_transpose, st.rest, st.wc, st.nb, &mone, hp[st.at + a], n,
                                             hp[st.at + a], n, &one, hp[st.at + na + a], n));
            }
        }
        if (o.rest2 > 0) {
            if (n < g_tune.batch_syrk_rows.load())
                KKT_RB(rocblas_dsyrk_batched(L->handle, rocblas_fill_lower, rocblas_operation_none, o.rest2, o.Jend - o.J0, &mone,
                                             (const double* const*)(dp + o.at), n, &one, dp + o.at + na, n, na));
            else
                for (int a = 0; a < na; ++a)
                    KKT_RB(rocblas_dsyrk(L->handle, rocblas_fill_lower, rocblas_operation_none, o.rest2, o.Jend - o.J0, &mone, hp[o.at + a], n, &one,
                                         hp[o.at + na + a], n));
        }
    }
    KKT_HIP(hipGetLastError());
    return EMI_OK;
}

}  // namespace

// n scenarios, one mesh.  pws[b]: the scenario's workspace slot (created here if empty); dD[b]: its context's differentiation matrix
// on the device; Qblk / Jblk / fixed / dc as kkt_factor, per scenario (host pointers).  info[b]: 0 factorised (Schur path), > 0 singular,
// -1: this scenario needs the single entry point (a node block not positive definite, or the regularisation ladder exhausted: the LU).
int kkt_factor_batch(int n, KktWorkspace** const* pws, hipStream_t stream, const double* const* dD, int M, int ns, int nv,
                     const double* const* Qblk, const double* const* Jblk, const unsigned char* const* fixed, const double* dc, int* info,
                     std::string* err) {
    if (n < 1 || nv > KKT_NV_MAX) { *err = "emi_kkt_factor_batch: bad batch"; return EMI_ERR_ARG; }
    const int nh = nv * (nv + 1) / 2, nz = nv * M, md = ns * M, npairs = ns * (ns + 1) / 2;
    std::vector<KktWorkspace*> W(n);
    for (int b = 0; b < n; ++b) {
        if (!*pws[b]) *pws[b] = new KktWorkspace();
        KktWorkspace* w = W[b] = *pws[b];
        w->factored = false;
        w->lr_active = false;
        w->linv_n = 0;
        w->reg_dc_applied = dc[b];
        w->reg_dw_applied = 0.0;
        if (int st = ws_prepare(w, stream, dD[b], M, ns, nv, err)) return st;
        KKT_HIP(hipMemcpyAsync(w->Q, Qblk[b], (size_t)nh * M * sizeof(double), hipMemcpyHostToDevice, stream));
        KKT_HIP(hipMemcpyAsync(w->J, Jblk[b], (size_t)ns * nv * M * sizeof(double), hipMemcpyHostToDevice, stream));
        KKT_HIP(hipMemcpyAsync(w->fixed, fixed[b], (size_t)nz, hipMemcpyHostToDevice, stream));
        info[b] = -1;
    }
    KktWorkspace* L = W[0];
    KKT_RB(rocblas_set_stream(L->handle, stream));
    // scratch: table, pointer arrays (S build: A, B, one C array per state pair; Cholesky: two arrays per step and per outer panel;
    // block inverses: two arrays per diagonal block), status words
    const int nsteps = (md + CHOL_NB - 1) / CHOL_NB, nblk = (md + TRSV_NB - 1) / TRSV_NB;
    const size_t ptr_count = (size_t)n * (2 + npairs + 2 * (nsteps + nsteps / 2 + 2) + 2 * nblk + 8);
    const size_t tab_bytes = (size_t)n * sizeof(KktDev), ptr_bytes = ptr_count * sizeof(double*), stat_bytes = (size_t)2 * n * sizeof(int);
    if (int st = batch_scratch(L, tab_bytes, ptr_bytes, stat_bytes, err)) return st;
    KktDev* h_tab = reinterpret_cast<KktDev*>(L->b_pin);
    double** h_ptr = reinterpret_cast<double**>(L->b_pin + tab_bytes);
    int* h_stat = reinterpret_cast<int*>(L->b_pin + tab_bytes + ptr_bytes);
    KktDev* d_tab = reinterpret_cast<KktDev*>(L->b_tab);

    static const double LV_DC[5] = {1.0, 1e3, 1e3, 1e3, 1e6}, LV_DW[5] = {0.0, 0.0, 1e-7, 1e-5, 1e-3};       // the ladder of kkt_factor
    constexpr int NLV = 5;
    const int max_lv = g_tune.primal_levels.load() ? NLV : 2;
    std::vector<int> level(n), first(n);
    std::vector<char> done(n, 0);
    for (int b = 0; b < n; ++b) {
        KktWorkspace* w = W[b];
        if (w->reg_M != M || w->reg_ns != ns || w->reg_nv != nv || !g_tune.sticky_reg.load()) {
            w->reg_level = w->reg_hits = 0;
            w->reg_M = M; w->reg_ns = ns; w->reg_nv = nv;
        } else if (w->reg_level > 0 && w->reg_hits >= 4) {
            --w->reg_level;
            w->reg_hits = 0;
        }
        level[b] = first[b] = std::min(w->reg_level, max_lv - 1);
    }
    const unsigned nb2 = (unsigned)(((size_t)M * M + 255) / 256);
    const double one = 1.0, zero = 0.0;
    std::vector<int> act;
    std::vector<double*> Sptr(n);
    for (int round = 0; round < NLV + 1; ++round) {
        act.clear();
        for (int b = 0; b < n; ++b)
            if (!done[b]) act.push_back(b);
        const int na = (int)act.size();
        if (na == 0) break;
        for (int a = 0; a < na; ++a) {
            KktWorkspace* w = W[act[a]];
            const double dc_base = dc[act[a]] > 1e-9 ? dc[act[a]] : 1e-9;
            KktDev t{};
            t.Q = w->Q; t.J = w->J; t.Pinv = w->Pinv; t.G = w->G; t.Rk = w->Rk; t.S = w->S; t.W = w->W; t.chol_blk = w->chol_blk;
            t.Doff = L->Doff;                       // one copy of the operand for the whole batch (same mesh: same matrix)
            t.fixed = w->fixed;
            t.info = L->b_stat + a;
            t.flag = L->b_stat + na + a;
            t.dw = LV_DW[level[act[a]]];
            t.dc = dc_base * LV_DC[level[act[a]]];
            h_tab[a] = t;
            Sptr[a] = w->S;
        }
        // pointer arrays of the S build
        size_t q = 0;
        const size_t at_A = q;  for (int a = 0; a < na; ++a) h_ptr[q++] = L->Doff;
        const size_t at_B = q;  for (int a = 0; a < na; ++a) h_ptr[q++] = W[act[a]]->W;
        const size_t at_C = q;
        for (int i = 0; i < ns; ++i)
            for (int ip = 0; ip <= i; ++ip)
                for (int a = 0; a < na; ++a) h_ptr[q++] = W[act[a]]->S + ((size_t)ip * M) * md + (size_t)i * M;
        KKT_HIP(hipMemcpyAsync(d_tab, h_tab, (size_t)na * sizeof(KktDev), hipMemcpyHostToDevice, stream));
        KKT_HIP(hipMemcpyAsync(L->b_ptrs, h_ptr, q * sizeof(double*), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(emi_kkt_zero_status_b_kernel, dim3((na + 63) / 64), dim3(64), 0, stream, (const KktDev*)d_tab, na);
        if (ns == 6 && nv == 8)
            hipLaunchKernelGGL((emi_kkt_node_inverse_fixed_b_kernel<6, 8>), dim3((M + 63) / 64, na), dim3(64), 0, stream, (const KktDev*)d_tab, M);
        else if (ns == 2 && nv == 4)
            hipLaunchKernelGGL((emi_kkt_node_inverse_fixed_b_kernel<2, 4>), dim3((M + 63) / 64, na), dim3(64), 0, stream, (const KktDev*)d_tab, M);
        else
            hipLaunchKernelGGL(emi_kkt_node_inverse_b_kernel, dim3((M + 63) / 64, na), dim3(64), 0, stream, (const KktDev*)d_tab, M, ns, nv);
        KKT_HIP(hipGetLastError());
        for (int i = 0, p = 0; i < ns; ++i)
            for (int ip = 0; ip <= i; ++ip, ++p) {
                hipLaunchKernelGGL(emi_kkt_scale_b_kernel, dim3(nb2, na), dim3(256), 0, stream, (const KktDev*)d_tab, M, (i * nv + ip) * M);
                if (md < g_tune.batch_gemm_rows.load())
                    KKT_RB(rocblas_dgemm_batched(L->handle, rocblas_operation_transpose, rocblas_operation_none, M, M, M, &one,
                                                 (const double* const*)(L->b_ptrs + at_A), M, (const double* const*)(L->b_ptrs + at_B), M, &zero,
                                                 L->b_ptrs + at_C + (size_t)p * na, (rocblas_int)md, na));
                else
                    for (int a = 0; a < na; ++a)
                        KKT_RB(rocblas_dgemm(L->handle, rocblas_operation_transpose, rocblas_operation_none, M, M, M, &one, L->Doff, M, W[act[a]]->W, M, &zero,
                                             h_ptr[at_C + (size_t)p * na + a], (rocblas_int)md));
                hipLaunchKernelGGL(emi_kkt_sblock_terms_b_kernel, dim3(nb2, na), dim3(256), 0, stream, (const KktDev*)d_tab, M, ns, i, ip);
            }
        KKT_HIP(hipGetLastError());
        size_t used = 0;
        if (int st = chol_batched(L, stream, d_tab, na, Sptr.data(), md, h_ptr + q, L->b_ptrs + q, &used, err)) return st;
        KKT_HIP(hipMemcpyAsync(h_stat, L->b_stat, (size_t)2 * na * sizeof(int), hipMemcpyDeviceToHost, stream));
        KKT_HIP(hipStreamSynchronize(stream));
        for (int a = 0; a < na; ++a) {
            const int b = act[a], hinfo = h_stat[a], hflag = h_stat[na + a];
            KktWorkspace* w = W[b];
            if (hinfo == 0 && hflag == 0) {
                done[b] = 1;
                info[b] = 0;
                if (level[b] == first[b]) ++w->reg_hits; else { w->reg_level = level[b]; w->reg_hits = 0; }
                w->reg_dc_applied = h_tab[a].dc;
                w->reg_dw_applied = h_tab[a].dw;
                w->factored = true;
                w->method_used = 1;
            } else if (hflag != 0 || level[b] + 1 >= max_lv) {
                done[b] = 1;                        // not the quasi-definite case, or no level helps: the single path (and its LU) decides
                info[b] = -1;
                if (hflag == 0) { w->reg_level = max_lv - 1; w->reg_hits = 0; }
                if (g_tune.debug.load())
                    fprintf(stderr, "emi_kkt_factor_batch: scenario %d of %d leaves the batch (block flag %d, potrf info %d, M %d)\n", b, n, hflag, hinfo, M);
            } else {
                if (g_tune.debug.load())
                    fprintf(stderr, "emi_kkt_factor_batch: S not positive definite at %d (scenario %d, dual regularisation %.1e, primal %.1e, M %d), retrying\n",
                            hinfo, b, h_tab[a].dc, h_tab[a].dw, M);
                ++level[b];
            }
        }
    }
    // block inverses for the single-right-hand-side solves of everything that was factorised
    if (g_tune.block_trsv.load() && md >= 2 * TRSV_NB) {
        act.clear();
        for (int b = 0; b < n; ++b)
            if (info[b] == 0) act.push_back(b);
        const int na = (int)act.size();
        if (na > 0) {
            const int full = md / TRSV_NB, tail = md - full * TRSV_NB;
            size_t q = 0;
            for (int a = 0; a < na; ++a) {
                KktWorkspace* w = W[act[a]];
                KKT_ENSURE(w->Linv, w->cap_Linv, (size_t)nblk * TRSV_NB * TRSV_NB * sizeof(double));
                KKT_ENSURE(w->LinvT, w->cap_LinvT, (size_t)nblk * TRSV_NB * TRSV_NB * sizeof(double));
                KKT_HIP(hipMemsetAsync(w->Linv, 0, (size_t)nblk * TRSV_NB * TRSV_NB * sizeof(double), stream));
                KktDev t{};
                t.Linv = w->Linv;
                t.LinvT = w->LinvT;
                h_tab[a] = t;
            }
            const size_t at_LA = q;
            for (int j = 0; j < full; ++j)
                for (int a = 0; a < na; ++a) h_ptr[q++] = W[act[a]]->S + (size_t)j * TRSV_NB * ((size_t)md + 1);
            const size_t at_LI = q;
            for (int j = 0; j < full; ++j)
                for (int a = 0; a < na; ++a) h_ptr[q++] = W[act[a]]->Linv + (size_t)j * TRSV_NB * TRSV_NB;
            const size_t at_TA = q;
            for (int a = 0; a < na; ++a) h_ptr[q++] = W[act[a]]->S + (size_t)full * TRSV_NB * ((size_t)md + 1);
            const size_t at_TI = q;
            for (int a = 0; a < na; ++a) h_ptr[q++] = W[act[a]]->Linv + (size_t)full * TRSV_NB * TRSV_NB;
            KKT_HIP(hipMemcpyAsync(d_tab, h_tab, (size_t)na * sizeof(KktDev), hipMemcpyHostToDevice, stream));
            KKT_HIP(hipMemcpyAsync(L->b_ptrs, h_ptr, q * sizeof(double*), hipMemcpyHostToDevice, stream));
            if (full > 0 && md < g_tune.batch_trtri_rows.load())
                KKT_RB(rocblas_dtrtri_batched(L->handle, rocblas_fill_lower, rocblas_diagonal_non_unit, TRSV_NB,
                                              (const double* const*)(L->b_ptrs + at_LA), md, L->b_ptrs + at_LI, TRSV_NB, full * na));
            else if (full > 0)
                for (int a = 0; a < na; ++a)
                    KKT_RB(rocblas_dtrtri_strided_batched(L->handle, rocblas_fill_lower, rocblas_diagonal_non_unit, TRSV_NB, W[act[a]]->S, md,
                                                          (rocblas_stride)TRSV_NB * (md + 1), W[act[a]]->Linv, TRSV_NB,
                                                          (rocblas_stride)TRSV_NB * TRSV_NB, full));
            if (tail > 0)
                KKT_RB(rocblas_dtrtri_batched(L->handle, rocblas_fill_lower, rocblas_diagonal_non_unit, tail,
                                              (const double* const*)(L->b_ptrs + at_TA), md, L->b_ptrs + at_TI, TRSV_NB, na));
            hipLaunchKernelGGL(emi_trsv_transpose_b_kernel, dim3(TRSV_NB / 32, TRSV_NB / 32, nblk * na), dim3(256), 0, stream, (const KktDev*)d_tab, nblk);
            KKT_HIP(hipGetLastError());
            KKT_HIP(hipStreamSynchronize(stream));          // (the pinned table is reused by the next call)
            for (int a = 0; a < na; ++a) W[act[a]]->linv_n = md;
        }
    }
    return EMI_OK;
}

// ---- batched solves with the refinement on the device ---------------------------------------------------------------------------
namespace {

// r = b - K x per scenario, K = [[Q - U Delta U^T (while a low-rank correction is active), J^T], [J, -dc I]] with the nominal node
// blocks Q and the nominal dc (what the caller means by "the matrix"; the factorisation may hold a regularised one).  Node part here,
// the two D products as batched GEMMs around it (kkt_residual).  xs = x with fixed variables zeroed (in T), Cb = Doff xs_states.
__global__ void emi_kkt_mask_copy_b_kernel(const KktDev* __restrict__ tab, int nz) {
    const KktDev t = tab[blockIdx.y];
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nz) t.T[q] = t.fixed[q] ? 0.0 : t.xx[q];
}
__global__ void emi_kkt_residual_node_b_kernel(const KktDev* __restrict__ tab, int M, int ns, int nv, double dc) {
    const KktDev t = tab[blockIdx.y];
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M) return;
    const int nz = nv * M;
    double x[KKT_NV_MAX], lam[KKT_NV_MAX];
    for (int q = 0; q < nv; ++q) x[q] = t.T[(size_t)q * M + k];
    for (int i = 0; i < ns; ++i) lam[i] = t.xx[(size_t)nz + (size_t)i * M + k];
    for (int i = 0; i < ns; ++i) {
        double acc = t.Cb[(size_t)i * M + k] - dc * lam[i];
        for (int v = 0; v < nv; ++v) acc += t.J[(size_t)(i * nv + v) * M + k] * x[v];
        t.rhs[(size_t)nz + (size_t)i * M + k] = t.bb[(size_t)nz + (size_t)i * M + k] - acc;
    }
    for (int v = 0; v < nv; ++v) {
        double acc = 0.0;
        for (int q = 0; q < nv; ++q) {
            const int hi = v > q ? v : q, lo = v > q ? q : v;
            acc += t.Q[(size_t)(hi * (hi + 1) / 2 + lo) * M + k] * x[q];
        }
        for (int i = 0; i < ns; ++i) acc += t.J[(size_t)(i * nv + v) * M + k] * lam[i];
        t.rhs[(size_t)v * M + k] = t.bb[(size_t)v * M + k] - acc;      // (fixed variables: masked after the D product)
    }
}
// + U Delta (U^T x) of the scenarios whose low-rank correction is active: one thread per FIRST column of a node walks the node's columns
// (the columns come node by node: fixed order of the additions)
__global__ void emi_kkt_residual_lr_b_kernel(const KktDev* __restrict__ tab, int M, int nv) {
    const KktDev t = tab[blockIdx.y];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= t.lr_r) return;
    const int node = t.lr_node[c];
    if (c > 0 && t.lr_node[c - 1] == node) return;
    for (int a = c; a < t.lr_r && t.lr_node[a] == node; ++a) {
        double dot = 0.0;
        for (int v = 0; v < nv; ++v) dot += t.lr_vec[(size_t)a * nv + v] * t.T[(size_t)v * M + node];
        dot *= t.lr_delta[a];
        for (int v = 0; v < nv; ++v) t.rhs[(size_t)v * M + node] += t.lr_vec[(size_t)a * nv + v] * dot;
    }
}
// stat[item] = max |v| over the N entries of the chosen vector (which: 0 rhs, 1 bb); NaN propagates as +inf
__global__ __launch_bounds__(256) void emi_kkt_absmax_b_kernel(const KktDev* __restrict__ tab, int N, int which, double* __restrict__ out) {
    __shared__ double red[256];
    const KktDev t = tab[blockIdx.x];
    const double* v = which ? t.bb : t.rhs;
    double m = 0.0;
    for (int q = threadIdx.x; q < N; q += 256) {
        const double a = fabs(v[q]);
        m = (a > m || a != a) ? (a != a ? INFINITY : a) : m;
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int sft = 128; sft > 0; sft >>= 1) {
        if (threadIdx.x < sft) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + sft]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}
// dst <- src (+ add) over N entries: sel 0: rhs <- bb; 1: xx <- rhs; 2: xp <- xx; 3: xx <- xp; 4: xx += rhs
__global__ void emi_kkt_vecop_b_kernel(const KktDev* __restrict__ tab, int N, int sel) {
    const KktDev t = tab[blockIdx.y];
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= N) return;
    if (sel == 0) t.rhs[q] = t.bb[q];
    else if (sel == 1) t.xx[q] = t.rhs[q];
    else if (sel == 2) t.xp[q] = t.xx[q];
    else if (sel == 3) t.xx[q] = t.xp[q];
    else t.xx[q] += t.rhs[q];
}

struct SolveShape { int M, ns, nv, N, nz, md, nblk; bool blk; };

void fill_solve_entry(KktDev& t, KktWorkspace* w, const KktWorkspace* L) {
    t.Q = w->Q; t.J = w->J; t.Pinv = w->Pinv; t.S = w->S; t.Linv = w->Linv; t.LinvT = w->LinvT; t.T = w->T; t.Cb = w->Cb; t.rhs = w->rhs;
    t.y = w->trsv_y; t.bb = w->ref_b; t.xx = w->ref_x; t.xp = w->ref_p;
    t.fixed = w->fixed;
    t.Doff = L->Doff;
    t.lr_r = w->lr_active ? w->lr_r : 0;
    t.lr_node = w->lr_node; t.lr_vec = w->lr_vec; t.lr_delta = w->lr_delta;
}

// table + the pointer arrays every batched call below needs, for the n workspaces of this step
struct SolveArrays {
    KktDev* d_tab;
    double** P;
    size_t at_D, at_T, at_Cb, at_X, at_S, at_Y, at_XX, at_lam;
    std::vector<size_t> at_LT, at_LI, at_xj, at_yj, at_Sr;
};
int solve_arrays(KktWorkspace* L, hipStream_t stream, const SolveShape& sh, int n, KktWorkspace* const* ws, SolveArrays* A, std::string* err) {
    const size_t ptr_count = (size_t)n * (8 + (sh.blk ? 5 * sh.nblk : 0));
    const size_t tab_bytes = (size_t)n * sizeof(KktDev), ptr_bytes = ptr_count * sizeof(double*), stat_bytes = (size_t)4 * n * sizeof(double);
    if (int st = batch_scratch(L, tab_bytes, ptr_bytes, stat_bytes, err)) return st;
    KktDev* h_tab = reinterpret_cast<KktDev*>(L->b_pin);
    double** h_ptr = reinterpret_cast<double**>(L->b_pin + tab_bytes);
    A->d_tab = reinterpret_cast<KktDev*>(L->b_tab);
    A->P = L->b_ptrs;
    for (int b = 0; b < n; ++b) {
        KktDev t{};
        fill_solve_entry(t, ws[b], L);
        h_tab[b] = t;
    }
    size_t q = 0;
    auto arr = [&](auto f) { const size_t at = q; for (int b = 0; b < n; ++b) h_ptr[q++] = f(ws[b]); return at; };
    A->at_D = arr([&](KktWorkspace*) { return L->Doff; });
    A->at_T = arr([](KktWorkspace* w) { return w->T; });
    A->at_Cb = arr([](KktWorkspace* w) { return w->Cb; });
    A->at_X = arr([](KktWorkspace* w) { return w->rhs; });
    A->at_S = arr([](KktWorkspace* w) { return w->S; });
    A->at_Y = arr([](KktWorkspace* w) { return w->trsv_y; });
    A->at_XX = arr([](KktWorkspace* w) { return w->ref_x; });
    A->at_lam = arr([&](KktWorkspace* w) { return w->ref_x + sh.nz; });
    A->at_LT.assign(sh.nblk, 0); A->at_LI.assign(sh.nblk, 0); A->at_xj.assign(sh.nblk, 0); A->at_yj.assign(sh.nblk, 0); A->at_Sr.assign(sh.nblk, 0);
    if (sh.blk)
        for (int j = 0; j < sh.nblk; ++j) {
            const size_t j0 = (size_t)j * TRSV_NB;
            A->at_LT[j] = arr([&](KktWorkspace* w) { return w->LinvT + (size_t)j * TRSV_NB * TRSV_NB; });
            A->at_LI[j] = arr([&](KktWorkspace* w) { return w->Linv + (size_t)j * TRSV_NB * TRSV_NB; });
            A->at_xj[j] = arr([&](KktWorkspace* w) { return w->Cb + j0; });
            A->at_yj[j] = arr([&](KktWorkspace* w) { return w->trsv_y + j0; });
            A->at_Sr[j] = arr([&](KktWorkspace* w) { return w->S + j0; });
        }
    // the previous step's launches read the device copies: they are done before the pinned staging is overwritten, because every
    // step of the callers ends with a stream synchronisation (status words) -- except the first, which has nothing in flight
    KKT_HIP(hipMemcpyAsync(A->d_tab, h_tab, (size_t)n * sizeof(KktDev), hipMemcpyHostToDevice, stream));
    KKT_HIP(hipMemcpyAsync(A->P, h_ptr, q * sizeof(double*), hipMemcpyHostToDevice, stream));
    return EMI_OK;
}

// rhs <- K~^-1 rhs (+ Woodbury term of the scenarios that hold a low-rank correction), in place on every ws[b]->rhs
int solve_core(KktWorkspace* L, hipStream_t stream, const SolveShape& sh, int n, KktWorkspace* const* ws, const SolveArrays& A, std::string* err) {
    const int M = sh.M, ns = sh.ns, nv = sh.nv, N = sh.N, nz = sh.nz, md = sh.md, nblk = sh.nblk;
    const KktDev* d_tab = A.d_tab;
    double** P = A.P;
    const double one = 1.0, zero = 0.0, mone = -1.0;
    dim3 gk((M + 127) / 128, n), bk(128);
    hipLaunchKernelGGL(emi_kkt_mask_rhs_b_kernel, dim3((nz + 255) / 256, n), dim3(256), 0, stream, d_tab, nz, N);
    hipLaunchKernelGGL(emi_kkt_apply_p_b_kernel, gk, bk, 0, stream, d_tab, M, nv, 0);           // t = P a
    KKT_HIP(hipGetLastError());
    KKT_RB(rocblas_dgemm_batched(L->handle, rocblas_operation_transpose, rocblas_operation_none, M, ns, M, &one, (const double* const*)(P + A.at_D), M,
                                 (const double* const*)(P + A.at_T), M, &zero, P + A.at_Cb, M, n));                 // Cb = Doff t_states
    hipLaunchKernelGGL(emi_kkt_jnode_minus_b_b_kernel, gk, bk, 0, stream, d_tab, M, ns, nv);    // Cb += J_node t - b
    KKT_HIP(hipGetLastError());
    if (sh.blk) {                                       // lambda = S^-1 Cb through the block inverses (blk_potrs, gemv form)
        for (int j = 0; j < nblk; ++j) {
            const int j0 = j * TRSV_NB, bs = std::min(TRSV_NB, md - j0), rest = md - j0 - bs;
            KKT_RB(rocblas_dgemv_batched(L->handle, rocblas_operation_transpose, bs, bs, &one, (const double* const*)(P + A.at_LT[j]), TRSV_NB,
                                         (const double* const*)(P + A.at_xj[j]), 1, &zero, P + A.at_yj[j], 1, n));
            if (rest > 0)
                hipLaunchKernelGGL(emi_trsv_update_b_kernel, dim3((rest + 63) / 64, n), dim3(256), 0, stream, d_tab, md, j0, bs, rest);
        }
        for (int j = nblk - 1; j >= 0; --j) {
            const int j0 = j * TRSV_NB, bs = std::min(TRSV_NB, md - j0);
            KKT_RB(rocblas_dgemv_batched(L->handle, rocblas_operation_transpose, bs, bs, &one, (const double* const*)(P + A.at_LI[j]), TRSV_NB,
                                         (const double* const*)(P + A.at_yj[j]), 1, &zero, P + A.at_xj[j], 1, n));
            if (j0 > 0)
                KKT_RB(rocblas_dgemv_batched(L->handle, rocblas_operation_transpose, bs, j0, &mone, (const double* const*)(P + A.at_Sr[j]), md,
                                             (const double* const*)(P + A.at_xj[j]), 1, &one, P + A.at_Y, 1, n));
        }
        KKT_HIP(hipGetLastError());
    } else {
        KKT_RB(rocsolver_dpotrs_batched(L->handle, rocblas_fill_lower, md, 1, P + A.at_S, md, P + A.at_Cb, md, n));
    }
    KKT_RB(rocblas_dgemm_batched(L->handle, rocblas_operation_none, rocblas_operation_none, M, ns, M, &mone, (const double* const*)(P + A.at_D), M,
                                 (const double* const*)(P + A.at_Cb), M, &one, P + A.at_X, M, n));                  // y = a - J^T lambda ...
    hipLaunchKernelGGL(emi_kkt_jnode_t_b_kernel, gk, bk, 0, stream, d_tab, M, ns, nv);
    hipLaunchKernelGGL(emi_kkt_apply_p_b_kernel, gk, bk, 0, stream, d_tab, M, nv, 1);           // ... x = P y
    hipLaunchKernelGGL(emi_kkt_finish_solve_b_kernel, dim3((N + 255) / 256, n), dim3(256), 0, stream, d_tab, nz, md);
    KKT_HIP(hipGetLastError());
    for (int b = 0; b < n; ++b) {
        KktWorkspace* w = ws[b];
        if (!w->lr_active) continue;    // x <- x + Y C^-1 (U^T x), with the scenario's own handle on this stream
        const int r = w->lr_r;
        KKT_RB(rocblas_set_stream(w->handle, stream));
        hipLaunchKernelGGL(emi_kkt_lr_utx_kernel, dim3((r + 63) / 64, 1), dim3(64), 0, stream, w->lrT, w->rhs, w->lr_node, w->lr_vec, r, N, M, nv);
        KKT_HIP(hipGetLastError());
        KKT_RB(rocsolver_dpotrs(w->handle, rocblas_fill_lower, r, 1, w->lrC, r, w->lrT, r));
        KKT_RB(rocblas_dgemv(w->handle, rocblas_operation_none, N, r, &one, w->lrY, N, w->lrT, 1, &one, w->rhs, 1));
    }
    return EMI_OK;
}

// rhs <- bb - K xx for every scenario of the table; out_max[b] = max |rhs_b| (device array of doubles)
int residual_core(KktWorkspace* L, hipStream_t stream, const SolveShape& sh, int n, const SolveArrays& A, double dc_nominal, double* d_max,
                  std::string* err) {
    const int M = sh.M, ns = sh.ns, nv = sh.nv, N = sh.N, nz = sh.nz;
    const double one = 1.0, zero = 0.0, mone = -1.0;
    hipLaunchKernelGGL(emi_kkt_mask_copy_b_kernel, dim3((nz + 255) / 256, n), dim3(256), 0, stream, (const KktDev*)A.d_tab, nz);
    KKT_RB(rocblas_dgemm_batched(L->handle, rocblas_operation_transpose, rocblas_operation_none, M, ns, M, &one, (const double* const*)(A.P + A.at_D), M,
                                 (const double* const*)(A.P + A.at_T), M, &zero, A.P + A.at_Cb, M, n));            // Cb = Doff xs_states
    hipLaunchKernelGGL(emi_kkt_residual_node_b_kernel, dim3((M + 127) / 128, n), dim3(128), 0, stream, (const KktDev*)A.d_tab, M, ns, nv, dc_nominal);
    KKT_HIP(hipGetLastError());
    KKT_RB(rocblas_dgemm_batched(L->handle, rocblas_operation_none, rocblas_operation_none, M, ns, M, &mone, (const double* const*)(A.P + A.at_D), M,
                                 (const double* const*)(A.P + A.at_lam), M, &one, A.P + A.at_X, M, n));            // states' rows -= Doff^T lambda
    hipLaunchKernelGGL(emi_kkt_residual_lr_b_kernel, dim3(64, n), dim3(64), 0, stream, (const KktDev*)A.d_tab, M, nv);
    hipLaunchKernelGGL(emi_kkt_mask_rhs_b_kernel, dim3((nz + 255) / 256, n), dim3(256), 0, stream, (const KktDev*)A.d_tab, nz, N);
    hipLaunchKernelGGL(emi_kkt_absmax_b_kernel, dim3(n), dim3(256), 0, stream, (const KktDev*)A.d_tab, N, 0, d_max);
    KKT_HIP(hipGetLastError());
    return EMI_OK;
}

int solve_buffers(KktWorkspace* w, int N, int nz, int md, bool refine, std::string* err) {
    if (w->rhs_elems < (size_t)N) {
        if (w->rhs) KKT_HIP(hipFree(w->rhs));
        w->rhs = nullptr; w->rhs_elems = 0;
        KKT_HIP(hipMalloc(&w->rhs, (size_t)N * sizeof(double)));
        w->rhs_elems = (size_t)N;
    }
    if (w->T_elems < (size_t)nz) {
        if (w->T) KKT_HIP(hipFree(w->T));
        w->T = nullptr; w->T_elems = 0;
        KKT_HIP(hipMalloc(&w->T, (size_t)nz * sizeof(double)));
        w->T_elems = (size_t)nz;
    }
    if (w->Cb_elems < (size_t)md) {
        if (w->Cb) KKT_HIP(hipFree(w->Cb));
        w->Cb = nullptr; w->Cb_elems = 0;
        KKT_HIP(hipMalloc(&w->Cb, (size_t)md * sizeof(double)));
        w->Cb_elems = (size_t)md;
    }
    KKT_ENSURE(w->trsv_y, w->cap_trsv_y, (size_t)md * sizeof(double));
    if (refine) {
        KKT_ENSURE(w->ref_b, w->cap_ref_b, (size_t)N * sizeof(double));
        KKT_ENSURE(w->ref_x, w->cap_ref_x, (size_t)N * sizeof(double));
        KKT_ENSURE(w->ref_p, w->cap_ref_p, (size_t)N * sizeof(double));
    }
    return EMI_OK;
}

int solve_shape(int n, KktWorkspace* const* ws, SolveShape* sh, std::string* err) {
    KktWorkspace* L = ws[0];
    sh->M = L->M; sh->ns = L->ns; sh->nv = L->nv; sh->N = L->N; sh->nz = L->nv * L->M; sh->md = L->ns * L->M;
    sh->nblk = (sh->md + TRSV_NB - 1) / TRSV_NB;
    sh->blk = true;
    for (int b = 0; b < n; ++b) {
        KktWorkspace* w = ws[b];
        if (!w || !w->factored || w->method_used != 1 || w->M != sh->M || w->ns != sh->ns || w->nv != sh->nv || w->N != sh->N) {
            *err = "batched solve: every scenario must hold a Schur factorisation on the same mesh";
            return EMI_ERR_STATE;
        }
        sh->blk = sh->blk && w->linv_n == sh->md;
    }
    return EMI_OK;
}

}  // namespace

// One right-hand side per scenario, all factorised on the same mesh (Schur path): rhs[b] [N] host, in place.  Scenarios with an active
// low-rank correction get their Woodbury term after the common part.
int kkt_solve_batch(int n, KktWorkspace* const* ws, hipStream_t stream, int nz, double* const* rhs, std::string* err) {
    if (n < 1) { *err = "emi_kkt_solve_batch: empty batch"; return EMI_ERR_ARG; }
    (void)nz;
    SolveShape sh;
    if (int st = solve_shape(n, ws, &sh, err)) return st;
    KktWorkspace* L = ws[0];
    for (int b = 0; b < n; ++b) {
        if (int st = solve_buffers(ws[b], sh.N, sh.nz, sh.md, false, err)) return st;
        KKT_HIP(hipMemcpyAsync(ws[b]->rhs, rhs[b], (size_t)sh.N * sizeof(double), hipMemcpyHostToDevice, stream));
    }
    KKT_RB(rocblas_set_stream(L->handle, stream));
    SolveArrays A;
    if (int st = solve_arrays(L, stream, sh, n, ws, &A, err)) return st;
    if (int st = solve_core(L, stream, sh, n, ws, A, err)) return st;
    for (int b = 0; b < n; ++b) KKT_HIP(hipMemcpyAsync(rhs[b], ws[b]->rhs, (size_t)sh.N * sizeof(double), hipMemcpyDeviceToHost, stream));
    KKT_HIP(hipStreamSynchronize(stream));
    return EMI_OK;
}

// The Newton step WITH its iterative refinement on the device: x = K~^-1 b, then up to max_steps rounds of  r = b - K x,  x += K~^-1 r
// against the NOMINAL matrix K (nominal node blocks and dc_nominal[b]; minus the low-rank term while a scenario's correction is
// active, i.e. the matrix its exact step belongs to) -- the loop of the host iteration (emi_nlp.cpp), rule for rule: stop when the
// residual is at round-off or no longer halves; a correction that made the residual WORSE is taken back.  Only the residual norms
// cross to the host (n doubles per round).  Out per scenario: rel[b] = final max |r| / max(1, max |b|), nsolve[b] = solves used,
// reverted[b] = 1 if the last correction was taken back, status[b] = 0 ok, 2 the first solution is not finite (the caller regularises).
int kkt_solve_refined_batch(int n, KktWorkspace* const* ws, hipStream_t stream, double* const* rhs, const double* dc_nominal, int max_steps,
                            double* rel, int* nsolve, int* reverted, int* status, std::string* err) {
    if (n < 1) { *err = "emi_kkt_solve_refined_batch: empty batch"; return EMI_ERR_ARG; }
    SolveShape sh;
    if (int st = solve_shape(n, ws, &sh, err)) return st;
    KktWorkspace* L = ws[0];
    const int N = sh.N;
    const double refine_rel = std::pow(10.0, -(double)g_tune.refine_exp.load());
    for (int b = 0; b < n; ++b) {
        if (int st = solve_buffers(ws[b], sh.N, sh.nz, sh.md, true, err)) return st;
        KKT_HIP(hipMemcpyAsync(ws[b]->ref_b, rhs[b], (size_t)N * sizeof(double), hipMemcpyHostToDevice, stream));
        rel[b] = 0.0; nsolve[b] = 0; reverted[b] = 0; status[b] = 0;
    }
    KKT_RB(rocblas_set_stream(L->handle, stream));
    double* d_stat = reinterpret_cast<double*>(L->b_stat);              // (batch_scratch sizes it for 4 n doubles)
    std::vector<double> h_max(n), bmax(n), prev(n, 1e300), rlast(n, 0.0);
    std::vector<char> have_prev(n, 0);
    std::vector<KktWorkspace*> act(ws, ws + n);
    std::vector<int> idx(n);
    for (int b = 0; b < n; ++b) idx[b] = b;
    SolveArrays A;
    if (int st = solve_arrays(L, stream, sh, n, act.data(), &A, err)) return st;
    d_stat = reinterpret_cast<double*>(L->b_stat);
    // mask the right-hand sides as the single path does (fixed variables: 0), |b|, first solve
    dim3 gN((N + 255) / 256, n);
    hipLaunchKernelGGL(emi_kkt_vecop_b_kernel, gN, dim3(256), 0, stream, (const KktDev*)A.d_tab, N, 0);                 // rhs <- bb
    hipLaunchKernelGGL(emi_kkt_mask_rhs_b_kernel, dim3((sh.nz + 255) / 256, n), dim3(256), 0, stream, (const KktDev*)A.d_tab, sh.nz, N);
    hipLaunchKernelGGL(emi_kkt_absmax_b_kernel, dim3(n), dim3(256), 0, stream, (const KktDev*)A.d_tab, N, 0, d_stat);    // max |b| (masked)
    KKT_HIP(hipMemcpyAsync(bmax.data(), d_stat, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, stream));
    // (bb itself stays unmasked in the fixed slots; the residual kernels mask their result, so those entries never matter)
    if (int st = solve_core(L, stream, sh, n, act.data(), A, err)) return st;
    hipLaunchKernelGGL(emi_kkt_vecop_b_kernel, gN, dim3(256), 0, stream, (const KktDev*)A.d_tab, N, 1);                 // xx <- rhs
    for (int b = 0; b < n; ++b) nsolve[b] = 1;
    for (int ir = 0; ir <= max_steps; ++ir) {
        const int na = (int)act.size();
        if (na == 0) break;
        double dcn = dc_nominal[idx[0]];
        bool same_dc = true;
        for (int a = 1; a < na; ++a) same_dc = same_dc && dc_nominal[idx[a]] == dcn;
        if (same_dc) {
            if (int st = residual_core(L, stream, sh, na, A, dcn, d_stat, err)) return st;
        } else {                                                        // (different nominal dc in one batch: one scenario at a time)
            for (int a = 0; a < na; ++a) {
                SolveArrays A1;
                KktWorkspace* one_ws = act[a];
                KKT_HIP(hipStreamSynchronize(stream));
                if (int st = solve_arrays(L, stream, sh, 1, &one_ws, &A1, err)) return st;
                if (int st = residual_core(L, stream, sh, 1, A1, dc_nominal[idx[a]], d_stat + a, err)) return st;
            }
            KKT_HIP(hipStreamSynchronize(stream));
            if (int st = solve_arrays(L, stream, sh, na, act.data(), &A, err)) return st;
        }
        KKT_HIP(hipMemcpyAsync(h_max.data(), d_stat, (size_t)na * sizeof(double), hipMemcpyDeviceToHost, stream));
        KKT_HIP(hipStreamSynchronize(stream));
        std::vector<KktWorkspace*> next;
        std::vector<int> next_idx, revert;
        for (int a = 0; a < na; ++a) {
            const int b = idx[a];
            const double rmax = h_max[a];
            if (ir == 0 && !std::isfinite(rmax)) { status[b] = 2; rlast[b] = rmax; continue; }
            if (have_prev[b] && !(rmax < prev[b])) {                     // the last correction did harm: undo it and stop
                revert.push_back(a);
                reverted[b] = 1;
                rlast[b] = prev[b];
                continue;
            }
            rlast[b] = rmax;
            if (ir == max_steps || !(rmax > refine_rel * std::max(1.0, bmax[b])) || !(rmax < 0.5 * prev[b])) continue;
            prev[b] = rmax;
            next.push_back(act[a]);
            next_idx.push_back(b);
        }
        if (!revert.empty()) {                                           // xx <- xp for those (their own small table)
            std::vector<KktWorkspace*> rv;
            for (int a : revert) rv.push_back(act[a]);
            SolveArrays Ar;
            if (int st = solve_arrays(L, stream, sh, (int)rv.size(), rv.data(), &Ar, err)) return st;
            hipLaunchKernelGGL(emi_kkt_vecop_b_kernel, dim3((N + 255) / 256, (unsigned)rv.size()), dim3(256), 0, stream, (const KktDev*)Ar.d_tab, N, 3);
            KKT_HIP(hipGetLastError());
            KKT_HIP(hipStreamSynchronize(stream));
        }
        act.swap(next);
        idx.swap(next_idx);
        const int nc = (int)act.size();
        if (nc == 0) break;
        if (int st = solve_arrays(L, stream, sh, nc, act.data(), &A, err)) return st;
        dim3 gc((N + 255) / 256, nc);
        hipLaunchKernelGGL(emi_kkt_vecop_b_kernel, gc, dim3(256), 0, stream, (const KktDev*)A.d_tab, N, 2);             // xp <- xx
        if (int st = solve_core(L, stream, sh, nc, act.data(), A, err)) return st;                                      // rhs (= r) <- K~^-1 r
        hipLaunchKernelGGL(emi_kkt_vecop_b_kernel, gc, dim3(256), 0, stream, (const KktDev*)A.d_tab, N, 4);             // xx += correction
        KKT_HIP(hipGetLastError());
        for (int a = 0; a < nc; ++a) { ++nsolve[idx[a]]; have_prev[idx[a]] = 1; }
    }
    for (int b = 0; b < n; ++b) {
        rel[b] = rlast[b] / std::max(1.0, bmax[b]);
        KKT_HIP(hipMemcpyAsync(rhs[b], ws[b]->ref_x, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, stream));
    }
    KKT_HIP(hipStreamSynchronize(stream));
    return EMI_OK;
#undef KKT_ENSURE
#undef KKT_HIP
#undef KKT_RB
}

}  // namespace emi
