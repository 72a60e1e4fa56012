/*
 * emi355x.h -- C ABI of libemi355x, the MI355X (gfx950) collocation evaluator.
 *
 * This is the drop-in boundary of the eMI355X eSolver.  It is the only thing
 * `src/eMI355X/eMI355X.cpp` (our ETOL::TrajectoryOptimizer peer of ePSOPT)
 * calls to reach the GPU.  Plain pointers and sizes only: no C++ types, no
 * torch types, no exceptions cross this boundary; every call returns an int
 * status (EMI_OK == 0) and the text of the last failure is kept per context.
 *
 * What each entry point replaces in the reference (paths under the ETOL tree;
 * "[PSOPT]" marks arithmetic that the reference delegates to PSOPT 5.0.0,
 * whose sources are not part of the reference tree):
 *
 *   emi_lgl            [PSOPT] LGL nodes / weights / differentiation matrix
 *                      selected by  src/ePSOPT/ePSOPT.cpp:68
 *                      (collocation_method = "Legendre"), node count :44-45
 *   emi_set_mesh       the (t0,tf) fixed-horizon mapping, ePSOPT.cpp:151-154
 *   emi_set_model      the per-node callbacks installed at ePSOPT.cpp:76-80
 *                      (integrand_cost :186-216, dae :218-276) -- the user's
 *                      f_t closures cannot run on the device, so a model id
 *                      + parameter block selects a hand-written kernel
 *   emi_set_model_source  the same callbacks for a model the library has no
 *                      kernel for: where ePSOPT records the user's closures on
 *                      an ADOL-C tape and interprets it at every evaluation
 *                      (derivatives = "automatic", ePSOPT.cpp:64), the host
 *                      records them once, differentiates the trace and hands
 *                      over the text of a model struct; it is compiled for
 *                      gfx950 at this call and runs in the same kernels
 *   emi_set_path       the path rows counted at ePSOPT.cpp:58 and produced
 *                      at :261-270; row formulas follow
 *                      src/Examples/PSOPT/etol_psopt_example1.cpp:163-182
 *                      (ellipse per polygon edge) and :243-247 (disc)
 *   emi_set_tracks     moving-disc centres, etol_psopt_example1.cpp:233-241
 *   emi_eval_*         one pass of the hot loop of ::psopt (ePSOPT.cpp:84):
 *                      dae + integrand_cost at every node, the defect
 *                      D.X - (tf-t0)/2.F, the cost quadrature, and what
 *                      ADOL-C's sparse_jac/gradient drivers return [PSOPT]
 *   emi_hess_*         what ADOL-C's sparse_hess returns for hessian="exact"
 *                      (ePSOPT.cpp:65) [PSOPT]
 *   emi_jac_structure  the sparsity pattern IPOPT is given [PSOPT]
 *   emi_kkt_factor/_solve  the linear solve of IPOPT's Newton step, reached
 *                      through ::psopt at ePSOPT.cpp:84 with nlp_method "IPOPT"
 *                      (:62) [IPOPT]: the primal-dual KKT matrix is assembled
 *                      and factorised on the device
 *   emi_kkt_solve_refined  the iterative refinement IPOPT wraps around that
 *                      solve [IPOPT]: residual, correction and the revert of
 *                      a correction that made it worse stay on the device
 *   emi_kkt_*_batch    no counterpart: the reference runs one trajectory per
 *                      process; the Newton steps of several scenarios on one
 *                      mesh go through one sequence of batched launches
 *   emi_plan_pass      no counterpart: the launch form emi_eval_* takes for a
 *                      batch size, so that tests and tools query the policy
 *                      instead of restating it
 *
 * Threading: a context must be driven by one caller thread at a time
 * (the reference is single-threaded throughout).
 * Different contexts may be driven by different threads concurrently (each
 * owns its streams, rocBLAS handle and workspaces); results do not depend
 * on what runs beside a context (tests/test_gpu_kkt.py,
 * test_concurrent_contexts_give_reproducible_factorisations).
 *
 * Data layout (all arrays dense, real type = double unless the context was
 * created with emi_create_f32):
 *   X     [B][ns][M]   state trajectories, node index fastest
 *   U     [B][nc][M]   control trajectories
 *   RES   [B][ns+np][M]  rows 0..ns-1  : defect  (D.X)_i,k - h f_i(x_k,u_k,t_k)
 *                        rows ns..     : path constraint values c_j(x_k,t_k)
 *   VALS  [B][nvals][M], nvals = ns*(ns+nc) + 2*np + (ns+nc):
 *           entry i*(ns+nc)+v        : d defect_(i,k) / d z_(v,k)
 *                                      = -h df_i/dz_v + (v==i ? D_kk : 0)
 *           entry ns*(ns+nc)+2j+{0,1}: d c_j / d (px, py) at node k   (rows of the record table, emi_set_path)
 *           then PW entries per row traced from callbacks (emi_set_model_source)
 *           last ns+nc entries        : d cost / d z_(v,k) = sgn h w_k dL/dz_v
 *   COST  [B]          sgn h sum_k w_k L(x_k,u_k)   (sgn=-1 when maximising,
 *                      ePSOPT.cpp:212-213)
 *   h = (tf - t0)/2,  z_(v,k): v<ns -> state v, else control v-ns.
 */
#ifndef EMI355X_H_
#define EMI355X_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMI_ABI_VERSION 2

typedef struct emi_ctx_s* emi_ctx_t;

/* status codes */
enum {
    EMI_OK = 0,
    EMI_ERR_ARG = 1,          /* bad argument (null pointer, size mismatch) */
    EMI_ERR_STATE = 2,        /* call made before its prerequisites */
    EMI_ERR_HIP = 3,          /* a HIP runtime call failed */
    EMI_ERR_NO_DEVICE = 4,    /* no usable gfx950 device */
    EMI_ERR_UNSUPPORTED = 5,  /* valid request this build has no kernel for */
    EMI_ERR_COMM = 6          /* RCCL failure */
};

/* built-in node models (emi_set_model) */
enum {
    EMI_MODEL_POINTMASS2D = 0,  /* ns=2 nc=2: etol_psopt_example1.cpp:101-138 */
    EMI_MODEL_QUADROTOR2D = 1,  /* ns=6 nc=2: planar quadrotor (build-defined) */
    EMI_MODEL_FIXEDWING12 = 2,  /* ns=12 nc=4: rigid-body fixed wing (build-defined) */
    EMI_MODEL_SOURCE = 100      /* installed by emi_set_model_source (emi_layout_t.model)  */
};

/* path-row kinds; one record = EMI_PATH_REC reals: {kind, c0..c6} */
enum {
    EMI_PATH_ELLIPSE = 0, /* c = {xc, yc, cos tt, sin tt, a^2, b^2, -}      */
    EMI_PATH_DISC = 1,    /* c = {xc, yc, r^2, -, -, -, -}                  */
    EMI_PATH_TRACK = 2    /* c = {track index, r^2, -, ...}: centre per node */
};
#define EMI_PATH_REC 8

/* emi_eval flags */
enum {
    EMI_EVAL_NODES = 1u,   /* K1+K2+K3+K5: node functions, Jacobian values   */
    EMI_EVAL_DEFECT = 2u,  /* K4: accumulate D.X into the defect rows        */
    EMI_EVAL_ALL = 3u,
    EMI_EVAL_NOJAC = 4u    /* values only (line-search evaluations)          */
};

typedef struct {
    int model, ns, nc, np, M, B;
    int nres;    /* ns + np                        */
    int nvals;   /* ns*(ns+nc) + 2*np_table + PW*np_traced + (ns+nc) */
    int nhess;   /* (ns+nc)*(ns+nc+1)/2            */
    int real_bytes; /* 8 (f64) or 4 (f32)          */
    int px, py;  /* state indices the keep-outs act on */
    double t0, tf;
} emi_layout_t;

/* ---- library ---------------------------------------------------------- */
int emi_abi_version(void);
const char* emi_status_string(int status);
int emi_device_count(int* count);

/* ---- host-side mesh construction (no device needed) -------------------- */
/* LGL nodes tau[M] (ascending, -1..1), weights w[M], D[M*M] row-major.     */
int emi_lgl(int M, double* tau, double* w, double* D);
int emi_model_dims(int model, int* ns, int* nc, int* nparams);
/* ellipse record from one polygon edge (a->b); follows the arithmetic of
 * etol_psopt_example1.cpp:163-182 term by term.                            */
int emi_edge_ellipse(double xa, double ya, double xb, double yb, double* rec8);
/* linear interpolation of a waypoint table at the node times; follows
 * TrajectoryOptimizer.hpp:239-258 (bracket search) term by term.           */
int emi_track_centres(int nway, const double* t, const double* x, const double* y,
                      int M, const double* node_t, double* xc, double* yc);

/* ---- context ------------------------------------------------------------ */
int emi_create(int device_id, emi_ctx_t* out);       /* f64 arithmetic */
int emi_create_f32(int device_id, emi_ctx_t* out);   /* f32 arithmetic */
int emi_destroy(emi_ctx_t ctx);
const char* emi_last_error(emi_ctx_t ctx);
int emi_set_stream(emi_ctx_t ctx, void* hip_stream); /* NULL = own stream */
int emi_get_stream(emi_ctx_t ctx, void** hip_stream);
int emi_synchronize(emi_ctx_t ctx);

/* ---- problem definition ------------------------------------------------- */
/* D may be NULL: a points-only mesh (abscissae tau and weights w without a
 * differentiation matrix), for evaluating the node functions BETWEEN the collocation
 * nodes (the ODE-error estimate of the mesh refinement); such a context accepts
 * EMI_EVAL_NODES only.                                                            */
int emi_set_mesh(emi_ctx_t ctx, int M, const double* tau, const double* w,
                 const double* D, double t0, double tf);
/* Selects a built-in model.  emi_set_model and emi_set_model_source DROP the path-row
 * table of the context (np = 0: its px / py name states of the previous model, which
 * the new one need not have): emi_set_path, and emi_set_tracks where rows of kind
 * EMI_PATH_TRACK are used, must follow every model change.                        */
int emi_set_model(emi_ctx_t ctx, int model, const double* params, int nparams,
                  int maximize);
/* Model given as C++ text: the definition of
 *     template <typename T> struct <struct_name> { NS, NC, NV, f, jac, cost, grad, hess };
 * with the interface of the built-in models (etol_amd/csrc/emi_models.hpp; the
 * text may use EMI_DEV, ModelParams<T> and the emi_sin/cos/tan/exp/log/sqrt/pow
 * helpers).  It is compiled for gfx950 here (hiprtc) together with the
 * library's kernel templates; on a compile error the status is EMI_ERR_ARG and
 * emi_last_error() holds the compiler log.  params (<= 16) reach the struct as
 * ModelParams<T>.  npath = number of path rows the struct computes itself
 * (NPATH, PW, pvar(), path() / path_hess(): constraint callbacks traced by the
 * host); they follow the rows of emi_set_path in RES.  A traced row may depend on
 * any states and controls of its node: path_vars[n_path_vars] (ascending variable
 * indices, states first) is the union over the rows, and VALS holds n_path_vars
 * partials per traced row (entry ns*(ns+nc) + 2*np_table + j*n_path_vars + q =
 * d c_j / d z_path_vars[q]) between the table rows' pairs and the cost gradient.
 * Replaces a previous emi_set_model / emi_set_model_source (and, like them, drops
 * the table of emi_set_path).                                                  */
int emi_set_model_source(emi_ctx_t ctx, const char* struct_name, const char* source,
                         int ns, int nc, int npath, const int* path_vars, int n_path_vars,
                         const double* params, int nparams, int maximize);
/* Compile-only check of such a text (no device needed): EMI_OK or EMI_ERR_ARG
 * with up to log_len-1 characters of the compiler log in log (may be NULL).   */
int emi_check_model_source(const char* struct_name, const char* source, int ns,
                           int nc, int npath, int n_path_vars, int f32, char* log, size_t log_len);
int emi_set_batch(emi_ctx_t ctx, int B);
/* Delayed states and controls -- what ePSOPT::dae appends to the callbacks' x and u through PSOPT's get_delayed_state /
 * get_delayed_control (reference src/ePSOPT/ePSOPT.cpp:231-248): x(t - i dt) of every state for i = 1 .. x_horizon - 1 and
 * u(t - i dt) of every control for i = 1 .. u_horizon.  On the device they are extra INPUTS of the node functions: the model
 * (emi_set_model_source) is written with nc_model = nc + (x_horizon - 1) ns + u_horizon nc controls, ordered
 *     [ u (nc) | x(t - dt) (ns) | .. | x(t - (x_horizon-1) dt) | u(t - dt) (nc) | .. | u(t - u_horizon dt) ],
 * the caller keeps passing U[B][nc][M], and every evaluation first forms the delayed rows as W(i dt) . (node values) on the
 * MFMA defect kernel: W(delay)[k][j] = Lagrange basis polynomial j of the LGL nodes at t_k - delay (the value of the
 * collocation polynomial -- PSOPT's "Legendre" interpolation), with t_k - delay CLAMPED to t0: PSOPT 5.0.0 is not part of the
 * reference tree, and this build takes the history of a delayed variable before t0 to be its value at t0.
 * VALS / H then hold partials with respect to the delayed inputs as they hold those of the controls (entries of the extended
 * node-variable vector); the total derivative with respect to the trajectory is (d . / d delayed) . W -- W stays an operator,
 * like the off-diagonal part of I (x) D.  emi_get_layout reports nc = nc_model; emi_get_delays the split.  Must follow
 * emi_set_model_source (a model change drops the delays).  Horizons 0 / 1 and 0: no delayed values, as the reference.    */
int emi_set_delays(emi_ctx_t ctx, int x_horizon, int u_horizon, double dt);
int emi_get_delays(emi_ctx_t ctx, int* x_horizon, int* u_horizon, int* n_delayed);
/* W(delay)[M][M] as above, on the host (no device needed)                                                                   */
int emi_delay_matrix(int M, const double* tau, const double* w, double t0, double tf, double delay, double* W);
/* recs: [nsets][np][EMI_PATH_REC]; nsets is 1 (shared) or B (per instance) */
int emi_set_path(emi_ctx_t ctx, int np, int nsets, const double* recs,
                 int px_state, int py_state);
/* xc, yc: [nsets][ntracks][M] disc centres at the node times               */
int emi_set_tracks(emi_ctx_t ctx, int ntracks, int nsets, const double* xc,
                   const double* yc);
int emi_get_layout(emi_ctx_t ctx, emi_layout_t* out);

/* ---- Newton step of the NLP iteration (batch of one, f64 contexts) --------
 * KKT matrix of one instance, N = (ns+nc+ns)*M rows:
 *     [ Q    J^T ]   Q: node-block-diagonal, Qblk [nhess][M] packed lower
 *     [ J  -dc I ]      triangles (layout of emi_hess_*: Hessian blocks plus
 *                       whatever diagonal / path-row terms the caller adds)
 *                    J: D (x) [I 0] off the node diagonal, Jblk [ns*(ns+nc)][M]
 *                       on it (the defect entries of VALS, which hold D_kk)
 * Unknown order: variables v*M+k, then defect multipliers i*M+k.  fixed
 * [(ns+nc)*M]: 1 = the variable does not move (identity row/column, rhs 0).
 * Host pointers.  *info = 0 factorised, > 0 singular.
 * Option "kkt_method" (emi_set_option): 1 (default) factorises the Schur
 * complement J Q^-1 J^T + dc I with a Cholesky -- valid when every Q block is
 * positive definite, which also fixes the inertia of K by construction; a
 * matrix that is not takes method 0 automatically: K assembled in HBM and
 * LU-factorised.  Neither reports an inertia; see emi_kkt_lowrank.           */
int emi_kkt_factor(emi_ctx_t ctx, const double* Qblk, const double* Jblk,
                   const unsigned char* fixed, double dc, int* info);
/* Low-rank correction of the factorised matrix: K = K~ - sum_c delta_c u_c u_c^T,
 * u_c = vec[c][0..nv) placed on the variables of node[c] (what the caller added
 * to make the Q blocks positive definite).  K~^-1 U and the Cholesky factor of
 * C = Delta^-1 - U^T K~^-1 U stay on the device.  *exact = 1 iff C is positive
 * definite, i.e. iff K has the inertia of K~; emi_kkt_solve then returns
 * solutions of K (Woodbury), else of K~.  r = 0 clears the correction.       */
int emi_kkt_lowrank(emi_ctx_t ctx, int r, const int* node, const double* vec,
                    const double* delta, int* exact);
/* The Newton steps of n scenarios at once: ctxs[b] are n DIFFERENT contexts on one
 * device with the same mesh size and model dimensions (one scenario of a
 * Monte-Carlo batch each, BASELINE configs[3]); arguments per scenario as the
 * single entry points take them (host pointers).  Every launch of the
 * factorisation then carries the whole batch -- the 96-step dependency chain of a
 * 1024-node Cholesky is paid once per batch instead of once per scenario -- on
 * ctxs[0]'s stream; each context keeps its own factors, so emi_kkt_lowrank /
 * emi_kkt_solve / emi_kkt_solve_batch may follow in any grouping.  A scenario
 * the batch cannot take (a node block that is not positive definite, or one
 * whose regularisation ladder is exhausted) is factorised through
 * emi_kkt_factor inside the call.  info[b] as emi_kkt_factor's.  What IPOPT does
 * once per scenario behind ePSOPT (reference src/ePSOPT/ePSOPT.cpp:62-66, 84).     */
int emi_kkt_factor_batch(int n, const emi_ctx_t* ctxs, const double* const* Qblk,
                         const double* const* Jblk, const unsigned char* const* fixed,
                         const double* dc, int* info);
/* one right-hand side [N] per scenario, in place (host pointers); low-rank
 * corrections of the scenarios that hold one are applied                          */
int emi_kkt_solve_batch(int n, const emi_ctx_t* ctxs, double* const* rhs);
/* 1 if the context holds a factorisation of the Schur path (what the batched and refined solves take), else 0 */
int emi_kkt_is_schur(emi_ctx_t ctx);
/* The Newton step WITH its iterative refinement on the device: x = K~^-1 b, then
 * up to max_steps rounds of  r = b - K x,  x += K~^-1 r  against the NOMINAL
 * matrix K -- the node blocks and dc_nominal the caller handed emi_kkt_factor
 * (the factorisation may hold a regularised matrix: emi_kkt_last_regularisation),
 * minus the low-rank term while emi_kkt_lowrank reported "exact".  A round stops
 * when the residual is at round-off or no longer halves; a correction that made
 * the residual worse is taken back.  Only residual norms cross to the host.  Out:
 * rel = final max|r| / max(1, max|b|), nsolve = solves with the factors used,
 * reverted = 1 if the last correction was taken back, status = 0 ok / 2 the first
 * solution is not finite.  Schur-path factorisations only (EMI_ERR_UNSUPPORTED
 * otherwise: refine around emi_kkt_solve).  What IPOPT's own iterative refinement
 * does behind ePSOPT (reference src/ePSOPT/ePSOPT.cpp:62-66).                     */
int emi_kkt_solve_refined(emi_ctx_t ctx, double* rhs, double dc_nominal, int max_steps,
                          double* rel, int* nsolve, int* reverted, int* status);
int emi_kkt_solve_refined_batch(int n, const emi_ctx_t* ctxs, double* const* rhs,
                                const double* dc_nominal, int max_steps, double* rel,
                                int* nsolve, int* reverted, int* status);
/* What the last emi_kkt_factor really factorised: [[Q + dw I_x, J^T], [J, -dc I]]
 * with dw on the diagonal of the free STATE variables.  dc >= the caller's and
 * dw >= 0; they exceed the nominal (dc, 0) when the Schur path had to climb its
 * regularisation ladder (csrc/emi_kkt.hip).  A caller that refines its step
 * against the nominal matrix reads them to know the step is inexact -- what
 * IPOPT's delta_w / delta_c tell its own iteration
 * (reference src/ePSOPT/ePSOPT.cpp:62-66: nlp_method "IPOPT").                */
int emi_kkt_last_regularisation(emi_ctx_t ctx, double* dc, double* dw);
/* rhs [nrhs][N] (one right-hand side after the other) in, solutions out;
 * may be called repeatedly after one factor.                               */
int emi_kkt_solve(emi_ctx_t ctx, double* rhs, int nrhs);

/* What emi_eval_dev's default dispatch would do with a batch of B instances on this
 * context (mesh, model, options as set): the one definition of the launch policy,
 * for reports, tools and tests (csrc/emi_api.hip: plan_pass, plan_piece).           */
typedef struct emi_pass_plan {
  int one_launch;      /* 1: the pass goes out as ONE launch (emi_pass_f64_kernel: MFMA-role + node-role workgroups) */
  int sw;              /* states per MFMA workgroup */
  int ksplit;          /* K slices per tile (> 1: combined in-kernel by ticket, in slice order) */
  int ring_stages;     /* operand ring stages of the MFMA role */
  int cpart, cx;       /* tile order: 0 plain, > 0 column partitions, < 0 grouped (-G instance groups per super-block); column tiles per block */
  int mfma_workgroups; /* workgroups of the MFMA role (tiles x K slices) */
  int store_mode;      /* node-role result stores: 0 plain, 1 sc1, 2 non-temporal, 3 nt sc1 */
  int block_order;     /* 1 MFMA workgroups first, 0 evenly interleaved, >= 100: MFMA workgroups at that % of the even density */
  int tiles16;         /* 16-instance x 128-node tiles of the (first) launch */
  int piece, tail;     /* > 0: the batch goes out in launches of `piece` instances and a last one of `tail` (0: none) */
  int k_tile;          /* depth of a K tile of the MFMA role: 8 or 16 */
  int column_tiles;    /* 64-column sub-tiles per MFMA workgroup: 1 or 2 */
  int k_halves;        /* 2: the K range of a tile in two halves inside a 512-thread workgroup */
} emi_pass_plan_t;
int emi_plan_pass(emi_ctx_t ctx, int B, emi_pass_plan_t* out);

/* COO pattern of VALS in per-instance NLP numbering:
 * rows/cols have nvals*M entries ordered like VALS; cost-gradient entries
 * carry row = -1.                                                          */
int emi_jac_structure(emi_ctx_t ctx, int* rows, int* cols);

/* ---- device memory helpers (for callers without their own allocator) ---- */
int emi_dev_alloc(emi_ctx_t ctx, size_t bytes, void** dptr);
int emi_dev_free(emi_ctx_t ctx, void* dptr);
int emi_h2d(emi_ctx_t ctx, void* dst, const void* src, size_t bytes);
int emi_d2h(emi_ctx_t ctx, void* dst, const void* src, size_t bytes);

/* ---- the hot path -------------------------------------------------------- */
/* Device-pointer form: every pointer is device memory of the layout above,
 * in the context's real type.  Asynchronous on the context's stream.        */
int emi_eval_dev(emi_ctx_t ctx, const void* dX, const void* dU, void* dRES,
                 void* dVALS, void* dCOST, unsigned flags);
/* Host-buffer form (double in/out whatever the arithmetic type): copies in,
 * evaluates, copies out, synchronises.  Output pointers may be NULL.        */
int emi_eval_host(emi_ctx_t ctx, const double* X, const double* U, double* RES,
                  double* VALS, double* COST, unsigned flags);

/* Lagrangian Hessian node blocks: H[B][nhess][M], lower triangle row-major
 * of  sigma*sgn*h*w_k*L_zz - h*sum_i lamF_i,k f_i,zz + sum_j lamC_j,k c_j,zz */
int emi_hess_dev(emi_ctx_t ctx, const void* dX, const void* dU,
                 const void* dLamF, const void* dLamC, double sigma, void* dH);
int emi_hess_host(emi_ctx_t ctx, const double* X, const double* U,
                  const double* LamF, const double* LamC, double sigma,
                  double* H);

/* ---- measurement --------------------------------------------------------- */
/* HIP-event timers on the context's stream.                                 */
int emi_timer_start(emi_ctx_t ctx);
int emi_timer_stop(emi_ctx_t ctx, float* elapsed_ms); /* synchronises */
/* Per-kernel event brackets inside emi_eval_dev, recorded on the stream the
 * kernel is launched on.  level 1: every bracket (both kernels and the whole
 * overlapped pass: eight events per pass, ~25 us of dependency latency on a
 * 0.25 ms pass -- for diagnosis); 2: the defect (MFMA) kernel only; 3: the
 * node kernel only (two events per pass); 0: off.                            */
int emi_profile_enable(emi_ctx_t ctx, int level);
int emi_profile_read(emi_ctx_t ctx, float* node_ms, int* node_launches,
                     float* defect_ms, int* defect_launches,
                     float* fused_ms, int* fused_launches); /* syncs+resets */

/* ---- kernel selection ------------------------------------------------------ */
/* "overlap" (default 1): emi_eval with EMI_EVAL_ALL runs the even/odd MFMA defect
 * kernel (emi_symdefect.hip) and the streaming node kernel CONCURRENTLY on two
 * streams when the mesh allows it (f64, M % 128 == 0, centro-antisymmetric D,
 * 2- or 6-state model); 0 forces the general node-then-defect sequence.
 * "sym_ct": MFMA kernel variant.  0 (default) = chosen from the batch size;
 * 3 = LDS-DMA operand ring, one workgroup per CU; 5 / 6 / 7 / 8 = the ring with
 * SW = NS / 2 / 1 / 3 states per workgroup, 8-deep K tiles, several workgroups per
 * CU; 4 = as 0; 1 / 2 = register-staged, 64 / 128 columns (2 needs M % 256 == 0).
 * "node_store": cache policy of the node kernel's result stores on the overlapped
 * path: -1 (default) non-temporal once a pass writes more than the Infinity Cache
 * holds, 0 plain, 1 write-through, 2 non-temporal.
 * "sym_order" (default 1): workgroup -> tile order within an XCD (sym_ct 1..3).
 * "overlap_mode": how the two kernels share the chip.  0 (default) = by batch size;
 * 2 = two streams (fork / join through events); 3 = ONE launch, MFMA-role and
 * node-role workgroups in one grid with COST finished in-kernel (what 0 chooses at
 * every batch size, one instance included, where the model has a pass instantiation);
 * 1 = one stream, back to back.
 * "sym_ksplit": K slices per tile of the state-split ring: 0 (default) = chosen by the
 * default dispatch for small batches (4 slices while the MFMA role stays within 256
 * workgroups, 2 within 512: up to 16 / 80 instances at 1024 nodes), 1 = never,
 * 2 / 4 / 8 forced; partial sums through a slab, summed in slice order (bitwise
 * reproducible, not bitwise the unsplit sum).
 * "small_rows" (default 24): up to this many rows B*ns the skinny streaming defect
 * kernel takes a pass that cannot go as one launch (0: never).  "sym_combine": 1
 * (default) the workgroup that draws a tile's last ticket adds the slices
 * in-kernel, 0 a second launch does; bitwise the same result.
 * "sym_cpart": tile order of the state-split ring.  0 (default) = by mesh and batch
 * size; -1 = plain (the column tiles of an X tile are neighbours, X tiles dealt over
 * the XCDs); 1 / 2 / 4 / 8 = the column tiles cut into that many partitions, each
 * worked on by 8 / cpart XCDs, so that an XCD's share of De / Do stays in its L2.
 * "sym_gblk" / "sym_cx": the GROUPED tile order (csrc/emi_args.hpp ring_tile_of): an XCD keeps a contiguous range of instance
 * groups -- the range whose node-role workgroups it also runs -- and walks it in super-blocks of sym_gblk 16-instance groups,
 * the column tiles in blocks of sym_cx (0: 2); X and U of a super-block then reach that XCD's L2 once for every consumer.
 * sym_gblk 0: off (unless the policy chooses it: inputs that do not stay in the Infinity Cache between passes).
 * "sym_nst": ring stages of the one-launch pass (3 default, 4).  "pass_order": the MFMA workgroups of an
 * XCD first in its share of the one-launch grid (1), interleaved with the node workgroups (0), interleaved
 * at value / 100 times the even MFMA density with the node workgroups at the tail (>= 100), or -1 (default)
 * by batch size: first for small batches (fewer than 128 tiles).
 * "kkt_*": process-wide switches of emi_kkt_factor ("kkt_block_trsv" 1 (default): single right-hand sides through the
 * library's block-inverse triangular solves instead of rocBLAS trsv; "kkt_primal_levels" 1 (default): primal regularisation levels behind the
 * dual ones before the LU fallback; "kkt_sticky_reg" 1 (default):
 * the Schur path starts at the dual regularisation level that worked last on this
 * mesh; "kkt_cholesky" 2 (default): the library's blocked Cholesky in two-level form from 1024 rows
 * (outer panels of "kkt_chol_outer" columns, default 768), 1: one level, 0: rocsolver_dpotrf;
 * "kkt_chol_diag" 2 (default): the 64 x 64 diagonal block of a Cholesky step by one wave with matrix-pipe block updates, 1: column by column by 256 threads;
 * "kkt_chol_panel" 2 (default): the panel solve of a Cholesky step as 16 x 16 block products on the matrix pipe, 1: one row per thread, 0: rocblas_dtrsm;
 * "kkt_debug", "kkt_batched_max_nodes", "kkt_potrf_lock": diagnostics, see csrc/emi_kkt.hip).
 * "slice": > 0: batches above 2 * slice instances are evaluated in pieces of `slice` instances; 0 (default): a batch above
 * 2048 instances goes as one launch over its multiple of 256 instances plus one for the remainder.
 * "sym_ablate": diagnostics only, results invalid.                             */
int emi_set_option(emi_ctx_t ctx, const char* name, int value);
/* 1 if emi_eval(EMI_EVAL_ALL) currently takes the overlapped path             */
int emi_last_path(emi_ctx_t ctx, int* fused);
/* Diagnostics, no device needed: the tile order the state-split MFMA defect kernel would use for (ns states, B instances,
 * M nodes) with sym_ct (0 / 5..8) and sym_cpart as emi_set_option takes them.  out_tile[t] = column_tile + ncoltiles * group
 * for every tile slot t of the launch (out_cap entries at most); *ntiles_total = number of slots, *cpart / *cx = the plan.
 * Every value 0 .. ntiles_total-1 must occur exactly once (tests/test_abi.py).                                               */
int emi_debug_tile_order(int ns, int B, int M, int sym_ct, int sym_cpart, int* out_tile, int out_cap, int* ntiles_total,
                         int* cpart, int* cx);
/* ... the same with the grouped order's options "sym_gblk" / "sym_cx" (*cpart comes back negative, -gblk, when the plan takes it) */
int emi_debug_tile_order2(int ns, int B, int M, int sym_ct, int sym_cpart, int sym_gblk, int sym_cx, int* out_tile, int out_cap,
                          int* ntiles_total, int* cpart, int* cx);
/* Diagnostics, no device needed: how the one-launch pass deals an XCD's nm MFMA-role and nn node-role blocks ("pass_order":
 * 0 evenly interleaved, 1 MFMA blocks first, >= 100 interleaved at order / 100 times the even MFMA density).  out_role[j] =
 * MFMA block index (>= 0) or -1 - node block index; every index of either role must occur exactly once.                      */
int emi_debug_pass_roles(int nm, int nn, int order, int* out_role, int out_cap);
/* name of the kernel that produced the defect rows in this context's last
 * emi_eval_dev (what a rocprofv3 kernel trace will show); "" before the first  */
const char* emi_last_defect_kernel(emi_ctx_t ctx);

/* ---- the one collective: gather of a batch's results over RCCL ---------------- */
/* SURVEY.md section 8e / north_star: instances shard over the GPUs of a node with no
 * communication while they are evaluated or solved; afterwards every rank hands its
 * block to the root: one group of point-to-point ncclSend/ncclRecv over xGMI.
 * (The reference has no distributed code; this replaces nothing, it is what lets a
 * caller of ETOL::eMI355X run a Monte-Carlo batch on 8 GPUs and end up with every
 * trajectory in one place.)  A communicator is its own handle, one per process/GPU.
 * The 128-byte id is made on one rank (emi_comm_unique_id) and carried to the others
 * by the launcher's means (a file, an environment variable, torch.distributed).     */
#define EMI_COMM_ID_BYTES 128
typedef struct emi_comm_s* emi_comm_t;
int emi_comm_unique_id(void* id /* EMI_COMM_ID_BYTES, out */);
int emi_comm_create(int device_id, int world, int rank, const void* id, emi_comm_t* out); /* collective */
/* every rank sends `bytes` bytes of device memory; on the root, drecv[world][bytes]
 * (device memory) receives them in rank order (the root's own block is copied).
 * hip_stream NULL: the communicator's own stream, synchronised before returning;
 * otherwise asynchronous on that stream.                                            */
int emi_comm_gather(emi_comm_t comm, const void* dsend, void* drecv, size_t bytes, int root, void* hip_stream);
int emi_comm_destroy(emi_comm_t comm);
const char* emi_comm_last_error(emi_comm_t comm); /* NULL: error of the last handle-less call of this thread */

#ifdef __cplusplus
}
#endif
#endif /* EMI355X_H_ */
