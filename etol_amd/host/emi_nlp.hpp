// emi_nlp.hpp -- the NLP iteration behind eMI355X::solve().
//
// ePSOPT hands the transcribed problem to IPOPT with ADOL-C derivatives and an
// exact Hessian (reference src/ePSOPT/ePSOPT.cpp:62-66, 84).  Neither exists
// here, so eMI355X owns the iteration: a primal-dual interior-point method
// (log barrier on variable and path-row bounds, Newton steps on the perturbed
// KKT system with inertia-correcting regularisation, l1-merit backtracking line
// search, Fiacco-McCormick barrier schedule).  Every function/Jacobian/Hessian
// value comes from the device through an NlpEvaluator; the linear algebra of
// the step is either a dense symmetric-indefinite LDL^T on the host (small
// problems such as the shipped resource/configs ones: exact inertia) or a
// KktBackend -- the device assembly + LU of etol_amd/csrc/emi_kkt.hip (SURVEY.md
// section 8f rank 1).  An LU reports no inertia, so that branch factorises a
// quasi-definite matrix (node blocks of Q made positive definite: inertia known
// by construction) and recovers the exact Newton step, together with an exact
// inertia test of the unmodified matrix, from a low-rank Woodbury correction.
#ifndef ETOL_MI355X_EMI_NLP_HPP_
#define ETOL_MI355X_EMI_NLP_HPP_

#include <string>
#include <utility>
#include <vector>

namespace ETOL {
namespace mi355x {

// Source of all problem functions (layouts of include/emi355x.h, batch of one).
class NlpEvaluator {
 public:
    virtual ~NlpEvaluator() {}
    // RES[ns+np][M], VALS[nvals][M] (may be NULL when !jac), COST[1]; returns 0 on success
    virtual int eval(const double* X, const double* U, double* RES, double* VALS, double* COST, bool jac) = 0;
    // H[nhess][M] packed lower triangles
    virtual int hess(const double* X, const double* U, const double* lamF, const double* lamC, double sigma,
                     double* H) = 0;
    virtual std::string last_error() const { return std::string(); }
};

// Linear algebra of the Newton step somewhere else than the host (eMI355X: emi_kkt_factor /
// emi_kkt_solve on the device).  KKT matrix [[Q, J^T], [J, -dc I]] in the layout of
// include/emi355x.h: Qblk [nh][M] packed node blocks, Jblk [ns*nv][M] defect Jacobian node
// entries, unknowns ordered variables v*M+k then defect multipliers i*M+k.
class KktBackend {
 public:
    virtual ~KktBackend() {}
    // 0: factorised; > 0: singular; < 0: failure (see last_error)
    virtual int factor(const double* Qblk, const double* Jblk, const unsigned char* fixed, double dc) = 0;
    // K = K~ - sum_c delta[c] u_c u_c^T with u_c = vec[c*nv .. ) on the variables of node[c] (what the caller
    // added to the Q blocks).  *exact = true iff K has the inertia of K~; solve() then answers for K (Woodbury
    // correction kept by the backend), otherwise for K~.  r = 0 clears.  0 on success.
    virtual int lowrank(int r, const int* node, const double* vec, const double* delta, bool* exact) = 0;
    virtual int solve(double* rhs, int nrhs) = 0;   // rhs [nrhs][nz+md], in place; 0 on success
    // The step with its iterative refinement done by the backend (against the nominal matrix with dual regularisation dc_nominal,
    // exact or convexified as the last lowrank() verdict says): rhs [nz+md] in place.  Returns 0 done (rel = final relative residual,
    // nsolve = solves used, reverted = a correction was taken back), 2 the first solution is not finite, < 0 failure,
    // 1 not offered by this backend (the caller refines around solve()).
    virtual int solve_refined(double* rhs, double dc_nominal, int max_steps, double* rel, int* nsolve, int* reverted) {
        (void)rhs; (void)dc_nominal; (void)max_steps; (void)rel; (void)nsolve; (void)reverted;
        return 1;
    }
    // what the last factor() really factorised: [[Q + dw I_x, J^T], [J, -dc I]] (dw on the free state variables).  A backend
    // that never regularises on its own leaves both untouched (the caller presets them to its nominal dc and 0).
    virtual void applied_regularisation(double* dc, double* dw) { (void)dc; (void)dw; }
    virtual std::string last_error() const { return std::string(); }
};

// A linear equality that couples a node variable to the whole trajectory of another one:
//     z[dst][k] - sum_j W[k][j] z[src][j] = 0     for every node k.
// This is how delayed states / controls (reference src/ePSOPT/ePSOPT.cpp:231-248: get_delayed_state / get_delayed_control
// values appended to the callbacks' x and u) enter the NLP: the delayed value is a variable of its node -- the node functions,
// their Jacobian entries and the packed Hessian blocks stay node-local, exactly what the device kernels produce on the extended
// node variables -- and W (the interpolation operator of the delay on the mesh, emi_delay_matrix) ties it to its source.
struct NlpLink {
    int dst = 0, src = 0;               // variable indices (states 0 .. ns-1, controls ns .. ns+nc-1)
    std::vector<double> W;              // M*M, row-major
};

struct NlpProblem {
    int ns = 0, nc = 0, np = 0, M = 0;
    int px = 0, py = 1;                 // states the path rows depend on when row_vars is empty (two partials per row)
    // per path row: (variable index, VALS entry) of every partial; empty = every row has the pair (px, py) at
    // entries ns*nv + 2j, + 2j + 1
    std::vector<std::vector<std::pair<int, int>>> row_vars;
    std::vector<double> D;              // M*M differentiation matrix (row-major)
    std::vector<double> zl, zu;         // (ns+nc)*M variable bounds, index v*M+k; zl==zu fixes a variable
    std::vector<double> cl, cu;         // np path-row bounds (same at every node)
    std::vector<double> cscale;         // np positive row scalings applied inside the iteration (empty = 1)
    std::vector<double> vscale;         // ns+nc positive variable scales: the iteration runs on z_v / vscale[v], defect rows of state i
                                        // on defect_i / vscale[i] (PSOPT's scaling = "automatic" with state-based defect scaling); empty = none
    bool jacobian_defect_scaling = false;   // PSOPT's defect_scaling = "jacobian-based" (reference src/Examples/PSOPT/etol_psopt_example1.cpp:90-91):
                                        // defect row r is weighted by s_r = 1 / max(1, ||row r of the constraint Jacobian at the first point||_inf)
                                        // wherever the iteration MEASURES the defects against each other or against the objective -- the l1 merit
                                        // function and its penalty weight (which then bounds the scaled multipliers lambda_r / s_r).  A row scaling
                                        // is diag(s) (D (x) I - h f_x): D stays an operator, and the Newton step is invariant under it, so the
                                        // linear algebra (node blocks, Schur complement, backend) is untouched; the convergence test keeps the
                                        // unscaled defects (answers are compared at 1e-6 in the caller's units)
    std::vector<double> lamF0, lamC0;   // optional warm start of the defect / path-row multipliers (ns*M, np*M)
    std::vector<NlpLink> links;         // linear coupling rows (delayed values); the dense host backend only (kkt must be null)
    NlpEvaluator* ev = nullptr;
    KktBackend* kkt = nullptr;          // null: dense LDL^T on the host (with inertia); else e.g. the device LU
};

struct NlpOptions {
    double tol = 1e-8;                  // scaled KKT error at mu = 0
    int max_iter = 200;
    int print_level = 0;
    double mu_init = 0.1;
    double bound_push = 1e-2, bound_frac = 1e-2;
    double max_cpu_time = 1e9;          // seconds
    int max_futile_escalations = 3;     // tenfold raises of the penalty weight beyond 1e5 without halving the largest elastic before giving up
    int max_shift_trials = 6;           // inertia search: trial shifts delta_w per iteration before the reflected step is taken (0: search off)
    double mu_restart = 0.0;            // > 0: the first time the iteration stagnates (below) with a barrier parameter under 1e-4, the parameter is
                                        // multiplied by this (capped at 1e-3) once per solve: a warm start sliding along its active keep-outs
    int stagnation_iters = 12;          // iterations without 10 % progress of the barrier KKT residual before the inertia search starts
    int crawl_limit = 3;                // consecutive short steps (alpha < crawl_frac * alpha_max) before the crawl rule acts
    double crawl_frac = 0.3;
    double rho_init = 10.0;             // exact-penalty weight of the elastic path rows (escalated x10 as needed)
    double acceptable_factor = 100.0;   // "acceptable": KKT error <= acceptable_factor * tol ...
    int acceptable_iter = 10;           // ... over this many consecutive iterations (as IPOPT's acceptable_*)
};

struct NlpResult {
    bool ok = false;
    std::string msg;
    int iterations = 0;
    int evaluations = 0;
    int newton_steps = 0;               // iterations accepted on the KKT residual alone (near the solution)
    int soc_steps = 0;                  // iterations accepted through a second-order correction
    double cost = 0, kkt_error = 0, constr_viol = 0;
    std::vector<double> z;              // (ns+nc)*M solution
    std::vector<double> lamF, lamC;     // multipliers of the defect and path rows
    std::vector<double> lamL;           // multipliers of the coupling rows (NlpProblem::links), links.size() * M
    double t_eval = 0, t_factor = 0, t_solve = 0, t_lowrank = 0, t_total = 0;   // seconds: evaluator, KKT factor, KKT solves, host low-rank algebra
    int n_factor = 0, n_solve = 0;      // factorisations (inertia-search trials included) and solve calls
    double t_jt = 0, t_matvec = 0, t_blocks = 0, t_hess = 0;   // host: J^T lambda, refinement matvecs, node-block assembly + eigen-decompositions; Hessian calls
    double rho = 0;                     // penalty weight the solve ended with (warm start of the next mesh)
    int n_backend_shifted = 0;          // factorisations the backend regularised beyond the nominal matrix (applied_regularisation)
    int n_refine_reverted = 0;          // refinement corrections taken back because they made the residual worse
    double worst_step_residual = 0;     // largest relative residual |b - K x| / max(1, |b|) a Newton step was used with
};

NlpResult solve_nlp(const NlpProblem& prob, const NlpOptions& opt, const std::vector<double>& z0);

// Dense symmetric-indefinite LDL^T (Bunch-Kaufman partial pivoting), lower
// triangle of a row-major n*n array, in place.  Exposed for the unit tests.
struct LdltFactor {
    int n = 0;
    std::vector<double> a;
    std::vector<int> ipiv;
    int npos = 0, nneg = 0, nzero = 0;
};
bool ldlt_factor(LdltFactor& F);                 // false on an exactly singular pivot
void ldlt_solve(const LdltFactor& F, double* b); // b <- A^{-1} b

}  // namespace mi355x
}  // namespace ETOL
#endif
