// eMI355X.hpp -- MI355X-native collocation eSolver: a peer of ePSOPT / eGLPK / eGurobi.
//
// Same role and shape as ETOL::ePSOPT (reference include/ETOL/ePSOPT.hpp:20-163,
// src/ePSOPT/ePSOPT.cpp): derives from TrajectoryOptimizer, overrides
// setup/solve/debug/close, exposes its knobs and results through
// getAlgorithm()/getSolution()/getProblem().  Where ePSOPT delegates to
// PSOPT + ADOL-C + IPOPT on the CPU, eMI355X owns mesh, defect, derivative
// evaluation and the NLP iteration, with every per-node evaluation on the GPU
// behind the C ABI of include/emi355x.h.
#ifndef ETOL_MI355X_EMI355X_HPP_
#define ETOL_MI355X_EMI355X_HPP_

#include <memory>
#include <string>
#include <vector>

#include <ETOL/TrajectoryOptimizer.hpp>
#include <ETOL/eMI355X_Types.hpp>

namespace ETOL {

namespace mi355x {

// Rendezvous of the Newton steps of CONCURRENT solves on one device (a Monte-Carlo batch: one host thread and one eMI355X per
// scenario in flight, BASELINE configs[3]).  Solvers that share a batcher hand their factorisations, low-rank corrections and
// single-right-hand-side solves to it instead of launching them themselves.  The batcher keeps one rendezvous per MESH SIZE (the
// solves in flight sit on different rungs of their mesh ladders, and only requests of one size share launches): the solves currently
// iterating on a mesh are its members, and whichever of them finds every member waiting (or the oldest request older than `flush_us`)
// runs what has gathered as ONE batched call (emi_kkt_factor_batch / emi_kkt_solve_refined_batch: every launch carries all scenarios)
// and hands the answers back.  The members of a mesh size therefore move in step, one Newton iteration per round -- the cheap
// coarse-mesh rounds, which are pure launch latency for a single scenario, at their own pace beside the expensive fine-mesh ones.
// The iteration of each scenario is untouched: same matrices, same steps, its own factors; only the launches are shared.
class KktBatcher {
 public:
    KktBatcher();
    ~KktBatcher();
    struct Member {                                 // (kept for callers of the first form: membership is per mesh now, OnMesh)
        explicit Member(const std::shared_ptr<KktBatcher>& b);
        ~Member();
        std::shared_ptr<KktBatcher> batcher;
    };
    struct OnMesh {                                 // RAII: the calling solve iterates on a mesh of `nodes` nodes (solve() holds one per NLP solve)
        OnMesh(KktBatcher* b, int nodes);
        ~OnMesh();
        KktBatcher* batcher;
        int nodes;
    };
    int flush_us = 20000;                           // a request older than this is run with whatever has gathered (safety: a member that
                                                    // spends unusually long between two Newton steps must not hold the others for good)
    // totals, for reports: batched calls, scenarios they carried, largest batch
    long factor_calls = 0, factor_items = 0, solve_calls = 0, solve_items = 0;
    int largest_batch = 0;
    struct Impl;
    Impl* impl;                                     // (used by the device adapter in eMI355X.cpp)
};

// Algorithm knobs (the fields of PSOPT's Alg that ePSOPT::setup sets,
// reference src/ePSOPT/ePSOPT.cpp:62-72, plus device selection).
struct Alg {
    std::string nlp_method = "interior-point (eMI355X)";
    std::string derivatives = "analytic (device kernels)";
    std::string hessian = "exact";
    std::string collocation_method = "Legendre";   // Legendre-Gauss-Lobatto, as PSOPT's "Legendre"
    std::string mesh_refinement = "automatic";     // "automatic" or "none" (ePSOPT.cpp:69)
    std::string scaling = "none";                  // ePSOPT.cpp:63 sets PSOPT's "automatic".  Here "automatic" is built (the NLP iteration runs on variables
                                                   // scaled by their bounds, z_v / max(|lower_v|, |upper_v|), and on defect rows scaled like their state:
                                                   // PSOPT's state-based defect scaling) and is an OPTION, not the default: measured on one box
                                                   // (profiles/r04_regress_probe.json) it takes the 1024-node / 20 keep-out solve from 12 to 58 last-mesh
                                                   // iterations (0.92 -> 3.05 s, and to a worse local optimum, 416.19 against 400.47) and the 129-node
                                                   // fixed wing from 11 to 79 (1.05 -> 2.32 s); on Monte-Carlo sets it saves ~10 % of the iterations.
                                                   // Results are always in the caller's units.
    std::string defect_scaling = "state-based";    // PSOPT's Alg::defect_scaling.  "state-based": defect rows follow their state's scale (what `scaling`
                                                   // does); "jacobian-based" (the shipped example sets it, etol_psopt_example1.cpp:90-91): every defect row is
                                                   // weighted by the reciprocal of its Jacobian row norm in the merit function of the NLP iteration
                                                   // (mi355x::NlpProblem::jacobian_defect_scaling: the Newton step is invariant under row scalings)
    int mr_max_iterations = 10;                    // ePSOPT.cpp:70
    double ode_tolerance = 1.e-4;                  // ePSOPT.cpp:71
    int mr_max_nodes = 513;                        // refinement stops adding nodes here
    bool inflate_keepouts = false;                 // on the sequencing ladder, grow the keep-outs by half the node spacing
                                                   // (against stepping over thin obstacles; measured: no gain on the
                                                   // Monte-Carlo sets, off by default)
    bool mesh_sequencing = true;                   // meshes above 80 nodes are reached through 33, 65, 129, ... nodes
    int ladder_ratio = 2;                          // ... each rung (ratio x nodes - (ratio - 1)): 2 gives 33, 65, 129, 257, 513; 4 gives 33, 129, 513
    int guess_retries = 4;                         // a locally infeasible cold start is repeated from up to this many bent lines
    double rung_tolerance = 1e-4;                  // NLP tolerance of the intermediate rungs of the mesh ladder: they only feed the next guess (the
                                                   // requested mesh is solved to nlp_tolerance).  1e-6 until round 4; 64 x 1024-node Monte-Carlo set, 8
                                                   // threads, rung patience 100: 2.59 solves/s and 151 mean iterations at 1e-6, 3.06 / 134 at 1e-4, 3.27 /
                                                   // 126 at 1e-3 (profiles/r04_montecarlo_ladder_knobs.jsonl)
    int rung_patience = 100;                       // a warm-started rung of the mesh ladder (65, 129, ... nodes) that is still iterating after this many
                                                   // iterations is given up like a failed one -- the ladder starts again from the next bent line (0: only
                                                   // nlp_iter_max applies).  Rungs that converge take 10 - 70 iterations; the ones that end "locally
                                                   // infeasible" (the interpolant sits on the wrong side of a keep-out) took 280 - 360 to say so, twice in
                                                   // a row in the two slowest scenarios of the 64 x 1024-node set (943 and 920 iterations): 1.93 solves/s
                                                   // without the rule, 2.50 - 2.60 with 60 - 100, 2.35 - 2.41 with 80 / 150 (paths differ) (r04_notes.md)
    double plan_clearance = 0.01;                  // clearance (x the span) at which a step of the planned route costs twice its length.  256-scenario set, planned
                                                   // route first: 0.1: mean cost 408.5, 105.6 iterations; 0.05: 402.3, 98.2; 0.02: 400.6, 97.3; 0.01: 400.3, 99.6
                                                   // (the straight line first: 400.5, 125.5) -- profiles/r04_notes.md section 27
    bool plan_first_start = true;                  // the planned route is the FIRST start of a climb of the mesh ladder, the straight line the second (needs
                                                   // plan_second_start; false: the straight line first).  Config 4 in full: 206 s against 279 s, mean 98 against
                                                   // 128.5 iterations, 90th percentile 126 against 270, mean cost of the solved 400.38 against 401.02; 1021 against
                                                   // 1023 of 1024 solved within the example's budget
    int target_patience = 200;                     // the warm start on the REQUESTED mesh gets this many iterations while the ladder still has an unused start
                                                   // (0: nlp_iter_max); see solve()
    double mu_restart = 10.0;                      // NlpOptions::mu_restart of the warm starts: barrier parameter x this (at most 1e-3), once, when a warm
                                                   // start stagnates at a small parameter (0: off; 10 and 100 measured: profiles/r02_notes.md section 12)
    int warm_patience = 0;                         // > 0: a warm start (interpolated guess) still running after this many iterations is
                                                   // started again from the same guess with a 10 x larger barrier parameter (0: off)
    bool warm_multipliers = false;                 // carry costate-mapped multipliers to the next mesh (measured: no gain, profiles/r01_notes.md)
    double warm_mu_init = 1e-5;                    // barrier parameter of a solve started from an interpolated solution
    double warm_bound_push = 1e-4;                 // ... and its bound push / fraction
    int max_shift_trials = 6, stagnation_iters = 12, crawl_limit = 3;   // inertia search / crawl rule of the NLP iteration (emi_nlp.hpp)
    double crawl_frac = 0.3;
    std::string linear_solver = "auto";            // Newton step: "host" (dense LDL^T), "device" (structured
                                                   // factorisation in HBM, emi_kkt_*), "auto" = device above 400 KKT rows
    int nlp_iter_max = 200;                        // per NLP solve (one mesh, one start), as ePSOPT.cpp:66
    bool plan_second_start = true;                 // the route planned through the free space of the static keep-outs (clearance-weighted: planned_path_guess) is the
                                                   // second start of a mesh-ladder climb, before the bent lines (false: the last cold-start attempt only).  256-scenario
                                                   // Monte-Carlo set: 125.5 against 135.2 iterations per scenario, 3.91 against 3.65 solves/s (profiles/r04_notes.md section 24)
    int nlp_iter_budget = 0;                       // > 0: iterations ONE solve() may spend over all its meshes, rungs and restarts; when they are used
                                                   // up the problem is reported unsolved ("iteration budget exhausted") instead of being carried on
                                                   // -- a Monte-Carlo batch waits for its slowest scenario, and the slowest are the ones that wander
                                                   // (profiles/r04_notes.md: median 105 iterations, 90th percentile 411, maximum 1028 at 1024 nodes)
    double nlp_tolerance = 1.e-6;
    double max_cpu_time = 1.e9;
    int print_level = 0;
    int device = 0;                                 // HIP device ordinal
    std::shared_ptr<KktBatcher> kkt_batcher;        // set: the device Newton steps of this solver go through the shared rendezvous (above)
};

struct Sol {
    int error_flag = 0;
    std::string error_msg;
    double cost = 0;
    int nlp_iterations = 0;         // of the last NLP solve
    int nlp_iterations_total = 0;   // over all meshes (sequencing + refinement)
    int evaluations = 0;
    double kkt_error = 0, constraint_violation = 0;
    std::string linear_solver;      // what the last solve used for the Newton step
    int mesh_iterations = 0;        // NLP solves performed (1 = no refinement happened)
    struct NlpRun {
        size_t nodes; int iterations; bool converged; double seconds;
        // of those seconds: evaluator calls, Hessian calls, Newton-step factorisations, solves, low-rank corrections, host node-block
        // assembly + eigen-decompositions, host J^T lambda, host refinement matvecs (the remainder is the iteration's own host arithmetic)
        double t_eval, t_hess, t_factor, t_solve, t_lowrank, t_blocks, t_jt, t_matvec;
        int factorisations, solves;
    };
    std::vector<NlpRun> nlp_runs;   // every NLP solve of this solve(), in order (ladder rungs, restarts, refinement): where the iterations went
    double ode_error = 0;           // relative local ODE error of the last mesh (integral of the ODE residual between nodes)
    size_t nstates = 0, ncontrols = 0, nodes = 0;
    std::vector<double> states;     // [nstates][nodes]
    std::vector<double> controls;   // [ncontrols][nodes]
    std::vector<double> time;       // [nodes]  LGL times h (tau_k + 1)
};

// The transcribed problem (what ePSOPT keeps in PSOPT's Prob).
struct Prob {
    std::string name = "ETOL Problem";
    size_t nstates = 0, ncontrols = 0, nodes = 0, npath = 0;
    // delayed values (ePSOPT::dae, ePSOPT.cpp:231-248): xhorizon - 1 delayed copies of every state, uhorizon of every control,
    // delay i * delay_dt; the device model sees them as ndelayed extra inputs behind the ncontrols controls
    size_t xhorizon = 0, uhorizon = 0, ndelayed = 0;
    double delay_dt = 0;
    bool lifted = false;                           // while solve() runs on a delayed problem: the ndelayed values are NLP variables of their
                                                   // node (controls ncontrols .. ncontrols + ndelayed - 1, unbounded), tied to their sources by
                                                   // linear coupling rows with the interpolation operators (mi355x::NlpLink); the device then
                                                   // evaluates the node functions on given delayed values instead of forming them itself
    std::vector<size_t> row_order;                 // evaluation-order row q is row row_order[q] of the callbacks
    std::vector<double> traced_scale;              // normalisation of the traced rows inside the NLP iteration
    size_t npath_traced = 0;                       // of npath: rows traced from constraint callbacks (they follow the table rows)
    std::vector<int> path_vars;                    // node variables the traced rows depend on (ascending; PW partials per traced row)
    int model = -1;
    std::vector<double> model_params;
    std::string model_source;                      // model == EMI_MODEL_SOURCE: struct generated from the traced callbacks
    double t0 = 0, tf = 0;
    std::vector<double> tau, w, D;                 // LGL mesh
    std::vector<double> path_records;              // [npath][EMI_PATH_REC]
    std::vector<double> track_x, track_y;          // [ntracks][nodes]
    std::vector<TrackTable> tracks;                // waypoint tables (re-tabulated when the mesh changes)
    size_t ntracks = 0, px = 0, py = 1;
    std::vector<double> state_lower, state_upper, control_lower, control_upper;
    std::vector<double> path_lower, path_upper;
    std::vector<double> event_lower, event_upper;  // [2*nstates]: x(t0) then x(tf)
    std::vector<double> guess_states, guess_controls;   // optional warm start, [n][nodes]
    std::vector<double> guess_lamF, guess_lamC;         // multipliers to go with it, [nstates][nodes] / [npath][nodes]
    double guess_bend = 0;                              // default guess: sideways offset of the straight line at mid-horizon
    bool guess_planned = false;                         // default guess: positions along a route through the free space of the static keep-outs
    double guess_clearance = 0.01;                      // ... whose steps cost twice their length at a clearance of this x the span (0: the shortest route)
                                                        // (set by solve() when it retries a locally infeasible start)
};

}  // namespace mi355x

class eMI355X : public TrajectoryOptimizer {
 public:
    eMI355X();
    virtual ~eMI355X();

    void setup();    // ETOL configuration -> transcribed NLP on the device
    void solve();    // NLP iteration; fills score and trajectories on success
    void debug();    // per-iteration log (print_level 5), call after setup()
    void close();    // hands the device context to this thread's pool of idle contexts (the next setup() reuses it)
    static void releaseDevices();   // destroys the calling thread's idle contexts (call before a worker thread ends)

    mi355x::Alg* getAlgorithm();
    mi355x::Sol* getSolution();
    mi355x::Prob* getProblem();
    // PSOPT-style relative local ODE error of a trajectory z = [X (nstates x nodes), U (ncontrols x nodes)] on the
    // current mesh (what the automatic mesh refinement compares with ode_tolerance); needs setup()
    double odeError(const std::vector<double>& z, std::vector<double>* z_fine = nullptr, size_t* nodes_fine = nullptr);
    // One evaluation pass of the transcribed problem at z = [X (nstates x nodes), U (ncontrols x nodes)] on the device -- what
    // the NLP solver asks ePSOPT for through PSOPT (dae + integrand_cost at every node, defects, quadrature, Jacobian
    // values): res [nstates + npath][nodes] (defect rows, then path rows), vals [nvals][nodes] (may be NULL), cost.  Layouts:
    // include/emi355x.h.  Needs setup().  With delayed states / controls the node variables of vals are
    // [x | u | delayed values] (emi_set_delays).
    void evaluate(const std::vector<double>& z, std::vector<double>* res, std::vector<double>* vals, double* cost);

 protected:
    mi355x::Alg _algorithm;
    mi355x::Sol _solution;
    mi355x::Prob _problem;

 private:
    struct Device;                       // emi_ctx_t + NlpEvaluator / KktBackend adapter
    std::unique_ptr<Device> _dev;
    static std::vector<Device*>& device_pool();
    void traceCallbacks();               // calls every f_t once (see eMI355X_Types.hpp)
    void addBounds();
    void getTraj();
    void setMesh(size_t nodes);          // LGL mesh + track tables for this node count
    void configureDevice(Device* dev);   // mesh, model, batch of one, path table -> device context
};

}  // namespace ETOL
#endif
