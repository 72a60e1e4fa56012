// eMI355X.cpp -- the MI355X eSolver: ETOL configuration -> LGL transcription on the GPU.
//
// Mirrors, member for member, what ePSOPT does on the CPU (reference
// src/ePSOPT/ePSOPT.cpp): setup() :40-81 (problem mapping, algorithm defaults),
// addBounds() :125-155, solve() :83-94, getTraj() :157-182, debug() :100-102,
// close() :107.  The per-node callbacks dae/integrand_cost/events (:186-291) have
// no host counterpart here: they are the kernels behind include/emi355x.h.
#include <ETOL/eMI355X.hpp>

#include <algorithm>
#include <chrono>
#include <queue>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <map>
#include <mutex>

#include "emi_nlp.hpp"
#include "emi_transcribe.hpp"
#include "emi_trace.hpp"

namespace ETOL {

// ---------------------------------------------------------------------------------------------
// keep-out row builders (eMI355X_Types.hpp)
// ---------------------------------------------------------------------------------------------
namespace mi355x {

namespace {
void check_xy(const Symbol& sx, const Symbol& sy, const char* who) {
    if (sx.kind != Symbol::STATE || sy.kind != Symbol::STATE || sx.index == sy.index) {
        fprintf(stderr, "%s: keep-out rows must act on two different states\n", who);
        exit(EXIT_FAILURE);
    }
}
}  // namespace

PathBlock ellipse_rows(const std::vector<border_t>& zones, const Symbol& sx, const Symbol& sy) {
    check_xy(sx, sy, "ellipse_rows");
    PathBlock out;
    out.px = sx.index;
    out.py = sy.index;
    for (const border_t& poly : zones) {
        for (auto a = poly.begin(); a != poly.end(); ++a) {
            auto b = std::next(a);
            if (b == poly.end()) b = poly.begin();
            std::array<double, EMI_PATH_REC> rec{};
            emi_edge_ellipse((*a)[0], (*a)[1], (*b)[0], (*b)[1], rec.data());
            out.rows.push_back(rec);
        }
    }
    return out;
}

PathBlock disc_rows(const std::vector<std::array<double, 3>>& discs, const Symbol& sx, const Symbol& sy) {
    check_xy(sx, sy, "disc_rows");
    PathBlock out;
    out.px = sx.index;
    out.py = sy.index;
    for (const auto& d : discs) {
        std::array<double, EMI_PATH_REC> rec{};
        rec[0] = (double)EMI_PATH_DISC;
        rec[1] = d[0];
        rec[2] = d[1];
        rec[3] = d[2] * d[2];
        out.rows.push_back(rec);
    }
    return out;
}

PathBlock track_rows(const std::list<track_t>& tracks, const Symbol& sx, const Symbol& sy) {
    check_xy(sx, sy, "track_rows");
    PathBlock out;
    out.px = sx.index;
    out.py = sy.index;
    for (const track_t& trk : tracks) {
        TrackTable tb;
        tb.radius = trk.radius;
        for (const traj_elem_t& wp : trk.trajectory) {
            tb.t.push_back(wp.first);
            tb.x.push_back(wp.second.at(0));
            tb.y.push_back(wp.second.at(1));
        }
        out.tracks.push_back(tb);
    }
    return out;
}

}  // namespace mi355x

// ---------------------------------------------------------------------------------------------
// device adapter: the only place that touches the C ABI
// ---------------------------------------------------------------------------------------------
struct eMI355X::Device : public mi355x::NlpEvaluator, public mi355x::KktBackend {
    emi_ctx_t ctx = nullptr;
    int device_id = -1;
    int nodes = 0;                       // mesh size the context holds (configureDevice)
    std::string installed_source;        // text of the traced model whose code object is loaded in ctx ("" = none)
    bool installed_maximize = false;     // ... and the objective sign it was installed with
    ~Device() override {
        if (ctx) emi_destroy(ctx);
    }
    int eval(const double* X, const double* U, double* RES, double* VALS, double* COST, bool jac) override {
        return emi_eval_host(ctx, X, U, RES, VALS, COST, EMI_EVAL_ALL | (jac ? 0u : (unsigned)EMI_EVAL_NOJAC));
    }
    int hess(const double* X, const double* U, const double* lamF, const double* lamC, double sigma,
             double* H) override {
        return emi_hess_host(ctx, X, U, lamF, lamC, sigma, H);
    }
    // Newton step on the device (emi_kkt.hip)
    int factor(const double* Qblk, const double* Jblk, const unsigned char* fixed, double dc) override {
        int info = -1;
        const int st = emi_kkt_factor(ctx, Qblk, Jblk, fixed, dc, &info);
        return st == EMI_OK ? info : -1;
    }
    int lowrank(int r, const int* node, const double* vec, const double* delta, bool* exact) override {
        int ex = 0;
        const int st = emi_kkt_lowrank(ctx, r, node, vec, delta, &ex);
        *exact = ex != 0;
        return st == EMI_OK ? 0 : -1;
    }
    int solve(double* rhs, int nrhs) override { return emi_kkt_solve(ctx, rhs, nrhs) == EMI_OK ? 0 : -1; }
    void applied_regularisation(double* dc, double* dw) override { (void)emi_kkt_last_regularisation(ctx, dc, dw); }
    int solve_refined(double* rhs, double dc_nominal, int max_steps, double* rel, int* nsolve, int* reverted) override {
        int status = 0;
        const int st = emi_kkt_solve_refined(ctx, rhs, dc_nominal, max_steps, rel, nsolve, reverted, &status);
        if (st == EMI_ERR_UNSUPPORTED) return 1;          // LU fallback: the host loop refines around solve()
        return st == EMI_OK ? status : -1;
    }
    std::string last_error() const override { return ctx ? emi_last_error(ctx) : "no device context"; }
};

// ---------------------------------------------------------------------------------------------
// KktBatcher: the rendezvous (see include/ETOL/eMI355X.hpp)
// ---------------------------------------------------------------------------------------------
namespace mi355x {

struct KktBatcher::Impl {
    struct Req {
        int op = 0;                                 // 0 factor, 1 solve (one right-hand side), 2 solve with the refinement on the device,
                                                    // 3 low-rank correction (one scenario after the other on the leader's thread: its kernels
                                                    // are large, r right-hand sides each, and do not gain from sharing launches -- but many
                                                    // host threads launching them at once take turns on the runtime's lock)
        mi355x::KktBackend* direct = nullptr;       // op 3: the scenario's own adapter
        int lr_r = 0;
        const int* lr_node = nullptr;
        const double *lr_vec = nullptr, *lr_delta = nullptr;
        bool lr_exact = false;
        emi_ctx_t ctx = nullptr;
        int nodes = 0;
        const double *Q = nullptr, *J = nullptr;
        const unsigned char* fixed = nullptr;
        double dc = 0;
        double* rhs = nullptr;
        double rel = 0;                             // op 2 (refined solve): relative residual, solves used, a correction was taken back
        int nsolve = 0, reverted = 0;
        int result = -1;
        bool done = false;
        std::chrono::steady_clock::time_point posted;
    };
    // One rendezvous per MESH SIZE: the solves in flight sit on different rungs of their mesh ladders (33, 65, ... nodes), and only
    // requests of one size share launches.  The members of a level are the solves currently iterating on that mesh; they move in step
    // (one Newton iteration per round), cheap coarse-mesh rounds at their own pace beside the expensive fine-mesh ones.
    struct Level {
        int members = 0;
        bool leading = false;
        std::vector<Req*> pending;
    };
    std::mutex m;
    std::condition_variable cv;
    std::map<int, Level> levels;
};

KktBatcher::KktBatcher() : impl(new Impl()) {}
KktBatcher::~KktBatcher() { delete impl; }
KktBatcher::Member::Member(const std::shared_ptr<KktBatcher>& b) : batcher(b) {}
KktBatcher::Member::~Member() {}
KktBatcher::OnMesh::OnMesh(KktBatcher* b, int nodes_) : batcher(b), nodes(nodes_) {
    if (!batcher) return;
    std::lock_guard<std::mutex> lk(batcher->impl->m);
    ++batcher->impl->levels[nodes].members;
}
KktBatcher::OnMesh::~OnMesh() {
    if (!batcher) return;
    {
        std::lock_guard<std::mutex> lk(batcher->impl->m);
        --batcher->impl->levels[nodes].members;
    }
    batcher->impl->cv.notify_all();                 // whoever waits for "everyone on this mesh is here" counts again
}

// what has gathered, as batched calls: one per operation and mesh size
static void run_batch(KktBatcher* B, std::vector<KktBatcher::Impl::Req*>& take) {
    using Req = KktBatcher::Impl::Req;
    std::sort(take.begin(), take.end(), [](const Req* a, const Req* b) { return a->op != b->op ? a->op < b->op : a->nodes < b->nodes; });
    size_t i = 0;
    while (i < take.size()) {
        size_t j = i;
        while (j < take.size() && take[j]->op == take[i]->op && take[j]->nodes == take[i]->nodes) ++j;
        const int n = (int)(j - i);
        std::vector<emi_ctx_t> ctxs(n);
        for (int b = 0; b < n; ++b) ctxs[b] = take[i + b]->ctx;
        if (take[i]->op == 0) {
            std::vector<const double*> Q(n), J(n);
            std::vector<const unsigned char*> fx(n);
            std::vector<double> dc(n);
            std::vector<int> info(n, -1);
            for (int b = 0; b < n; ++b) { Q[b] = take[i + b]->Q; J[b] = take[i + b]->J; fx[b] = take[i + b]->fixed; dc[b] = take[i + b]->dc; }
            int st = n == 1 ? emi_kkt_factor(ctxs[0], Q[0], J[0], fx[0], dc[0], &info[0])
                            : emi_kkt_factor_batch(n, ctxs.data(), Q.data(), J.data(), fx.data(), dc.data(), info.data());
            if (st == EMI_ERR_UNSUPPORTED && n > 1) {               // (a context forced onto the LU method: one by one)
                st = EMI_OK;
                for (int b = 0; b < n && st == EMI_OK; ++b) st = emi_kkt_factor(ctxs[b], Q[b], J[b], fx[b], dc[b], &info[b]);
            }
            for (int b = 0; b < n; ++b) take[i + b]->result = st == EMI_OK ? info[b] : -1;
            __sync_fetch_and_add(&B->factor_calls, 1L);
            __sync_fetch_and_add(&B->factor_items, (long)n);
        } else if (take[i]->op == 3) {
            for (int b = 0; b < n; ++b) {
                KktBatcher::Impl::Req* q = take[i + b];
                q->result = q->direct->lowrank(q->lr_r, q->lr_node, q->lr_vec, q->lr_delta, &q->lr_exact);
            }
        } else if (take[i]->op == 1) {
            std::vector<double*> rhs(n);
            for (int b = 0; b < n; ++b) rhs[b] = take[i + b]->rhs;
            const int st = n == 1 ? emi_kkt_solve(ctxs[0], rhs[0], 1) : emi_kkt_solve_batch(n, ctxs.data(), rhs.data());
            for (int b = 0; b < n; ++b) take[i + b]->result = st == EMI_OK ? 0 : -1;
            __sync_fetch_and_add(&B->solve_calls, 1L);
            __sync_fetch_and_add(&B->solve_items, (long)n);
        } else {
            // refined solves: the scenarios whose factorisation is the LU fallback answer "not offered" (1) and refine on the host
            std::vector<double*> rhs;
            std::vector<double> dcn, rel;
            std::vector<emi_ctx_t> cs;
            std::vector<int> which;
            for (int b = 0; b < n; ++b) {
                double d0 = 0, d1 = 0;
                if (emi_kkt_last_regularisation(ctxs[b], &d0, &d1) == EMI_OK && emi_kkt_is_schur(ctxs[b])) {
                    which.push_back(b); cs.push_back(ctxs[b]); rhs.push_back(take[i + b]->rhs); dcn.push_back(take[i + b]->dc);
                } else {
                    take[i + b]->result = 1;
                }
            }
            const int m = (int)which.size();
            if (m > 0) {
                rel.assign(m, 0.0);
                std::vector<int> nsv(m, 0), rev(m, 0), stat(m, 0);
                const int st = emi_kkt_solve_refined_batch(m, cs.data(), rhs.data(), dcn.data(), 8, rel.data(), nsv.data(), rev.data(), stat.data());
                for (int a = 0; a < m; ++a) {
                    KktBatcher::Impl::Req* q = take[i + which[a]];
                    q->result = st == EMI_OK ? stat[a] : -1;
                    q->rel = rel[a]; q->nsolve = nsv[a]; q->reverted = rev[a];
                }
                __sync_fetch_and_add(&B->solve_calls, 1L);
                __sync_fetch_and_add(&B->solve_items, (long)m);
            }
        }
        if (n > B->largest_batch) B->largest_batch = n;
        i = j;
    }
}

static int submit(KktBatcher* B, KktBatcher::Impl::Req& r) {
    KktBatcher::Impl* I = B->impl;
    std::unique_lock<std::mutex> lk(I->m);
    KktBatcher::Impl::Level& Lv = I->levels[r.nodes];          // (std::map: references stay valid)
    r.posted = std::chrono::steady_clock::now();
    Lv.pending.push_back(&r);
    // Nobody is woken by an arrival: the thread that completes the set finds that out itself, right here; the others sleep until the
    // leader of their batch is through (notify_all), a member leaves their mesh (notify_all), or their own request turns stale.
    // (The first form polled every 50 us from every waiting thread: 64 workers kept the 16 host cores of a GPU box busy with wake-ups
    // and the set ran at a third of the speed of 8 free-running threads, profiles/r04_notes.md.)
    const auto deadline = r.posted + std::chrono::microseconds(B->flush_us);
    for (;;) {
        if (r.done) return r.result;
        if (!Lv.leading && !Lv.pending.empty()) {
            const bool everyone = (int)Lv.pending.size() >= Lv.members;
            bool stale = false;
            if (!everyone) {
                const auto now = std::chrono::steady_clock::now();
                for (const KktBatcher::Impl::Req* q : Lv.pending) stale = stale || now - q->posted > std::chrono::microseconds(B->flush_us);
            }
            if (everyone || stale) {
                Lv.leading = true;
                std::vector<KktBatcher::Impl::Req*> take;
                take.swap(Lv.pending);
                lk.unlock();
                run_batch(B, take);
                lk.lock();
                for (KktBatcher::Impl::Req* q : take) q->done = true;
                Lv.leading = false;
                I->cv.notify_all();
                continue;
            }
        }
        if (std::chrono::steady_clock::now() >= deadline) I->cv.wait_for(lk, std::chrono::microseconds(B->flush_us));   // (stale already, a leader is busy)
        else I->cv.wait_until(lk, deadline);
    }
}
}  // namespace mi355x

// The device Newton step of a solver that shares a KktBatcher: factorisations and single-right-hand-side solves go to the
// rendezvous, everything else (the low-rank correction with its many right-hand sides) straight to the solver's own context.
struct BatchedKkt : public mi355x::KktBackend {
    mi355x::KktBatcher* B;
    mi355x::KktBackend* direct;          // the solver's own device adapter
    emi_ctx_t ctx;
    int nodes;
    mi355x::KktBatcher::OnMesh on_mesh;  // this solve counts as a member of its mesh size while it iterates
    BatchedKkt(mi355x::KktBatcher* b, mi355x::KktBackend* d, emi_ctx_t c, int n) : B(b), direct(d), ctx(c), nodes(n), on_mesh(b, n) {}
    int factor(const double* Qblk, const double* Jblk, const unsigned char* fixed, double dc) override {
        mi355x::KktBatcher::Impl::Req r;
        r.op = 0; r.ctx = ctx; r.nodes = nodes; r.Q = Qblk; r.J = Jblk; r.fixed = fixed; r.dc = dc;
        return mi355x::submit(B, r);
    }
    int lowrank(int r, const int* node, const double* vec, const double* delta, bool* exact) override {
        if (r == 0) return direct->lowrank(r, node, vec, delta, exact);        // (clears the correction: no device work)
        mi355x::KktBatcher::Impl::Req q;
        q.op = 3; q.ctx = ctx; q.nodes = nodes; q.direct = direct; q.lr_r = r; q.lr_node = node; q.lr_vec = vec; q.lr_delta = delta;
        const int rc = mi355x::submit(B, q);
        *exact = q.lr_exact;
        return rc;
    }
    int solve(double* rhs, int nrhs) override {
        if (nrhs != 1) return direct->solve(rhs, nrhs);
        mi355x::KktBatcher::Impl::Req r;
        r.op = 1; r.ctx = ctx; r.nodes = nodes; r.rhs = rhs;
        return mi355x::submit(B, r);
    }
    int solve_refined(double* rhs, double dc_nominal, int max_steps, double* rel, int* nsolve, int* reverted) override {
        (void)max_steps;
        mi355x::KktBatcher::Impl::Req r;
        r.op = 2; r.ctx = ctx; r.nodes = nodes; r.rhs = rhs; r.dc = dc_nominal;
        const int rc = mi355x::submit(B, r);
        *rel = r.rel; *nsolve = r.nsolve; *reverted = r.reverted;
        return rc;
    }
    void applied_regularisation(double* dc, double* dw) override { direct->applied_regularisation(dc, dw); }
    std::string last_error() const override { return direct->last_error(); }
};

namespace {
[[noreturn]] void die(const std::string& msg) {
    fprintf(stderr, "eMI355X: %s\n", msg.c_str());
    exit(EXIT_FAILURE);
}
void must(int status, emi_ctx_t ctx, const char* what) {
    if (status == EMI_OK) return;
    die(std::string(what) + ": " + emi_status_string(status) + (ctx ? std::string(" - ") + emi_last_error(ctx) : ""));
}
}  // namespace

eMI355X::eMI355X() : TrajectoryOptimizer() {}
eMI355X::~eMI355X() {}

mi355x::Alg* eMI355X::getAlgorithm() { return &_algorithm; }
mi355x::Sol* eMI355X::getSolution() { return &_solution; }
mi355x::Prob* eMI355X::getProblem() { return &_problem; }

// One call per callback, with Symbols in the anys; collects the model and the path table.
void eMI355X::traceCallbacks() {
    mi355x::Prob& P = _problem;
    mi355x::Trace::active().clear();     // the handles below are the inputs of a fresh trace
    vector_t x, u;
    for (size_t i = 0; i < getNStates(); ++i) x.push_back(mi355x::Symbol{mi355x::Symbol::STATE, i});
    for (size_t j = 0; j < getNControls(); ++j) u.push_back(mi355x::Symbol{mi355x::Symbol::CONTROL, j});
    // Delayed values, appended in ePSOPT::dae's order (ePSOPT.cpp:231-248): x(t - i dt) of every state for i = 1 ..
    // Xrhorizon - 1 behind the states, u(t - i dt) of every control for i = 1 .. Urhorizon behind the controls.  For the
    // device they are extra inputs of the node functions: handles on control slots nc .. nc + ndelayed - 1, states' delayed
    // copies first (the layout emi_set_delays documents).
    const size_t nsx = getNStates(), ncx = getNControls();
    const size_t nxd = getXrhorizon() > 1 ? (getXrhorizon() - 1) * nsx : 0, nud = getUrhorizon() * ncx;
    for (size_t q = 0; q < nxd; ++q) x.push_back(mi355x::Symbol{mi355x::Symbol::CONTROL, ncx + q});
    for (size_t q = 0; q < nud; ++q) u.push_back(mi355x::Symbol{mi355x::Symbol::CONTROL, ncx + nxd + q});
    const int nc_model = (int)(ncx + nxd + nud);        // controls of the device model: free controls + delayed values
    const std::any tsym = mi355x::Symbol{mi355x::Symbol::TIME, 0};
    if (_objective == NULL) die("no objective function set");
    if (_gradient.size() != getNStates()) die("setGradient needs one function per state");
    P.model_source.clear();
    try {
        bool traced = false;
        int cost_node = -1;
        std::vector<int> f_nodes;
        {
            vector_t params = {std::string()};
            std::vector<std::string> pnames = {std::string("")};
            const std::any out = (*_objective)(x, u, params, pnames, tsym, getDt());
            if (out.type() == typeid(mi355x::Var)) {
                // computed with the handles: a traced model (the ePSOPT way of writing callbacks)
                traced = true;
                cost_node = std::any_cast<mi355x::Var>(out).node;
                if (cost_node < 0) die("the objective callback returned a Var that is not part of the trace");
            } else {
                const mi355x::ModelTerm t = std::any_cast<mi355x::ModelTerm>(out);
                if (t.row != -1) die("the objective callback must return mi355x::objective(...) or a traced Var");
                if (nxd + nud > 0) die("delayed states / controls (rhorizon) need callbacks computed with the mi355x::Var handles; the hand-written device models take none");
                P.model = t.model;
                P.model_params = t.params;
            }
        }
        for (size_t i = 0; i < getNStates(); ++i) {
            vector_t params = {std::string()};
            std::vector<std::string> pnames = {std::string("")};
            const std::any out = (*_gradient.at(i))(x, u, params, pnames, tsym, getDt());
            if (traced) {
                if (out.type() != typeid(mi355x::Var))
                    die("gradient callback " + std::to_string(i) + " must return a traced Var like the objective");
                f_nodes.push_back(std::any_cast<mi355x::Var>(out).node);
                if (f_nodes.back() < 0) die("gradient callback " + std::to_string(i) + " returned a Var outside the trace");
                continue;
            }
            const mi355x::ModelTerm t = std::any_cast<mi355x::ModelTerm>(out);
            if (t.row != (int)i || t.model != P.model || t.params != P.model_params)
                die("gradient callback " + std::to_string(i) + " does not describe state derivative " +
                    std::to_string(i) + " of the objective's model");
        }
        P.path_records.clear();
        P.tracks.clear();
        P.ntracks = 0;
        P.row_order.clear();
        P.traced_scale.clear();
        bool have_xy = false;
        std::vector<int> path_nodes;            // traced rows, in callback order
        std::vector<size_t> table_pos, traced_pos;   // position of every row in callback (= parameter) order
        size_t row_counter = 0;
        for (size_t c = 0; c < _constraints.size(); ++c) {
            vector_t params = {std::string()};
            std::vector<std::string> pnames = {std::string("")};
            const std::any cout_ = (*_constraints.at(c))(x, u, params, pnames, tsym, getDt());
            if (cout_.type() == typeid(fout_mi355x_vars_t)) {
                // rows computed with the handles, as ePSOPT's fout_psopt_t of adoubles (etol_psopt_example1.cpp:153-190)
                if (!traced) die("traced constraint rows need a traced objective and traced dynamics (return mi355x::Var there too)");
                for (const mi355x::Var& v : std::any_cast<fout_mi355x_vars_t>(cout_)) {
                    if (v.node < 0) die("constraint callback " + std::to_string(c) + " returned a Var outside the trace");
                    path_nodes.push_back(v.node);
                    traced_pos.push_back(row_counter++);
                }
                continue;
            }
            const fout_mi355x_t blk = std::any_cast<fout_mi355x_t>(cout_);
            for (size_t q = 0; q < blk.rows.size() + blk.tracks.size(); ++q) table_pos.push_back(row_counter++);
            if (blk.rows.empty() && blk.tracks.empty()) continue;
            if (have_xy && (blk.px != P.px || blk.py != P.py))
                die("all keep-out rows must act on the same two states");
            P.px = blk.px;
            P.py = blk.py;
            have_xy = true;
            for (const auto& r : blk.rows) P.path_records.insert(P.path_records.end(), r.begin(), r.end());
            for (const mi355x::TrackTable& tb : blk.tracks) {
                if (tb.t.size() < 2) die("a moving exclusion zone needs at least two waypoints");
                P.tracks.push_back(tb);
                std::array<double, EMI_PATH_REC> rec{};
                rec[0] = (double)EMI_PATH_TRACK;
                rec[1] = (double)P.ntracks++;
                rec[2] = tb.radius * tb.radius;
                P.path_records.insert(P.path_records.end(), rec.begin(), rec.end());
            }
        }
        P.npath_traced = path_nodes.size();
        P.path_vars.clear();
        if (!path_nodes.empty()) {
            // Traced rows may depend on any states and controls of their node; the keep-out geometry (guess repair,
            // bent starts) works on the two position states: those of the table rows, or else the first two states
            // the traced rows use.
            mi355x::Trace& tr = mi355x::Trace::active();
            std::vector<int> used;
            for (int n : path_nodes)
                for (int v : tr.dependencies(n, (int)getNStates(), nc_model))
                    if (std::find(used.begin(), used.end(), v) == used.end()) used.push_back(v);
            std::sort(used.begin(), used.end());
            if (!have_xy) {
                std::vector<int> st;
                for (int v : used) if (v < (int)getNStates()) st.push_back(v);
                P.px = st.empty() ? 0 : st[0];
                P.py = st.size() > 1 ? st[1] : (P.px + 1 < getNStates() ? P.px + 1 : (P.px > 0 ? P.px - 1 : 0));
            }
            // row normalisation (the iteration works on sigma_j c_j): largest value along the straight line between
            // the boundary states, which for a keep-out row is reached where the line passes closest to its centre
            std::vector<double> xs(getNStates()), us(nc_model, 0.0);
            for (int n : path_nodes) {
                double ref = 0;
                for (int q = 0; q <= 32; ++q) {
                    for (size_t i = 0; i < getNStates(); ++i) xs[i] = getX0()[i] + (getXf()[i] - getX0()[i]) * q / 32.0;
                    ref = std::max(ref, std::fabs(tr.eval(n, xs, us, P.tf * q / 32.0)));
                }
                P.traced_scale.push_back(ref > 0 && std::isfinite(ref) ? 1.0 / ref : 1.0);
            }
        }
        if (traced) {
            // derivatives and code for the device, once (ePSOPT: ADOL-C tape re-interpreted per evaluation)
            P.model = EMI_MODEL_SOURCE;
            P.model_params.clear();
            std::string gerr;
            P.model_source = mi355x::Trace::active().generate_model("TracedModel", (int)getNStates(), nc_model, f_nodes,
                                                                     cost_node, path_nodes, &P.path_vars, &gerr);
            if (P.model_source.empty()) die(gerr);
        }
        // rows are evaluated table rows first, traced rows after them; bounds arrive in callback order
        P.row_order = table_pos;
        P.row_order.insert(P.row_order.end(), traced_pos.begin(), traced_pos.end());
    } catch (std::bad_any_cast& e) {
        _eAny = &e;
        std::cout << "Error in eMI355X callback trace" << std::endl;
        errorHandler();
    }
    P.npath = P.path_records.size() / EMI_PATH_REC + P.npath_traced;
}

// ETOL bounds -> NLP bounds, as ePSOPT::addBounds (reference :125-155): state/control
// boxes at every node, hard initial state, boxed terminal state, one [lbnd,ubnd] pair
// per path row taken from _parameters in map (name-sorted) order, fixed horizon.
void eMI355X::addBounds() {
    mi355x::Prob& P = _problem;
    const size_t ns = getNStates(), nc = getNControls();
    if (getXlower().size() < ns || getXupper().size() < ns || getX0().size() < ns || getXf().size() < ns ||
        getXtol().size() < ns || getUlower().size() < nc || getUupper().size() < nc)
        die("bounds / initial / terminal vectors are shorter than the state or control count");
    P.state_lower.assign(getXlower().begin(), getXlower().begin() + ns);
    P.state_upper.assign(getXupper().begin(), getXupper().begin() + ns);
    P.control_lower.assign(getUlower().begin(), getUlower().begin() + nc);
    P.control_upper.assign(getUupper().begin(), getUupper().begin() + nc);
    P.event_lower.assign(2 * ns, 0.0);
    P.event_upper.assign(2 * ns, 0.0);
    for (size_t i = 0; i < ns; ++i) {
        P.event_lower[i] = getX0()[i];
        P.event_upper[i] = getX0()[i];
        P.event_lower[i + ns] = getXf()[i] - getXtol()[i];
        P.event_upper[i + ns] = getXf()[i] + getXtol()[i];
    }
    // one [lbnd, ubnd] pair per path row, taken from _parameters in map order for the rows in callback order
    // (ePSOPT.cpp:144-149), then put into evaluation order (table rows first, traced rows after them)
    std::vector<double> lo, up;
    for (const auto& kv : _parameters) {
        lo.push_back(kv.second.lbnd);
        up.push_back(kv.second.ubnd);
    }
    P.path_lower = lo;
    P.path_upper = up;
    if (P.row_order.size() == lo.size())
        for (size_t q = 0; q < lo.size(); ++q) {
            P.path_lower[q] = lo[P.row_order[q]];
            P.path_upper[q] = up[P.row_order[q]];
        }
}

void eMI355X::setup() {
    mi355x::Prob& P = _problem;
    // ePSOPT::dae appends delayed states for i = 1 .. Xrhorizon-1 and delayed controls for i = 1 .. Urhorizon
    // (ePSOPT.cpp:231-248): a state horizon of 0 or 1 adds nothing (the shipped mip_2d_ex1.xml has rhorizon="1")
    P.nstates = getNStates();
    P.ncontrols = getNControls();
    P.xhorizon = getXrhorizon();
    P.uhorizon = getUrhorizon();
    P.ndelayed = (P.xhorizon > 1 ? (P.xhorizon - 1) * P.nstates : 0) + P.uhorizon * P.ncontrols;
    P.delay_dt = getDt();                            // delay = getDt() * i, ePSOPT.cpp:232, 241
    P.nodes = getNSteps() + 1;                       // ePSOPT.cpp:44-45
    P.t0 = 0.;
    P.tf = getNSteps() * getDt();                    // fixed horizon, ePSOPT.cpp:151-154
    if (P.nodes < 2 || !(P.tf > 0)) die("nsteps and dt must be positive");
    // a guess the user put into getProblem() before setup() is kept, as ePSOPT keeps a pre-filled
    // phases(1).guess (ePSOPT.cpp:47-56); solve() uses it when it has the size of the first mesh

    traceCallbacks();
    setMesh(P.nodes);
    int ns = (int)P.nstates, nc = (int)P.ncontrols, npar = 0;
    if (P.model != EMI_MODEL_SOURCE && emi_model_dims(P.model, &ns, &nc, &npar) != EMI_OK) die("unknown device model");
    if ((size_t)ns != P.nstates || (size_t)nc != P.ncontrols)
        die("the device model has " + std::to_string(ns) + " states / " + std::to_string(nc) +
            " controls, the configuration has " + std::to_string(P.nstates) + " / " + std::to_string(P.ncontrols));
    if (P.npath != _parameters.size())   // ePSOPT sizes npath from _parameters (:58)
        die("constraint callbacks returned " + std::to_string(P.npath) + " rows but " +
            std::to_string(_parameters.size()) + " were registered with addParams");
    addBounds();

    // a device context (stream, rocBLAS handle, workspaces, compiled model) is expensive to build and cheap to
    // re-point at another problem: solvers of one thread hand theirs on (Monte-Carlo runs set up thousands)
    _dev.reset();
    std::vector<Device*>& pool = device_pool();
    for (size_t i = 0; i < pool.size(); ++i)
        if (pool[i]->device_id == _algorithm.device) {
            _dev.reset(pool[i]);
            pool.erase(pool.begin() + i);
            break;
        }
    if (!_dev) _dev.reset(new Device());
    _dev->device_id = _algorithm.device;
    configureDevice(_dev.get());

    // algorithm defaults of ePSOPT::setup (:62-72) that have a meaning here
    _algorithm.nlp_iter_max = 200;
    _algorithm.nlp_tolerance = 1.e-6;
    _algorithm.mesh_refinement = "automatic";
    _algorithm.mr_max_iterations = 10;
    _algorithm.ode_tolerance = 1.e-4;
    _algorithm.print_level = 0;
}

// LGL nodes / weights / D for `nodes` collocation points and the moving-zone centres at the
// node times (PSOPT builds the same per mesh-refinement iteration).
void eMI355X::setMesh(size_t nodes) {
    mi355x::Prob& P = _problem;
    P.nodes = nodes;
    P.tau.resize(nodes);
    P.w.resize(nodes);
    P.D.resize(nodes * nodes);
    must(emi_lgl((int)nodes, P.tau.data(), P.w.data(), P.D.data()), nullptr, "emi_lgl");
    std::vector<double> node_t(nodes);
    for (size_t k = 0; k < nodes; ++k) node_t[k] = P.t0 + (P.tf - P.t0) / 2.0 * (P.tau[k] + 1.0);
    P.track_x.assign(P.tracks.size() * nodes, 0.0);
    P.track_y.assign(P.tracks.size() * nodes, 0.0);
    for (size_t i = 0; i < P.tracks.size(); ++i) {
        const mi355x::TrackTable& tb = P.tracks[i];
        must(emi_track_centres((int)tb.t.size(), tb.t.data(), tb.x.data(), tb.y.data(), (int)nodes, node_t.data(),
                               &P.track_x[i * nodes], &P.track_y[i * nodes]), nullptr, "emi_track_centres");
    }
}

void eMI355X::configureDevice(Device* dev) {
    mi355x::Prob& P = _problem;
    if (!dev->ctx) {
        const int st = emi_create(_algorithm.device, &dev->ctx);
        if (st != EMI_OK)
            die(std::string("cannot open MI355X device ") + std::to_string(_algorithm.device) + ": " +
                emi_status_string(st) + " (there is no CPU fallback)");
    }
    emi_ctx_t c = dev->ctx;
    must(emi_set_mesh(c, (int)P.nodes, P.tau.data(), P.w.data(), P.D.data(), P.t0, P.tf), c, "emi_set_mesh");
    dev->nodes = (int)P.nodes;
    if (P.model == EMI_MODEL_SOURCE) {
        // compiled for gfx950 once per context; meshes come and go.  The call also fixes the objective sign.
        if (dev->installed_source != P.model_source || dev->installed_maximize != isMaximized())
            must(emi_set_model_source(c, "TracedModel", P.model_source.c_str(), (int)P.nstates, (int)(P.ncontrols + P.ndelayed),
                                      (int)P.npath_traced, P.path_vars.data(), (int)P.path_vars.size(), nullptr, 0,
                                      isMaximized() ? 1 : 0), c, "emi_set_model_source");
        dev->installed_source = P.model_source;
        dev->installed_maximize = isMaximized();
    } else {
        must(emi_set_model(c, P.model, P.model_params.data(), (int)P.model_params.size(), isMaximized() ? 1 : 0), c,
             "emi_set_model");
        dev->installed_source.clear();
    }
    // (lifted: solve() iterates on the delayed values as variables of their node and ties them to their sources itself)
    if (P.lifted) must(emi_set_delays(c, 0, 0, 1.0), c, "emi_set_delays");
    else must(emi_set_delays(c, (int)P.xhorizon, (int)P.uhorizon, P.delay_dt > 0 ? P.delay_dt : 1.0), c, "emi_set_delays");
    must(emi_set_batch(c, 1), c, "emi_set_batch");
    if (P.ntracks)
        must(emi_set_tracks(c, (int)P.ntracks, 1, P.track_x.data(), P.track_y.data()), c, "emi_set_tracks");
    must(emi_set_path(c, (int)(P.npath - P.npath_traced), 1, P.path_records.data(), (int)P.px, (int)P.py), c, "emi_set_path");
}

namespace {
// Barycentric Lagrange interpolation from the LGL nodes (tau, w) to the points t2.  For LGL
// nodes the barycentric weights are (-1)^j sqrt(w_j) up to a common factor, because
// w_j = 2 / (N (N+1) P_N(tau_j)^2) and the Lagrange weights are proportional to 1 / P_N(tau_j).
void interp_lgl(const std::vector<double>& tau, const std::vector<double>& w, const double* v, size_t M,
                const std::vector<double>& t2, double* out) {
    for (size_t q = 0; q < t2.size(); ++q) {
        double num = 0, den = 0;
        bool hit = false;
        for (size_t j = 0; j < M; ++j) {
            const double d = t2[q] - tau[j];
            if (d == 0.0) { out[q] = v[j]; hit = true; break; }
            const double lam = ((j & 1) ? -1.0 : 1.0) * std::sqrt(w[j]) / d;
            num += lam * v[j];
            den += lam;
        }
        if (!hit) out[q] = num / den;
    }
}
}  // namespace

// Relative local ODE error of the solution z on the current mesh, the quantity PSOPT's automatic mesh refinement
// compares with ode_tolerance (ePSOPT.cpp:69-71; PSOPT 5.0.0 follows Betts' relative local error): between every two
// consecutive nodes the ODE residual of the INTERPOLATED solution is integrated,
//     eta_(i,k) = int_{t_k}^{t_k+1} | d/dt x~_i(s) - f_i(x~(s), u~(s), s) | ds ,
// and the estimate is  max_(i,k) eta_(i,k) / (w_i + 1),  w_i = max_k max(|x~_i(t_k)|, |d/dt x~_i(t_k)|).
// x~, u~ are the Lagrange interpolants through the LGL nodes (barycentric form); the integral is a 5-point
// Gauss-Legendre rule per interval (PSOPT uses a Romberg rule; PSOPT is not in the reference tree, so this follows
// the published definition, not its source).  f at the (M-1) x 5 quadrature points is evaluated ON THE DEVICE: the
// node kernel on a points-only mesh (emi_set_mesh with D = NULL), values only.
double eMI355X::odeError(const std::vector<double>& z, std::vector<double>* z_fine, size_t* nodes_fine) {
    mi355x::Prob& P = _problem;
    const size_t ns = P.nstates, nc = P.ncontrols, nv = ns + nc, M = P.nodes;
    static const double gx[5] = {-0.9061798459386640, -0.5384693101056831, 0.0, 0.5384693101056831, 0.9061798459386640};
    static const double gw[5] = {0.2369268850561891, 0.4786286704993665, 0.5688888888888889, 0.4786286704993665, 0.2369268850561891};
    const size_t Q = 5, Mq = (M - 1) * Q;
    const double h = (P.tf - P.t0) / 2.0;
    std::vector<double> tq(Mq), wq(Mq, 0.0);
    for (size_t k = 0; k + 1 < M; ++k)
        for (size_t q = 0; q < Q; ++q) tq[k * Q + q] = P.tau[k] + 0.5 * (P.tau[k + 1] - P.tau[k]) * (gx[q] + 1.0);
    // interpolant and its tau-derivative at the quadrature points (barycentric: weights (-1)^j sqrt(w_j))
    std::vector<double> zq(nv * Mq), dq(ns * Mq);
    for (size_t p = 0; p < Mq; ++p) {
        double den = 0;
        std::vector<double> lam(M);
        for (size_t j = 0; j < M; ++j) {
            lam[j] = ((j & 1) ? -1.0 : 1.0) * std::sqrt(P.w[j]) / (tq[p] - P.tau[j]);
            den += lam[j];
        }
        for (size_t v = 0; v < nv; ++v) {
            double num = 0;
            for (size_t j = 0; j < M; ++j) num += lam[j] * z[v * M + j];
            const double val = num / den;
            zq[v * Mq + p] = val;
            if (v < ns) {      // p'(s) = sum_j lam_j (p(s) - x_j) / (s - tau_j) / sum_j lam_j
                double dn = 0;
                for (size_t j = 0; j < M; ++j) dn += lam[j] * (val - z[v * M + j]) / (tq[p] - P.tau[j]);
                dq[v * Mq + p] = dn / den;
            }
        }
    }
    for (size_t j = 0; j < nc; ++j)      // the interpolant of a bounded control may overshoot: clip
        for (size_t p = 0; p < Mq; ++p)
            zq[(ns + j) * Mq + p] = std::min(std::max(zq[(ns + j) * Mq + p], P.control_lower[j]), P.control_upper[j]);
    // moving-zone centres at the quadrature times, then f at the points on the device
    emi_ctx_t c = _dev->ctx;
    must(emi_set_mesh(c, (int)Mq, tq.data(), wq.data(), nullptr, P.t0, P.tf), c, "emi_set_mesh (points)");
    if (P.ntracks) {
        std::vector<double> tt(Mq), tx(P.tracks.size() * Mq), ty(P.tracks.size() * Mq);
        for (size_t p = 0; p < Mq; ++p) tt[p] = P.t0 + h * (tq[p] + 1.0);
        for (size_t i = 0; i < P.tracks.size(); ++i) {
            const mi355x::TrackTable& tb = P.tracks[i];
            must(emi_track_centres((int)tb.t.size(), tb.t.data(), tb.x.data(), tb.y.data(), (int)Mq, tt.data(), &tx[i * Mq], &ty[i * Mq]),
                 nullptr, "emi_track_centres");
        }
        must(emi_set_tracks(c, (int)P.ntracks, 1, tx.data(), ty.data()), c, "emi_set_tracks");
    }
    std::vector<double> RES((ns + P.npath) * Mq), cost(1);
    must(emi_eval_host(c, zq.data(), zq.data() + ns * Mq, RES.data(), nullptr, cost.data(), EMI_EVAL_NODES | EMI_EVAL_NOJAC), c,
         "emi_eval_host (ODE error)");
    // weights w_i from the nodes: |x_ik| and |(D x)_ik| / h
    double err = 0;
    for (size_t i = 0; i < ns; ++i) {
        double wi = 0;
        for (size_t k = 0; k < M; ++k) {
            double dx = 0;
            for (size_t j = 0; j < M; ++j) dx += P.D[k * M + j] * z[i * M + j];
            wi = std::max(wi, std::max(std::fabs(z[i * M + k]), std::fabs(dx) / h));
        }
        for (size_t k = 0; k + 1 < M; ++k) {
            double eta = 0;
            for (size_t q = 0; q < Q; ++q) {
                const size_t p = k * Q + q;
                const double xdot = dq[i * Mq + p] / h, f = -RES[i * Mq + p] / h;      // RES defect rows hold -h f
                eta += gw[q] * std::fabs(xdot - f);
            }
            eta *= 0.5 * (P.tau[k + 1] - P.tau[k]) * h;
            err = std::max(err, eta / (wi + 1.0));
        }
    }
    if (z_fine) z_fine->clear();
    if (nodes_fine) *nodes_fine = 0;
    configureDevice(_dev.get());         // back to the collocation mesh
    return err;
}

namespace mi355x {

NlpProblem make_nlp(const Prob& P, NlpEvaluator* ev) {
    const size_t ns = P.nstates, ncf = P.ncontrols, nc = ncf + (P.lifted ? P.ndelayed : 0), M = P.nodes, nv = ns + nc;
    NlpProblem nlp;
    nlp.ns = (int)ns; nlp.nc = (int)nc; nlp.np = (int)P.npath; nlp.M = (int)M;
    nlp.px = (int)P.px; nlp.py = (int)P.py;
    // (variable, VALS entry) pairs of every path row: table rows two partials on (px, py), traced rows one per variable
    // of the model's list
    {
        const int nv = (int)(P.nstates + nc), base = (int)P.nstates * nv;
        const int ntab = (int)(P.npath - P.npath_traced), pw = (int)P.path_vars.size();
        nlp.row_vars.clear();
        for (int j = 0; j < ntab; ++j) nlp.row_vars.push_back({{(int)P.px, base + 2 * j}, {(int)P.py, base + 2 * j + 1}});
        for (int j = 0; j < (int)P.npath_traced; ++j) {
            std::vector<std::pair<int, int>> rv;
            for (int q = 0; q < pw; ++q) rv.push_back({P.path_vars[q], base + 2 * ntab + j * pw + q});
            nlp.row_vars.push_back(rv);
        }
    }
    nlp.D = P.D;
    nlp.ev = ev;
    nlp.zl.resize(nv * M);
    nlp.zu.resize(nv * M);
    for (size_t i = 0; i < ns; ++i)
        for (size_t k = 0; k < M; ++k) {
            double l = P.state_lower[i], u = P.state_upper[i];
            if (k == 0) { l = std::max(l, P.event_lower[i]); u = std::min(u, P.event_upper[i]); }
            if (k == M - 1) { l = std::max(l, P.event_lower[ns + i]); u = std::min(u, P.event_upper[ns + i]); }
            nlp.zl[i * M + k] = l;
            nlp.zu[i * M + k] = u;
        }
    for (size_t j = 0; j < nc; ++j)
        for (size_t k = 0; k < M; ++k) {
            nlp.zl[(ns + j) * M + k] = j < ncf ? P.control_lower[j] : -1e20;        // delayed values: free
            nlp.zu[(ns + j) * M + k] = j < ncf ? P.control_upper[j] : 1e20;
        }
    if (P.lifted) {
        // coupling rows, in the order ePSOPT::dae appends the delayed values (reference ePSOPT.cpp:231-248; emi_set_delays):
        // x(t - i dt) of every state for i = 1 .. xhorizon - 1, then u(t - i dt) of every control for i = 1 .. uhorizon
        std::vector<double> W(M * M);
        size_t slot = ns + ncf;
        auto add = [&](size_t src, double delay) {
            if (emi_delay_matrix((int)M, P.tau.data(), P.w.data(), P.t0, P.tf, delay, W.data()) != EMI_OK) return;
            NlpLink L;
            L.dst = (int)slot++;
            L.src = (int)src;
            L.W = W;
            nlp.links.push_back(std::move(L));
        };
        for (size_t i = 1; i < P.xhorizon; ++i)
            for (size_t st = 0; st < ns; ++st) add(st, (double)i * P.delay_dt);
        for (size_t i = 1; i <= P.uhorizon; ++i)
            for (size_t c = 0; c < ncf; ++c) add(ns + c, (double)i * P.delay_dt);
    }
    nlp.cl = P.path_lower;
    nlp.cu = P.path_upper;
    // keep-out rows are iterated on normalised: ellipse / (a^2 b^2), disc / r^2
    nlp.cscale.assign(P.npath, 1.0);
    for (size_t j = 0; j < P.npath_traced && j < P.traced_scale.size(); ++j)
        nlp.cscale[P.npath - P.npath_traced + j] = P.traced_scale[j];
    for (size_t j = 0; j < P.npath - P.npath_traced; ++j) {
        const double* r = &P.path_records[j * EMI_PATH_REC];
        const int kind = (int)r[0];
        const double ref = kind == EMI_PATH_ELLIPSE ? r[5] * r[6] : (kind == EMI_PATH_DISC ? r[3] : r[2]);
        if (ref > 0 && std::isfinite(ref)) nlp.cscale[j] = 1.0 / ref;
    }
    return nlp;
}

// Variable scales of PSOPT's automatic scaling: the larger magnitude of a variable's two bounds (event bounds of the end nodes
// included), 1 where it has no finite bound or is pinned at zero.
std::vector<double> bound_scales(const Prob& P) {
    const size_t ns = P.nstates, nc = P.ncontrols;
    std::vector<double> s(ns + nc, 1.0);
    auto mag = [](double a, double m) { return std::isfinite(a) && std::fabs(a) < 1e19 ? std::max(m, std::fabs(a)) : m; };
    for (size_t i = 0; i < ns; ++i) {
        double m = 0;
        m = mag(P.state_lower[i], m);
        m = mag(P.state_upper[i], m);
        if (P.event_lower.size() == 2 * ns && P.event_upper.size() == 2 * ns)
            for (size_t e : {i, ns + i}) { m = mag(P.event_lower[e], m); m = mag(P.event_upper[e], m); }
        if (m > 0) s[i] = m;
    }
    for (size_t j = 0; j < nc; ++j) {
        double m = 0;
        m = mag(P.control_lower[j], m);
        m = mag(P.control_upper[j], m);
        if (m > 0) s[ns + j] = m;
    }
    if (P.lifted) {          // a delayed value is scaled like its source, so its coupling row keeps the interpolation operator as it is
        for (size_t i = 1; i < P.xhorizon; ++i)
            for (size_t st = 0; st < ns; ++st) s.push_back(s[st]);
        for (size_t i = 1; i <= P.uhorizon; ++i)
            for (size_t c = 0; c < nc; ++c) s.push_back(s[ns + c]);
    }
    return s;
}

// Nodes of a guess that fall inside a keep-out of the record table are moved radially out of it (ellipse: offset
// from the centre scaled until the quadratic form reaches 1+margin).  Inside a keep-out the row function is concave
// with a vanishing gradient at the centre, which is the worst place to start a Newton-type iteration from.
void repair_guess(const Prob& P, double* xs, double* ys) {
    const size_t M = P.nodes;
    if (P.npath - P.npath_traced == 0 || M <= 2) return;
    const double margin = 0.05;
    for (int sweep = 0; sweep < 50; ++sweep) {
        bool moved = false;
        for (size_t k = 1; k + 1 < M; ++k) {
            for (size_t j = 0; j < P.npath - P.npath_traced; ++j) {       // rows of the record table (traced rows: no geometry known)
                const double* r = &P.path_records[j * EMI_PATH_REC];
                const int kind = (int)r[0];
                double xc, yc, ct = 1, st = 0, asq, bsq;
                if (kind == EMI_PATH_ELLIPSE) { xc = r[1]; yc = r[2]; ct = r[3]; st = r[4]; asq = r[5]; bsq = r[6]; }
                else if (kind == EMI_PATH_DISC) { xc = r[1]; yc = r[2]; asq = bsq = r[3]; }
                else { const size_t t = (size_t)r[1]; xc = P.track_x[t * M + k]; yc = P.track_y[t * M + k]; asq = bsq = r[2]; }
                if (!(asq > 0) || !(bsq > 0)) continue;
                const double dx = xs[k] - xc, dy = ys[k] - yc;
                double ex = ct * dx - st * dy, ey = st * dx + ct * dy;
                const double q = ex * ex / asq + ey * ey / bsq;
                if (q >= 1.0 + 0.5 * margin) continue;
                if (q < 1e-12) { ex = 0; ey = std::sqrt(bsq * (1.0 + margin)); }      // dead centre: minor axis
                else { const double g = std::sqrt((1.0 + margin) / q); ex *= g; ey *= g; }
                xs[k] = xc + ct * ex + st * ey;
                ys[k] = yc - st * ex + ct * ey;
                moved = true;
            }
        }
        if (!moved) break;
    }
}

// A path from the start position to the end position through the free space of the STATIC keep-outs of the record table (discs and
// ellipses, grown by `grow` of their size; tracks move and are left to repair_guess): Dijkstra over an 8-connected grid on the box of the
// two position states, then the positions of the nodes spread along it by arc length in node time.  The last of the cold-start guesses
// (solve_cold_with_retries): a bent straight line only reaches the homotopy classes next to the straight one -- a start that sits behind a
// wall of overlapping keep-outs with one gap (Monte-Carlo scenario 938 of config 4, profiles/r04_notes.md section 22) needs a route.
bool planned_path_guess(const Prob& P, double* xs, double* ys, double clearance_weight) {
    const size_t M = P.nodes, ns = P.nstates;
    if (P.px >= ns || P.py >= ns || M < 3 || P.event_lower.size() != 2 * ns) return false;
    const size_t nrec = P.npath - P.npath_traced;
    if (nrec == 0) return false;
    const double x0 = 0.5 * (P.event_lower[P.px] + P.event_upper[P.px]), y0 = 0.5 * (P.event_lower[P.py] + P.event_upper[P.py]);
    const double x1 = 0.5 * (P.event_lower[ns + P.px] + P.event_upper[ns + P.px]), y1 = 0.5 * (P.event_lower[ns + P.py] + P.event_upper[ns + P.py]);
    double xl = P.state_lower[P.px], xu = P.state_upper[P.px], yl = P.state_lower[P.py], yu = P.state_upper[P.py];
    const double span = std::max(std::fabs(x1 - x0), std::fabs(y1 - y0));
    if (!(span > 0)) return false;
    // unbounded positions: a box of twice the span round the end points
    if (!(xl > -1e18) || !(xu < 1e18)) { xl = std::min(x0, x1) - span; xu = std::max(x0, x1) + span; }
    if (!(yl > -1e18) || !(yu < 1e18)) { yl = std::min(y0, y1) - span; yu = std::max(y0, y1) + span; }
    const int N = 161;
    const double hx = (xu - xl) / (N - 1), hy = (yu - yl) / (N - 1);
    if (!(hx > 0) || !(hy > 0)) return false;
    auto cell = [&](double x, double y, int* i, int* j) {
        *i = std::min(N - 1, std::max(0, (int)std::lround((x - xl) / hx)));
        *j = std::min(N - 1, std::max(0, (int)std::lround((y - yl) / hy)));
    };
    int is, js, it, jt;
    cell(x0, y0, &is, &js);
    cell(x1, y1, &it, &jt);
    // clearance of every cell from the nearest static keep-out (exact for discs, from the quadratic form for ellipses): steps through
    // narrow places are charged more, so that the route prefers the middle of a gap to the edge of a keep-out -- a start for an
    // interior-point iteration wants room, not the shortest way
    std::vector<double> clear((size_t)N * N, 1e300);
    for (size_t j = 0; j < nrec; ++j) {
        const double* r = &P.path_records[j * EMI_PATH_REC];
        const int kind = (int)r[0];
        double xc, yc, ct = 1, st = 0, asq, bsq;
        if (kind == EMI_PATH_ELLIPSE) { xc = r[1]; yc = r[2]; ct = r[3]; st = r[4]; asq = r[5]; bsq = r[6]; }
        else if (kind == EMI_PATH_DISC) { xc = r[1]; yc = r[2]; asq = bsq = r[3]; }
        else continue;
        if (!(asq > 0) || !(bsq > 0)) continue;
        const double rmin = std::sqrt(std::min(asq, bsq));
        for (int i = 0; i < N; ++i)
            for (int q = 0; q < N; ++q) {
                const double dx = xl + i * hx - xc, dy = yl + q * hy - yc;
                const double ex = ct * dx - st * dy, ey = st * dx + ct * dy;
                const double c = (std::sqrt(ex * ex / asq + ey * ey / bsq) - 1.0) * rmin;
                clear[(size_t)i * N + q] = std::min(clear[(size_t)i * N + q], c);
            }
    }
    bool any_static = false;
    for (double c : clear) if (c < 1e299) { any_static = true; break; }
    if (!any_static) return false;                        // moving keep-outs only: nothing to plan round (the straight line is the route)
    const double room = clearance_weight * span;          // a step at this clearance costs twice its length
    for (double grow : {0.25, 0.1, 0.0}) {
        std::vector<char> blocked((size_t)N * N, 0);
        for (size_t j = 0; j < nrec; ++j) {
            const double* r = &P.path_records[j * EMI_PATH_REC];
            const int kind = (int)r[0];
            double xc, yc, ct = 1, st = 0, asq, bsq;
            if (kind == EMI_PATH_ELLIPSE) { xc = r[1]; yc = r[2]; ct = r[3]; st = r[4]; asq = r[5]; bsq = r[6]; }
            else if (kind == EMI_PATH_DISC) { xc = r[1]; yc = r[2]; asq = bsq = r[3]; }
            else continue;
            if (!(asq > 0) || !(bsq > 0)) continue;
            const double g = (1.0 + grow) * (1.0 + grow), reach = std::sqrt(std::max(asq, bsq) * g);
            int ia, ja, ib, jb;
            cell(xc - reach, yc - reach, &ia, &ja);
            cell(xc + reach, yc + reach, &ib, &jb);
            for (int i = ia; i <= ib; ++i)
                for (int q = ja; q <= jb; ++q) {
                    const double dx = xl + i * hx - xc, dy = yl + q * hy - yc;
                    const double ex = ct * dx - st * dy, ey = st * dx + ct * dy;
                    if (ex * ex / (asq * g) + ey * ey / (bsq * g) < 1.0) blocked[(size_t)i * N + q] = 1;
                }
        }
        blocked[(size_t)is * N + js] = blocked[(size_t)it * N + jt] = 0;
        // Dijkstra (8 neighbours, Euclidean step lengths)
        std::vector<double> dist((size_t)N * N, 1e300);
        std::vector<int> prev((size_t)N * N, -1);
        typedef std::pair<double, int> QE;
        std::priority_queue<QE, std::vector<QE>, std::greater<QE>> pq;
        dist[(size_t)is * N + js] = 0;
        pq.push({0.0, is * N + js});
        const int goal = it * N + jt;
        while (!pq.empty()) {
            const QE e = pq.top();
            pq.pop();
            if (e.first > dist[e.second]) continue;
            if (e.second == goal) break;
            const int i = e.second / N, q = e.second % N;
            for (int di = -1; di <= 1; ++di)
                for (int dj = -1; dj <= 1; ++dj) {
                    if (!di && !dj) continue;
                    const int a = i + di, b = q + dj;
                    if (a < 0 || b < 0 || a >= N || b >= N || blocked[(size_t)a * N + b]) continue;
                    const double cl = std::max(clear[(size_t)a * N + b], 1e-3 * span);
                    const double d = e.first + std::sqrt(di * di * hx * hx + dj * dj * hy * hy) * (1.0 + (room > 0 ? (room / cl) * (room / cl) : 0.0));
                    if (d < dist[(size_t)a * N + b]) { dist[(size_t)a * N + b] = d; prev[(size_t)a * N + b] = e.second; pq.push({d, a * N + b}); }
                }
        }
        if (!(dist[goal] < 1e299)) continue;                 // no route with the keep-outs grown this much: try them smaller
        std::vector<double> px, py;
        for (int c = goal; c >= 0; c = prev[c]) { px.push_back(xl + (c / N) * hx); py.push_back(yl + (c % N) * hy); }
        std::reverse(px.begin(), px.end());
        std::reverse(py.begin(), py.end());
        px.front() = x0; py.front() = y0; px.back() = x1; py.back() = y1;
        // three passes of neighbour averaging take the grid's staircase out (end points fixed)
        for (int pass = 0; pass < 3; ++pass)
            for (size_t c = 1; c + 1 < px.size(); ++c) { px[c] = 0.25 * px[c - 1] + 0.5 * px[c] + 0.25 * px[c + 1]; py[c] = 0.25 * py[c - 1] + 0.5 * py[c] + 0.25 * py[c + 1]; }
        std::vector<double> arc(px.size(), 0.0);
        for (size_t c = 1; c < px.size(); ++c) arc[c] = arc[c - 1] + std::hypot(px[c] - px[c - 1], py[c] - py[c - 1]);
        if (!(arc.back() > 0)) return false;
        size_t seg = 0;
        for (size_t k = 0; k < M; ++k) {
            const double s = 0.5 * (P.tau[k] + 1.0) * arc.back();
            while (seg + 2 < px.size() && arc[seg + 1] < s) ++seg;
            const double w = arc[seg + 1] > arc[seg] ? (s - arc[seg]) / (arc[seg + 1] - arc[seg]) : 0.0;
            xs[k] = px[seg] + std::min(1.0, std::max(0.0, w)) * (px[seg + 1] - px[seg]);
            ys[k] = py[seg] + std::min(1.0, std::max(0.0, w)) * (py[seg + 1] - py[seg]);
        }
        return true;
    }
    return false;
}

std::vector<double> initial_guess(const Prob& P) {
    const size_t ns = P.nstates, nc = P.ncontrols, M = P.nodes;
    std::vector<double> z0((ns + nc) * M, 0.0);
    // ePSOPT cold-starts every state at 0 (reference ePSOPT.cpp:47-56) and lets IPOPT's
    // restoration phase recover; this iteration has no restoration phase, so the default
    // state guess is the straight line between the boundary states (it satisfies the event
    // bounds; the defects are then O(1) instead of O(N^2 |x|)).  Controls start at 0.
    for (size_t i = 0; i < ns; ++i) {
        const double a = 0.5 * (P.event_lower[i] + P.event_upper[i]);
        const double b = 0.5 * (P.event_lower[ns + i] + P.event_upper[ns + i]);
        for (size_t k = 0; k < M; ++k) z0[i * M + k] = a + (b - a) * 0.5 * (P.tau[k] + 1.0);
    }
    bool planned = false;
    if (P.guess_planned && P.px < ns && P.py < ns) planned = planned_path_guess(P, &z0[P.px * M], &z0[P.py * M], P.guess_clearance);
    if (!planned && P.guess_bend != 0 && P.px < ns && P.py < ns) {
        // the line bent sideways (half a sine wave along it): another homotopy class round the keep-outs
        const double dx = z0[P.px * M + M - 1] - z0[P.px * M], dy = z0[P.py * M + M - 1] - z0[P.py * M];
        const double len = std::sqrt(dx * dx + dy * dy);
        if (len > 0)
            for (size_t k = 0; k < M; ++k) {
                const double s = std::sin(1.5707963267948966 * (P.tau[k] + 1.0));
                double x = z0[P.px * M + k] - P.guess_bend * dy / len * s, y = z0[P.py * M + k] + P.guess_bend * dx / len * s;
                z0[P.px * M + k] = std::min(std::max(x, P.state_lower[P.px]), P.state_upper[P.px]);
                z0[P.py * M + k] = std::min(std::max(y, P.state_lower[P.py]), P.state_upper[P.py]);
            }
    }
    repair_guess(P, &z0[P.px * M], &z0[P.py * M]);
    if (P.guess_states.size() == ns * M) std::copy(P.guess_states.begin(), P.guess_states.end(), z0.begin());
    if (P.guess_controls.size() == nc * M)
        std::copy(P.guess_controls.begin(), P.guess_controls.end(), z0.begin() + ns * M);
    if (P.lifted && P.ndelayed > 0) {
        // delayed values start consistent with their sources (coupling rows satisfied): W(i dt) . guess
        z0.resize((ns + nc + P.ndelayed) * M, 0.0);
        std::vector<double> W(M * M);
        size_t slot = ns + nc;
        auto fill = [&](size_t src, double delay) {
            if (emi_delay_matrix((int)M, P.tau.data(), P.w.data(), P.t0, P.tf, delay, W.data()) != EMI_OK) return;
            for (size_t k = 0; k < M; ++k) {
                double acc = 0;
                for (size_t j = 0; j < M; ++j) acc += W[k * M + j] * z0[src * M + j];
                z0[slot * M + k] = acc;
            }
            ++slot;
        };
        for (size_t i = 1; i < P.xhorizon; ++i)
            for (size_t st = 0; st < ns; ++st) fill(st, (double)i * P.delay_dt);
        for (size_t i = 1; i <= P.uhorizon; ++i)
            for (size_t c = 0; c < nc; ++c) fill(ns + c, (double)i * P.delay_dt);
    }
    return z0;
}

}  // namespace mi355x

void eMI355X::evaluate(const std::vector<double>& z, std::vector<double>* res, std::vector<double>* vals, double* cost) {
    if (!_dev || !_dev->ctx) die("evaluate() called before setup()");
    const mi355x::Prob& P = _problem;
    const size_t ns = P.nstates, nc = P.ncontrols, M = P.nodes;
    if (z.size() != (ns + nc) * M) die("evaluate(): z must hold nstates + ncontrols rows of `nodes` values");
    emi_layout_t lay;
    must(emi_get_layout(_dev->ctx, &lay), _dev->ctx, "emi_get_layout");
    std::vector<double> r((size_t)lay.nres * M), v((size_t)lay.nvals * M);
    double c = 0;
    must(emi_eval_host(_dev->ctx, z.data(), z.data() + ns * M, r.data(), v.data(), &c, vals ? EMI_EVAL_ALL : (EMI_EVAL_ALL | EMI_EVAL_NOJAC)),
         _dev->ctx, "emi_eval_host");
    if (res) *res = r;
    if (vals) *vals = v;
    if (cost) *cost = c;
}

void eMI355X::solve() {
    if (!_dev || !_dev->ctx) die("solve() called before setup()");
    mi355x::Prob& P = _problem;
    P.guess_clearance = _algorithm.plan_clearance;
    // Delayed states / controls (reference ePSOPT.cpp:231-248).  ePSOPT hands them to IPOPT through PSOPT like any other dependency of
    // the node functions.  Here the delayed values become variables of their node for the duration of the solve ("lifted"), tied to their
    // sources by linear coupling rows with the interpolation operators W(i dt) of the mesh: node functions, Jacobian entries and Hessian
    // blocks stay node-local (the device kernels on the extended node variables, unchanged), W appears as constant rows of the KKT matrix.
    // The Newton step of such a problem runs on the dense host backend (the structured device step has no place for a second operator
    // beside D yet), so the mesh is bounded; mesh sequencing and refinement are off (the ODE-error estimate evaluates between the nodes,
    // where the delayed values of a points-only mesh are not defined).
    struct Lift {
        eMI355X* self;
        bool on;
        ~Lift() {
            if (!on) return;
            self->_problem.lifted = false;
            self->configureDevice(self->_dev.get());          // evaluate() forms the delayed values on the device again
        }
    } lift{this, P.ndelayed > 0};
    if (lift.on) {
        const size_t rows = (2 * P.nstates + P.ncontrols + 2 * P.ndelayed) * P.nodes;
        if (rows > 4000)
            die("solve(): a problem with delayed states / controls is solved with the dense host backend, which takes up to 4000 KKT rows; "
                "this one has " + std::to_string(rows) + " ((2 nstates + ncontrols + 2 ndelayed) x nodes): use fewer nodes");
        P.lifted = true;
        configureDevice(_dev.get());
    }
    const size_t ns = P.nstates, nc = P.ncontrols;

    mi355x::NlpOptions opt;
    opt.tol = _algorithm.nlp_tolerance;
    opt.max_iter = _algorithm.nlp_iter_max;
    opt.print_level = _algorithm.print_level;
    opt.max_cpu_time = _algorithm.max_cpu_time;
    opt.max_shift_trials = _algorithm.max_shift_trials;
    opt.stagnation_iters = _algorithm.stagnation_iters;
    opt.crawl_limit = _algorithm.crawl_limit;
    opt.crawl_frac = _algorithm.crawl_frac;

    // the guess vectors of the problem are working storage of the mesh loop below; the caller's own guess
    // (ePSOPT.cpp:47-56) is put back when solve() returns, so that a second solve() starts from it again
    struct KeepGuess {
        mi355x::Prob& P;
        std::vector<double> gs, gc, lf, lc;
        explicit KeepGuess(mi355x::Prob& p) : P(p), gs(p.guess_states), gc(p.guess_controls), lf(p.guess_lamF), lc(p.guess_lamC) {}
        ~KeepGuess() { P.guess_states = gs; P.guess_controls = gc; P.guess_lamF = lf; P.guess_lamC = lc; }
    } keep_guess(P);

    mi355x::NlpResult r;
    auto solve_current_mesh = [&](const mi355x::NlpOptions& o) {
        mi355x::NlpProblem nlp = mi355x::make_nlp(P, _dev.get());
        // Newton-step linear algebra: small KKT systems on the host, the rest on the device
        const size_t kkt_rows = (2 * ns + nc) * P.nodes;
        const bool dev_kkt = !P.lifted && (_algorithm.linear_solver == "device" ||
                                           (_algorithm.linear_solver == "auto" && kkt_rows > 400));
        BatchedKkt shared(dev_kkt ? _algorithm.kkt_batcher.get() : nullptr, static_cast<mi355x::KktBackend*>(_dev.get()), _dev->ctx, _dev->nodes);
        nlp.kkt = dev_kkt ? (_algorithm.kkt_batcher ? static_cast<mi355x::KktBackend*>(&shared) : static_cast<mi355x::KktBackend*>(_dev.get()))
                          : nullptr;
        if (_algorithm.scaling == "automatic") nlp.vscale = mi355x::bound_scales(P);
        else if (_algorithm.scaling != "none") die("Alg::scaling must be \"automatic\" or \"none\"");
        if (_algorithm.defect_scaling == "jacobian-based") nlp.jacobian_defect_scaling = true;
        else if (_algorithm.defect_scaling != "state-based") die("Alg::defect_scaling must be \"state-based\" or \"jacobian-based\"");
        _solution.linear_solver = dev_kkt ? "device: structured KKT factorisation (Schur complement + Cholesky), Woodbury-corrected" : "host LDL^T";
        if (P.guess_lamF.size() == ns * P.nodes) nlp.lamF0 = P.guess_lamF;
        if (P.guess_lamC.size() == P.npath * P.nodes) nlp.lamC0 = P.guess_lamC;
        mi355x::NlpOptions ob = o;
        if (_algorithm.nlp_iter_budget > 0) {           // what is left of the solve()'s iteration budget
            const int left = _algorithm.nlp_iter_budget - _solution.nlp_iterations_total;
            if (left <= 0) {
                r = mi355x::NlpResult();
                r.msg = "iteration budget exhausted (" + std::to_string(_solution.nlp_iterations_total) + " iterations over all meshes and restarts)";
                return;
            }
            ob.max_iter = std::min(ob.max_iter, left);
        }
        const auto t_run = std::chrono::steady_clock::now();
        r = mi355x::solve_nlp(nlp, ob, mi355x::initial_guess(P));
        _solution.nlp_iterations_total += r.iterations;
        _solution.nlp_runs.push_back({P.nodes, r.iterations, r.ok, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_run).count(),
                                      r.t_eval, r.t_hess, r.t_factor, r.t_solve, r.t_lowrank, r.t_blocks, r.t_jt, r.t_matvec, r.n_factor, r.n_solve});
        if (!r.ok && _algorithm.nlp_iter_budget > 0 && _solution.nlp_iterations_total >= _algorithm.nlp_iter_budget)
            r.msg = "iteration budget exhausted (" + std::to_string(_solution.nlp_iterations_total) + " iterations over all meshes and restarts); last: " + r.msg;
    };
    // A warm start that wanders (no failure, just hundreds of regularised steps along a flat valley; Monte-Carlo scenario 10 of
    // profiles/r02_notes.md section 12) is cut off after Alg::warm_patience iterations and repeated from the same interpolated
    // guess with a 10 x larger barrier parameter
    auto solve_warm = [&](const mi355x::NlpOptions& o) {
        if (_algorithm.warm_patience <= 0 || _algorithm.warm_patience >= o.max_iter) { solve_current_mesh(o); return; }
        mi355x::NlpOptions first = o;
        first.max_iter = _algorithm.warm_patience;
        solve_current_mesh(first);
        if (r.ok || r.iterations < first.max_iter) return;                  // converged, or ended for another reason
        if (_algorithm.print_level >= 5)
            printf("warm start still running after %d iterations: again from the same guess with barrier parameter %.1e\n",
                   first.max_iter, 10.0 * o.mu_init);
        mi355x::NlpOptions again = o;
        again.mu_init = 10.0 * o.mu_init;
        solve_current_mesh(again);
    };
    // A cold start that ends locally infeasible (the path rows stay violated whatever the penalty weight: the iterate
    // sits on the wrong side of a keep-out) is repeated from the straight line bent to either side, by 15 % and 35 %
    // of its length.  IPOPT's restoration phase does this job for ePSOPT; here it is a search over homotopy classes,
    // decided by the first start that converges.  Only for the default guess: a user's guess is taken as given.
    auto solve_cold_with_retries = [&](const mi355x::NlpOptions& o) {
        solve_current_mesh(o);
        if (r.ok || P.npath == 0 || !P.guess_states.empty() || _algorithm.guess_retries <= 0) return;
        const bool outer_planned = P.guess_planned;          // (a ladder climb that starts from the planned route: the bends follow without it)
        P.guess_planned = false;
        double span = 0;
        if (P.event_lower.size() == 2 * ns) {
            const double dx = 0.5 * (P.event_lower[ns + P.px] + P.event_upper[ns + P.px]) - 0.5 * (P.event_lower[P.px] + P.event_upper[P.px]);
            const double dy = 0.5 * (P.event_lower[ns + P.py] + P.event_upper[ns + P.py]) - 0.5 * (P.event_lower[P.py] + P.event_upper[P.py]);
            span = std::sqrt(dx * dx + dy * dy);
        }
        const double bends[4] = {0.15, -0.15, 0.35, -0.35};
        for (int t = 0; t < 4 && t < _algorithm.guess_retries && !r.ok && span > 0; ++t) {
            P.guess_bend = bends[t] * span;
            if (_algorithm.print_level >= 5) printf("cold start failed (%s): retrying from the line bent by %+.3f\n", r.msg.c_str(), P.guess_bend);
            solve_current_mesh(o);
        }
        P.guess_bend = 0;
        // ... and last from a route planned through the free space of the static keep-outs (planned_path_guess), unless this cold start
        // already began from it (a ladder climb's second start, Alg::plan_second_start).
        if (!r.ok && span > 0 && _algorithm.guess_retries > 0 && !outer_planned) {
            P.guess_planned = true;
            if (_algorithm.print_level >= 5) printf("cold start failed (%s): retrying from a path planned through the free space\n", r.msg.c_str());
            solve_current_mesh(o);
            P.guess_planned = false;
        }
        P.guess_planned = outer_planned;
    };
    // Multipliers are NOT carried to the next mesh by default: measured over 32 Monte-Carlo scenarios at 257 nodes the
    // costate-mapped warm start needed 112 iterations on average against 103 from zero multipliers (interior-point
    // warm starts want centred pairs, which interpolated multipliers are not).  Alg::warm_multipliers enables it.
    const bool warm_multipliers = _algorithm.warm_multipliers;
    // the solution on the current mesh, interpolated to Mnew LGL nodes, becomes the guess there
    auto remesh_with_guess = [&](size_t Mnew) {
        const std::vector<double> tau = P.tau, w = P.w;
        const size_t M = P.nodes;
        setMesh(Mnew);
        configureDevice(_dev.get());
        P.guess_states.assign(ns * Mnew, 0.0);
        P.guess_controls.assign(nc * Mnew, 0.0);
        for (size_t i = 0; i < ns; ++i) interp_lgl(tau, w, &r.z[i * M], M, P.tau, &P.guess_states[i * Mnew]);
        // the interpolant may cut through a keep-out between two old nodes
        mi355x::repair_guess(P, &P.guess_states[P.px * Mnew], &P.guess_states[P.py * Mnew]);
        for (size_t j = 0; j < nc; ++j) {
            interp_lgl(tau, w, &r.z[(ns + j) * M], M, P.tau, &P.guess_controls[j * Mnew]);
            for (size_t k = 0; k < Mnew; ++k)
                P.guess_controls[j * Mnew + k] =
                    std::min(std::max(P.guess_controls[j * Mnew + k], P.control_lower[j]), P.control_upper[j]);
        }
        // multipliers: lambda_k / w_k samples the costate (covector mapping of pseudospectral methods), which is what
        // interpolates between meshes; the row multipliers of the keep-outs likewise
        P.guess_lamF.clear();
        P.guess_lamC.clear();
        if (warm_multipliers && r.lamF.size() == ns * M && r.lamC.size() == P.npath * M) {
            std::vector<double> tmp(M);
            P.guess_lamF.assign(ns * Mnew, 0.0);
            for (size_t i = 0; i < ns; ++i) {
                for (size_t k = 0; k < M; ++k) tmp[k] = r.lamF[i * M + k] / w[k];
                interp_lgl(tau, w, tmp.data(), M, P.tau, &P.guess_lamF[i * Mnew]);
                for (size_t k = 0; k < Mnew; ++k) P.guess_lamF[i * Mnew + k] *= P.w[k];
            }
            P.guess_lamC.assign(P.npath * Mnew, 0.0);
            for (size_t j = 0; j < P.npath; ++j) {
                for (size_t k = 0; k < M; ++k) tmp[k] = r.lamC[j * M + k] / w[k];
                interp_lgl(tau, w, tmp.data(), M, P.tau, &P.guess_lamC[j * Mnew]);
                for (size_t k = 0; k < Mnew; ++k) P.guess_lamC[j * Mnew + k] *= P.w[k];
            }
        }
    };
    mi355x::NlpOptions warm = opt;      // started from an interpolated solution: stay close to it
    warm.mu_init = _algorithm.warm_mu_init;               // (1e-3 .. 1e-5 measured, profiles/r01_notes.md)
    warm.bound_push = warm.bound_frac = _algorithm.warm_bound_push;
    warm.mu_restart = _algorithm.mu_restart;
    _solution.mesh_iterations = 0;
    _solution.nlp_iterations_total = 0;
    _solution.nlp_runs.clear();
    _solution.ode_error = 0;
    bool sequenced = false;             // the requested mesh is started from the sequencing ladder's solution
    std::function<bool(double, bool)> climb;  // the ladder from its coarsest mesh with the straight-line guess bent by so much: true if every rung converged
    double ladder_span = 0;
    int ladder_next_start = 1;          // index of the first start (below) no climb has used yet
    bool ladder_has_plan = false;       // Alg::plan_second_start, and the static keep-outs leave a route that planned_path_guess finds
    bool ladder_first_rung_failed = false;
    const double ladder_bends[4] = {0.15, -0.15, 0.35, -0.35};
    // start k of a climb: 0 the straight line, then (Alg::plan_second_start, if there is one) the planned route, then the bends
    // (Alg::plan_first_start, an experiment: the planned route FIRST and the straight line second)
    auto ladder_start_planned = [&](int k) { return ladder_has_plan && k == (_algorithm.plan_first_start ? 0 : 1); };
    auto ladder_start_bend = [&](int k) { const int b = k - 1 - (ladder_has_plan ? 1 : 0); return (b >= 0 && b < 4) ? ladder_bends[b] : 0.0; };
    auto ladder_starts = [&]() { return 1 + (ladder_has_plan ? 1 : 0) + 4; };

    // Mesh sequencing: a fine global mesh is reached through coarse ones (33, 65, 129, ... nodes), each
    // solve started from the interpolated previous solution.  An interior-point iteration from a cold
    // straight-line guess needs hundreds of Newton steps on a 1000-node mesh; from the interpolant of the
    // next-coarser solution it needs a few dozen, and the coarse solves cost next to nothing.
    // (PSOPT's own remedy is the same idea driven by the error estimate: start coarse, refine.)
    // (function scope: the ladder is climbed again, from another guess, if the warm start on the requested mesh fails)
    const size_t target = P.nodes;
    std::vector<size_t> ladder;
    const std::vector<double> true_records = P.path_records;
    double span = 0;
    if (_algorithm.mesh_sequencing && P.nodes > 80 && P.guess_states.empty() && !P.lifted) {
        const size_t lr = (size_t)std::max(2, _algorithm.ladder_ratio);
        for (size_t m = 33; m < target; m = lr * m - (lr - 1)) ladder.push_back(m);
        // Constraints hold at the nodes only, so a coarse mesh can step over a thin keep-out ("tunnelling") and leave
        // the finer meshes a start on the wrong side of it.  On the ladder the keep-outs of the record table are
        // therefore inflated by half the largest node spacing of the straight line between the boundary positions
        // (LGL nodes are (pi/2) / (m-1) of the span apart at mid-horizon); the requested mesh gets the true sizes back.
        if (P.event_lower.size() == 2 * ns) {
            const double dx = 0.5 * (P.event_lower[ns + P.px] + P.event_upper[ns + P.px]) - 0.5 * (P.event_lower[P.px] + P.event_upper[P.px]);
            const double dy = 0.5 * (P.event_lower[ns + P.py] + P.event_upper[ns + P.py]) - 0.5 * (P.event_lower[P.py] + P.event_upper[P.py]);
            span = std::sqrt(dx * dx + dy * dy);
        }
        auto inflate_records = [&, this](size_t m) {
            P.path_records = true_records;
            if (!_algorithm.inflate_keepouts || !(span > 0)) return;
            const double delta = 0.5 * span * 1.5707963267948966 / (double)(m - 1);
            for (size_t j = 0; j < P.npath - P.npath_traced; ++j) {
                double* rec = &P.path_records[j * EMI_PATH_REC];
                const int kind = (int)rec[0];
                auto grow = [&](double& sq) { const double r0 = std::sqrt(std::max(sq, 0.0)) + delta; sq = r0 * r0; };
                if (kind == EMI_PATH_ELLIPSE) { grow(rec[5]); grow(rec[6]); }
                else if (kind == EMI_PATH_DISC) grow(rec[3]);
                else grow(rec[2]);
            }
        };
        // A rung that fails after the coarsest one converged (the interpolant runs into a corner the coarse mesh did not
        // see) sends the whole ladder back to its start with the straight-line guess bent to one side: a cold start on
        // 33 nodes costs a quarter of a second, whereas the fallback below -- cold starts on the requested mesh -- spent
        // 4 x 400 iterations of a 513-node problem on one Monte-Carlo scenario (36 of its 41 s, profiles/r02_notes.md).
        // Alg::plan_second_start: the route planned through the free space as the second start of a climb, before the bends.  (The SHORTEST
        // route, which hugs the keep-outs, made things worse in that place: 138.7 against 129.7 iterations per scenario on the 64-scenario
        // set, scenario 960 lost; charged for narrow places it takes that set to 127.0, the 256-scenario set from 135.2 to 125.5, scenario
        // 960 from 870 to 235 iterations and scenario 17 from 685 to 313: profiles/r04_notes.md section 24.)
        if (_algorithm.plan_second_start) {
            std::vector<double> px(P.nodes), py(P.nodes);
            ladder_has_plan = P.npath > 0 && span > 0 && _algorithm.guess_retries > 0 && mi355x::planned_path_guess(P, px.data(), py.data(), _algorithm.plan_clearance);
        }
        const int ladder_tries = (P.npath > 0 && span > 0) ? 1 + (ladder_has_plan ? 1 : 0) + std::min(4, std::max(0, _algorithm.guess_retries)) : 1;
        ladder_span = ladder_tries > 1 ? span : 0.0;
        climb = [&, this, inflate_records](double bend, bool planned) -> bool {
            P.guess_states.clear();
            P.guess_controls.clear();
            P.guess_lamF.clear();
            P.guess_lamC.clear();
            warm.rho_init = opt.rho_init;
            bool ok = true;
            for (size_t li = 0; li < ladder.size() && ok; ++li) {
                inflate_records(ladder[li]);
                if (li == 0) setMesh(ladder[0]);
                configureDevice(_dev.get());
                mi355x::NlpOptions o = li == 0 ? opt : warm;
                o.tol = std::max(opt.tol, _algorithm.rung_tolerance);          // intermediate meshes only feed the next guess
                if (li > 0 && _algorithm.rung_patience > 0) o.max_iter = std::min(o.max_iter, _algorithm.rung_patience);
                if (li == 0) {
                    P.guess_bend = bend;
                    P.guess_planned = planned;
                    solve_cold_with_retries(o);
                    P.guess_bend = 0;
                    P.guess_planned = false;
                } else {
                    solve_warm(o);
                }
                ++_solution.mesh_iterations;
                if (_algorithm.print_level >= 5)
                    printf("mesh sequencing: %zu nodes, %d iterations, cost %.10e (%s)\n", P.nodes, r.iterations, r.cost,
                           r.msg.c_str());
                ok = r.ok;
                warm.rho_init = std::max(warm.rho_init, r.rho);     // a penalty weight found too small stays raised
                if (ok) {
                    if (li + 1 == ladder.size()) P.path_records = true_records;       // the guess repair below sees the true sizes
                    remesh_with_guess(li + 1 < ladder.size() ? ladder[li + 1] : target);
                } else if (li == 0) {
                    ladder_first_rung_failed = true;                   // the cold start already went through its own bends
                }
            }
            P.path_records = true_records;
            return ok;
        };
        bool chain_ok = false;
        for (int ca = 0; ca < ladder_tries && !chain_ok && !ladder_first_rung_failed; ++ca) {
            const bool planned = ladder_start_planned(ca);
            const double bend = ladder_start_bend(ca) * span;
            if (ca > 0 && _algorithm.print_level >= 5) {
                if (planned) printf("mesh sequencing: a rung failed (%s), ladder restarted from the route planned through the free space\n", r.msg.c_str());
                else printf("mesh sequencing: a rung failed (%s), ladder restarted from the line bent by %+.3f\n", r.msg.c_str(), bend);
            }
            chain_ok = climb(bend, planned);
            ladder_next_start = ca + 1;             // starts 0 .. ca have been used (the climb is deterministic: none is worth repeating)
        }
        P.path_records = true_records;
        if (chain_ok) {
            sequenced = true;
        } else {                                       // fall back to the cold start on the requested mesh
            setMesh(target);
            configureDevice(_dev.get());
            P.guess_states.clear();
            P.guess_controls.clear();
            P.guess_lamF.clear();
            P.guess_lamC.clear();
        }
    }

    // PSOPT's mesh refinement ("automatic", ePSOPT.cpp:69-71): solve, estimate the ODE error,
    // add nodes and re-solve from the interpolated solution until the tolerance is met.
    const bool refine = _algorithm.mesh_refinement == "automatic" && !P.lifted;
    mi355x::NlpResult r_good;           // last converged solution and its mesh
    size_t M_good = 0;
    for (int mr = 0;; ++mr) {
        // The warm start on the requested mesh, while the ladder still has a start it has not used: at most Alg::target_patience iterations
        // (converged ones take 11 at the median, 24 at the 90th percentile, 212 at most over 256 Monte-Carlo scenarios; the one that fails used
        // all 400 of nlp_iter_max -- 20 s of a 1024-node problem -- before the next start got its turn: scenario 558, r04_notes.md section 26)
        auto target_warm = [&]() {
            mi355x::NlpOptions w = warm;
            const bool more_starts = climb && ladder_span > 0 && !ladder_first_rung_failed && ladder_next_start < ladder_starts() &&
                                     ladder_next_start - (ladder_has_plan ? 2 : 1) < _algorithm.guess_retries;
            if (_algorithm.target_patience > 0 && more_starts) w.max_iter = std::min(w.max_iter, _algorithm.target_patience);
            solve_warm(w);
        };
        if (mr == 0 && !sequenced) solve_cold_with_retries(opt);
        else if (sequenced && mr == 0) target_warm();
        else solve_current_mesh(opt);
        ++_solution.mesh_iterations;
        if (!r.ok && sequenced && mr == 0) {
            // the ladder led into a corner (typically an interpolant cutting through a keep-out the coarse meshes
            // did not see): start over on the requested mesh from the default guess
            const size_t target_nodes = P.nodes;
            // (from the first bend the ladder has not been climbed with yet: a Monte-Carlo scenario whose climb from the +15 % bend ended
            // in a failed warm start used to climb from +15 % AGAIN, to the same failure, 117 iterations later)
            for (int ca = ladder_next_start; climb && ca < ladder_starts() && ca - (ladder_has_plan ? 2 : 1) < _algorithm.guess_retries && ladder_span > 0 && !r.ok && !ladder_first_rung_failed; ++ca) {
                const bool planned = ladder_start_planned(ca);
                const double bend = ladder_start_bend(ca) * ladder_span;
                if (_algorithm.print_level >= 5) {
                    if (planned) printf("mesh sequencing: warm start on %zu nodes failed (%s), ladder restarted from the planned route\n", target_nodes, r.msg.c_str());
                    else printf("mesh sequencing: warm start on %zu nodes failed (%s), ladder restarted from the line bent by %+.3f\n", target_nodes, r.msg.c_str(), bend);
                }
                ladder_next_start = ca + 1;
                if (climb(bend, planned)) {
                    target_warm();
                    ++_solution.mesh_iterations;
                } else {
                    r.ok = false;
                }
            }
            if (r.ok) goto target_solved;
            if (P.nodes != target_nodes) { setMesh(target_nodes); configureDevice(_dev.get()); }
            if (_algorithm.print_level >= 5) printf("mesh sequencing: warm start failed (%s), cold start on %zu nodes\n", r.msg.c_str(), P.nodes);
            P.guess_states.clear();
            P.guess_controls.clear();
            P.guess_lamF.clear();
            P.guess_lamC.clear();
            solve_cold_with_retries(opt);
            ++_solution.mesh_iterations;
        }
    target_solved:
        if (!r.ok && mr > 0 && r_good.ok) {
            // a refinement solve that fails does not take the converged coarser solution with it
            if (_algorithm.print_level >= 5) printf("mesh iteration %d failed (%s): keeping the %zu-node solution\n", mr, r.msg.c_str(), M_good);
            setMesh(M_good);
            configureDevice(_dev.get());
            r = r_good;
            break;
        }
        if (!r.ok || !refine) break;
        std::vector<double> zf;
        size_t M2 = 0;
        _solution.ode_error = odeError(r.z, &zf, &M2);
        if (_algorithm.print_level >= 5)
            printf("mesh iteration %d: %zu nodes, cost %.10e, relative ODE error %.3e\n", mr, P.nodes, r.cost,
                   _solution.ode_error);
        if (_solution.ode_error <= _algorithm.ode_tolerance || mr + 1 >= _algorithm.mr_max_iterations) break;
        const size_t Mnew = std::min((size_t)_algorithm.mr_max_nodes, P.nodes + std::max<size_t>(4, (P.nodes - 1) / 2));
        if (Mnew <= P.nodes) break;
        // refined meshes start from the interpolated solution but with the cold-start barrier settings: the
        // interpolant may cut through keep-outs between the old nodes, and the elastic rows need room to move
        r_good = r;
        M_good = P.nodes;
        remesh_with_guess(Mnew);
    }
    const size_t M = P.nodes;

    _solution.error_flag = r.ok ? 0 : 1;
    _solution.error_msg = r.msg;
    _solution.nlp_iterations = r.iterations;
    _solution.evaluations = r.evaluations;
    _solution.kkt_error = r.kkt_error;
    _solution.constraint_violation = r.constr_viol;
    if (_solution.error_flag) {
        std::cout << "!!!!!Problem failed!!!!!" << std::endl << _solution.error_msg << std::endl;
        return;
    }
    _solution.cost = r.cost;
    _solution.nstates = ns;
    _solution.ncontrols = nc;
    _solution.nodes = M;
    _solution.states.assign(r.z.begin(), r.z.begin() + ns * M);
    _solution.controls.assign(r.z.begin() + ns * M, r.z.begin() + (ns + nc) * M);      // (a lifted solve carries the delayed values behind them)
    _solution.time.resize(M);
    for (size_t k = 0; k < M; ++k) _solution.time[k] = P.t0 + (P.tf - P.t0) / 2.0 * (P.tau[k] + 1.0);
    setScore(isMaximized() ? -_solution.cost : _solution.cost);
    getTraj();
}

// one (t_k, values) element per LGL node, states and controls (ePSOPT.cpp:157-182)
void eMI355X::getTraj() {
    traj_t* xt = getXtraj();
    traj_t* ut = getUtraj();
    xt->clear();
    ut->clear();
    const size_t M = _solution.nodes;
    for (size_t k = 0; k < M; ++k) {
        state_t xs, us;
        for (size_t i = 0; i < _solution.nstates; ++i) xs.push_back(_solution.states[i * M + k]);
        for (size_t j = 0; j < _solution.ncontrols; ++j) us.push_back(_solution.controls[j * M + k]);
        xt->push_back(traj_elem_t(_solution.time[k], xs));
        ut->push_back(traj_elem_t(_solution.time[k], us));
    }
}

void eMI355X::debug() { _algorithm.print_level = 5; }

// per-thread pool of idle device contexts (raw pointers on purpose: nothing is torn down behind the back of the HIP
// runtime at thread or process exit; releaseDevices() does it explicitly)
std::vector<eMI355X::Device*>& eMI355X::device_pool() {
    static thread_local std::vector<Device*> pool;
    return pool;
}

void eMI355X::releaseDevices() {
    std::vector<Device*>& pool = device_pool();
    for (Device* d : pool) delete d;
    pool.clear();
}

void eMI355X::close() {
    if (_dev && _dev->ctx && device_pool().size() < 2) device_pool().push_back(_dev.release());
    _dev.reset();
}

}  // namespace ETOL
