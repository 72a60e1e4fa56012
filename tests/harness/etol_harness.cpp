// etol_harness.cpp -- extern "C" shim that lets the Python tests drive the C++ host
// library (ETOL::TrajectoryOptimizer base + ETOL::eMI355X).  Test infrastructure.
#include <ETOL/eMI355X.hpp>

#include <cstring>
#include <sstream>
#include <string>

#include <dlfcn.h>

#include "emi_nlp.hpp"
#include "emi_transcribe.hpp"
#include "emi_trace.hpp"
#include <cmath>

namespace mx = ETOL::mi355x;

namespace {

std::string g_out, g_out2;
std::string g_linear_solver = "auto";   // mx::Alg::linear_solver for the harness solves

// a TrajectoryOptimizer that can be instantiated without a GPU (for the loader tests)
class Plain : public ETOL::TrajectoryOptimizer {
 public:
    void setup() override {}
    void solve() override {}
    void debug() override {}
    void close() override {}
    ETOL::paramset_t& params() { return _parameters; }
};

template <class V> void arr(std::ostringstream& o, const char* key, const V& v, bool last = false) {
    o << "\"" << key << "\":[";
    bool first = true;
    for (auto e : v) { o << (first ? "" : ",") << e; first = false; }
    o << "]" << (last ? "" : ",");
}

void dump_configs(ETOL::TrajectoryOptimizer& t, std::ostringstream& o) {
    o.precision(17);
    o << "{\"nsteps\":" << t.getNSteps() << ",\"dt\":" << t.getDt() << ",\"nstates\":" << t.getNStates()
      << ",\"ncontrols\":" << t.getNControls() << ",\"xrhorizon\":" << t.getXrhorizon() << ",\"urhorizon\":"
      << t.getUrhorizon() << ",\"rhorizon\":" << t.getRhorizon() << ",";
    arr(o, "xlower", t.getXlower()); arr(o, "xupper", t.getXupper()); arr(o, "x0", t.getX0());
    arr(o, "xf", t.getXf()); arr(o, "xtol", t.getXtol()); arr(o, "ulower", t.getUlower()); arr(o, "uupper", t.getUupper());
    std::vector<int> xv, uv;
    for (auto v : t.getXvartype()) xv.push_back((int)v);
    for (auto v : t.getUvartype()) uv.push_back((int)v);
    arr(o, "xvartype", xv); arr(o, "uvartype", uv);
    o << "\"nexclzones\":" << t.getNExclZones() << ",\"zones\":[";
    bool fz = true;
    for (const auto& b : *t.getObstacles_Raw()) {
        o << (fz ? "" : ",") << "["; fz = false;
        bool fc = true;
        for (const auto& c : b) { o << (fc ? "" : ",") << "[" << c[0] << "," << c[1] << "," << c[2] << "]"; fc = false; }
        o << "]";
    }
    o << "],\"tracks\":[";
    bool ft = true;
    for (const auto& tr : *t.getTracks()) {
        o << (ft ? "" : ",") << "{\"radius\":" << tr.radius << ",\"waypoints\":["; ft = false;
        bool fw = true;
        for (const auto& w : tr.trajectory) {
            o << (fw ? "" : ",") << "[" << w.first; fw = false;
            for (double d : w.second) o << "," << d;
            o << "]";
        }
        o << "]}";
    }
    o << "]}";
}

// the example-1 problem on an eMI355X, optionally without keep-outs
struct Ex1 {
    ETOL::eMI355X solver;
    ETOL::f_t obj, xdot, ydot, obs, saa;
};

}  // namespace

// ---- the NLP iteration on the CPU oracle (solver-logic tests only; never the product path) ----
namespace {
typedef int (*orc_eval_t)(int, const double*, int, int, int, const double*, const double*, const double*, double, double,
                          int, int, const double*, int, int, int, int, const double*, const double*, const double*,
                          const double*, double*, double*, double*);
typedef int (*orc_hess_t)(int, const double*, int, int, int, const double*, double, double, int, int, const double*, int,
                          int, int, int, const double*, const double*, const double*, const double*, const double*,
                          const double*, double, double*);
struct OracleEval : public mx::NlpEvaluator {
    const mx::Prob* P = nullptr;
    orc_eval_t ev = nullptr;
    orc_hess_t hs = nullptr;
    std::vector<double> vals;
    int eval(const double* X, const double* U, double* RES, double* VALS, double* COST, bool jac) override {
        return ev(P->model, P->model_params.data(), 0, (int)P->nodes, 1, P->tau.data(), P->w.data(), P->D.data(), P->t0,
                  P->tf, (int)P->npath, 1, P->path_records.data(), (int)P->px, (int)P->py, (int)P->ntracks, 1,
                  P->track_x.data(), P->track_y.data(), X, U, RES, jac ? VALS : nullptr, COST);
    }
    int hess(const double* X, const double* U, const double* lamF, const double* lamC, double sigma, double* H) override {
        return hs(P->model, P->model_params.data(), 0, (int)P->nodes, 1, P->w.data(), P->t0, P->tf, (int)P->npath, 1,
                  P->path_records.data(), (int)P->px, (int)P->py, (int)P->ntracks, 1, P->track_x.data(),
                  P->track_y.data(), X, U, lamF, lamC, sigma, H);
    }
};

// CPU stand-in for the device Newton step (emi_kkt.hip): same matrix, same unknown order, dense
// factorisation on the host.  Lets the CPU tests drive the KktBackend branch of solve_nlp.
struct HostKkt : public mx::KktBackend {
    const mx::Prob* P = nullptr;
    mx::LdltFactor F;
    std::vector<unsigned char> fixed;
    bool ok = false;
    int factor(const double* Qblk, const double* Jblk, const unsigned char* fx, double dc) override {
        const int M = (int)P->nodes, ns = (int)P->nstates, nv = ns + (int)P->ncontrols, nz = nv * M, N = nz + ns * M;
        F.n = N;
        F.a.assign((size_t)N * N, 0.0);
        fixed.assign(fx, fx + nz);
        auto put = [&](int r, int c, double v) { (r >= c ? F.a[(size_t)r * N + c] : F.a[(size_t)c * N + r]) = v; };
        for (int k = 0; k < M; ++k)
            for (int v = 0; v < nv; ++v)
                for (int q = 0; q <= v; ++q) put(v * M + k, q * M + k, Qblk[(size_t)(v * (v + 1) / 2 + q) * M + k]);
        for (int i = 0; i < ns; ++i)
            for (int k = 0; k < M; ++k) {
                const int R = nz + i * M + k;
                for (int j = 0; j < M; ++j) put(R, i * M + j, P->D[(size_t)k * M + j]);
                for (int v = 0; v < nv; ++v) put(R, v * M + k, Jblk[(size_t)(i * nv + v) * M + k]);
                put(R, R, -dc);
            }
        for (int q = 0; q < nz; ++q) {
            if (!fixed[q]) continue;
            for (int r = 0; r < N; ++r) put(r, q, 0.0);
            put(q, q, 1.0);
        }
        A0 = F.a;                 // K~ before the factorisation overwrites it
        use_exact = false;
        ok = mx::ldlt_factor(F) && F.nzero == 0;
        inertia_ok = ok && F.npos == nz && F.nneg == ns * M;
        if (ok && !inertia_ok) ++wrong_inertia;   // must never happen: Q is convexified by the caller
        return ok ? 0 : 1;
    }
    // Independent of the Woodbury route the product takes: build K = K~ - U Delta U^T densely, factorise it and
    // COUNT its inertia; the verdict must be what the r x r Cholesky test of the product says.
    int lowrank(int r, const int* node, const double* vec, const double* delta, bool* exact) override {
        use_exact = false;
        *exact = r == 0;
        if (r == 0) return 0;
        const int M = (int)P->nodes, ns = (int)P->nstates, nv = ns + (int)P->ncontrols, nz = nv * M, N = F.n;
        F2.n = N;
        F2.a = A0;
        for (int c = 0; c < r; ++c)
            for (int v = 0; v < nv; ++v)
                for (int q = 0; q <= v; ++q) {
                    const int rr = v * M + node[c], cc = q * M + node[c];
                    if (fixed[rr] || fixed[cc]) continue;
                    F2.a[(size_t)rr * N + cc] -= delta[c] * vec[(size_t)c * nv + v] * vec[(size_t)c * nv + q];
                }
        const bool ok2 = mx::ldlt_factor(F2) && F2.nzero == 0;
        use_exact = ok2 && F2.npos == nz && F2.nneg == ns * M;
        *exact = use_exact;
        ++lowrank_calls;
        if (use_exact) ++lowrank_exact;
        return 0;
    }
    int solve(double* rhs, int nrhs) override {
        if (!ok) return -1;
        for (int c = 0; c < nrhs; ++c) {
            double* b = rhs + (size_t)c * F.n;
            for (size_t q = 0; q < fixed.size(); ++q)
                if (fixed[q]) b[q] = 0.0;
            mx::ldlt_solve(use_exact ? F2 : F, b);
        }
        return 0;
    }
    mx::LdltFactor F2;
    std::vector<double> A0;
    bool use_exact = false;
    int lowrank_calls = 0, lowrank_exact = 0;
    bool inertia_ok = false;
    int wrong_inertia = 0;
};
HostKkt g_host_kkt;
}  // namespace

int g_scaling = -1;               // -1: defaults (oracle-evaluator solves unscaled, eMI355X as Alg::scaling says); 0 / 1: scaling "none" / "automatic"
extern "C" void harness_set_scaling(int on) { g_scaling = on; }
int g_defect_scaling = 0;         // 1: Alg::defect_scaling = "jacobian-based" (etol_psopt_example1.cpp:90-91) on every eMI355X the harness sets up
extern "C" void harness_set_defect_scaling(int jacobian_based) { g_defect_scaling = jacobian_based; }
extern "C" int harness_kkt_standin_wrong_inertia(void) { return g_host_kkt.wrong_inertia; }

extern "C" int harness_solve_example1_oracle(const char* xml, const char* oracle_so, int with_obstacles, double tol,
                                             int print_level, int max_iter, double* cost, int* M, double* X, double* U,
                                             int cap, int* iters) {
    void* h = dlopen(oracle_so, RTLD_NOW);
    if (!h) { g_out = dlerror(); return 3; }
    OracleEval oe;
    oe.ev = (orc_eval_t)dlsym(h, "orc_eval");
    oe.hs = (orc_hess_t)dlsym(h, "orc_hess");
    Plain t;
    t.loadConfigs(xml);
    mx::Prob P;
    P.nstates = 2; P.ncontrols = 2; P.nodes = t.getNSteps() + 1; P.t0 = 0; P.tf = t.getNSteps() * t.getDt();
    P.model = EMI_MODEL_POINTMASS2D;
    P.tau.resize(P.nodes); P.w.resize(P.nodes); P.D.resize(P.nodes * P.nodes);
    emi_lgl((int)P.nodes, P.tau.data(), P.w.data(), P.D.data());
    std::vector<double> node_t(P.nodes);
    for (size_t k = 0; k < P.nodes; ++k) node_t[k] = P.tf / 2.0 * (P.tau[k] + 1.0);
    if (with_obstacles) {
        for (const auto& poly : *t.getObstacles_Raw())
            for (auto a = poly.begin(); a != poly.end(); ++a) {
                auto b = std::next(a);
                if (b == poly.end()) b = poly.begin();
                double rec[8];
                emi_edge_ellipse((*a)[0], (*a)[1], (*b)[0], (*b)[1], rec);
                P.path_records.insert(P.path_records.end(), rec, rec + 8);
            }
        for (const auto& trk : *t.getTracks()) {
            std::vector<double> tt, xx, yy, xc(P.nodes), yc(P.nodes);
            for (const auto& wp : trk.trajectory) { tt.push_back(wp.first); xx.push_back(wp.second[0]); yy.push_back(wp.second[1]); }
            emi_track_centres((int)tt.size(), tt.data(), xx.data(), yy.data(), (int)P.nodes, node_t.data(), xc.data(), yc.data());
            P.track_x.insert(P.track_x.end(), xc.begin(), xc.end());
            P.track_y.insert(P.track_y.end(), yc.begin(), yc.end());
            double rec[8] = {(double)EMI_PATH_TRACK, (double)P.ntracks++, trk.radius * trk.radius, 0, 0, 0, 0, 0};
            P.path_records.insert(P.path_records.end(), rec, rec + 8);
        }
    }
    P.npath = P.path_records.size() / 8;
    P.state_lower = t.getXlower(); P.state_upper = t.getXupper();
    P.control_lower = t.getUlower(); P.control_upper = t.getUupper();
    for (int i = 0; i < 2; ++i) { P.event_lower.push_back(t.getX0()[i]); P.event_upper.push_back(t.getX0()[i]); }
    for (int i = 0; i < 2; ++i) { P.event_lower.push_back(t.getXf()[i] - t.getXtol()[i]); P.event_upper.push_back(t.getXf()[i] + t.getXtol()[i]); }
    P.path_lower.assign(P.npath, -1000.0); P.path_upper.assign(P.npath, 0.0);
    oe.P = &P;
    mx::NlpProblem nlp = mx::make_nlp(P, &oe);
    if (g_scaling == 1) nlp.vscale = mx::bound_scales(P);
    if (g_linear_solver == "device") {   // CPU tests: the KktBackend branch with the host stand-in
        g_host_kkt.P = &P;
        g_host_kkt.wrong_inertia = 0;
        nlp.kkt = &g_host_kkt;
    }
    mx::NlpOptions opt;
    opt.tol = tol; opt.print_level = print_level; opt.max_iter = max_iter;
    mx::NlpResult r = mx::solve_nlp(nlp, opt, mx::initial_guess(P));
    *iters = r.iterations;
    g_out = r.msg;
    if (!r.ok) { dlclose(h); return 1; }
    const int m = (int)P.nodes;
    if (m > cap) { dlclose(h); return 2; }
    *M = m; *cost = r.cost;
    for (int i = 0; i < 2 * m; ++i) { X[i] = r.z[i]; U[i] = r.z[2 * m + i]; }
    dlclose(h);
    return 0;
}

// ---- the delayed problem of harness_delay_demo, lifted (delayed values as node variables + coupling rows) and solved on the CPU
// oracle: model 3 of oracle/emi_oracle.c takes [u | x(t-dt) | x(t-2dt) | u(t-dt)] as its 8 controls, which is the lifted node
// variable list.  Solver-logic test of mi355x::NlpLink / make_nlp / solve_nlp without a GPU.  Z out: [X (2 x M) | U (2 x M) | delayed (6 x M)].
extern "C" int harness_solve_delay_demo_oracle(const char* oracle_so, int nsteps, double dt, double disc_r, double tol, int print_level,
                                               int scaling, double* cost, double* Z, int* iters) {
    void* h = dlopen(oracle_so, RTLD_NOW);
    if (!h) { g_out = dlerror(); return 3; }
    OracleEval oe;
    oe.ev = (orc_eval_t)dlsym(h, "orc_eval");
    oe.hs = (orc_hess_t)dlsym(h, "orc_hess");
    mx::Prob P;
    P.nstates = 2; P.ncontrols = 2; P.xhorizon = 3; P.uhorizon = 1; P.ndelayed = 6; P.delay_dt = dt; P.lifted = true;
    P.nodes = nsteps + 1; P.t0 = 0; P.tf = nsteps * dt;
    P.model = 3;
    P.model_params = {0.7, 0.3};
    P.tau.resize(P.nodes); P.w.resize(P.nodes); P.D.resize(P.nodes * P.nodes);
    emi_lgl((int)P.nodes, P.tau.data(), P.w.data(), P.D.data());
    if (disc_r > 0) P.path_records = {(double)EMI_PATH_DISC, 2.0, 1.5, disc_r * disc_r, 0, 0, 0, 0};
    P.npath = P.path_records.size() / 8;
    P.px = 0; P.py = 1;
    P.state_lower = {-10, -10}; P.state_upper = {10, 10}; P.control_lower = {-5, -5}; P.control_upper = {5, 5};
    P.event_lower = {1, 2, 3 - 0.01, 1 - 0.01}; P.event_upper = {1, 2, 3 + 0.01, 1 + 0.01};
    P.path_lower.assign(P.npath, -1000.0); P.path_upper.assign(P.npath, 0.0);
    oe.P = &P;
    mx::NlpProblem nlp = mx::make_nlp(P, &oe);
    if (nlp.links.size() != 6) { g_out = "make_nlp built " + std::to_string(nlp.links.size()) + " coupling rows, expected 6"; dlclose(h); return 4; }
    if (scaling) nlp.vscale = mx::bound_scales(P);
    mx::NlpOptions opt;
    opt.tol = tol; opt.print_level = print_level; opt.max_iter = 400;
    mx::NlpResult r = mx::solve_nlp(nlp, opt, mx::initial_guess(P));
    *iters = r.iterations;
    g_out = r.msg;
    if (!r.ok) { dlclose(h); return 1; }
    *cost = r.cost;
    std::copy(r.z.begin(), r.z.end(), Z);
    dlclose(h);
    return 0;
}

// ---- the planned cold-start guess (mi355x::planned_path_guess): discs (xc, yc, r) x ndisc, start / end, box, M LGL nodes -> xs, ys
extern "C" int harness_planned_path(int ndisc, const double* discs, double x0, double y0, double x1, double y1, double lo, double hi, int M,
                                    double* xs, double* ys) {
    mx::Prob P;
    P.nstates = 2; P.ncontrols = 1;
    P.nodes = (size_t)M;
    P.tau.resize(M); P.w.resize(M); P.D.resize((size_t)M * M);
    emi_lgl(M, P.tau.data(), P.w.data(), P.D.data());
    for (int i = 0; i < ndisc; ++i) {
        const double rec[EMI_PATH_REC] = {(double)EMI_PATH_DISC, discs[3 * i], discs[3 * i + 1], discs[3 * i + 2] * discs[3 * i + 2], 0, 0, 0, 0};
        P.path_records.insert(P.path_records.end(), rec, rec + EMI_PATH_REC);
    }
    P.npath = (size_t)ndisc;
    P.px = 0; P.py = 1;
    P.state_lower = {lo, lo}; P.state_upper = {hi, hi};
    P.event_lower = {x0, y0, x1, y1}; P.event_upper = {x0, y0, x1, y1};
    return mx::planned_path_guess(P, xs, ys) ? 0 : 1;
}

// ---- traced models: the quadrotor written with mi355x::Var arithmetic, as a user would -----------
namespace {
// x = (px, pz, theta, vx, vz, omega), u = (T, tau); same equations as EMI_MODEL_QUADROTOR2D
mx::Var traced_quad_rhs(const std::vector<mx::Var>& x, const std::vector<mx::Var>& u, int i) {
    const double m = 1.0, I = 0.01, g = 9.81;
    switch (i) {
        case 0: return x[3];
        case 1: return x[4];
        case 2: return x[5];
        case 3: return -(u[0] / m) * mx::sin(x[2]);
        case 4: return (u[0] / m) * mx::cos(x[2]) - g;
        default: return u[1] / I;
    }
}
mx::Var traced_quad_cost(const std::vector<mx::Var>& u) { return 1.0 * u[0] * u[0] + 1.0 * u[1] * u[1]; }
}  // namespace

// ---- delayed states / controls (ePSOPT::dae, ePSOPT.cpp:231-248): a 2-state problem with a state horizon of 3 and a control
// horizon of 1, set up through the public ETOL API; the callbacks compute with the handles as an ePSOPT user's compute with
// adoubles, reading the delayed values from the tails of x and u exactly where ePSOPT::dae appends them.  Same functions as
// model 3 of oracle/emi_oracle.c.  Returns the device's evaluation at z = [X (2 x M) | U (2 x M)].
namespace {
struct DelayDemo {           // owns the callbacks: the optimiser keeps raw f_t pointers (TrajectoryOptimizer.hpp:686-690)
    ETOL::f_t obj, f0, f1, obs;
};
void configure_delay_demo(ETOL::TrajectoryOptimizer* t, DelayDemo& d, int nsteps, double dt, int xh, int uh, int with_disc, double disc_r = 0.5) {
    t->setNSteps(nsteps); t->setDt(dt); t->setNStates(2); t->setNControls(2);
    t->setXrhorizon(xh); t->setUrhorizon(uh);
    t->setX0({1, 2}); t->setXf({3, 1}); t->setXtol({0.01, 0.01});
    t->setXlower({-10, -10}); t->setXupper({10, 10}); t->setUlower({-5, -5}); t->setUupper({5, 5});
    t->setMaximize(false);
    const double p0 = 0.7, p1 = 0.3;
    auto V = [](const std::any& a) { return std::any_cast<mx::Var>(a); };
    // x = [x0 x1 | x(t-dt) | x(t-2dt)], u = [u0 u1 | u(t-dt)]  (ePSOPT.cpp:225-248)
    d.obj = [=](F_ARGS) -> ETOL::scalar_t {
        return V(u.at(0)) * V(u.at(0)) + V(u.at(1)) * V(u.at(1)) + p1 * V(x.at(2)) * V(x.at(4)) + 0.05 * V(u.at(2)) * V(u.at(2));
    };
    d.f0 = [=](F_ARGS) -> ETOL::scalar_t { return -p0 * V(x.at(2)) + V(u.at(0)) + 0.1 * V(u.at(3)) * V(x.at(1)); };
    d.f1 = [=](F_ARGS) -> ETOL::scalar_t { return V(x.at(0)) * V(x.at(5)) - mx::sin(V(x.at(3))) + V(u.at(1)) * V(u.at(2)); };
    t->setObjective(&d.obj);
    t->setGradient({&d.f0, &d.f1});
    d.obs = [disc_r](F_ARGS) -> ETOL::scalar_t {
        return mx::disc_rows({{2.0, 1.5, disc_r}}, std::any_cast<mx::Symbol>(x.at(0)), std::any_cast<mx::Symbol>(x.at(1)));
    };
    if (with_disc) {
        t->addParams({std::pair<PARAM_PAIR>("disc_0", {ETOL::var_t::CONTINUOUS, -1000., 0., 0., nsteps * dt})});
        t->setConstraints({&d.obs});
    }
}
}  // namespace

extern "C" int harness_delay_demo(int nsteps, double dt, int xh, int uh, int with_disc, const double* z, double* res, int res_cap,
                                  double* vals, int vals_cap, double* cost, int* nres, int* nvals) {
    ETOL::eMI355X solver;
    ETOL::TrajectoryOptimizer* t = &solver;
    DelayDemo d;
    configure_delay_demo(t, d, nsteps, dt, xh, uh, with_disc);
    t->setup();
    const size_t M = nsteps + 1;
    std::vector<double> zz(z, z + 4 * M), r, v;
    solver.evaluate(zz, &r, &v, cost);
    *nres = (int)(r.size() / M);
    *nvals = (int)(v.size() / M);
    if ((int)r.size() > res_cap || (int)v.size() > vals_cap) return 2;
    std::copy(r.begin(), r.end(), res);
    std::copy(v.begin(), v.end(), vals);
    g_out = solver.getProblem()->model_source;
    t->close();
    return 0;
}

// The same problem SOLVED through the ETOL API (setup(), solve(), getXtraj()/getUtraj(), getScore()): X [2][M], U [2][M] at the LGL
// nodes.  After the solve one more evaluate() at the solution (the device forms the delayed values itself again): its defect rows go
// to defect_max -- the trajectory must satisfy the DELAYED dynamics, not the lifted problem's.
extern "C" int harness_solve_delay_demo(int nsteps, double dt, int xh, int uh, double disc_r, double tol, int print_level, double* cost,
                                        double* X, double* U, int* iters, double* defect_max) {
    ETOL::eMI355X solver;
    ETOL::TrajectoryOptimizer* t = &solver;
    DelayDemo d;
    configure_delay_demo(t, d, nsteps, dt, xh, uh, disc_r > 0, disc_r);
    t->setup();
    solver.getAlgorithm()->nlp_tolerance = tol;
    solver.getAlgorithm()->nlp_iter_max = 400;
    solver.getAlgorithm()->print_level = print_level;
    t->solve();
    g_out = solver.getSolution()->error_msg;
    if (solver.getSolution()->error_flag) { t->close(); return 1; }
    const size_t M = nsteps + 1;
    if (t->getXtraj()->size() != M || t->getUtraj()->size() != M) { t->close(); return 2; }
    std::vector<double> zz(4 * M);
    for (size_t k = 0; k < M; ++k)
        for (int i = 0; i < 2; ++i) {
            X[i * M + k] = zz[i * M + k] = t->getXtraj()->at(k).second.at(i);
            U[i * M + k] = zz[(2 + i) * M + k] = t->getUtraj()->at(k).second.at(i);
        }
    *cost = t->getScore();
    *iters = solver.getSolution()->nlp_iterations;
    std::vector<double> r, v;
    double c2 = 0;
    solver.evaluate(zz, &r, &v, &c2);
    double dm = 0;
    for (size_t q = 0; q < 2 * M; ++q) dm = std::max(dm, std::fabs(r[q]));
    *defect_max = dm;
    if (std::fabs(c2 - *cost) > 1e-9 * std::max(1.0, std::fabs(c2))) { g_out = "cost of the solve and of evaluate() at the solution differ"; t->close(); return 3; }
    t->close();
    return 0;
}

// ---- quadrotor VGP (the headline model) as an ETOL problem set up through the public API --------
namespace {
int g_traced = 0;               // 1: callbacks compute with mx::Var handles (traced model) instead of naming a built-in
double g_quad_tau_max = 1.0;   // torque bound of the quadrotor test problem (harness_set_quad_tau_max)
double g_speed_limit = 0.0;    // > 0 (traced callbacks only): rows vx^2 + vz^2 <= v^2 and T sin(theta) <= 0.6 v^2, traced
struct QuadSetup {
    std::vector<std::array<double, 3>> discs;
    ETOL::f_t obj, obs, spd;
    std::vector<ETOL::f_t> grad;
    std::vector<double> params{1.0, 0.01, 9.81, 1.0, 1.0};
};
void configure_quadrotor(ETOL::TrajectoryOptimizer* t, QuadSetup& q, int nsteps, double dt, int ndiscs) {
    t->setNSteps(nsteps); t->setDt(dt); t->setNStates(6); t->setNControls(2);
    t->setX0({1, 1, 0, 0, 0, 0}); t->setXf({8, 6, 0, 0, 0, 0}); t->setXtol({0.01, 0.01, 0.01, 0.05, 0.05, 0.05});
    t->setXlower({0, 0, -1.2, -6, -6, -4}); t->setXupper({10, 10, 1.2, 6, 6, 4});
    t->setUlower({0, -g_quad_tau_max}); t->setUupper({25, g_quad_tau_max});
    t->setMaximize(false);
    // up to 20 keep-out discs (config 3 has 20): three large ones near the straight line, the rest scattered
    const std::array<double, 3> all[20] = {{4.0, 3.2, 0.8}, {6.3, 4.4, 0.7}, {2.5, 1.2, 0.4}, {1.6, 3.4, 0.35}, {3.1, 5.2, 0.30},
                                           {5.2, 1.4, 0.35}, {7.4, 2.6, 0.30}, {8.6, 4.2, 0.25}, {5.0, 6.3, 0.35}, {2.2, 7.1, 0.30},
                                           {6.9, 7.4, 0.35}, {8.9, 7.9, 0.30}, {0.9, 5.6, 0.25}, {3.9, 8.4, 0.30}, {9.2, 1.3, 0.30},
                                           {7.0, 0.8, 0.25}, {4.6, 4.9, 0.20}, {2.9, 2.9, 0.20}, {5.6, 3.0, 0.20}, {7.6, 5.4, 0.20}};
    if (ndiscs > 20) ndiscs = 20;
    for (int i = 0; i < ndiscs; ++i) q.discs.push_back(all[i]);
    auto mp = q.params;
    const bool traced = g_traced != 0;
    // traced: the way an ePSOPT user writes callbacks (arithmetic on the solver's own scalar type,
    // reference etol_psopt_example1.cpp:101-138), here on mx::Var
    q.obj = [mp, traced](F_ARGS) -> ETOL::scalar_t {
        if (!traced) return mx::objective(EMI_MODEL_QUADROTOR2D, mp);
        std::vector<mx::Var> uu = {std::any_cast<mx::Var>(u.at(0)), std::any_cast<mx::Var>(u.at(1))};
        return traced_quad_cost(uu);
    };
    t->setObjective(&q.obj);
    q.grad.resize(6);
    std::vector<ETOL::f_t*> gp;
    for (int i = 0; i < 6; ++i) {
        q.grad[i] = [mp, i, traced](F_ARGS) -> ETOL::scalar_t {
            if (!traced) return mx::derivative(EMI_MODEL_QUADROTOR2D, i, mp);
            std::vector<mx::Var> xx, uu;
            for (const auto& a : x) xx.push_back(std::any_cast<mx::Var>(a));
            for (const auto& a : u) uu.push_back(std::any_cast<mx::Var>(a));
            return traced_quad_rhs(xx, uu, i);
        };
        gp.push_back(&q.grad[i]);
    }
    t->setGradient(gp);
    if (ndiscs > 0) {
        for (int i = 0; i < ndiscs; ++i)
            t->addParams({std::pair<PARAM_PAIR>("disc_" + std::to_string(i), {ETOL::var_t::CONTINUOUS, -1000., 0., 0., nsteps * dt})});
        auto discs = q.discs;
        q.obs = [discs](F_ARGS) -> ETOL::scalar_t {
            return mx::disc_rows(discs, std::any_cast<mx::Symbol>(x.at(0)), std::any_cast<mx::Symbol>(x.at(1)));
        };
    }
    std::vector<ETOL::f_t*> cons;
    if (ndiscs > 0) cons.push_back(&q.obs);
    if (traced && g_speed_limit > 0) {
        // rows on other variables than the keep-outs' positions, a control among them (traced, PW = 4: 2 3 4 6)
        const double v2 = g_speed_limit * g_speed_limit;
        t->addParams({std::pair<PARAM_PAIR>("speed_0", {ETOL::var_t::CONTINUOUS, -1000., 0., 0., nsteps * dt})});
        t->addParams({std::pair<PARAM_PAIR>("tilt_0", {ETOL::var_t::CONTINUOUS, -1000., 0., 0., nsteps * dt})});
        q.spd = [v2](F_ARGS) -> ETOL::scalar_t {
            const mx::Var th = std::any_cast<mx::Var>(x.at(2)), vx = std::any_cast<mx::Var>(x.at(3)), vz = std::any_cast<mx::Var>(x.at(4));
            const mx::Var T = std::any_cast<mx::Var>(u.at(0));
            return ETOL::fout_mi355x_vars_t{vx * vx + vz * vz - v2, T * mx::sin(th) - 0.6 * v2};
        };
        cons.push_back(&q.spd);
    }
    if (!cons.empty()) t->setConstraints(cons);
}
}  // namespace

extern "C" void harness_set_quad_tau_max(double v) { g_quad_tau_max = v; }
extern "C" void harness_set_traced(int on) { g_traced = on; }
extern "C" void harness_set_speed_limit(double v) { g_speed_limit = v; }
int g_refine = -1;              // harness_solve_example1: -1 = the setup() default ("automatic"), 0 = "none", 1 = "automatic"
extern "C" void harness_set_refine(int mode) { g_refine = mode; }
extern "C" void harness_set_linear_solver(const char* name) { g_linear_solver = name; }
extern "C" const char* harness_last_linear_solver(void) { return g_out2.c_str(); }

// ETOL::eMI355X::odeError (the PSOPT-style relative local error) of a given trajectory of the quadrotor problem
extern "C" int harness_ode_error_quadrotor(int nsteps, double dt, int ndiscs, const double* X, const double* U, double* err) {
    ETOL::eMI355X solver;
    QuadSetup q;
    configure_quadrotor(&solver, q, nsteps, dt, ndiscs);
    solver.setup();
    const size_t M = nsteps + 1;
    std::vector<double> z(8 * M);
    std::copy(X, X + 6 * M, z.begin());
    std::copy(U, U + 2 * M, z.begin() + 6 * M);
    *err = solver.odeError(z);
    solver.close();
    return 0;
}

// Solve the quadrotor VGP on the GPU through ETOL::eMI355X.  Outputs X[6][M], U[2][M].
extern "C" int harness_solve_quadrotor(int nsteps, double dt, int ndiscs, double tol, int print_level, int refine,
                                       double ode_tol, double* cost, int* M, double* X, double* U, int cap, int* iters,
                                       int* mesh_iters, double* ode_err) {
    ETOL::eMI355X solver;
    QuadSetup q;
    configure_quadrotor(&solver, q, nsteps, dt, ndiscs);
    solver.setup();
    solver.getAlgorithm()->nlp_tolerance = tol;
    solver.getAlgorithm()->print_level = print_level;
    solver.getAlgorithm()->nlp_iter_max = 400;
    solver.getAlgorithm()->mesh_refinement = refine ? "automatic" : "none";
    solver.getAlgorithm()->ode_tolerance = ode_tol;
    solver.getAlgorithm()->linear_solver = g_linear_solver;
    solver.getAlgorithm()->scaling = g_scaling < 0 ? solver.getAlgorithm()->scaling : (g_scaling ? "automatic" : "none");
    if (g_defect_scaling) solver.getAlgorithm()->defect_scaling = "jacobian-based";
    solver.solve();
    g_out2 = solver.getSolution()->linear_solver;
    const mx::Sol* s = solver.getSolution();
    *iters = s->nlp_iterations;
    *mesh_iters = s->mesh_iterations;
    *ode_err = s->ode_error;
    g_out = s->error_msg;
    if (s->error_flag) return 1;
    const int m = (int)s->nodes;
    if (m > cap) return 2;
    *M = m; *cost = solver.getScore();
    for (int k = 0; k < m; ++k) {
        for (int i = 0; i < 6; ++i) X[i * m + k] = (*solver.getXtraj())[k].second[i];
        for (int i = 0; i < 2; ++i) U[i * m + k] = (*solver.getUtraj())[k].second[i];
    }
    solver.close();
    return 0;
}

// Same problem, NLP iteration driven by the CPU oracle (solver-logic tests; never the product path).
extern "C" int harness_solve_quadrotor_oracle(const char* oracle_so, int nsteps, double dt, int ndiscs, double tol,
                                              int print_level, double* cost, int* M, double* X, double* U, int cap,
                                              int* iters) {
    void* h = dlopen(oracle_so, RTLD_NOW);
    if (!h) { g_out = dlerror(); return 3; }
    OracleEval oe;
    oe.ev = (orc_eval_t)dlsym(h, "orc_eval");
    oe.hs = (orc_hess_t)dlsym(h, "orc_hess");
    Plain t;
    QuadSetup q;
    configure_quadrotor(&t, q, nsteps, dt, ndiscs);
    mx::Prob P;
    P.nstates = 6; P.ncontrols = 2; P.nodes = nsteps + 1; P.t0 = 0; P.tf = nsteps * dt;
    P.model = EMI_MODEL_QUADROTOR2D; P.model_params = q.params;
    P.tau.resize(P.nodes); P.w.resize(P.nodes); P.D.resize(P.nodes * P.nodes);
    emi_lgl((int)P.nodes, P.tau.data(), P.w.data(), P.D.data());
    for (const auto& d : q.discs) {
        double rec[8] = {(double)EMI_PATH_DISC, d[0], d[1], d[2] * d[2], 0, 0, 0, 0};
        P.path_records.insert(P.path_records.end(), rec, rec + 8);
    }
    P.npath = q.discs.size();
    P.state_lower = t.getXlower(); P.state_upper = t.getXupper();
    P.control_lower = t.getUlower(); P.control_upper = t.getUupper();
    for (int i = 0; i < 6; ++i) { P.event_lower.push_back(t.getX0()[i]); P.event_upper.push_back(t.getX0()[i]); }
    for (int i = 0; i < 6; ++i) { P.event_lower.push_back(t.getXf()[i] - t.getXtol()[i]); P.event_upper.push_back(t.getXf()[i] + t.getXtol()[i]); }
    P.path_lower.assign(P.npath, -1000.0); P.path_upper.assign(P.npath, 0.0);
    oe.P = &P;
    mx::NlpProblem nlp = mx::make_nlp(P, &oe);
    if (g_scaling == 1) nlp.vscale = mx::bound_scales(P);
    if (g_linear_solver == "device") {   // CPU tests: the KktBackend branch with the host stand-in
        g_host_kkt.P = &P;
        g_host_kkt.wrong_inertia = 0;
        nlp.kkt = &g_host_kkt;
    }
    mx::NlpOptions opt;
    opt.tol = tol; opt.print_level = print_level; opt.max_iter = 400;
    mx::NlpResult r = mx::solve_nlp(nlp, opt, mx::initial_guess(P));
    *iters = r.iterations;
    g_out = r.msg;
    dlclose(h);
    if (!r.ok) return 1;
    const int m = (int)P.nodes;
    if (m > cap) return 2;
    *M = m; *cost = r.cost;
    for (int i = 0; i < 6 * m; ++i) X[i] = r.z[i];
    for (int i = 0; i < 2 * m; ++i) U[i] = r.z[6 * m + i];
    return 0;
}

// Body of FixedWing12::hess (etol_amd/csrc/emi_models.hpp), generated: the model's equations written once more on
// mx::Var with the parameter block as IN_PARAM inputs, second derivatives by the trace.  The text is pasted into
// emi_models.hpp; tests/test_trace.py checks that what is checked in there is what this generates.
extern "C" const char* harness_fixedwing_hess_body(void) {
    mx::Trace& tr = mx::Trace::active();
    tr.clear();
    auto par = [&](int i) { mx::Var v; v.kind = mx::Var::EXPR; v.node = tr.input(mx::Trace::IN_PARAM, i); return v; };
    std::vector<mx::Var> z;
    for (size_t i = 0; i < 12; ++i) z.push_back(mx::Var(mx::Var::STATE, i));
    for (size_t j = 0; j < 4; ++j) z.push_back(mx::Var(mx::Var::CONTROL, j));
    const mx::Var sph = mx::sin(z[3]), cph = mx::cos(z[3]), sth = mx::sin(z[4]), cth = mx::cos(z[4]);
    const mx::Var sps = mx::sin(z[5]), cps = mx::cos(z[5]);
    const mx::Var icth = 1.0 / cth, tth = sth * icth;
    const mx::Var u = z[6], v = z[7], w = z[8], p = z[9], qq = z[10], r = z[11];
    const mx::Var m = par(0), Ixx = par(1), Iyy = par(2), Izz = par(3), g = par(4), qS = par(5), iV = 1.0 / par(13), damp = par(14);
    std::vector<mx::Var> f(12);
    f[0] = cth * cps * u + (sph * sth * cps - cph * sps) * v + (cph * sth * cps + sph * sps) * w;
    f[1] = cth * sps * u + (sph * sth * sps + cph * cps) * v + (cph * sth * sps - sph * cps) * w;
    f[2] = -sth * u + sph * cth * v + cph * cth * w;
    f[3] = p + tth * (sph * qq + cph * r);
    f[4] = cph * qq - sph * r;
    f[5] = (sph * qq + cph * r) * icth;
    const mx::Var al = w * iV, CL = par(6) + par(7) * al, CD = par(8) + par(9) * CL * CL;
    const mx::Var X = z[12] - qS * CD, Y = -damp * v, Z = -qS * CL;
    f[6] = r * v - qq * w - g * sth + X / m;
    f[7] = p * w - r * u + g * sph * cth + Y / m;
    f[8] = qq * u - p * v + g * cph * cth + Z / m;
    const mx::Var Lm = qS * par(10) * z[13] - damp * p, Mm = qS * par(11) * z[14] - damp * qq, Nm = qS * par(12) * z[15] - damp * r;
    f[9] = ((Iyy - Izz) * qq * r + Lm) / Ixx;
    f[10] = ((Izz - Ixx) * p * r + Mm) / Iyy;
    f[11] = ((Ixx - Iyy) * p * qq + Nm) / Izz;
    const mx::Var L = par(15) * (z[12] * z[12] + z[13] * z[13] + z[14] * z[14] + z[15] * z[15]);
    std::vector<int> fn;
    for (const mx::Var& fi : f) fn.push_back(fi.node);
    g_out = tr.generate_hess_body(12, 4, fn, L.node);
    return g_out.c_str();
}

// ---- 12-state fixed wing (the config-5 model) as a guidance problem: level flight at trim, lateral offset ---------
namespace {
const std::vector<double> kFwParams = {10.0, 0.8, 1.1, 1.8, 9.81, 120.0, 0.3, 4.5, 0.03, 0.05, 0.08, -0.6, 0.06, 25.0, 0.9, 1.0};
struct FwSetup {
    ETOL::f_t obj;
    std::vector<ETOL::f_t> grad;
};
// trim of the model at 25 m/s: lift = weight fixes alpha (w = alpha V), thrust = drag
void fixedwing_trim(double* w_trim, double* theta_trim, double* thrust_trim) {
    const std::vector<double>& p = kFwParams;
    const double CL = p[0] * p[4] / p[5], alpha = (CL - p[6]) / p[7];
    *w_trim = alpha * p[13];
    *theta_trim = alpha;
    *thrust_trim = p[5] * (p[8] + p[9] * CL * CL);
}
void configure_fixedwing(ETOL::TrajectoryOptimizer* t, FwSetup& q, int nsteps, double tf, double lateral) {
    double wt, tht, thr;
    fixedwing_trim(&wt, &tht, &thr);
    const double V = kFwParams[13];
    t->setNSteps(nsteps); t->setDt(tf / nsteps); t->setNStates(12); t->setNControls(4);
    //            pn      pe       pd    phi  theta psi  ub  vb  wb  p  q  r
    t->setX0({0.0, 0.0, -100.0, 0.0, tht, 0.0, V, 0.0, wt, 0, 0, 0});
    t->setXf({V * tf, lateral, -100.0, 0.0, tht, 0.0, V, 0.0, wt, 0, 0, 0});
    t->setXtol({5.0, 0.5, 2.0, 0.05, 0.05, 0.1, 2.0, 1.0, 1.0, 0.2, 0.2, 0.2});
    t->setXlower({-50, -200, -200, -1.0, -0.6, -1.5, 10, -10, -10, -2, -2, -2});
    t->setXupper({2000, 200, -10, 1.0, 0.6, 1.5, 40, 10, 10, 2, 2, 2});
    t->setUlower({0, -0.5, -0.5, -0.5}); t->setUupper({60, 0.5, 0.5, 0.5});
    t->setMaximize(false);
    const std::vector<double> mp = kFwParams;
    q.obj = [mp](F_ARGS) -> ETOL::scalar_t { return mx::objective(EMI_MODEL_FIXEDWING12, mp); };
    t->setObjective(&q.obj);
    q.grad.resize(12);
    std::vector<ETOL::f_t*> gp;
    for (int i = 0; i < 12; ++i) {
        q.grad[i] = [mp, i](F_ARGS) -> ETOL::scalar_t { return mx::derivative(EMI_MODEL_FIXEDWING12, i, mp); };
        gp.push_back(&q.grad[i]);
    }
    t->setGradient(gp);
}
}  // namespace

// fixed-wing lateral-offset manoeuvre, NLP iteration driven by the CPU oracle (solver-logic test)
extern "C" int harness_solve_fixedwing_oracle(const char* oracle_so, int nsteps, double tf, double lateral, double tol,
                                              int print_level, double* cost, int* M, double* X, double* U, int cap, int* iters) {
    void* h = dlopen(oracle_so, RTLD_NOW);
    if (!h) { g_out = dlerror(); return 3; }
    OracleEval oe;
    oe.ev = (orc_eval_t)dlsym(h, "orc_eval");
    oe.hs = (orc_hess_t)dlsym(h, "orc_hess");
    Plain t;
    FwSetup q;
    configure_fixedwing(&t, q, nsteps, tf, lateral);
    mx::Prob P;
    P.nstates = 12; P.ncontrols = 4; P.nodes = nsteps + 1; P.t0 = 0; P.tf = tf;
    P.model = EMI_MODEL_FIXEDWING12; P.model_params = kFwParams;
    P.tau.resize(P.nodes); P.w.resize(P.nodes); P.D.resize(P.nodes * P.nodes);
    emi_lgl((int)P.nodes, P.tau.data(), P.w.data(), P.D.data());
    P.npath = 0;
    P.state_lower = t.getXlower(); P.state_upper = t.getXupper();
    P.control_lower = t.getUlower(); P.control_upper = t.getUupper();
    for (int i = 0; i < 12; ++i) { P.event_lower.push_back(t.getX0()[i]); P.event_upper.push_back(t.getX0()[i]); }
    for (int i = 0; i < 12; ++i) { P.event_lower.push_back(t.getXf()[i] - t.getXtol()[i]); P.event_upper.push_back(t.getXf()[i] + t.getXtol()[i]); }
    oe.P = &P;
    mx::NlpProblem nlp = mx::make_nlp(P, &oe);
    if (g_scaling == 1) nlp.vscale = mx::bound_scales(P);
    mx::NlpOptions opt;
    opt.tol = tol; opt.print_level = print_level; opt.max_iter = 300;
    mx::NlpResult r = mx::solve_nlp(nlp, opt, mx::initial_guess(P));
    *iters = r.iterations;
    g_out = r.msg;
    dlclose(h);
    if (!r.ok) return 1;
    const int m = (int)P.nodes;
    if (m > cap) return 2;
    *M = m; *cost = r.cost;
    for (int i = 0; i < 12 * m; ++i) X[i] = r.z[i];
    for (int i = 0; i < 4 * m; ++i) U[i] = r.z[12 * m + i];
    return 0;
}

// the same problem through ETOL::eMI355X on the GPU
extern "C" int harness_solve_fixedwing(int nsteps, double tf, double lateral, double tol, int print_level, double* cost, int* M,
                                       double* X, double* U, int cap, int* iters) {
    ETOL::eMI355X solver;
    FwSetup q;
    configure_fixedwing(&solver, q, nsteps, tf, lateral);
    solver.setup();
    solver.getAlgorithm()->nlp_tolerance = tol;
    solver.getAlgorithm()->print_level = print_level;
    solver.getAlgorithm()->nlp_iter_max = 300;
    solver.getAlgorithm()->mesh_refinement = "none";
    solver.getAlgorithm()->linear_solver = g_linear_solver;
    solver.getAlgorithm()->scaling = g_scaling < 0 ? solver.getAlgorithm()->scaling : (g_scaling ? "automatic" : "none");
    if (g_defect_scaling) solver.getAlgorithm()->defect_scaling = "jacobian-based";
    solver.solve();
    const mx::Sol* s = solver.getSolution();
    *iters = s->nlp_iterations;
    g_out = s->error_msg;
    g_out2 = s->linear_solver;
    if (s->error_flag) return 1;
    const int m = (int)s->nodes;
    if (m > cap) return 2;
    *M = m; *cost = solver.getScore();
    for (int k = 0; k < m; ++k) {
        for (int i = 0; i < 12; ++i) X[i * m + k] = (*solver.getXtraj())[k].second[i];
        for (int i = 0; i < 4; ++i) U[i * m + k] = (*solver.getUtraj())[k].second[i];
    }
    solver.close();
    return 0;
}

// Source of the generated model struct for the traced quadrotor (or, with which=1, a model that
// exercises every traced operation).  Host-only: used to check trace + derivatives + code generation.
extern "C" const char* harness_traced_model_source(int which) {
    mx::Trace& tr = mx::Trace::active();
    tr.clear();
    if (which == 0) {
        std::vector<mx::Var> x, u;
        for (size_t i = 0; i < 6; ++i) x.push_back(mx::Var(mx::Var::STATE, i));
        for (size_t j = 0; j < 2; ++j) u.push_back(mx::Var(mx::Var::CONTROL, j));
        std::vector<int> f;
        for (int i = 0; i < 6; ++i) f.push_back(traced_quad_rhs(x, u, i).node);
        g_out = tr.generate_model("TracedModel", 6, 2, f, traced_quad_cost(u).node);
    } else if (which == 4) {
        // point mass + moving-disc rows whose centres are waypoint tables interpolated at the node time:
        // the two tracks of the shipped problem (two waypoints each) and one with four waypoints
        std::vector<mx::Var> x = {mx::Var(mx::Var::STATE, 0), mx::Var(mx::Var::STATE, 1)};
        std::vector<mx::Var> u = {mx::Var(mx::Var::CONTROL, 0), mx::Var(mx::Var::CONTROL, 1)};
        const mx::Var tk(mx::Var::TIME, 0);
        struct Trk { double r; std::vector<double> t, x, y; };
        const Trk trks[3] = {{0.5, {0.0, 32.0}, {1.51, 2.00}, {2.00, 2.00}},
                             {0.5, {0.0, 32.0}, {1.00, 1.00}, {4.00, 3.00}},
                             {0.4, {0.0, 4.0, 9.0, 16.0}, {3.0, 3.5, 2.5, 4.0}, {1.0, 2.5, 3.0, 2.0}}};
        std::vector<int> rows;
        for (const Trk& k : trks) {
            const mx::Var ox = x[0] - mx::interp1(k.t, k.x, tk), oy = x[1] - mx::interp1(k.t, k.y, tk);
            rows.push_back((k.r * k.r - (ox * ox + oy * oy)).node);
        }
        std::string err;
        g_out = tr.generate_model("TracedModel", 2, 2, {u[0].node, u[1].node}, (u[0] * u[0] + u[1] * u[1]).node, rows, nullptr, &err);
        if (g_out.empty()) g_out = "ERROR: " + err;
    } else if (which == 3) {
        // rows on more than two variables, controls included: a disc keep-out on (x, z), a speed limit on (vx, vz),
        // a thrust-tilt coupling on (theta, thrust) and a row of time and one state.  PW = 6: variables 0 1 2 3 4 6
        std::vector<mx::Var> x, u;
        for (size_t i = 0; i < 6; ++i) x.push_back(mx::Var(mx::Var::STATE, i));
        for (size_t j = 0; j < 2; ++j) u.push_back(mx::Var(mx::Var::CONTROL, j));
        const mx::Var tk(mx::Var::TIME, 0);
        std::vector<int> f;
        for (int i = 0; i < 6; ++i) f.push_back(traced_quad_rhs(x, u, i).node);
        std::vector<int> rows;
        rows.push_back((0.64 - ((x[0] - 4.0) * (x[0] - 4.0) + (x[1] - 3.2) * (x[1] - 3.2))).node);
        rows.push_back((x[3] * x[3] + x[4] * x[4] - 9.0).node);
        rows.push_back((u[0] * mx::sin(x[2]) - 6.0).node);
        rows.push_back((x[1] * mx::cos(0.3 * tk) - 9.5).node);
        std::string err;
        g_out = tr.generate_model("TracedModel", 6, 2, f, traced_quad_cost(u).node, rows, nullptr, &err);
        if (g_out.empty()) g_out = "ERROR: " + err;
    } else if (which == 2) {
        // quadrotor + path rows written as arithmetic: a disc (etol_psopt_example1.cpp:243-247) and the ellipse of the
        // polygon edge (3.2,2.5)-(3.4,2.6) (:163-182, constants from emi_edge_ellipse); both act on states 0, 1
        std::vector<mx::Var> x, u;
        for (size_t i = 0; i < 6; ++i) x.push_back(mx::Var(mx::Var::STATE, i));
        for (size_t j = 0; j < 2; ++j) u.push_back(mx::Var(mx::Var::CONTROL, j));
        std::vector<int> f;
        for (int i = 0; i < 6; ++i) f.push_back(traced_quad_rhs(x, u, i).node);
        std::vector<int> rows;
        {
            const double xc = 4.0, yc = 3.2, r = 0.8;
            const mx::Var dx = x[0] - xc, dy = x[1] - yc;
            rows.push_back((r * r - (dx * dx + dy * dy)).node);
        }
        {
            double rec[EMI_PATH_REC];
            emi_edge_ellipse(3.2, 2.5, 3.4, 2.6, rec);
            const mx::Var ox = x[0] - rec[1], oy = x[1] - rec[2];
            const mx::Var delx = rec[3] * ox - rec[4] * oy, dely = rec[4] * ox + rec[3] * oy;
            rows.push_back((rec[5] * rec[6] - (rec[6] * mx::pow(delx, 2.) + rec[5] * mx::pow(dely, 2.))).node);
        }
        std::string err;
        g_out = tr.generate_model("TracedModel", 6, 2, f, traced_quad_cost(u).node, rows, nullptr, &err);
        if (g_out.empty()) g_out = "ERROR: " + err;
    } else {
        mx::Var a(mx::Var::STATE, 0), b(mx::Var::STATE, 1), c(mx::Var::CONTROL, 0), t(mx::Var::TIME, 0);
        std::vector<int> f;
        f.push_back((mx::exp(a * 0.3) / (1.0 + b * b) + mx::tan(c) * mx::sqrt(2.0 + a * a) - t).node);
        f.push_back((mx::log(3.0 + a * b + c * c) * mx::pow(2.0 + b, 1.5) - mx::cos(a - c) / mx::sin(1.0 + b * 0.1)).node);
        g_out = tr.generate_model("TracedModel", 2, 1, f, (c * c * mx::exp(-a) + b * mx::sin(a * c)).node);
    }
    return g_out.c_str();
}

extern "C" {

// parsed configuration of an XML file as JSON
const char* harness_load_configs(const char* xml) {
    Plain p;
    p.loadConfigs(xml);
    std::ostringstream o;
    dump_configs(p, o);
    g_out = o.str();
    return g_out.c_str();
}

// load -> saveConfigs -> load again; JSON of the second load
const char* harness_roundtrip_configs(const char* xml, const char* tmp_xml) {
    Plain p;
    p.loadConfigs(xml);
    p.saveConfigs(tmp_xml);
    Plain q;
    q.loadConfigs(tmp_xml);
    std::ostringstream o;
    dump_configs(q, o);
    g_out = o.str();
    return g_out.c_str();
}

// CSV writer: writes a 3-row 2-column trajectory to `path`, returns the name used
const char* harness_save_csv(const char* path) {
    ETOL::traj_t tr = {{0.0, {1.0, 2.5}}, {0.5, {1.25, -3.0}}, {1.0, {1e-7, 4.0}}};
    g_out = ETOL::TrajectoryOptimizer::save(&tr, path);
    return g_out.c_str();
}

// template helpers: linear_interpolation at n points
void harness_lin_interp(int n, const double* t, int nt, const double* tv, const double* ref, double* out) {
    ETOL::state_t a(tv, tv + nt), b(ref, ref + nt);
    for (int i = 0; i < n; ++i) out[i] = ETOL::TrajectoryOptimizer::linear_interpolation<double>(t[i], a, b);
}

// template helpers extractTraj / scaleTraj / offsetTraj on an R x (1+C) row-major table (time first);
// outputs: ext[R][1+nidx], sc[R][1+C], of[R][1+C]
void harness_traj_templates(int R, int Cc, const double* tab, int nidx, const int* idxs, int nsc, const double* scv,
                            int nof, const double* ofv, double* ext, double* sc, double* of) {
    ETOL::traj_t tr;
    for (int r = 0; r < R; ++r) tr.push_back({tab[r * (1 + Cc)], ETOL::state_t(tab + r * (1 + Cc) + 1, tab + (r + 1) * (1 + Cc))});
    std::vector<size_t> ix(idxs, idxs + nidx);
    ETOL::traj_t e = ETOL::TrajectoryOptimizer::extractTraj(tr, ix);
    ETOL::traj_t s2 = tr, o2 = tr;
    ETOL::TrajectoryOptimizer::scaleTraj(&s2, std::vector<double>(scv, scv + nsc));
    ETOL::TrajectoryOptimizer::offsetTraj(&o2, std::vector<double>(ofv, ofv + nof));
    for (int r = 0; r < R; ++r) {
        ext[r * (1 + nidx)] = e[r].first;
        for (int i = 0; i < nidx; ++i) ext[r * (1 + nidx) + 1 + i] = e[r].second[i];
        sc[r * (1 + Cc)] = s2[r].first;
        of[r * (1 + Cc)] = o2[r].first;
        for (int c = 0; c < Cc; ++c) { sc[r * (1 + Cc) + 1 + c] = s2[r].second[c]; of[r * (1 + Cc) + 1 + c] = o2[r].second[c]; }
    }
}

// dense LDL^T: factor + solve + inertia, for the unit test of the KKT factorisation
int harness_ldlt(int n, const double* A, double* b, int* inertia) {
    mx::LdltFactor F;
    F.n = n;
    F.a.assign(A, A + (size_t)n * n);
    const bool ok = mx::ldlt_factor(F);
    mx::ldlt_solve(F, b);
    inertia[0] = F.npos; inertia[1] = F.nneg; inertia[2] = F.nzero;
    return ok ? 0 : 1;
}

// Solve example 1 on the GPU.  with_obstacles=0 drops both constraint groups
// (the analytic-optimum case).  Outputs: cost, M, then X[2][M], U[2][M], t[M].
int harness_solve_example1(const char* xml, int with_obstacles, double tol, int print_level, double* cost, int* M,
                           double* X, double* U, double* T, int cap, int* iters) {
    Ex1 e;
    ETOL::TrajectoryOptimizer* t = &e.solver;
    t->loadConfigs(xml);
    t->setMaximize(false);
    if (g_traced) {
        // the reference's own callbacks (etol_psopt_example1.cpp:101-138), on mx::Var instead of adouble
        e.obj = [](F_ARGS) -> ETOL::scalar_t {
            const mx::Var u0 = std::any_cast<mx::Var>(u.at(0)), u1 = std::any_cast<mx::Var>(u.at(1));
            return u0 * u0 + u1 * u1;
        };
        e.xdot = [](F_ARGS) -> ETOL::scalar_t { return std::any_cast<mx::Var>(u.at(0)); };
        e.ydot = [](F_ARGS) -> ETOL::scalar_t { return std::any_cast<mx::Var>(u.at(1)); };
    } else {
        e.obj = [](F_ARGS) -> ETOL::scalar_t { return mx::objective(EMI_MODEL_POINTMASS2D); };
        e.xdot = [](F_ARGS) -> ETOL::scalar_t { return mx::derivative(EMI_MODEL_POINTMASS2D, 0); };
        e.ydot = [](F_ARGS) -> ETOL::scalar_t { return mx::derivative(EMI_MODEL_POINTMASS2D, 1); };
    }
    t->setObjective(&e.obj);
    t->setGradient({&e.xdot, &e.ydot});
    if (with_obstacles) {
        const double tspan = t->getDt() * t->getNSteps();
        const std::vector<ETOL::border_t>* zones = t->getObstacles_Raw();
        size_t i = 0;
        for (const auto& z : *zones) {
            for (size_t j = 0; j < z.size(); ++j)
                t->addParams({std::pair<PARAM_PAIR>("side_" + std::to_string(i) + "_" + std::to_string(j) + "_0",
                                                    {ETOL::var_t::CONTINUOUS, -1000., 0., 0., tspan})});
            ++i;
        }
        const std::list<ETOL::track_t>* tracks = t->getTracks();
        for (size_t k = 0; k < tracks->size(); ++k)
            t->addParams({std::pair<PARAM_PAIR>("ball_" + std::to_string(k) + "_0_0",
                                                {ETOL::var_t::CONTINUOUS, -1000., 0., 0., tspan})});
        const bool traced_rows = g_traced >= 2;
        e.obs = [zones, traced_rows](F_ARGS) -> ETOL::scalar_t {
            if (!traced_rows)
                return mx::ellipse_rows(*zones, std::any_cast<mx::Symbol>(x.at(0)), std::any_cast<mx::Symbol>(x.at(1)));
            // the rows of the reference's obstacle callback (etol_psopt_example1.cpp:153-190) as arithmetic on the
            // handles: per polygon edge the ellipse constants (emi_edge_ellipse follows the reference term by term),
            // then  a^2 b^2 - (b^2 delx^2 + a^2 dely^2)  with (delx, dely) the offset rotated into the edge frame
            ETOL::fout_mi355x_vars_t fout;
            const mx::Var px = std::any_cast<mx::Var>(x.at(0)), py = std::any_cast<mx::Var>(x.at(1));
            for (const auto& poly : *zones) {
                std::vector<ETOL::corner_t> corner(poly.begin(), poly.end());
                for (size_t i = 0; i < corner.size(); ++i) {
                    const ETOL::corner_t& p0 = corner[i];
                    const ETOL::corner_t& p1 = corner[(i + 1) % corner.size()];
                    double rec[EMI_PATH_REC];
                    emi_edge_ellipse(p0.at(0), p0.at(1), p1.at(0), p1.at(1), rec);
                    const double ct = rec[3], st = rec[4], asq = rec[5], bsq = rec[6];
                    const mx::Var ox = px - rec[1], oy = py - rec[2];
                    const mx::Var delx = ct * ox - st * oy, dely = st * ox + ct * oy;
                    fout.push_back(asq * bsq - (bsq * mx::pow(delx, 2.) + asq * mx::pow(dely, 2.)));
                }
            }
            return fout;
        };
        const bool traced_tracks = g_traced >= 3;
        e.saa = [tracks, traced_tracks](F_ARGS) -> ETOL::scalar_t {
            if (!traced_tracks)
                return mx::track_rows(*tracks, std::any_cast<mx::Symbol>(x.at(0)), std::any_cast<mx::Symbol>(x.at(1)));
            // moving discs as arithmetic on the handles: centre = waypoint table interpolated at the node time
            // (etol_psopt_example1.cpp:233-247), row  r^2 - ((x - xc(t))^2 + (y - yc(t))^2)
            ETOL::fout_mi355x_vars_t rows;
            const mx::Var px = std::any_cast<mx::Var>(x.at(0)), py = std::any_cast<mx::Var>(x.at(1));
            const mx::Var tk = std::any_cast<mx::Var>(k);
            for (const ETOL::track_t& trk : *tracks) {
                std::vector<double> tw, xw, yw;
                for (const ETOL::traj_elem_t& wp : trk.trajectory) {
                    tw.push_back(wp.first);
                    xw.push_back(wp.second.at(0));
                    yw.push_back(wp.second.at(1));
                }
                const mx::Var ox = px - mx::interp1(tw, xw, tk), oy = py - mx::interp1(tw, yw, tk);
                rows.push_back(trk.radius * trk.radius - (ox * ox + oy * oy));
            }
            return rows;
        };
        t->setConstraints({&e.obs, &e.saa});
    }
    t->setup();
    e.solver.getAlgorithm()->nlp_tolerance = tol;
    e.solver.getAlgorithm()->print_level = print_level;
    e.solver.getAlgorithm()->linear_solver = g_linear_solver;
    e.solver.getAlgorithm()->scaling = g_scaling < 0 ? e.solver.getAlgorithm()->scaling : (g_scaling ? "automatic" : "none");
    if (g_defect_scaling) e.solver.getAlgorithm()->defect_scaling = "jacobian-based";
    if (g_refine >= 0) e.solver.getAlgorithm()->mesh_refinement = g_refine ? "automatic" : "none";
    t->solve();
    g_out2 = e.solver.getSolution()->linear_solver;
    const mx::Sol* s = e.solver.getSolution();
    *iters = s->nlp_iterations;
    if (s->error_flag) { g_out = s->error_msg; return 1; }
    const int m = (int)s->nodes;
    if (m > cap) return 2;
    *M = m;
    *cost = t->getScore();
    // read back through the public trajectory API
    const ETOL::traj_t* xt = t->getXtraj();
    const ETOL::traj_t* ut = t->getUtraj();
    for (int k = 0; k < m; ++k) {
        T[k] = (*xt)[k].first;
        for (int i = 0; i < 2; ++i) { X[i * m + k] = (*xt)[k].second[i]; U[i * m + k] = (*ut)[k].second[i]; }
    }
    t->close();
    return 0;
}

const char* harness_last_message(void) { return g_out.c_str(); }

}  // extern "C"
