// epsopt_style.cpp -- "ePSOPT-style" CPU evaluator.  TEST INFRASTRUCTURE ONLY
// (used by tests/ as a third opinion and by bench.py's cpu_baseline leg).
//
// Parity status: as the header of emi_oracle.c states it (node callbacks pinned by reference-executed
// vectors, LGL / defect by closed forms, ePSOPT's solved trajectories unpinned).  This file is a PORT,
// not the reference binary: real ePSOPT needs PSOPT 5.0.0 + ADOL-C + IPOPT, none of which exist here.
//
// It reproduces the *shape* of the reference's per-node work so that the CPU
// number printed beside the GPU number carries the same overheads:
//   * one std::vector<std::any> for x and one for u, filled by push_back of
//     pointers to the active scalars           (src/ePSOPT/ePSOPT.cpp:222-229)
//   * a fresh params / pnames vector per state derivative and per constraint
//     group                                    (ePSOPT.cpp:254-255, 264-265)
//   * one std::function call per state derivative (:252-260), one per
//     constraint group returning a vector of active scalars (:261-270), one for
//     the integrand cost (:196-203, sign flip :212-213)
//   * callbacks any_cast their inputs          (etol_psopt_example1.cpp:104-106)
// In place of ADOL-C's adouble (tape + sparse_jac) the active scalar is a
// forward-mode vector dual carrying d/dz for the node's own (x,u): one sweep
// gives the value and the whole node Jacobian block.
// The defect D.X - h.F, the quadrature and the assembly into RES/VALS/COST use
// the layouts of include/emi355x.h.
#include <any>
#include <cmath>
#include <cstring>
#include <functional>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

constexpr int MAXV = 16;
struct Dual {
    double v = 0;
    double d[MAXV] = {0};
    int n = 0;
};
inline Dual mk(double v, int n) { Dual r; r.v = v; r.n = n; return r; }
inline Dual operator+(const Dual& a, const Dual& b) { Dual r = mk(a.v + b.v, a.n > b.n ? a.n : b.n); for (int i = 0; i < r.n; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
inline Dual operator-(const Dual& a, const Dual& b) { Dual r = mk(a.v - b.v, a.n > b.n ? a.n : b.n); for (int i = 0; i < r.n; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
inline Dual operator*(const Dual& a, const Dual& b) { Dual r = mk(a.v * b.v, a.n > b.n ? a.n : b.n); for (int i = 0; i < r.n; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
inline Dual operator/(const Dual& a, const Dual& b) { Dual r = mk(a.v / b.v, a.n > b.n ? a.n : b.n); for (int i = 0; i < r.n; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v; return r; }
inline Dual operator-(const Dual& a) { Dual r = mk(-a.v, a.n); for (int i = 0; i < r.n; ++i) r.d[i] = -a.d[i]; return r; }
inline Dual operator*(double s, const Dual& a) { Dual r = mk(s * a.v, a.n); for (int i = 0; i < r.n; ++i) r.d[i] = s * a.d[i]; return r; }
inline Dual operator*(const Dual& a, double s) { return s * a; }
inline Dual operator/(const Dual& a, double s) { return (1.0 / s) * a; }
inline Dual operator+(const Dual& a, double s) { Dual r = a; r.v += s; return r; }
inline Dual operator-(const Dual& a, double s) { Dual r = a; r.v -= s; return r; }
inline Dual operator+(double s, const Dual& a) { return a + s; }
inline Dual operator-(double s, const Dual& a) { return (-a) + s; }
inline Dual sin(const Dual& a) { Dual r = mk(std::sin(a.v), a.n); const double c = std::cos(a.v); for (int i = 0; i < r.n; ++i) r.d[i] = c * a.d[i]; return r; }
inline Dual cos(const Dual& a) { Dual r = mk(std::cos(a.v), a.n); const double s = -std::sin(a.v); for (int i = 0; i < r.n; ++i) r.d[i] = s * a.d[i]; return r; }
inline Dual pow2(const Dual& a) { return a * a; }

using scalar_t = std::any;
using vector_t = std::vector<scalar_t>;
using fout_t = std::vector<Dual>;
using f_t = std::function<scalar_t(vector_t x, vector_t u, vector_t params,
                                   std::vector<std::string> pnames, std::any k, std::any dt)>;

struct Problem {
    int model, ns, nc, np;
    const double* p;
    f_t objective;
    std::vector<f_t> gradient;
    std::vector<f_t> constraints;
};

// per-thread "globals" the constraint closures read, as the reference example
// reads file-scope globals (etol_psopt_example1.cpp:23-24)
struct NodeCtx {
    const double* rec = nullptr;
    int np = 0, px = 0, py = 1;
    const double* txc = nullptr;
    const double* tyc = nullptr;  // per row, this node
};
thread_local NodeCtx g_ctx;

inline const Dual& X(const vector_t& x, int i) { return *std::any_cast<Dual*>(x.at(i)); }

void build(Problem& P) {
    const double* p = P.p;
    if (P.model == 0) {
        P.objective = [](vector_t x, vector_t u, vector_t, std::vector<std::string>, std::any, std::any) -> scalar_t {
            Dual u0 = X(u, 0), u1 = X(u, 1);
            return u0 * u0 + u1 * u1;
        };
        for (int i = 0; i < 2; ++i)
            P.gradient.push_back([i](vector_t, vector_t u, vector_t, std::vector<std::string>, std::any, std::any) -> scalar_t {
                return X(u, i);
            });
    } else if (P.model == 1) {
        P.objective = [p](vector_t, vector_t u, vector_t, std::vector<std::string>, std::any, std::any) -> scalar_t {
            Dual T = X(u, 0), tq = X(u, 1);
            return p[3] * T * T + p[4] * tq * tq;
        };
        for (int i = 0; i < 3; ++i)
            P.gradient.push_back([i](vector_t x, vector_t, vector_t, std::vector<std::string>, std::any, std::any) -> scalar_t {
                return X(x, 3 + i);
            });
        P.gradient.push_back([p](vector_t x, vector_t u, vector_t, std::vector<std::string>, std::any, std::any) -> scalar_t {
            return -(X(u, 0) / p[0]) * sin(X(x, 2));
        });
        P.gradient.push_back([p](vector_t x, vector_t u, vector_t, std::vector<std::string>, std::any, std::any) -> scalar_t {
            return (X(u, 0) / p[0]) * cos(X(x, 2)) - p[2];
        });
        P.gradient.push_back([p](vector_t, vector_t u, vector_t, std::vector<std::string>, std::any, std::any) -> scalar_t {
            return X(u, 1) / p[1];
        });
    } else {
        P.objective = [p](vector_t, vector_t u, vector_t, std::vector<std::string>, std::any, std::any) -> scalar_t {
            Dual s = X(u, 0) * X(u, 0) + X(u, 1) * X(u, 1) + X(u, 2) * X(u, 2) + X(u, 3) * X(u, 3);
            return p[15] * s;
        };
        for (int i = 0; i < 12; ++i)
            P.gradient.push_back([p, i](vector_t x, vector_t u, vector_t, std::vector<std::string>, std::any, std::any) -> scalar_t {
                const double m = p[0], Ixx = p[1], Iyy = p[2], Izz = p[3], g = p[4], qS = p[5];
                const Dual ph = X(x, 3), th = X(x, 4), ps = X(x, 5), ub = X(x, 6), vb = X(x, 7), wb = X(x, 8);
                const Dual pr = X(x, 9), qr = X(x, 10), rr = X(x, 11);
                const Dual sph = sin(ph), cph = cos(ph), sth = sin(th), cth = cos(th), sps = sin(ps), cps = cos(ps);
                switch (i) {
                    case 0: return cth * cps * ub + (sph * sth * cps - cph * sps) * vb + (cph * sth * cps + sph * sps) * wb;
                    case 1: return cth * sps * ub + (sph * sth * sps + cph * cps) * vb + (cph * sth * sps - sph * cps) * wb;
                    case 2: return -sth * ub + sph * cth * vb + cph * cth * wb;
                    case 3: return pr + (sth / cth) * (sph * qr + cph * rr);
                    case 4: return cph * qr - sph * rr;
                    case 5: return (sph * qr + cph * rr) / cth;
                    default: break;
                }
                const Dual al = wb / p[13];
                const Dual CL = p[6] + p[7] * al;
                const Dual CD = p[8] + p[9] * CL * CL;
                switch (i) {
                    case 6: return rr * vb - qr * wb - g * sth + (X(u, 0) - qS * CD) / m;
                    case 7: return pr * wb - rr * ub + g * sph * cth + (-p[14] * vb) / m;
                    case 8: return qr * ub - pr * vb + g * cph * cth + (-qS * CL) / m;
                    case 9: return ((Iyy - Izz) * qr * rr + (qS * p[10] * X(u, 1) - p[14] * pr)) / Ixx;
                    case 10: return ((Izz - Ixx) * pr * rr + (qS * p[11] * X(u, 2) - p[14] * qr)) / Iyy;
                    default: return ((Ixx - Iyy) * pr * qr + (qS * p[12] * X(u, 3) - p[14] * rr)) / Izz;
                }
            });
    }
    if (P.np > 0) {
        // one closure for every keep-out row, as obs / saa in the example
        P.constraints.push_back([](vector_t x, vector_t, vector_t, std::vector<std::string>, std::any, std::any) -> scalar_t {
            fout_t fout;
            const NodeCtx& c = g_ctx;
            const Dual xk = X(x, c.px), yk = X(x, c.py);
            for (int j = 0; j < c.np; ++j) {
                const double* r = c.rec + j * 8;
                const int kind = (int)r[0];
                if (kind == 0) {
                    Dual dx = xk - r[1], dy = yk - r[2];
                    Dual delx = r[3] * dx - r[4] * dy;
                    Dual dely = r[4] * dx + r[3] * dy;
                    fout.push_back(r[5] * r[6] - (r[6] * pow2(delx) + r[5] * pow2(dely)));
                } else {
                    const double xc = kind == 1 ? r[1] : c.txc[j], yc = kind == 1 ? r[2] : c.tyc[j];
                    const double rsq = kind == 1 ? r[3] : r[2];
                    Dual dx = xk - xc, dy = yk - yc;
                    Dual dist = pow2(dx) + pow2(dy);
                    fout.push_back(dist * (-1.) + rsq);
                }
            }
            return fout;
        });
    }
}

}  // namespace

extern "C" int eps_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

// Same contract as orc_eval (emi_oracle.c); nthreads <= 0 means all cores.
extern "C" int eps_eval(int model, const double* params, int maximize, int M, int B, const double* tau,
                        const double* w, const double* D, double t0, double tf, int np, int path_sets,
                        const double* recs, int px, int py, int ntracks, int track_sets,
                        const double* trkx, const double* trky, const double* Xa, const double* Ua,
                        double* RES, double* VALS, double* COST, int nthreads) {
    Problem P;
    P.model = model;
    P.p = params;
    P.np = np;
    if (model == 0) { P.ns = 2; P.nc = 2; }
    else if (model == 1) { P.ns = 6; P.nc = 2; }
    else if (model == 2) { P.ns = 12; P.nc = 4; }
    else return 1;
    build(P);
    const int ns = P.ns, nc = P.nc, nv = ns + nc, nres = ns + np, nvals = ns * nv + 2 * np + nv;
    const double h = (tf - t0) / 2.0, sgn = maximize ? -1.0 : 1.0;
    (void)tau;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static)
#endif
    for (int b = 0; b < B; ++b) {
        const double* Xb = Xa + (size_t)b * ns * M;
        const double* Ub = Ua + (size_t)b * nc * M;
        double* Rb = RES + (size_t)b * nres * M;
        double* Vb = VALS + (size_t)b * nvals * M;
        std::vector<double> txc(np + 1), tyc(np + 1);
        NodeCtx& ctx = g_ctx;
        ctx.rec = np ? recs + (size_t)(path_sets > 1 ? b : 0) * np * 8 : nullptr;
        ctx.np = np; ctx.px = px; ctx.py = py; ctx.txc = txc.data(); ctx.tyc = tyc.data();
        double cost = 0;
        for (int k = 0; k < M; ++k) {
            Dual states[12], controls[4], tnode = mk(t0 + h * (tau[k] + 1.0), nv);
            for (int i = 0; i < ns; ++i) { states[i] = mk(Xb[(size_t)i * M + k], nv); states[i].d[i] = 1; }
            for (int c = 0; c < nc; ++c) { controls[c] = mk(Ub[(size_t)c * M + k], nv); controls[c].d[ns + c] = 1; }
            for (int j = 0; j < np; ++j)
                if ((int)ctx.rec[j * 8] == 2) {
                    size_t off = ((size_t)(track_sets > 1 ? b : 0) * ntracks + (int)ctx.rec[j * 8 + 1]) * M + k;
                    txc[j] = trkx[off]; tyc[j] = trky[off];
                }
            // --- dae (ePSOPT.cpp:218-276)
            vector_t x, u;
            for (int i = 0; i < ns; ++i) x.push_back(&states[i]);
            for (int c = 0; c < nc; ++c) u.push_back(&controls[c]);
            Dual* tval = &tnode;
            for (int i = 0; i < ns; ++i) {
                vector_t params_v = {std::string()};
                std::vector<std::string> pnames = {std::string("")};
                scalar_t fv = P.gradient.at(i)(x, u, params_v, pnames, tval, (tf - t0) / (M - 1));
                Dual out = std::any_cast<Dual>(fv);
                Rb[(size_t)i * M + k] = -h * out.v;
                for (int v = 0; v < nv; ++v) Vb[(size_t)(i * nv + v) * M + k] = -h * out.d[v];
            }
            size_t j = 0;
            for (size_t ci = 0; ci < P.constraints.size(); ++ci) {
                vector_t params_v = {std::string()};
                std::vector<std::string> pnames = {std::string("")};
                scalar_t pv = P.constraints.at(ci)(x, u, params_v, pnames, tval, (tf - t0) / (M - 1));
                fout_t out = std::any_cast<fout_t>(pv);
                for (const Dual& val : out) {
                    Rb[(size_t)(ns + j) * M + k] = val.v;
                    Vb[(size_t)(ns * nv + 2 * j) * M + k] = val.d[px];
                    Vb[(size_t)(ns * nv + 2 * j + 1) * M + k] = val.d[py];
                    ++j;
                }
            }
            // --- integrand_cost (ePSOPT.cpp:186-216)
            {
                vector_t x2, u2;
                for (int i = 0; i < ns; ++i) x2.push_back(&states[i]);
                for (int c = 0; c < nc; ++c) u2.push_back(&controls[c]);
                vector_t params_v = {std::string()};
                std::vector<std::string> pnames = {std::string("")};
                Dual L = std::any_cast<Dual>(P.objective(x2, u2, params_v, pnames, tnode, (tf - t0) / (M - 1)));
                if (maximize) L = -1.0 * L;
                cost += w[k] * L.v;
                for (int v = 0; v < nv; ++v) Vb[(size_t)(ns * nv + 2 * np + v) * M + k] = h * w[k] * L.d[v];
            }
        }
        // --- defect D.X - h.F and the D_kk term of the assembled diagonal
        for (int i = 0; i < ns; ++i)
            for (int k = 0; k < M; ++k) {
                const double* Dk = D + (size_t)k * M;
                const double* xi = Xb + (size_t)i * M;
                double s = 0;
                for (int jn = 0; jn < M; ++jn) s += Dk[jn] * xi[jn];
                Rb[(size_t)i * M + k] += s;
                Vb[(size_t)(i * nv + i) * M + k] += Dk[k];
            }
        COST[b] = h * cost;
        (void)sgn;
    }
    return 0;
}
