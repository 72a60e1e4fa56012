// emi_nlp.cpp -- primal-dual interior-point iteration for the transcribed VGP.
// See emi_nlp.hpp for what this stands in for in the reference (IPOPT behind
// PSOPT, src/ePSOPT/ePSOPT.cpp:62-66,84).  Written from the published
// algorithm (Waechter & Biegler 2006: barrier subproblems, fraction-to-the-
// boundary rule, second-order correction, acceptable-level termination), with
// an l1 merit function instead of a filter, and with the inertia of the KKT
// matrix fixed by construction instead of by trial factorisations: the node
// blocks of the Hessian are made positive definite (quasi-definite matrix for
// the backend: Cholesky of a Schur complement on the device) and the exact
// matrix comes back, together with an exact inertia verdict, through a
// low-rank correction.
//
// NLP in per-instance numbering (DESIGN.md "NLP layout"):
//   variables   z (states then controls, index v*M+k), slacks s for the path rows
//   equalities  defect_(i,k)(z) = 0,   c_(j,k)(z) - s_(j,k) = 0
//   bounds      zl <= z <= zu (zl==zu removes the variable), cl_j <= s_(j,k) <= cu_j
// The boundary conditions of ePSOPT::events (ePSOPT.cpp:137-141, 281-291) act
// directly on node variables, so they arrive here as variable bounds.
#include "emi_nlp.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace ETOL {
namespace mi355x {

// ---------------------------------------------------------------------------------------------
// Bunch-Kaufman LDL^T, lower triangle, unblocked
// ---------------------------------------------------------------------------------------------
bool ldlt_factor(LdltFactor& F) {
    const int n = F.n;
    double* A = F.a.data();
    F.ipiv.assign(n, 0);
    F.npos = F.nneg = F.nzero = 0;
    const double alpha = (1.0 + std::sqrt(17.0)) / 8.0;
    auto at = [&](int i, int j) -> double& { return A[(size_t)i * n + j]; };
    bool ok = true;
    int k = 0;
    while (k < n) {
        int kstep = 1, kp = k;
        const double absakk = std::fabs(at(k, k));
        int imax = k;
        double colmax = 0.0;
        for (int i = k + 1; i < n; ++i)
            if (std::fabs(at(i, k)) > colmax) { colmax = std::fabs(at(i, k)); imax = i; }
        if (std::max(absakk, colmax) == 0.0) {
            ok = false;
            F.ipiv[k] = k;
            ++F.nzero;
            ++k;
            continue;
        }
        if (absakk < alpha * colmax) {
            double rowmax = 0.0;
            for (int j = k; j < imax; ++j) rowmax = std::max(rowmax, std::fabs(at(imax, j)));
            for (int i = imax + 1; i < n; ++i) rowmax = std::max(rowmax, std::fabs(at(i, imax)));
            if (absakk >= alpha * colmax * (colmax / rowmax)) kp = k;
            else if (std::fabs(at(imax, imax)) >= alpha * rowmax) kp = imax;
            else { kp = imax; kstep = 2; }
        }
        const int kk = k + kstep - 1;
        if (kp != kk) {   // symmetric interchange of rows/columns kk and kp in the trailing block
            for (int i = kp + 1; i < n; ++i) std::swap(at(i, kk), at(i, kp));
            for (int j = kk + 1; j < kp; ++j) std::swap(at(j, kk), at(kp, j));
            std::swap(at(kk, kk), at(kp, kp));
            if (kstep == 2) std::swap(at(k + 1, k), at(kp, k));
        }
        if (kstep == 1) {
            const double piv = at(k, k);
            (piv > 0 ? F.npos : F.nneg)++;
            const double r = 1.0 / piv;
            for (int j = k + 1; j < n; ++j) {
                const double ajk = at(j, k) * r;
                if (ajk != 0.0)
                    for (int i = j; i < n; ++i) at(i, j) -= at(i, k) * ajk;
            }
            for (int i = k + 1; i < n; ++i) at(i, k) *= r;
            F.ipiv[k] = kp;
        } else {
            const double a11 = at(k, k), a21 = at(k + 1, k), a22 = at(k + 1, k + 1);
            const double det = a11 * a22 - a21 * a21, tr = a11 + a22;
            if (det < 0) { ++F.npos; ++F.nneg; }
            else if (det > 0) { (tr > 0 ? F.npos : F.nneg) += 2; }
            else { ++F.nzero; (tr > 0 ? F.npos : F.nneg)++; }
            if (k + 2 < n) {
                const double d11 = a22 / a21, d22 = a11 / a21;
                const double t = 1.0 / (d11 * d22 - 1.0), d21 = t / a21;
                for (int j = k + 2; j < n; ++j) {
                    const double wk = d21 * (d11 * at(j, k) - at(j, k + 1));
                    const double wk1 = d21 * (d22 * at(j, k + 1) - at(j, k));
                    for (int i = j; i < n; ++i) at(i, j) -= at(i, k) * wk + at(i, k + 1) * wk1;
                    at(j, k) = wk;
                    at(j, k + 1) = wk1;
                }
            }
            F.ipiv[k] = F.ipiv[k + 1] = -(kp + 1);
        }
        k += kstep;
    }
    return ok;
}

void ldlt_solve(const LdltFactor& F, double* b) {
    const int n = F.n;
    const double* A = F.a.data();
    auto at = [&](int i, int j) -> double { return A[(size_t)i * n + j]; };
    int k = 0;
    while (k < n) {   // L D y = P b
        if (F.ipiv[k] >= 0) {
            const int kp = F.ipiv[k];
            if (kp != k) std::swap(b[k], b[kp]);
            for (int i = k + 1; i < n; ++i) b[i] -= at(i, k) * b[k];
            b[k] /= at(k, k);
            ++k;
        } else {
            const int kp = -F.ipiv[k] - 1;
            if (kp != k + 1) std::swap(b[k + 1], b[kp]);
            for (int i = k + 2; i < n; ++i) b[i] -= at(i, k) * b[k] + at(i, k + 1) * b[k + 1];
            const double a21 = at(k + 1, k);
            const double akm1 = at(k, k) / a21, ak = at(k + 1, k + 1) / a21;
            const double den = akm1 * ak - 1.0;
            const double bkm1 = b[k] / a21, bk = b[k + 1] / a21;
            b[k] = (ak * bkm1 - bk) / den;
            b[k + 1] = (akm1 * bk - bkm1) / den;
            k += 2;
        }
    }
    k = n - 1;
    while (k >= 0) {   // L^T x = y, undo P
        if (F.ipiv[k] >= 0) {
            double s = b[k];
            for (int i = k + 1; i < n; ++i) s -= at(i, k) * b[i];
            b[k] = s;
            const int kp = F.ipiv[k];
            if (kp != k) std::swap(b[k], b[kp]);
            --k;
        } else {
            double s0 = b[k - 1], s1 = b[k];
            for (int i = k + 1; i < n; ++i) {
                s0 -= at(i, k - 1) * b[i];
                s1 -= at(i, k) * b[i];
            }
            b[k - 1] = s0;
            b[k] = s1;
            const int kp = -F.ipiv[k] - 1;
            if (kp != k) std::swap(b[k], b[kp]);
            k -= 2;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// interior point
// ---------------------------------------------------------------------------------------------
// Path rows are elastic:  sigma_j c_j(z) - s - e+ + e- = 0,  cl <= s <= cu,  e+- >= 0, with the
// exact penalty rho (e+ + e-) in the objective.  A strictly interior start then always exists
// (a straight-line guess usually crosses keep-outs, where a plain slack formulation jams against
// the slack bound), and for rho above the row multipliers the minimiser has e = 0, i.e. it is a
// KKT point of the original problem.  rho is raised and the iteration continued if some e stays
// positive at convergence.
namespace {

constexpr double INF_BOUND = 1e19;

struct Iterate {
    std::vector<double> z, s, e1, e2;             // primal: variables, row slacks, elastics
    std::vector<double> lam, y;                   // equality multipliers (defects, path rows)
    std::vector<double> zL, zU, vL, vU, w1, w2;   // bound multipliers (0 where the bound is infinite)
};

struct Eval {
    std::vector<double> RES, VALS, H;
    std::vector<double> LNK;          // residuals of the coupling rows (NlpProblem::links), [link][M]
    double cost = 0;
};

// Makes every node block of Q (packed lower triangles, Qblk[nh][M]) positive definite over its free
// variables: cyclic Jacobi eigen-decomposition of the (at most 16 x 16) block; if an eigenvalue is
// below eps * max|lambda|, the block is rebuilt from max(|lambda|, floor).  Returns the largest
// shift applied to an eigenvalue (0: nothing was modified).
struct BlockMod {          // one modified eigenpair of one node block:  Q~_k = Q_k + delta v v^T
    int node;
    double delta;
    double v[16];
};
double convexify_node_blocks(double* Qblk, const unsigned char* fixed, int nv, int M, std::vector<BlockMod>* mods) {
    constexpr int NMAX = 16;
    mods->clear();
    if (nv > NMAX) return 0.0;
    double worst = 0.0;
    double A[NMAX][NMAX], Vv[NMAX][NMAX], lam[NMAX], d[NMAX];
    // thresholds in the diagonally scaled block (unit diagonal): a block mixes barrier terms of 1e10 with
    // curvatures of 1e-2, and only after scaling is "zero" distinguishable from "negative"
    const double fl = 1e-9;
    for (int k = 0; k < M; ++k) {
        for (int v = 0; v < nv; ++v)
            for (int q = 0; q <= v; ++q) {
                const bool fx = fixed[v * M + k] || fixed[q * M + k];
                A[v][q] = A[q][v] = fx ? (v == q ? 1.0 : 0.0) : Qblk[(size_t)(v * (v + 1) / 2 + q) * M + k];
            }
        double amax = 0;
        for (int v = 0; v < nv; ++v) amax = std::max(amax, std::fabs(A[v][v]));
        for (int v = 0; v < nv; ++v)
            for (int q = 0; q < v; ++q) amax = std::max(amax, std::fabs(A[v][q]));
        if (amax == 0.0) amax = 1.0;
        for (int v = 0; v < nv; ++v) d[v] = std::sqrt(std::max(std::fabs(A[v][v]), 1e-12 * amax));
        for (int v = 0; v < nv; ++v)
            for (int q = 0; q < nv; ++q) A[v][q] /= d[v] * d[q];          // congruence: inertia unchanged
        // cheap screen: a block whose Cholesky runs through with comfortable pivots is left alone
        {
            double Lc[NMAX][NMAX];
            bool pd = true;
            for (int i = 0; i < nv && pd; ++i)
                for (int j = 0; j <= i; ++j) {
                    double sum = A[i][j];
                    for (int t = 0; t < j; ++t) sum -= Lc[i][t] * Lc[j][t];
                    if (i == j) {
                        if (!(sum > 10.0 * fl)) { pd = false; break; }
                        Lc[i][i] = std::sqrt(sum);
                    } else {
                        Lc[i][j] = sum / Lc[j][j];
                    }
                }
            if (pd) continue;
        }
        for (int i = 0; i < nv; ++i)
            for (int j = 0; j < nv; ++j) Vv[i][j] = i == j ? 1.0 : 0.0;
        for (int sweep = 0; sweep < 30; ++sweep) {
            double off = 0, dia = 0;
            for (int i = 0; i < nv; ++i) {
                dia += A[i][i] * A[i][i];
                for (int j = 0; j < i; ++j) off += A[i][j] * A[i][j];
            }
            if (off < 1e-32 * std::max(1.0, dia)) break;
            for (int p = 0; p < nv; ++p)
                for (int q = p + 1; q < nv; ++q) {
                    if (A[p][q] == 0.0) continue;
                    const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                    const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                    const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                    for (int r = 0; r < nv; ++r) {
                        const double arp = A[r][p], arq = A[r][q];
                        A[r][p] = c * arp - sn * arq;
                        A[r][q] = sn * arp + c * arq;
                    }
                    for (int r = 0; r < nv; ++r) {
                        const double apr = A[p][r], aqr = A[q][r];
                        A[p][r] = c * apr - sn * aqr;
                        A[q][r] = sn * apr + c * aqr;
                    }
                    for (int r = 0; r < nv; ++r) {
                        const double vrp = Vv[r][p], vrq = Vv[r][q];
                        Vv[r][p] = c * vrp - sn * vrq;
                        Vv[r][q] = sn * vrp + c * vrq;
                    }
                }
        }
        for (int i = 0; i < nv; ++i) {
            lam[i] = A[i][i];
            const double nl = std::max(std::fabs(lam[i]), fl);
            // a (numerically) zero eigenvalue is floored without bookkeeping -- a perturbation of 1e-9 of
            // the diagonal; reflected, genuinely negative ones become columns of the low-rank correction
            if (lam[i] < -fl) {
                BlockMod m;
                m.node = k;
                m.delta = nl - lam[i];
                for (int r = 0; r < nv; ++r) m.v[r] = d[r] * Vv[r][i];
                mods->push_back(m);
                worst = std::max(worst, m.delta * d[0] * d[0]);
            }
            lam[i] = nl;
        }
        for (int v = 0; v < nv; ++v)
            for (int q = 0; q <= v; ++q) {
                if (fixed[v * M + k] || fixed[q * M + k]) continue;
                double sum = 0;
                for (int e = 0; e < nv; ++e) sum += Vv[v][e] * lam[e] * Vv[q][e];
                Qblk[(size_t)(v * (v + 1) / 2 + q) * M + k] = d[v] * d[q] * sum;
            }
    }
    return worst;
}

// Host backend: the same matrix as etol_amd/csrc/emi_kkt.hip assembles, dense LDL^T (Bunch-Kaufman).
class DenseHostKkt : public KktBackend {
 public:
    explicit DenseHostKkt(const NlpProblem& P) : _P(P) {}
    int factor(const double* Qblk, const double* Jblk, const unsigned char* fx, double dc) override {
        const int M = _P.M, ns = _P.ns, nv = ns + _P.nc, nz = nv * M, md = ns * M, N = nz + md + (int)_P.links.size() * M;
        _F.n = N;
        _r = 0;
        _F.a.assign((size_t)N * N, 0.0);
        _fixed.assign(fx, fx + nz);
        double* a = _F.a.data();
        for (int k = 0; k < M; ++k)
            for (int v = 0; v < nv; ++v) {
                if (_fixed[v * M + k]) { a[(size_t)(v * M + k) * N + v * M + k] = 1.0; continue; }
                for (int q = 0; q <= v; ++q)
                    if (!_fixed[q * M + k]) a[(size_t)(v * M + k) * N + q * M + k] = Qblk[(size_t)(v * (v + 1) / 2 + q) * M + k];
            }
        for (int i = 0; i < ns; ++i)
            for (int k = 0; k < M; ++k) {
                double* row = a + (size_t)(nz + i * M + k) * N;
                for (int j = 0; j < M; ++j)
                    if (!_fixed[i * M + j]) row[i * M + j] = _P.D[(size_t)k * M + j];
                for (int v = 0; v < nv; ++v)
                    if (!_fixed[v * M + k]) row[v * M + k] = Jblk[(size_t)(i * nv + v) * M + k];
                row[nz + i * M + k] = -dc;
            }
        // coupling rows  z[dst][k] - sum_j W[k][j] z[src][j] = 0  (delayed values): constant entries, regularised like the defects
        for (size_t l = 0; l < _P.links.size(); ++l) {
            const NlpLink& L = _P.links[l];
            for (int k = 0; k < M; ++k) {
                double* row = a + (size_t)(nz + md + (int)l * M + k) * N;
                for (int j = 0; j < M; ++j)
                    if (!_fixed[L.src * M + j]) row[L.src * M + j] = -L.W[(size_t)k * M + j];
                if (!_fixed[L.dst * M + k]) row[L.dst * M + k] += 1.0;
                row[nz + md + (int)l * M + k] = -dc;
            }
        }
        _ok = ldlt_factor(_F) && _F.nzero == 0;
        return _ok ? 0 : 1;
    }
    int lowrank(int r, const int* node, const double* vec, const double* delta, bool* exact) override {
        _r = 0;
        *exact = r == 0;
        if (r == 0) return 0;
        if (!_ok) return -1;
        const int M = _P.M, nv = _P.ns + _P.nc, N = _F.n;
        _node.assign(node, node + r);
        _vec.assign(vec, vec + (size_t)r * nv);
        _Y.assign((size_t)N * r, 0.0);
        for (int c = 0; c < r; ++c) {
            double* y = &_Y[(size_t)c * N];
            for (int v = 0; v < nv; ++v) y[v * M + node[c]] = vec[(size_t)c * nv + v];
            for (size_t q = 0; q < _fixed.size(); ++q)
                if (_fixed[q]) y[q] = 0.0;
            ldlt_solve(_F, y);
        }
        // C = Delta^-1 - U^T Y, row-major lower triangle; right-looking Cholesky
        _C.assign((size_t)r * r, 0.0);
        for (int c = 0; c < r; ++c) {
            const double* y = &_Y[(size_t)c * N];
            for (int a = c; a < r; ++a) {
                double dot = 0;
                for (int v = 0; v < nv; ++v) dot += vec[(size_t)a * nv + v] * y[v * M + node[a]];
                _C[(size_t)a * r + c] = -dot;
            }
            _C[(size_t)c * r + c] += 1.0 / delta[c];
        }
        std::vector<double> colj(r);
        for (int j = 0; j < r; ++j) {
            const double djj = _C[(size_t)j * r + j];
            if (!(djj > 1e-14 * (1.0 / delta[j]))) return 0;          // not positive definite: K~ answers stay
            const double ljj = std::sqrt(djj);
            _C[(size_t)j * r + j] = ljj;
            for (int i = j + 1; i < r; ++i) {
                _C[(size_t)i * r + j] /= ljj;
                colj[i] = _C[(size_t)i * r + j];
            }
            for (int i = j + 1; i < r; ++i) {
                const double lij = colj[i];
                double* row = &_C[(size_t)i * r];
                for (int t = j + 1; t <= i; ++t) row[t] -= lij * colj[t];
            }
        }
        _r = r;
        *exact = true;
        return 0;
    }
    int solve(double* rhs, int nrhs) override {
        if (!_ok) return -1;
        const int M = _P.M, nv = _P.ns + _P.nc, N = _F.n;
        std::vector<double> t(_r);
        for (int c = 0; c < nrhs; ++c) {
            double* b = rhs + (size_t)c * N;
            for (size_t q = 0; q < _fixed.size(); ++q)
                if (_fixed[q]) b[q] = 0.0;
            ldlt_solve(_F, b);
            if (_r == 0) continue;
            // b <- b + Y C^-1 (U^T b)
            for (int a = 0; a < _r; ++a) {
                double dot = 0;
                for (int v = 0; v < nv; ++v) dot += _vec[(size_t)a * nv + v] * b[v * M + _node[a]];
                t[a] = dot;
            }
            for (int i = 0; i < _r; ++i) {
                double sum = t[i];
                for (int q = 0; q < i; ++q) sum -= _C[(size_t)i * _r + q] * t[q];
                t[i] = sum / _C[(size_t)i * _r + i];
            }
            for (int i = _r - 1; i >= 0; --i) {
                double sum = t[i];
                for (int q = i + 1; q < _r; ++q) sum -= _C[(size_t)q * _r + i] * t[q];
                t[i] = sum / _C[(size_t)i * _r + i];
            }
            for (int a = 0; a < _r; ++a) {
                const double* y = &_Y[(size_t)a * N];
                const double ta = t[a];
                for (int q = 0; q < N; ++q) b[q] += ta * y[q];
            }
        }
        return 0;
    }
    std::string last_error() const override { return "dense host factorisation"; }

 private:
    const NlpProblem& _P;
    LdltFactor _F;
    std::vector<unsigned char> _fixed;
    bool _ok = false;
    int _r = 0;                       // active low-rank correction (0: none)
    std::vector<int> _node;
    std::vector<double> _vec, _Y, _C;
};

}  // namespace

// ---- variable scaling (PSOPT's scaling = "automatic", reference src/ePSOPT/ePSOPT.cpp:63) ------------------------------
// PSOPT iterates on  z~_v = z_v / s_v  with s_v taken from the variable's bounds, and scales the defect rows of state i like
// state i ("state-based" defect scaling, its default).  With one scale per state / control (the same at every node) the scaled
// transcription has the SAME shape as the unscaled one:  (D x_i - h f_i) / s_i = D x~_i - h f_i / s_i,  so D, the node-diagonal
// Jacobian entries and the packed node blocks of the Hessian keep their layouts and only their values change:
//     defect rows / s_i;   f-partial (i, v) * s_v / s_i  (the entry (i, i) carries D_kk: factor 1);   path partials and the cost
//     gradient * s_v;   Hessian entry (v, q) * s_v s_q, evaluated with the multipliers lambda~_i / s_i.
// The evaluator and the KKT backend of the caller are used as they are (the backend reads D from its own context).
// The example's defect_scaling = "jacobian-based" (etol_psopt_example1.cpp:91: one scale per defect ROW) would break the
// D (x) I structure the Newton step is built on and is not offered.
class ScaledEvaluator : public NlpEvaluator {
 public:
    ScaledEvaluator(const NlpProblem& P, const std::vector<std::vector<std::pair<int, int>>>& rv, int npart)
        : in_(P.ev), s_(P.vscale), rv_(rv), ns_(P.ns), nc_(P.nc), M_(P.M), npart_(npart), z_((size_t)(P.ns + P.nc) * P.M),
          lam_((size_t)P.ns * P.M) {}
    int eval(const double* X, const double* U, double* RES, double* VALS, double* COST, bool jac) override {
        unscale(X, U);
        const int rc = in_->eval(z_.data(), z_.data() + (size_t)ns_ * M_, RES, VALS, COST, jac);
        if (rc != 0) return rc;
        const int nv = ns_ + nc_;
        for (int i = 0; i < ns_; ++i) {
            const double inv = 1.0 / s_[i];
            for (int k = 0; k < M_; ++k) RES[(size_t)i * M_ + k] *= inv;
        }
        if (jac && VALS) {
            for (int i = 0; i < ns_; ++i)
                for (int v = 0; v < nv; ++v) {
                    if (v == i) continue;
                    const double f = s_[v] / s_[i];
                    double* e = VALS + (size_t)(i * nv + v) * M_;
                    for (int k = 0; k < M_; ++k) e[k] *= f;
                }
            for (const auto& row : rv_)
                for (const auto& ve : row) {
                    double* e = VALS + (size_t)ve.second * M_;
                    for (int k = 0; k < M_; ++k) e[k] *= s_[ve.first];
                }
            for (int v = 0; v < nv; ++v) {
                double* e = VALS + (size_t)(ns_ * nv + npart_ + v) * M_;
                for (int k = 0; k < M_; ++k) e[k] *= s_[v];
            }
        }
        return 0;
    }
    int hess(const double* X, const double* U, const double* lamF, const double* lamC, double sigma, double* H) override {
        unscale(X, U);
        for (int i = 0; i < ns_; ++i)
            for (int k = 0; k < M_; ++k) lam_[(size_t)i * M_ + k] = lamF[(size_t)i * M_ + k] / s_[i];
        const int rc = in_->hess(z_.data(), z_.data() + (size_t)ns_ * M_, lam_.data(), lamC, sigma, H);
        if (rc != 0) return rc;
        const int nv = ns_ + nc_;
        for (int hi = 0; hi < nv; ++hi)
            for (int lo = 0; lo <= hi; ++lo) {
                const double f = s_[hi] * s_[lo];
                if (f == 1.0) continue;
                double* e = H + (size_t)(hi * (hi + 1) / 2 + lo) * M_;
                for (int k = 0; k < M_; ++k) e[k] *= f;
            }
        return 0;
    }
    std::string last_error() const override { return in_->last_error(); }

 private:
    void unscale(const double* X, const double* U) {
        for (int v = 0; v < ns_; ++v)
            for (int k = 0; k < M_; ++k) z_[(size_t)v * M_ + k] = X[(size_t)v * M_ + k] * s_[v];
        for (int j = 0; j < nc_; ++j)
            for (int k = 0; k < M_; ++k) z_[(size_t)(ns_ + j) * M_ + k] = U[(size_t)j * M_ + k] * s_[ns_ + j];
    }
    NlpEvaluator* in_;
    std::vector<double> s_;
    std::vector<std::vector<std::pair<int, int>>> rv_;
    int ns_, nc_, M_, npart_;
    std::vector<double> z_, lam_;
};

NlpResult solve_nlp(const NlpProblem& P, const NlpOptions& opt, const std::vector<double>& z0) {
    NlpResult R;
    const int ns = P.ns, nc = P.nc, np = P.np, M = P.M, nv = ns + nc;
    const int nz = nv * M, md = ns * M, mc = np * M, nh = nv * (nv + 1) / 2;
    const int nl = (int)P.links.size(), ml = nl * M, me = md + ml;      // coupling rows behind the defects: me equality multipliers
    for (const NlpLink& L : P.links)
        if (L.dst < 0 || L.dst >= nv || L.src < 0 || L.src >= nv || L.dst == L.src || (int)L.W.size() != M * M) {
            R.msg = "solve_nlp: a coupling row names variables outside the problem or has no M x M operator";
            return R;
        }
    if (nl > 0 && P.kkt) { R.msg = "solve_nlp: coupling rows (delayed values) are solved with the dense host backend only"; return R; }
    // (variable, VALS entry) pairs of every path row
    std::vector<std::vector<std::pair<int, int>>> rv = P.row_vars;
    if (rv.empty())
        for (int j = 0; j < np; ++j) rv.push_back({{P.px, ns * nv + 2 * j}, {P.py, ns * nv + 2 * j + 1}});
    if ((int)rv.size() != np) { R.msg = "solve_nlp: row_vars has the wrong length"; return R; }
    int npart = 0;
    for (const auto& r : rv) npart += (int)r.size();
    const int nvals = ns * nv + npart + nv;
    if (!P.vscale.empty()) {            // iterate on the scaled variables (ScaledEvaluator above), answer in the caller's
        if ((int)P.vscale.size() != nv || (int)z0.size() != nz || (int)P.zl.size() != nz || (int)P.zu.size() != nz || !P.ev) {
            R.msg = "solve_nlp: inconsistent problem sizes (vscale)";
            return R;
        }
        for (double sv : P.vscale)
            if (!(sv > 0) || !std::isfinite(sv)) { R.msg = "solve_nlp: vscale must be positive"; return R; }
        for (const NlpLink& L : P.links)        // (d - W z) / s keeps W only if both ends carry the same scale
            if (P.vscale[L.dst] != P.vscale[L.src]) { R.msg = "solve_nlp: a coupled variable must be scaled like its source"; return R; }
        NlpProblem Q = P;
        Q.vscale.clear();
        Q.row_vars = rv;
        ScaledEvaluator sev(P, rv, npart);
        Q.ev = &sev;
        std::vector<double> zs(z0);
        for (int v = 0; v < nv; ++v)
            for (int k = 0; k < M; ++k) {
                const size_t q = (size_t)v * M + k;
                const double inv = 1.0 / P.vscale[v];
                zs[q] *= inv;
                if (Q.zl[q] > -INF_BOUND) Q.zl[q] *= inv;
                if (Q.zu[q] < INF_BOUND) Q.zu[q] *= inv;
                if (P.zl[q] == P.zu[q]) Q.zu[q] = Q.zl[q];
            }
        if ((int)Q.lamF0.size() == md)
            for (int i = 0; i < ns; ++i)
                for (int k = 0; k < M; ++k) Q.lamF0[(size_t)i * M + k] *= P.vscale[i];
        NlpResult S = solve_nlp(Q, opt, zs);
        for (int v = 0; v < nv && (int)S.z.size() == nz; ++v)
            for (int k = 0; k < M; ++k) S.z[(size_t)v * M + k] *= P.vscale[v];
        for (int i = 0; i < ns && (int)S.lamF.size() == md; ++i)
            for (int k = 0; k < M; ++k) S.lamF[(size_t)i * M + k] /= P.vscale[i];
        return S;
    }
    if (!P.ev || (int)P.zl.size() != nz || (int)P.zu.size() != nz || (int)P.D.size() != M * M ||
        (int)P.cl.size() != np || (int)P.cu.size() != np || (int)z0.size() != nz) {
        R.msg = "solve_nlp: inconsistent problem sizes";
        return R;
    }
    const auto tstart = std::chrono::steady_clock::now();
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double>(b - a).count();
    };

    // free-variable map
    std::vector<int> fidx(nz, -1);
    int nf = 0;
    for (int q = 0; q < nz; ++q)
        if (P.zu[q] > P.zl[q]) fidx[q] = nf++;
    auto hasL = [&](int q) { return P.zl[q] > -INF_BOUND; };
    auto hasU = [&](int q) { return P.zu[q] < INF_BOUND; };
    // path rows are iterated on in scaled form  sigma_j * c_j  (same KKT points, better
    // balanced against the defects: a keep-out value is O(a^2 b^2) ~ 1e-3 unscaled)
    std::vector<double> sig(np, 1.0);
    if (!P.cscale.empty()) {
        if ((int)P.cscale.size() != np) { R.msg = "solve_nlp: cscale has the wrong length"; return R; }
        for (int j = 0; j < np; ++j) sig[j] = P.cscale[j] > 0 ? P.cscale[j] : 1.0;
    }
    for (int j = 0; j < np; ++j)
        if (!(P.cl[j] > -INF_BOUND) && !(P.cu[j] < INF_BOUND)) { R.msg = "solve_nlp: path row without any bound"; return R; }
    auto shasL = [&](int r) { return P.cl[r / M] > -INF_BOUND; };
    auto shasU = [&](int r) { return P.cu[r / M] < INF_BOUND; };
    auto cL = [&](int r) { return shasL(r) ? sig[r / M] * P.cl[r / M] : P.cl[r / M]; };
    auto cU = [&](int r) { return shasU(r) ? sig[r / M] * P.cu[r / M] : P.cu[r / M]; };

    Iterate it;
    it.z = z0;
    // push the start into the interior of the bounds
    for (int q = 0; q < nz; ++q) {
        const double l = P.zl[q], u = P.zu[q];
        if (fidx[q] < 0) { it.z[q] = l; continue; }
        double pl = 0, pu = 0;
        if (hasL(q)) pl = opt.bound_push * std::max(1.0, std::fabs(l));
        if (hasU(q)) pu = opt.bound_push * std::max(1.0, std::fabs(u));
        if (hasL(q) && hasU(q)) {
            pl = std::min(pl, opt.bound_frac * (u - l));
            pu = std::min(pu, opt.bound_frac * (u - l));
        }
        if (hasL(q)) it.z[q] = std::max(it.z[q], l + pl);
        if (hasU(q)) it.z[q] = std::min(it.z[q], u - pu);
    }

    Eval E;
    E.RES.resize((size_t)(ns + np) * M);
    E.VALS.resize((size_t)nvals * M);
    E.H.resize((size_t)nh * M);
    auto evaluate = [&](const std::vector<double>& z, Eval& e, bool jac) -> bool {
        ++R.evaluations;
        const auto te = now();
        const int est = P.ev->eval(z.data(), z.data() + (size_t)ns * M, e.RES.data(), jac ? e.VALS.data() : nullptr, &e.cost, jac);
        R.t_eval += secs(te, now());
        if (est != 0) return false;
        e.LNK.resize(ml);
        for (int l = 0; l < nl; ++l) {
            const NlpLink& L = P.links[l];
            const double* src = &z[(size_t)L.src * M];
            for (int k = 0; k < M; ++k) {
                const double* Wk = &L.W[(size_t)k * M];
                double acc = 0;
                for (int j = 0; j < M; ++j) acc += Wk[j] * src[j];
                e.LNK[(size_t)l * M + k] = z[(size_t)L.dst * M + k] - acc;
            }
        }
        for (int j = 0; j < np; ++j) {
            if (sig[j] == 1.0) continue;
            for (int k = 0; k < M; ++k) {
                e.RES[(size_t)(ns + j) * M + k] *= sig[j];
                if (jac) {
                    e.VALS[(size_t)(ns * nv + 2 * j) * M + k] *= sig[j];
                    e.VALS[(size_t)(ns * nv + 2 * j + 1) * M + k] *= sig[j];
                }
            }
        }
        return true;
    };
    if (!evaluate(it.z, E, true)) { R.msg = "evaluator failed: " + P.ev->last_error(); return R; }

    double rho = opt.rho_init > 0 ? opt.rho_init : 10.0;
    it.s.assign(mc, 0.0);
    it.e1.assign(mc, 0.0);
    it.e2.assign(mc, 0.0);
    for (int r = 0; r < mc; ++r) {
        const double c0 = E.RES[(size_t)md + r];
        double v = c0;
        const double l = cL(r), u = cU(r);
        double pl = shasL(r) ? opt.bound_push * std::max(1.0, std::fabs(l)) : 0;
        double pu = shasU(r) ? opt.bound_push * std::max(1.0, std::fabs(u)) : 0;
        if (shasL(r) && shasU(r)) { pl = std::min(pl, opt.bound_frac * (u - l)); pu = std::min(pu, opt.bound_frac * (u - l)); }
        if (shasL(r)) v = std::max(v, l + pl);
        if (shasU(r)) v = std::min(v, u - pu);
        it.s[r] = v;
        const double gap = c0 - v, ee = opt.bound_push * std::max(1.0, std::fabs(gap));
        it.e1[r] = std::max(gap, 0.0) + ee;     // residual c - s - e1 + e2 starts at exactly 0
        it.e2[r] = std::max(-gap, 0.0) + ee;
    }
    it.lam.assign(me, 0.0);
    it.y.assign(mc, 0.0);
    if ((int)P.lamF0.size() == md) std::copy(P.lamF0.begin(), P.lamF0.end(), it.lam.begin());
    auto eqr = [&](const Eval& e, int r) { return r < md ? e.RES[r] : e.LNK[r - md]; };      // residual of equality row r
    // row weights of the merit function (jacobian_defect_scaling): 1 / max(1, inf-norm of the defect row of the Jacobian at the start)
    std::vector<double> rs(me, 1.0);
    if (P.jacobian_defect_scaling) {
        const double* V0 = E.VALS.data();
        for (int k = 0; k < M; ++k) {
            double dmax = 0;
            for (int j = 0; j < M; ++j)
                if (j != k) dmax = std::max(dmax, std::fabs(P.D[(size_t)k * M + j]));
            for (int i = 0; i < ns; ++i) {
                double nrm = dmax;
                for (int v = 0; v < nv; ++v) nrm = std::max(nrm, std::fabs(V0[(size_t)(i * nv + v) * M + k]));
                rs[i * M + k] = 1.0 / std::max(1.0, nrm);
            }
        }
    }
    if ((int)P.lamC0.size() == mc)      // iterated on in scaled form; inside the penalty box
        for (int r = 0; r < mc; ++r) it.y[r] = std::min(std::max(P.lamC0[r] / sig[r / M], -0.9 * rho), 0.9 * rho);
    it.zL.assign(nz, 0.0); it.zU.assign(nz, 0.0);
    it.vL.assign(mc, 0.0); it.vU.assign(mc, 0.0);
    it.w1.assign(mc, rho); it.w2.assign(mc, rho);
    for (int r = 0; r < mc; ++r) { it.w1[r] = std::max(1e-8, rho - it.y[r]); it.w2[r] = std::max(1e-8, rho + it.y[r]); }
    for (int q = 0; q < nz; ++q) if (fidx[q] >= 0) { if (hasL(q)) it.zL[q] = 1.0; if (hasU(q)) it.zU[q] = 1.0; }
    for (int r = 0; r < mc; ++r) { if (shasL(r)) it.vL[r] = 1.0; if (shasU(r)) it.vU[r] = 1.0; }

    double mu = opt.mu_init, nu = 1.0;
    double dw_used = 0.0;   // delta_w of the last exact step, or the largest reflected eigenvalue shift (log only)
    double dw_last_ok = 0.0;   // last nonzero delta_w that gave the right inertia
    const double tau_min = 0.99, kappa_eps = 10.0, kappa_mu = 0.2, theta_mu = 1.5, kappa_sigma = 1e10;
    std::vector<double> y_unscaled(mc);

    // --- pieces of the KKT residual at the current point --------------------------------------
    std::vector<double> gradf(nz), jtl(nz);
    auto grad_and_jt = [&](const Iterate& I) {
        const double* V = E.VALS.data();
        for (int v = 0; v < nv; ++v)
            for (int k = 0; k < M; ++k) gradf[v * M + k] = V[(size_t)(ns * nv + npart + v) * M + k];
        std::fill(jtl.begin(), jtl.end(), 0.0);
        // J_d^T lam: off-diagonal D part, then the node blocks (which hold D_kk)
        for (int i = 0; i < ns; ++i)
            for (int k = 0; k < M; ++k) {
                const double l = I.lam[i * M + k];
                if (l == 0.0) continue;
                const double* Dk = &P.D[(size_t)k * M];
                double* col = &jtl[(size_t)i * M];
                for (int j = 0; j < M; ++j) col[j] += Dk[j] * l;
                col[k] -= Dk[k] * l;
                for (int v = 0; v < nv; ++v) jtl[v * M + k] += V[(size_t)(i * nv + v) * M + k] * l;
            }
        for (int j = 0; j < np; ++j)
            for (int k = 0; k < M; ++k) {
                const double yy = I.y[j * M + k];
                for (const auto& ve : rv[j]) jtl[ve.first * M + k] += V[(size_t)ve.second * M + k] * yy;
            }
        for (int l = 0; l < nl; ++l) {          // coupling rows: + nu on the coupled variable, - W^T nu on its source
            const NlpLink& L = P.links[l];
            double* col = &jtl[(size_t)L.src * M];
            for (int k = 0; k < M; ++k) {
                const double nu_k = I.lam[md + l * M + k];
                if (nu_k == 0.0) continue;
                jtl[(size_t)L.dst * M + k] += nu_k;
                const double* Wk = &L.W[(size_t)k * M];
                for (int j = 0; j < M; ++j) col[j] -= Wk[j] * nu_k;
            }
        }
    };
    auto row_res = [&](const Eval& e, const std::vector<double>& s, const std::vector<double>& e1,
                       const std::vector<double>& e2, int r) { return e.RES[(size_t)md + r] - s[r] - e1[r] + e2[r]; };
    auto kkt_error = [&](const Iterate& I, double mu_t, double* viol_out, double* emax_out) {
        double sumz = 0, summ = 0;
        int cntz = 0;
        for (int q = 0; q < nz; ++q) { sumz += I.zL[q] + I.zU[q]; cntz += (I.zL[q] > 0) + (I.zU[q] > 0); }
        for (int r = 0; r < mc; ++r) {
            sumz += I.vL[r] + I.vU[r] + I.w1[r] + I.w2[r];
            cntz += (I.vL[r] > 0) + (I.vU[r] > 0) + 2;
            summ += std::fabs(I.y[r]);
        }
        for (int r = 0; r < me; ++r) summ += std::fabs(I.lam[r]);
        const double smax = 100.0;
        const double sd = std::max(smax, (summ + sumz) / std::max(1, me + mc + cntz)) / smax;
        const double sc = std::max(smax, sumz / std::max(1, cntz)) / smax;
        double ed = 0, ep = 0, ec = 0, emax = 0;
        for (int q = 0; q < nz; ++q)
            if (fidx[q] >= 0) ed = std::max(ed, std::fabs(gradf[q] + jtl[q] - I.zL[q] + I.zU[q]));
        for (int r = 0; r < mc; ++r) {
            ed = std::max(ed, std::fabs(-I.y[r] - I.vL[r] + I.vU[r]));
            ed = std::max(ed, std::fabs(rho - I.y[r] - I.w1[r]));
            ed = std::max(ed, std::fabs(rho + I.y[r] - I.w2[r]));
        }
        for (int r = 0; r < me; ++r) ep = std::max(ep, std::fabs(eqr(E, r)));
        for (int r = 0; r < mc; ++r) {
            ep = std::max(ep, std::fabs(row_res(E, I.s, I.e1, I.e2, r)));
            emax = std::max(emax, std::max(I.e1[r], I.e2[r]));
        }
        for (int q = 0; q < nz; ++q) {
            if (fidx[q] < 0) continue;
            if (hasL(q)) ec = std::max(ec, std::fabs((I.z[q] - P.zl[q]) * I.zL[q] - mu_t));
            if (hasU(q)) ec = std::max(ec, std::fabs((P.zu[q] - I.z[q]) * I.zU[q] - mu_t));
        }
        for (int r = 0; r < mc; ++r) {
            if (shasL(r)) ec = std::max(ec, std::fabs((I.s[r] - cL(r)) * I.vL[r] - mu_t));
            if (shasU(r)) ec = std::max(ec, std::fabs((cU(r) - I.s[r]) * I.vU[r] - mu_t));
            ec = std::max(ec, std::fabs(I.e1[r] * I.w1[r] - mu_t));
            ec = std::max(ec, std::fabs(I.e2[r] * I.w2[r] - mu_t));
        }
        if (viol_out) *viol_out = ep;
        if (emax_out) *emax_out = emax;
        return std::max(std::max(ed / sd, ep), ec / sc);
    };
    auto barrier_merit = [&](const std::vector<double>& z, const std::vector<double>& s, const std::vector<double>& e1,
                             const std::vector<double>& e2, const Eval& e, double mu_t, double nu_t, double* infeas) {
        double phi = e.cost, viol = 0;
        for (int q = 0; q < nz; ++q) {
            if (fidx[q] < 0) continue;
            if (hasL(q)) phi -= mu_t * std::log(z[q] - P.zl[q]);
            if (hasU(q)) phi -= mu_t * std::log(P.zu[q] - z[q]);
        }
        for (int r = 0; r < mc; ++r) {
            if (shasL(r)) phi -= mu_t * std::log(s[r] - cL(r));
            if (shasU(r)) phi -= mu_t * std::log(cU(r) - s[r]);
            phi += rho * (e1[r] + e2[r]) - mu_t * (std::log(e1[r]) + std::log(e2[r]));
            viol += std::fabs(row_res(e, s, e1, e2, r));
        }
        for (int r = 0; r < me; ++r) viol += rs[r] * std::fabs(eqr(e, r));
        if (infeas) *infeas = viol;
        return phi + nu_t * viol;
    };

    // Newton-step linear algebra: the caller's backend (eMI355X: the device) or the dense host one
    const size_t NN0 = (size_t)nz + me;
    DenseHostKkt host_kkt(P);
    KktBackend* kkt = P.kkt ? P.kkt : &host_kkt;
    std::vector<double> Qblk((size_t)nh * M), rhs_full((size_t)nz + me), eqres(me);
    std::vector<unsigned char> fixed_mask(nz);
    for (int q = 0; q < nz; ++q) fixed_mask[q] = fidx[q] < 0 ? 1 : 0;
    std::vector<BlockMod> mods;
    std::vector<double> Qexact, rhs_keep(NN0), resid(NN0), x_prev(NN0);
    bool exact_step = false;
    const int max_lowrank = 4096;        // more modified eigenpairs than this: take the modified step untested
    std::vector<double> dz(nz), ds(mc), de1(mc), de2(mc), dlam(me), dy(mc), dzL(nz), dzU(nz), dvL(mc), dvU(mc),
        dw1(mc), dw2(mc);
    std::vector<double> sig_t(mc), r_t(mc), sig_s(mc), rhat_s(mc);
    Eval Et, Ekeep;
    Et.RES.resize(E.RES.size());
    Et.VALS.resize(E.VALS.size());
    std::vector<double> zt(nz), st(mc), e1t(mc), e2t(mc);

    // ---- pieces of one Newton step, shared by the regular step and the second-order correction ----
    const size_t NN = (size_t)nz + me;
    int r_mod = 0;
    // r_t of the eliminated path rows for given row residuals  c - s - e1 + e2
    auto fill_rt = [&](const std::vector<double>& rowres) {
        for (int r = 0; r < mc; ++r) {
            const double a1 = it.e1[r] / it.w1[r], a2 = it.e2[r] / it.w2[r];
            r_t[r] = rowres[r] + rhat_s[r] / sig_s[r] - a1 * (it.y[r] - rho + mu / it.e1[r]) -
                     a2 * (it.y[r] + rho - mu / it.e2[r]);
        }
    };
    // right-hand side of the reduced KKT system in full indexing (fixed variables: 0), for defect residuals defres
    auto build_rhs = [&](double* out, const double* defres) {
        const double* V = E.VALS.data();
        std::fill(out, out + NN, 0.0);
        for (int q = 0; q < nz; ++q) {
            if (fidx[q] < 0) continue;
            double r = gradf[q] + jtl[q];
            if (hasL(q)) r -= mu / (it.z[q] - P.zl[q]);
            if (hasU(q)) r += mu / (P.zu[q] - it.z[q]);
            out[q] = -r;
        }
        for (int j = 0; j < np; ++j)
            for (int k = 0; k < M; ++k) {
                const double t = sig_t[j * M + k] * r_t[j * M + k];
                for (const auto& ve : rv[j])
                    if (fidx[ve.first * M + k] >= 0) out[ve.first * M + k] -= V[(size_t)ve.second * M + k] * t;
            }
        for (int r = 0; r < me; ++r) out[nz + r] = -defres[r];
    };
    std::vector<double> soc_def(me), soc_row(mc), soc_rhs(NN), lr_vec, lr_delta;
    std::vector<int> lr_node;
    // y = [[Q, J^T], [J, -dc I]] x  with the node blocks Qb (fixed variables: identity rows/columns)
    std::vector<double> xfree;
    auto kkt_matvec = [&](const double* Qb, const double* x, double* y, double dcv) {
        const double* V = E.VALS.data();
        std::fill(y, y + NN, 0.0);
        for (int k = 0; k < M; ++k)
            for (int v = 0; v < nv; ++v) {
                if (fixed_mask[v * M + k]) continue;
                double acc = 0;
                for (int q = 0; q < nv; ++q) {
                    if (fixed_mask[q * M + k]) continue;
                    const int hi = std::max(v, q), lo = std::min(v, q);
                    acc += Qb[(size_t)(hi * (hi + 1) / 2 + lo) * M + k] * x[q * M + k];
                }
                y[v * M + k] = acc;
            }
        // D part: plain dot / axpy over whole rows of D (vectorisable), with x of fixed variables taken as 0
        // and the node-diagonal term (which lives in the node blocks V) taken out again
        xfree.assign(x, x + nz);
        for (int q = 0; q < nz; ++q)
            if (fixed_mask[q]) xfree[q] = 0.0;
        for (int i = 0; i < ns; ++i) {
            const double* xi = &xfree[(size_t)i * M];
            double* yi = &y[(size_t)i * M];
            for (int k = 0; k < M; ++k) {
                const int R = nz + i * M + k;
                const double* Dk = &P.D[(size_t)k * M];
                const double xr = x[R];
                double acc = 0;
#pragma omp simd reduction(+ : acc)
                for (int j = 0; j < M; ++j) acc += Dk[j] * xi[j];
#pragma omp simd
                for (int j = 0; j < M; ++j) yi[j] += Dk[j] * xr;
                acc -= Dk[k] * xi[k];
                yi[k] -= Dk[k] * xr;
                for (int v = 0; v < nv; ++v) {
                    if (fixed_mask[v * M + k]) continue;
                    const double jv = V[(size_t)(i * nv + v) * M + k];
                    acc += jv * x[v * M + k];
                    y[v * M + k] += jv * xr;
                }
                y[R] = acc - dcv * xr;
            }
        }
        for (int l = 0; l < nl; ++l) {          // coupling rows
            const NlpLink& L = P.links[l];
            const double* xs = &xfree[(size_t)L.src * M];
            double* ys = &y[(size_t)L.src * M];
            for (int k = 0; k < M; ++k) {
                const int Rr = nz + md + l * M + k;
                const double* Wk = &L.W[(size_t)k * M];
                const double xr = x[Rr];
                double acc = xfree[(size_t)L.dst * M + k];
                for (int j = 0; j < M; ++j) acc -= Wk[j] * xs[j];
                for (int j = 0; j < M; ++j) ys[j] -= Wk[j] * xr;
                y[(size_t)L.dst * M + k] += xr;
                y[Rr] = acc - dcv * xr;
            }
        }
        for (int q = 0; q < nz; ++q)
            if (fixed_mask[q]) y[q] = x[q];
    };
    // everything that was eliminated from the system, from dz (uses r_t)
    auto expand_step = [&]() {
        const double* V = E.VALS.data();
        for (int j = 0; j < np; ++j)
            for (int k = 0; k < M; ++k) {
                const int r = j * M + k;
                double jcdz = 0.0;
                for (const auto& ve : rv[j]) jcdz += V[(size_t)ve.second * M + k] * dz[ve.first * M + k];
                dy[r] = sig_t[r] * (jcdz + r_t[r]);
                ds[r] = (dy[r] - rhat_s[r]) / sig_s[r];
                de1[r] = it.e1[r] / it.w1[r] * (dy[r] + it.y[r] - rho + mu / it.e1[r]);
                de2[r] = it.e2[r] / it.w2[r] * (-dy[r] - it.y[r] - rho + mu / it.e2[r]);
                dvL[r] = dvU[r] = 0;
                if (shasL(r)) { const double g = it.s[r] - cL(r); dvL[r] = mu / g - it.vL[r] - it.vL[r] / g * ds[r]; }
                if (shasU(r)) { const double g = cU(r) - it.s[r]; dvU[r] = mu / g - it.vU[r] + it.vU[r] / g * ds[r]; }
                dw1[r] = mu / it.e1[r] - it.w1[r] - it.w1[r] / it.e1[r] * de1[r];
                dw2[r] = mu / it.e2[r] - it.w2[r] - it.w2[r] / it.e2[r] * de2[r];
            }
        for (int q = 0; q < nz; ++q) {
            dzL[q] = dzU[q] = 0;
            if (fidx[q] < 0) continue;
            if (hasL(q)) { const double g = it.z[q] - P.zl[q]; dzL[q] = mu / g - it.zL[q] - it.zL[q] / g * dz[q]; }
            if (hasU(q)) { const double g = P.zu[q] - it.z[q]; dzU[q] = mu / g - it.zU[q] + it.zU[q] / g * dz[q]; }
        }
    };

    { const auto tj = now(); grad_and_jt(it); R.t_jt += secs(tj, now()); }
    int n_acceptable = 0;
    bool force_modified = false;
    bool search_on = false, last_step_reflected = false;
    // Raising the penalty weight is the answer to a relaxed path row only while it helps: beyond 1e5,
    // `max_futile_escalations` tenfold raises in a row that have not halved the largest elastic variable end the solve (a keep-out that cannot be
    // cleared from this side: the Monte-Carlo scenario of this kind used to burn 950 iterations up to rho = 1e11).
    bool locally_infeasible = false;
    int futile = 0;
    double emax_ref = 1e300;
    auto futile_escalation = [&](double emax_now) {
        if (rho < 1e5) return false;         // weights a multiplier of a scaled row can plausibly need: keep raising
        if (emax_now < 0.5 * emax_ref) { emax_ref = emax_now; futile = 0; return false; }
        return ++futile >= opt.max_futile_escalations;
    };
    int stagn = 0, crawl = 0;
    bool mu_restarted = false;
    // (experiments, profiles/r01_notes.md) max_shift_trials = 0 switches the inertia search off, crawl_limit = 1000 the crawl rule
    const int max_shift_trials = opt.max_shift_trials;
    const int stagn_limit = opt.stagnation_iters;
    const int crawl_limit = opt.crawl_limit;
    double stagn_ref = 1e300;
    for (int iter = 0;; ++iter) {
        R.iterations = iter;
        double viol = 0, emax = 0;
        const double err0 = kkt_error(it, 0.0, &viol, &emax);
        R.kkt_error = err0;
        R.constr_viol = viol;
        if (opt.print_level >= 5)
            printf("iter %3d  cost %.10e  inf_pr %.2e  kkt %.2e  mu %.1e  dw %.1e  nu %.1e  emax %.1e  rho %.0e  r %d%s\n", iter,
                   E.cost, viol, err0, mu, dw_used, nu, emax, rho, (int)mods.size(), exact_step ? " exact" : "");
        if (err0 <= opt.tol) {
            if (emax <= std::max(opt.tol, 1e-9) * 10.0 || mc == 0) { R.ok = true; R.msg = "converged"; break; }
            // a path row is still relaxed: the penalty was too small for it
            if (rho >= 1e12 || futile_escalation(emax)) {
                R.msg = "converged to a point that violates the path rows (locally infeasible)";
                break;
            }
            rho *= 10.0;
            mu = std::max(mu, 1e-2);
            for (int r = 0; r < mc; ++r) { it.w1[r] = std::max(1e-8, rho - it.y[r]); it.w2[r] = std::max(1e-8, rho + it.y[r]); }
        }
        // stagnation at round-off above the tolerance (large meshes): accept like IPOPT's acceptable level
        if (err0 <= opt.acceptable_factor * opt.tol && (mc == 0 || emax <= 1e-6)) {
            if (++n_acceptable >= opt.acceptable_iter) { R.ok = true; R.msg = "converged to acceptable level"; break; }
        } else {
            n_acceptable = 0;
        }
        // reflected steps that have not reduced the KKT residual of the barrier problem by 10 % over `stagn_limit` iterations:
        // switch the inertia search on (measured, profiles/r01_notes.md: 4 costs the keep-out Monte-Carlo sets half their
        // throughput in trial factorisations, 12 keeps it and still rescues the fixed-wing problems)
        {
            const double err_mu_now = kkt_error(it, mu, nullptr, nullptr);
            if (last_step_reflected && err_mu_now > 0.9 * stagn_ref) {
                if (++stagn >= stagn_limit) {
                    search_on = true;
                    if (opt.mu_restart > 1.0 && !mu_restarted && mu < 1e-4) {
                        mu_restarted = true;
                        mu = std::min(1e-3, mu * opt.mu_restart);
                        nu = 1.0;
                        stagn = 0;
                        stagn_ref = 1e300;
                        if (opt.print_level >= 5) printf("stagnation at a small barrier parameter: raised to %.1e\n", mu);
                    }
                }
            } else {
                stagn = 0;
                stagn_ref = err_mu_now;
            }
        }
        if (iter >= opt.max_iter) { R.msg = "maximum number of iterations exceeded"; break; }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - tstart).count() > opt.max_cpu_time) {
            R.msg = "time limit exceeded";
            break;
        }
        // barrier update (may fire several times in a row)
        while (mu > opt.tol / 10.0 && kkt_error(it, mu, nullptr, nullptr) <= kappa_eps * mu) {
            // this barrier problem is solved.  A row multiplier at the penalty weight (or, equivalently, an
            // elastic still far above mu / rho) means the weight is too small for that row: raise it now
            // instead of converging to a relaxed point first
            double ymax = 0;
            for (int r = 0; r < mc; ++r) ymax = std::max(ymax, std::fabs(it.y[r]));
            if (mc > 0 && (emax > std::max(1e-6, 100.0 * mu) || ymax > 0.9 * rho) && rho < 1e12) {
                if (futile_escalation(emax)) { locally_infeasible = true; break; }
                rho *= 10.0;
                mu = std::max(mu, 1e-2);
                for (int r = 0; r < mc; ++r) { it.w1[r] = std::max(1e-8, rho - it.y[r]); it.w2[r] = std::max(1e-8, rho + it.y[r]); }
                nu = 1.0;
                break;
            }
            mu = std::max(opt.tol / 10.0, std::min(kappa_mu * mu, std::pow(mu, theta_mu)));
            nu = 1.0;    // a new barrier problem: the penalty weight is rebuilt from its multipliers, not inherited
        }
        if (locally_infeasible) {
            R.msg = "the path rows stay violated while the penalty weight grows (locally infeasible)";
            break;
        }
        const double tau = std::max(tau_min, 1.0 - mu);

        // exact Lagrangian Hessian blocks from the device
        for (int r = 0; r < mc; ++r) y_unscaled[r] = sig[r / M] * it.y[r];
        const auto th0 = now();
        const int hess_rc = P.ev->hess(it.z.data(), it.z.data() + (size_t)ns * M, it.lam.data(), np ? y_unscaled.data() : nullptr, 1.0,
                                       E.H.data());
        R.t_hess += secs(th0, now());
        if (hess_rc != 0) {
            R.msg = "Hessian evaluation failed: " + P.ev->last_error();
            break;
        }
        // eliminate (s, e+, e-) of every path row:  dy = sig_t (J_c dz + r_t)
        for (int r = 0; r < mc; ++r) {
            double sg = 0, rh = -it.y[r];
            if (shasL(r)) { const double g = it.s[r] - cL(r); sg += it.vL[r] / g; rh -= mu / g; }
            if (shasU(r)) { const double g = cU(r) - it.s[r]; sg += it.vU[r] / g; rh += mu / g; }
            sig_s[r] = sg;
            rhat_s[r] = rh;
            const double a1 = it.e1[r] / it.w1[r], a2 = it.e2[r] / it.w2[r];
            sig_t[r] = 1.0 / (1.0 / sg + a1 + a2);
            r_t[r] = row_res(E, it.s, it.e1, it.e2, r) + rh / sg - a1 * (it.y[r] - rho + mu / it.e1[r]) -
                     a2 * (it.y[r] + rho - mu / it.e2[r]);
        }
        // factor with inertia correction
        bool factored = false;
        double dw = 0.0, dc = 0.0;
        double dw_shift = 0.0;      // primal regularisation delta_w I of the inertia search (below)
        int shift_trials = 0;
        for (int attempt = 0; attempt < 24; ++attempt) {
            // The device factorisation is an LU and reports no inertia, so the matrix handed to the
            // backend has its inertia by construction: every node block
            //     Q_k = H_k + Sigma_k + sum_j sig_t g_j g_j^T
            // is made positive definite (negative eigenvalues reflected, Q~_k = Q_k + sum delta v v^T),
            // which makes K~ quasi-definite: exactly nz positive and md negative eigenvalues.  With
            // U = [v; 0] (r columns, one per modified eigenpair) the true matrix is K = K~ - U Delta U^T and
            //     inertia(K) = (nz - r, md, 0) + inertia(C),   C = Delta^-1 - U^T K~^-1 U   (r x r)
            // (both Schur complements of [[K~, U], [U^T, Delta^-1]]).  So r extra solves with the same
            // factors decide EXACTLY whether the unmodified K has the right inertia; if it has, the
            // Woodbury identity turns the solve with K~ into the exact Newton step (quadratic
            // convergence is kept); if not, the step of K~ is the inertia-corrected one.  Y = K~^-1 U and
            // the factor of C live with the backend (KktBackend::lowrank): on the device for eMI355X.
            const double* V = E.VALS.data();
            const auto tb0 = now();
            std::copy(E.H.begin(), E.H.end(), Qblk.begin());
            for (int v = 0; v < nv; ++v)
                for (int k = 0; k < M; ++k) {
                    const int qq = v * M + k;
                    double sg = 0.0;
                    if (fidx[qq] >= 0) {
                        if (hasL(qq)) sg += it.zL[qq] / (it.z[qq] - P.zl[qq]);
                        if (hasU(qq)) sg += it.zU[qq] / (P.zu[qq] - it.z[qq]);
                    }
                    Qblk[(size_t)(v * (v + 1) / 2 + v) * M + k] += sg + (fidx[qq] >= 0 ? dw_shift : 0.0);
                }
            // eliminated path rows: sum_j sig_j (grad c_j)(grad c_j)^T on the variables each row depends on
            for (int j = 0; j < np; ++j)
                for (size_t a = 0; a < rv[j].size(); ++a)
                    for (size_t b = 0; b <= a; ++b) {
                        const int va = rv[j][a].first, vb = rv[j][b].first;
                        const int hi = std::max(va, vb), lo = std::min(va, vb);
                        double* q = &Qblk[(size_t)(hi * (hi + 1) / 2 + lo) * M];
                        const double* ga = &V[(size_t)rv[j][a].second * M];
                        const double* gb = &V[(size_t)rv[j][b].second * M];
                        const double* sg = &sig_t[(size_t)j * M];
                        for (int k = 0; k < M; ++k) q[k] += sg[k] * ga[k] * gb[k];
                    }
            Qexact = Qblk;
            dw = convexify_node_blocks(Qblk.data(), fixed_mask.data(), nv, M, &mods);
            R.t_blocks += secs(tb0, now());
            const auto tf0 = now();
            const int info = kkt->factor(Qblk.data(), V, fixed_mask.data(), dc);
            R.t_factor += secs(tf0, now());
            ++R.n_factor;
            if (info < 0) { R.msg = "KKT factorisation failed: " + kkt->last_error(); return R; }
            if (info > 0) {   // exactly singular: the defect Jacobian lost rank; regularise the dual block
                dc = dc == 0.0 ? 1e-8 * std::pow(mu, 0.25) : dc * 100.0;
                continue;
            }
            // the low-rank correction (kept by the backend, next to its factors) and the inertia verdict
            r_mod = (int)mods.size() <= max_lowrank ? (int)mods.size() : 0;
            lr_node.resize(r_mod);
            lr_delta.resize(r_mod);
            lr_vec.resize((size_t)r_mod * nv);
            for (int c = 0; c < r_mod; ++c) {
                lr_node[c] = mods[c].node;
                lr_delta[c] = mods[c].delta;
                for (int v = 0; v < nv; ++v) lr_vec[(size_t)c * nv + v] = mods[c].v[v];
            }
            const auto tl0 = now();
            bool lr_exact = false;
            if (force_modified) r_mod = 0;      // the exact step failed the line search here: take the convexified one
            if (kkt->lowrank(r_mod, lr_node.data(), lr_vec.data(), lr_delta.data(), &lr_exact) != 0) {
                R.msg = "KKT low-rank correction failed: " + kkt->last_error();
                return R;
            }
            R.t_lowrank += secs(tl0, now());
            exact_step = !force_modified && lr_exact && r_mod == (int)mods.size();
            // Inertia search (IPOPT's delta_w): the unmodified K has the wrong inertia.  The step of the reflected
            // blocks is the cheap answer and usually a good one; where it stagnates (search_on, set below) look for
            // the smallest shift K + delta_w I_z whose inertia is right instead -- the verdict for every trial comes
            // from the same low-rank test, at the price of a factorisation each.
            if (search_on && !exact_step && !force_modified && !mods.empty() && r_mod == (int)mods.size() &&
                shift_trials < max_shift_trials) {
                if (dw_shift == 0.0) dw_shift = dw_last_ok == 0.0 ? 1e-4 : std::max(1e-20, dw_last_ok / 3.0);
                else dw_shift *= (dw_last_ok == 0.0 ? 100.0 : 8.0);
                ++shift_trials;
                continue;
            }
            if (exact_step && dw_shift > 0.0) dw_last_ok = dw_shift;
            for (int r = 0; r < me; ++r) eqres[r] = eqr(E, r);
            build_rhs(rhs_full.data(), eqres.data());
            std::copy(rhs_full.begin(), rhs_full.begin() + NN, rhs_keep.begin());
            const auto ts0 = now();
            // a backend that refines on its own (the device: residuals and corrections never leave HBM) ...
            double rel_dev = 0.0;
            int ns_dev = 0, rev_dev = 0;
            const int rr = kkt->solve_refined(rhs_full.data(), dc, 8, &rel_dev, &ns_dev, &rev_dev);
            if (rr == 0 || rr == 2) {
                R.t_solve += secs(ts0, now());
                R.n_solve += ns_dev;
                if (rr == 2) { dc = dc == 0.0 ? 1e-8 * std::pow(mu, 0.25) : dc * 100.0; continue; }
                bool fin = true;
                for (size_t r = 0; r < NN && fin; ++r) fin = std::isfinite(rhs_full[r]);
                if (!fin) { dc = dc == 0.0 ? 1e-8 * std::pow(mu, 0.25) : dc * 100.0; continue; }
                double dc_applied = dc, dw_applied = 0.0;
                kkt->applied_regularisation(&dc_applied, &dw_applied);
                const bool shifted = dw_applied > 0.0 || dc_applied > std::max(dc, 1e-9) * 1.0001;
                if (shifted) ++R.n_backend_shifted;
                R.n_refine_reverted += rev_dev;
                R.worst_step_residual = std::max(R.worst_step_residual, rel_dev);
                if (opt.print_level >= 5 && shifted && rel_dev > 1e-6)
                    printf("          backend factorised with dc %.1e dw %.1e; step used with relative residual %.2e\n", dc_applied, dw_applied, rel_dev);
                if (exact_step) dw = dw_shift;
                factored = true;
                break;
            }
            if (rr < 0) { R.msg = "KKT solve failed: " + kkt->last_error(); return R; }
            // ... otherwise: solve, then refine against the host's own matrix-vector product
            const int sst = kkt->solve(rhs_full.data(), 1);
            R.t_solve += secs(ts0, now());
            ++R.n_solve;
            if (sst != 0) { R.msg = "KKT solve failed: " + kkt->last_error(); return R; }
            bool finite = true;
            for (size_t r = 0; r < NN && finite; ++r) finite = std::isfinite(rhs_full[r]);
            if (!finite) { dc = dc == 0.0 ? 1e-8 * std::pow(mu, 0.25) : dc * 100.0; continue; }
            // iterative refinement against the matrix the step belongs to (K if exact, K~ otherwise): the
            // factorisation of a 1000-node KKT matrix leaves residuals that would stall the Newton
            // iteration some orders above the requested tolerance.  The backend may have factorised a MORE regularised
            // matrix than it was given (applied_regularisation: the Schur path's ladder); the refinement then is a
            // stationary iteration with (K + E)^-1 that need not contract, so a correction that makes the residual worse
            // is taken back, and the residual the step is finally used with is recorded (an inexact Newton step: the
            // line search below still decides; R.worst_step_residual lets a caller see how inexact).
            {
                const std::vector<double>& Qm = exact_step ? Qexact : Qblk;
                double dc_applied = dc, dw_applied = 0.0;
                kkt->applied_regularisation(&dc_applied, &dw_applied);
                const bool shifted = dw_applied > 0.0 || dc_applied > std::max(dc, 1e-9) * 1.0001;
                if (shifted) ++R.n_backend_shifted;
                double bmax = 0;
                for (size_t r = 0; r < NN; ++r) bmax = std::max(bmax, std::fabs(rhs_keep[r]));
                double prev = 1e300, rlast = 0.0;
                bool have_prev = false;
                for (int ir = 0; ir < 8; ++ir) {      // (8 since the backend may have regularised the factorised matrix: linear convergence)
                    { const auto tm = now(); kkt_matvec(Qm.data(), rhs_full.data(), resid.data(), dc); R.t_matvec += secs(tm, now()); }
                    double rmax = 0;
                    for (size_t r = 0; r < NN; ++r) { resid[r] = rhs_keep[r] - resid[r]; rmax = std::max(rmax, std::fabs(resid[r])); }
                    if (have_prev && !(rmax < prev)) {          // the last correction did harm: undo it and stop
                        for (size_t r = 0; r < NN; ++r) rhs_full[r] = x_prev[r];
                        ++R.n_refine_reverted;
                        rlast = prev;
                        break;
                    }
                    rlast = rmax;
                    if (!(rmax > 1e-14 * std::max(1.0, bmax)) || !(rmax < 0.5 * prev)) break;
                    prev = rmax;
                    { const auto ts = now(); const int rs = kkt->solve(resid.data(), 1); R.t_solve += secs(ts, now()); ++R.n_solve; if (rs != 0) break; }
                    bool fin = true;
                    for (size_t r = 0; r < NN && fin; ++r) fin = std::isfinite(resid[r]);
                    if (!fin) break;
                    x_prev.assign(rhs_full.begin(), rhs_full.begin() + NN);
                    have_prev = true;
                    for (size_t r = 0; r < NN; ++r) rhs_full[r] += resid[r];
                }
                const double rel = rlast / std::max(1.0, bmax);
                R.worst_step_residual = std::max(R.worst_step_residual, rel);
                if (opt.print_level >= 5 && shifted && rel > 1e-6)
                    printf("          backend factorised with dc %.1e dw %.1e; step used with relative residual %.2e\n", dc_applied, dw_applied, rel);
            }
            if (exact_step) dw = dw_shift;     // the log shows delta_w of an exact step, else the largest reflected shift
            factored = true;
            break;
        }
        if (!factored) { R.msg = "KKT matrix could not be factorised (still singular after dual regularisation)"; break; }
        dw_used = dw;
        last_step_reflected = !exact_step;
        if (exact_step && dw_shift == 0.0) search_on = false;      // the plain Newton matrix is fine again

        // the step in the eliminated quantities
        for (int q = 0; q < nz; ++q) dz[q] = fidx[q] >= 0 ? rhs_full[q] : 0.0;
        for (int r = 0; r < me; ++r) dlam[r] = rhs_full[nz + r];
        expand_step();
        // fraction to the boundary
        double apr = 1.0, adu = 1.0;
        auto step_lengths = [&]() {
            apr = 1.0;
            adu = 1.0;
            for (int q = 0; q < nz; ++q) {
                if (fidx[q] < 0) continue;
                if (hasL(q) && dz[q] < 0) apr = std::min(apr, -tau * (it.z[q] - P.zl[q]) / dz[q]);
                if (hasU(q) && dz[q] > 0) apr = std::min(apr, tau * (P.zu[q] - it.z[q]) / dz[q]);
                if (dzL[q] < 0) adu = std::min(adu, -tau * it.zL[q] / dzL[q]);
                if (dzU[q] < 0) adu = std::min(adu, -tau * it.zU[q] / dzU[q]);
            }
            for (int r = 0; r < mc; ++r) {
                if (shasL(r) && ds[r] < 0) apr = std::min(apr, -tau * (it.s[r] - cL(r)) / ds[r]);
                if (shasU(r) && ds[r] > 0) apr = std::min(apr, tau * (cU(r) - it.s[r]) / ds[r]);
                if (de1[r] < 0) apr = std::min(apr, -tau * it.e1[r] / de1[r]);
                if (de2[r] < 0) apr = std::min(apr, -tau * it.e2[r] / de2[r]);
                if (dvL[r] < 0) adu = std::min(adu, -tau * it.vL[r] / dvL[r]);
                if (dvU[r] < 0) adu = std::min(adu, -tau * it.vU[r] / dvU[r]);
                if (dw1[r] < 0) adu = std::min(adu, -tau * it.w1[r] / dw1[r]);
                if (dw2[r] < 0) adu = std::min(adu, -tau * it.w2[r] / dw2[r]);
            }
        };
        step_lengths();
        // l1 merit: directional derivative of the barrier function and the penalty weight
        double dphi = 0, infeas0 = 0;
        for (int q = 0; q < nz; ++q) {
            if (fidx[q] < 0) continue;
            double g = gradf[q];
            if (hasL(q)) g -= mu / (it.z[q] - P.zl[q]);
            if (hasU(q)) g += mu / (P.zu[q] - it.z[q]);
            dphi += g * dz[q];
        }
        for (int r = 0; r < mc; ++r) {
            double g = 0;
            if (shasL(r)) g -= mu / (it.s[r] - cL(r));
            if (shasU(r)) g += mu / (cU(r) - it.s[r]);
            dphi += g * ds[r] + (rho - mu / it.e1[r]) * de1[r] + (rho - mu / it.e2[r]) * de2[r];
        }
        const double phi0_base = barrier_merit(it.z, it.s, it.e1, it.e2, E, mu, 0.0, &infeas0);
        // penalty weight of the l1 merit function: what the current multipliers and the descent condition
        // ask for.  It may come down again (at most halving per iteration): the multipliers of the first,
        // far-from-feasible iterations are orders of magnitude above those near the solution, and a weight
        // frozen at that level rejects every step whose constraint curvature shows at all.
        double mmax = 0;
        for (int r = 0; r < me; ++r) mmax = std::max(mmax, std::fabs(it.lam[r] + dlam[r]) / rs[r]);
        for (int r = 0; r < mc; ++r) mmax = std::max(mmax, std::fabs(it.y[r] + dy[r]));
        double nu_want = std::max(1.0, std::min(1.1 * mmax, 1e8));
        if (infeas0 > 0) nu_want = std::max(nu_want, dphi / (0.9 * infeas0) + 1.0);
        nu = std::max(nu_want, 0.5 * nu);
        const double phi0 = phi0_base + nu * infeas0;
        const double slope = dphi - nu * infeas0;
        auto slack_reset = [&]() {
            // slack reset: a row's slack may jump to the value that closes its residual whenever
            // that lowers the merit function (the keep-out rows are strongly curved, and a step
            // along a keep-out boundary otherwise shows up as an equality residual c - s)
            for (int r = 0; r < mc; ++r) {
                const double target = Et.RES[(size_t)md + r] - e1t[r] + e2t[r];
                const double lo = shasL(r) ? cL(r) : -INF_BOUND, hi = shasU(r) ? cU(r) : INF_BOUND;
                if (!(target > lo) || !(target < hi)) continue;
                double keep = nu * std::fabs(target - st[r]), take = 0.0;
                if (shasL(r)) { keep -= mu * std::log(st[r] - lo); take -= mu * std::log(target - lo); }
                if (shasU(r)) { keep -= mu * std::log(hi - st[r]); take -= mu * std::log(hi - target); }
                if (take < keep) st[r] = target;
            }
        };
        // the iterate after a step of length a_pr (primal: zt, st, e1t, e2t hold the trial point) / a_du (bound
        // multipliers), re-evaluated with derivatives
        auto take_step = [&](double a_pr, double a_du) -> bool {
            it.z = zt;
            it.s = st;
            it.e1 = e1t;
            it.e2 = e2t;
            auto clampm = [&](double m, double g) { return std::max(std::min(m, kappa_sigma * mu / g), mu / (kappa_sigma * g)); };
            for (int r = 0; r < me; ++r) it.lam[r] += a_pr * dlam[r];
            for (int r = 0; r < mc; ++r) {
                it.y[r] += a_pr * dy[r];
                it.vL[r] += a_du * dvL[r];
                it.vU[r] += a_du * dvU[r];
                it.w1[r] += a_du * dw1[r];
                it.w2[r] += a_du * dw2[r];
                if (shasL(r)) it.vL[r] = clampm(it.vL[r], it.s[r] - cL(r));
                if (shasU(r)) it.vU[r] = clampm(it.vU[r], cU(r) - it.s[r]);
                it.w1[r] = clampm(it.w1[r], it.e1[r]);
                it.w2[r] = clampm(it.w2[r], it.e2[r]);
            }
            for (int q = 0; q < nz; ++q) {
                if (fidx[q] < 0) continue;
                it.zL[q] += a_du * dzL[q];
                it.zU[q] += a_du * dzU[q];
                if (hasL(q)) it.zL[q] = clampm(it.zL[q], it.z[q] - P.zl[q]);
                if (hasU(q)) it.zU[q] = clampm(it.zU[q], P.zu[q] - it.z[q]);
            }
            if (!evaluate(it.z, E, true)) return false;
            { const auto tj = now(); grad_and_jt(it); R.t_jt += secs(tj, now()); }
            return true;
        };
        // backtracking
        double alpha = apr;
        bool accepted = false;
        bool newton_accepted = false;
        for (int ls = 0; ls < 40; ++ls) {
            for (int q = 0; q < nz; ++q) zt[q] = it.z[q] + alpha * dz[q];
            for (int r = 0; r < mc; ++r) {
                st[r] = it.s[r] + alpha * ds[r];
                e1t[r] = it.e1[r] + alpha * de1[r];
                e2t[r] = it.e2[r] + alpha * de2[r];
            }
            if (!evaluate(zt, Et, false)) { R.msg = "evaluator failed: " + P.ev->last_error(); return R; }
            slack_reset();
            const double phit = barrier_merit(zt, st, e1t, e2t, Et, mu, nu, nullptr);
            if (std::isfinite(phit) && phit <= phi0 + 1e-4 * alpha * std::min(slope, 0.0) + 1e-13 * std::fabs(phi0)) {
                accepted = true;
                break;
            }
            // Second-order correction (as in IPOPT's line search): the first trial point was rejected
            // and is less feasible than the current one -- the curvature of the constraints, not the
            // direction, is to blame (Maratos effect; without this the l1 merit function lets the
            // iteration crawl along the strongly curved defect / keep-out rows with steps of 2^-8).
            // Re-solve with the SAME factorisation for the constraint values seen at the trial point.
            if (ls == 0) {
                double infeas_t = 0;
                barrier_merit(zt, st, e1t, e2t, Et, mu, 0.0, &infeas_t);
                if (std::isfinite(infeas_t) && infeas_t >= infeas0) {
                    const std::vector<double> b_dz = dz, b_ds = ds, b_de1 = de1, b_de2 = de2, b_dlam = dlam, b_dy = dy,
                                              b_dzL = dzL, b_dzU = dzU, b_dvL = dvL, b_dvU = dvU, b_dw1 = dw1, b_dw2 = dw2,
                                              b_rt = r_t;
                    const double b_apr = apr, b_adu = adu;
                    for (int r = 0; r < me; ++r) soc_def[r] = alpha * eqr(E, r) + eqr(Et, r);
                    for (int r = 0; r < mc; ++r)
                        soc_row[r] = alpha * row_res(E, it.s, it.e1, it.e2, r) + row_res(Et, st, e1t, e2t, r);
                    double infeas_old = infeas_t;
                    for (int pc = 0; pc < 4 && !accepted; ++pc) {
                        fill_rt(soc_row);
                        build_rhs(soc_rhs.data(), soc_def.data());
                        if (kkt->solve(soc_rhs.data(), 1) != 0) break;
                        bool fin = true;
                        for (size_t r = 0; r < NN && fin; ++r) fin = std::isfinite(soc_rhs[r]);
                        if (!fin) break;
                        for (int q = 0; q < nz; ++q) dz[q] = fidx[q] >= 0 ? soc_rhs[q] : 0.0;
                        for (int r = 0; r < me; ++r) dlam[r] = soc_rhs[nz + r];
                        expand_step();
                        step_lengths();
                        const double asoc = apr;
                        for (int q = 0; q < nz; ++q) zt[q] = it.z[q] + asoc * dz[q];
                        for (int r = 0; r < mc; ++r) {
                            st[r] = it.s[r] + asoc * ds[r];
                            e1t[r] = it.e1[r] + asoc * de1[r];
                            e2t[r] = it.e2[r] + asoc * de2[r];
                        }
                        if (!evaluate(zt, Et, false)) { R.msg = "evaluator failed: " + P.ev->last_error(); return R; }
                        slack_reset();
                        double infeas_s = 0;
                        const double phis = barrier_merit(zt, st, e1t, e2t, Et, mu, nu, &infeas_s);
                        if (std::isfinite(phis) && phis <= phi0 + 1e-4 * asoc * std::min(slope, 0.0) + 1e-13 * std::fabs(phi0)) {
                            accepted = true;
                            alpha = asoc;
                            ++R.soc_steps;
                            break;
                        }
                        if (!std::isfinite(infeas_s) || infeas_s > 0.99 * infeas_old) break;
                        infeas_old = infeas_s;
                        for (int r = 0; r < me; ++r) soc_def[r] = asoc * soc_def[r] + eqr(Et, r);
                        for (int r = 0; r < mc; ++r) soc_row[r] = asoc * soc_row[r] + row_res(Et, st, e1t, e2t, r);
                    }
                    if (accepted) break;
                    dz = b_dz; ds = b_ds; de1 = b_de1; de2 = b_de2; dlam = b_dlam; dy = b_dy; dzL = b_dzL; dzU = b_dzU;
                    dvL = b_dvL; dvU = b_dvU; dw1 = b_dw1; dw2 = b_dw2; r_t = b_rt;
                    apr = b_apr; adu = b_adu;
                }
            }
            // Close to a solution the merit function stops resolving progress: the full Newton step changes it
            // by less than constraint curvature and round-off move it, and the backtracking then crawls with steps
            // of 1e-6 for a hundred iterations.  There -- and wherever the line search has just cut three steps in a
            // row below 30 % of the longest admissible step (`crawl`: the same effect further out, with a penalty
            // weight the far-from-feasible start left behind) -- the KKT residual itself is the better judge: take
            // the full step if it reduces the residual of the current barrier problem (else undo and backtrack).
            if (ls == 0 && (err0 <= 1e-2 || crawl >= crawl_limit)) {
                const double err_mu = kkt_error(it, mu, nullptr, nullptr);
                const Iterate it_keep = it;
                const std::vector<double> gradf_keep = gradf, jtl_keep = jtl;
                Ekeep.RES = E.RES; Ekeep.VALS = E.VALS; Ekeep.LNK = E.LNK; Ekeep.cost = E.cost;
                for (int q = 0; q < nz; ++q) zt[q] = it.z[q] + apr * dz[q];
                for (int r = 0; r < mc; ++r) {
                    st[r] = it.s[r] + apr * ds[r];
                    e1t[r] = it.e1[r] + apr * de1[r];
                    e2t[r] = it.e2[r] + apr * de2[r];
                }
                if (!take_step(apr, adu)) { R.msg = "evaluator failed: " + P.ev->last_error(); return R; }
                const double err_tr = kkt_error(it, mu, nullptr, nullptr);
                if (std::isfinite(err_tr) && err_tr <= 0.9 * err_mu) {
                    newton_accepted = true;
                    ++R.newton_steps;
                    break;
                }
                it = it_keep;
                gradf = gradf_keep;
                jtl = jtl_keep;
                E.RES = Ekeep.RES; E.VALS = Ekeep.VALS; E.LNK = Ekeep.LNK; E.cost = Ekeep.cost;
            }
            alpha *= 0.5;
        }
        if (opt.print_level >= 6)
            printf("          apr %.3e  alpha %.3e  adu %.3e  dphi %.3e  infeas1 %.3e  slope %.3e\n", apr, alpha, adu, dphi,
                   infeas0, slope);
        if (newton_accepted) {       // the iterate is already updated and re-evaluated
            crawl = 0;
            force_modified = false;
            continue;
        }
        if (!accepted && exact_step && !mods.empty() && !force_modified) {
            // the exact Newton direction is not a descent direction the merit function accepts at this point:
            // redo the iteration with the step of the convexified matrix (a descent direction by construction)
            force_modified = true;
            continue;
        }
        force_modified = false;
        if (!accepted) {
            R.msg = "line search failed";
            // IPOPT's rule for a search that can go no further: the point is a solution only if it meets the
            // acceptable level (the same acceptable_factor as the iteration-count rule above); otherwise the
            // failure is reported, with kkt_error / constr_viol for the caller to judge
            if (err0 <= opt.acceptable_factor * opt.tol && (mc == 0 || emax <= 1e-6)) {
                R.ok = true;
                R.msg = "converged to acceptable level (line search at round-off)";
            }
            break;
        }
        // accept
        const double crawl_frac = opt.crawl_frac;
        crawl = alpha < crawl_frac * apr ? crawl + 1 : 0;
        if (!take_step(alpha, adu)) { R.msg = "evaluator failed: " + P.ev->last_error(); return R; }
    }
    R.cost = E.cost;
    R.rho = rho;
    R.t_total = secs(tstart, now());
    if (opt.print_level >= 5)
        printf("time: total %.2f s = evaluator %.2f + KKT factor %.2f (%d factorisations) + KKT solves %.2f (%d calls) + low-rank/refinement (host) %.2f + rest"
               " (of the host part: J^T lambda %.2f, refinement matvecs %.2f, node blocks %.2f; Hessian calls %.2f)\n",
               R.t_total, R.t_eval, R.t_factor, R.n_factor, R.t_solve, R.n_solve, R.t_lowrank, R.t_jt, R.t_matvec, R.t_blocks, R.t_hess);
    R.z = it.z;
    R.lamF.assign(it.lam.begin(), it.lam.begin() + md);
    R.lamL.assign(it.lam.begin() + md, it.lam.end());
    R.lamC.resize(mc);
    for (int r = 0; r < mc; ++r) R.lamC[r] = sig[r / M] * it.y[r];
    return R;
}

}  // namespace mi355x
}  // namespace ETOL
