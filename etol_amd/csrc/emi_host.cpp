// emi_host.cpp -- host-only parts of the C ABI: mesh construction and the
// per-problem constants the kernels consume.  No device code here.
//
// Legendre-Gauss-Lobatto transcription: PSOPT 5.0.0's collocation_method =
// "Legendre" (reference src/ePSOPT/ePSOPT.cpp:68) places the nodes at the
// zeros of (1-tau^2) P'_N(tau), N = nodes-1 (ePSOPT.cpp:44-45), which include
// the end points that ePSOPT::events reads (ePSOPT.cpp:281-291).  PSOPT's
// sources are not part of the reference tree, so the construction below is the
// textbook one, written from the defining formulas.
#include <cmath>
#include <cstring>
#include <vector>

#include "emi355x.h"

namespace {

// P_N(x) and P_{N-1}(x) by the three-term recurrence, in extended precision.
inline void legendre_pair(int N, long double x, long double* pn, long double* pnm1) {
    long double p0 = 1.0L, p1 = x;
    if (N == 0) { *pn = 1.0L; *pnm1 = 0.0L; return; }
    for (int n = 1; n < N; ++n) {
        const long double p2 = ((2 * n + 1) * x * p1 - n * p0) / (n + 1);
        p0 = p1;
        p1 = p2;
    }
    *pn = p1;
    *pnm1 = p0;
}

}  // namespace

extern "C" {

int emi_abi_version(void) { return EMI_ABI_VERSION; }

const char* emi_status_string(int status) {
    switch (status) {
        case EMI_OK: return "ok";
        case EMI_ERR_ARG: return "invalid argument";
        case EMI_ERR_STATE: return "call out of order";
        case EMI_ERR_HIP: return "HIP runtime error";
        case EMI_ERR_NO_DEVICE: return "no gfx950 device";
        case EMI_ERR_UNSUPPORTED: return "unsupported in this build";
        case EMI_ERR_COMM: return "RCCL error";
    }
    return "unknown status";
}

int emi_lgl(int M, double* tau, double* w, double* D) {
    if (M < 2 || !tau || !w) return EMI_ERR_ARG;
    const int N = M - 1;
    const long double pi = 3.14159265358979323846264338327950288L;
    std::vector<long double> x(M), pN(M);
    x[0] = -1.0L;
    x[N] = 1.0L;
    // interior zeros of P'_N by Newton on q(x) = (1-x^2) P'_N = N (P_{N-1} - x P_N),
    // q'(x) = -N (N+1) P_N  (Legendre's equation), from Chebyshev-Lobatto guesses
    for (int k = 1; k <= N / 2; ++k) {
        long double xk = -cosl(pi * k / N);
        for (int it = 0; it < 100; ++it) {
            long double pn, pnm1;
            legendre_pair(N, xk, &pn, &pnm1);
            const long double dx = (xk * pn - pnm1) / ((N + 1) * pn);
            xk -= dx;
            if (fabsl(dx) <= 4.0L * 1.0842021724855044e-19L * fmaxl(1.0L, fabsl(xk))) break;
        }
        x[k] = xk;
        x[N - k] = -xk;  // the node set is symmetric about 0
    }
    if (N % 2 == 0) x[N / 2] = 0.0L;
    for (int k = 0; k < M; ++k) {
        long double pn, pnm1;
        legendre_pair(N, x[k], &pn, &pnm1);
        pN[k] = pn;
        tau[k] = (double)x[k];
        w[k] = (double)(2.0L / ((long double)N * (N + 1) * pn * pn));
    }
    if (D) {
        // D_ij = P_N(x_i) / (P_N(x_j) (x_i - x_j)), i != j;  the diagonal is the
        // negative row sum so that D.1 = 0 holds to rounding (the closed-form
        // diagonal already loses 1e-6 at N = 255).
        for (int i = 0; i < M; ++i) {
            long double rs = 0.0L;
            for (int j = 0; j < M; ++j) {
                if (i == j) continue;
                const long double d = pN[i] / (pN[j] * (x[i] - x[j]));
                D[(size_t)i * M + j] = (double)d;
                rs += (long double)D[(size_t)i * M + j];
            }
            D[(size_t)i * M + i] = (double)(-rs);
        }
        // The node set is symmetric, so D is centro-antisymmetric: D[N-i][N-j] = -D[i][j].
        // The off-diagonal entries satisfy this bit for bit; make the summed diagonal do so
        // too (the kernels split D.X into even and odd halves when it holds exactly).
        for (int i = 0; i < M / 2; ++i) D[(size_t)(N - i) * M + (N - i)] = -D[(size_t)i * M + i];
        if (M % 2 == 1) D[(size_t)(N / 2) * M + N / 2] = 0.0;
    }
    return EMI_OK;
}

int emi_model_dims(int model, int* ns, int* nc, int* nparams) {
    int s, c, p;
    switch (model) {
        case EMI_MODEL_POINTMASS2D: s = 2; c = 2; p = 0; break;
        case EMI_MODEL_QUADROTOR2D: s = 6; c = 2; p = 5; break;
        case EMI_MODEL_FIXEDWING12: s = 12; c = 4; p = 16; break;
        default: return EMI_ERR_ARG;
    }
    if (ns) *ns = s;
    if (nc) *nc = c;
    if (nparams) *nparams = p;
    return EMI_OK;
}

// One polygon edge a->b as an ellipse keep-out record.  The reference evaluates
// these constants inside the node loop (etol_psopt_example1.cpp:163-176); they
// depend only on the XML, so they are computed once here -- same operations,
// same order: centre (midpoint in x, point on the edge's line in y), squared
// half-length, rotation angle, b^2 = 0.2 a^2.  A vertical edge divides by zero
// exactly as the reference does (:169).
int emi_edge_ellipse(double xa, double ya, double xb, double yb, double* rec8) {
    if (!rec8) return EMI_ERR_ARG;
    const double xc = (xb + xa) / 2.;
    const double m = (yb - ya) / (xb - xa);
    const double yc = ya + m * (xc - xa);
    const double radsq = std::pow(xc - xa, 2.0) + std::pow(yc - ya, 2.0);
    const double tt = -1.0 * std::atan2(yc - ya, xc - xa);
    rec8[0] = (double)EMI_PATH_ELLIPSE;
    rec8[1] = xc;
    rec8[2] = yc;
    rec8[3] = std::cos(tt);
    rec8[4] = std::sin(tt);
    rec8[5] = radsq;
    rec8[6] = .2 * radsq;
    rec8[7] = 0.0;
    return EMI_OK;
}

// Waypoint table -> disc centre at each node time.  Bracket rule of the
// reference's linear_interpolation (TrajectoryOptimizer.hpp:239-258): before
// the table use segment 0, after it the last segment, inside it the LAST
// segment whose closed interval contains t.
int emi_track_centres(int nway, const double* t, const double* x, const double* y, int M,
                      const double* node_t, double* xc, double* yc) {
    if (nway < 2 || !t || !x || !y || M < 1 || !node_t || !xc || !yc) return EMI_ERR_ARG;
    for (int k = 0; k < M; ++k) {
        const double tv = node_t[k];
        int j = 0;
        if (tv > t[nway - 1]) {
            j = nway - 2;
        } else if (tv >= t[0]) {
            for (int s = 0; s + 1 < nway; ++s)
                if (tv >= t[s] && tv <= t[s + 1]) j = s;
        }
        const double dt = t[j + 1] - t[j];
        xc[k] = (tv - t[j]) * (x[j + 1] - x[j]) / dt + x[j];
        yc[k] = (tv - t[j]) * (y[j + 1] - y[j]) / dt + y[j];
    }
    return EMI_OK;
}

}  // extern "C"
