/*
 * emi_oracle.c -- CPU restatement of the collocation hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's library; nothing under etol_amd/ or include/ does.
 *
 * PARITY STATUS, precisely (tests/golden/README.md):
 *   PINNED BY REFERENCE-EXECUTED VECTORS (oracle/_ref, built from the reference's own sources where they lie):
 *     the node callbacks of the shipped problem -- objFunction, dxConstraint, dyConstraint, obsConstraint,
 *     saaConstraint of src/Examples/Dymos/etol_dymos_example1.cpp, values and partials -- and the waypoint
 *     interpolation of include/ETOL/TrajectoryOptimizer.hpp:239-258 (orc_track_centres is bit-exact with it):
 *     tests/golden/ref_dymos_ex1.json, ref_interp.json, ref_traj.json, tests/test_ref_vectors.py.
 *   PINNED BY A RESTATEMENT ONLY: the per-edge constants (xc, yc, radsq, tt) that orc_edge_ellipse computes follow
 *     setExz (etol_dymos_example1.cpp:316-326), which needs CGAL-linked objects and cannot run here; the generator
 *     restates its four formulas in Python (tests/golden/gen_ref_vectors.py:63-71), so orc_edge_ellipse is checked
 *     against that restatement and against the reference callbacks FED with it, not against setExz output.
 *   PINNED BY CLOSED FORMS ONLY: LGL nodes / weights / D, defect D.X - h F, quadrature (PSOPT 5.0.0 is not in the
 *     reference tree): 50-digit mpmath values and invariants; the build's quadrotor / fixed-wing models: sympy.
 *   UNPINNED: ePSOPT's solved trajectories (PSOPT / ADOL-C / IPOPT absent, ordinary missing dependencies: SURVEY.md
 *     section 8c); the analytic optimum of the obstacle-free shipped problem and an independent CPU optimiser
 *     (tests/indep_nlp.py) stand in.
 *
 * What is restated, and from where (paths in the reference tree):
 *   node loop, evaluation order     src/ePSOPT/ePSOPT.cpp:218-276 (dae),
 *                                   :186-216 (integrand_cost, sign :212-213)
 *   2-state node functions          src/Examples/PSOPT/etol_psopt_example1.cpp
 *                                   :101-114 (L), :116-138 (xdot, ydot)
 *   ellipse keep-out per edge       etol_psopt_example1.cpp:163-182
 *   moving-disc keep-out            etol_psopt_example1.cpp:243-247
 *   waypoint interpolation          include/ETOL/TrajectoryOptimizer.hpp:239-258
 *   horizon / node count            ePSOPT.cpp:44-45, :151-154
 *   LGL nodes, weights, D, defect D.X - h.F, cost h.sum w L : PSOPT's Legendre
 *   transcription selected at ePSOPT.cpp:68; PSOPT is not in the tree, so these
 *   follow the published definitions (Legendre's equation, Lobatto quadrature).
 *
 * Deliberately a DIFFERENT route from the product code: first derivatives come
 * from the complex-step method on the restated functions (no hand-derived
 * Jacobian here), the D.X product is accumulated in long double, and second
 * derivatives are central differences of complex-step gradients.
 */
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef double complex cplx;

enum { M_POINTMASS = 0, M_QUADROTOR = 1, M_FIXEDWING = 2, M_DELAYDEMO = 3 };
enum { P_ELLIPSE = 0, P_DISC = 1, P_TRACK = 2 };
#define REC 8

static int model_dims(int model, int* ns, int* nc) {
    switch (model) {
        case M_POINTMASS: *ns = 2; *nc = 2; return 0;
        case M_QUADROTOR: *ns = 6; *nc = 2; return 0;
        case M_FIXEDWING: *ns = 12; *nc = 4; return 0;
        /* 2 states, 2 controls, state horizon 3, control horizon 1 (ePSOPT.cpp:231-248): the node functions see
         * z = [x0 x1 | u0 u1 | x(t-dt) | x(t-2dt) | u(t-dt)], i.e. 8 "controls" of which the last 6 are delayed values */
        case M_DELAYDEMO: *ns = 2; *nc = 8; return 0;
    }
    return 1;
}

/* ---- LGL ------------------------------------------------------------------ */
static void leg(int N, long double x, long double* pn, long double* dpn) {
    /* P_N and P'_N by recurrence; P'_N from (x^2-1) P'_N = N (x P_N - P_{N-1}) */
    long double p0 = 1.0L, p1 = x;
    if (N == 0) { *pn = 1; *dpn = 0; return; }
    for (int n = 1; n < N; ++n) {
        long double p2 = ((2 * n + 1) * x * p1 - n * p0) / (n + 1);
        p0 = p1; p1 = p2;
    }
    *pn = p1;
    *dpn = (fabsl(x) == 1.0L) ? 0.0L : N * (x * p1 - p0) / (x * x - 1.0L);
}

int orc_lgl(int M, double* tau, double* w, double* D) {
    if (M < 2) return 1;
    const int N = M - 1;
    long double* x = (long double*)malloc(sizeof(long double) * M);
    long double* p = (long double*)malloc(sizeof(long double) * M);
    const long double pi = acosl(-1.0L);
    x[0] = -1.0L; x[N] = 1.0L;
    /* zeros of P'_N: Newton with P''_N from Legendre's equation
       (1-x^2) P'' = 2 x P' - N(N+1) P */
    for (int k = 1; k < N; ++k) {
        long double xk = -cosl(pi * k / N);
        for (int it = 0; it < 200; ++it) {
            long double pn, dpn;
            leg(N, xk, &pn, &dpn);
            long double d2 = (2 * xk * dpn - (long double)N * (N + 1) * pn) / (1 - xk * xk);
            long double dx = dpn / d2;
            xk -= dx;
            if (fabsl(dx) < 1e-19L) break;
        }
        x[k] = xk;
    }
    for (int k = 0; k < M; ++k) {
        long double dpn;
        leg(N, x[k], &p[k], &dpn);
        tau[k] = (double)x[k];
        w[k] = (double)(2.0L / ((long double)N * (N + 1) * p[k] * p[k]));
    }
    if (D) {
        for (int i = 0; i < M; ++i) {
            long double rs = 0;
            for (int j = 0; j < M; ++j) {
                if (i == j) continue;
                long double d = (p[i] / p[j]) / (x[i] - x[j]);
                D[(size_t)i * M + j] = (double)d;
                rs += d;
            }
            D[(size_t)i * M + i] = (double)(-rs);
        }
    }
    free(x); free(p);
    return 0;
}

/* ---- delayed values ---------------------------------------------------------- */
/* ePSOPT::dae appends get_delayed_state / get_delayed_control values (ePSOPT.cpp:231-248).  PSOPT 5.0.0 is not in the
 * reference tree; per its published description the delayed value is the collocation polynomial (Lagrange interpolation
 * through the node values for "Legendre" collocation) evaluated at t - delay.  Times before t0 are clamped to t0 (an
 * assumption of this build, the same one as the product: include/emi355x.h).  Route deliberately different from the
 * product's barycentric form: the plain product formula of the Lagrange basis, in long double.
 * W[k][j] = l_j(x_k*),  x_k* = node coordinate of max(t_k - delay, t0).                                                  */
int orc_delay_matrix(int M, const double* tau, double t0, double tf, double delay, double* W) {
    if (M < 2 || !(tf > t0) || delay < 0) return 1;
    const long double hh = ((long double)tf - t0) / 2;
    for (int k = 0; k < M; ++k) {
        long double ts = t0 + hh * ((long double)tau[k] + 1) - delay;
        if (ts < t0) ts = t0;
        const long double x = (ts - t0) / hh - 1;
        for (int j = 0; j < M; ++j) {
            long double l = 1;
            for (int m = 0; m < M; ++m)
                if (m != j) l *= (x - tau[m]) / ((long double)tau[j] - tau[m]);
            W[(size_t)k * M + j] = (double)l;
        }
    }
    return 0;
}

/* ---- keep-out constants ----------------------------------------------------- */
/* etol_psopt_example1.cpp:163-176, operation for operation */
void orc_edge_ellipse(double xa, double ya, double xb, double yb, double* rec) {
    double xc = (xb + xa) / 2.;
    double m = (yb - ya) / (xb - xa);
    double yc = ya + m * (xc - xa);
    double radsq = pow(xc - xa, 2.0) + pow(yc - ya, 2.0);
    double tt = -1.0 * atan2(yc - ya, xc - xa);
    rec[0] = P_ELLIPSE; rec[1] = xc; rec[2] = yc; rec[3] = cos(tt); rec[4] = sin(tt);
    rec[5] = radsq; rec[6] = .2 * radsq; rec[7] = 0;
}

/* TrajectoryOptimizer.hpp:239-258 */
static double lin_interp(double tval, int n, const double* tv, const double* ref) {
    int j = 0;
    if (tval > tv[n - 1]) {
        j = n - 2;
    } else if (tval >= tv[0]) {
        for (int s = 0; s + 1 < n; ++s)
            if (tval >= tv[s] && tval <= tv[s + 1]) j = s;
    }
    return (tval - tv[j]) * (ref[j + 1] - ref[j]) / (tv[j + 1] - tv[j]) + ref[j];
}

void orc_track_centres(int nway, const double* t, const double* x, const double* y, int M,
                       const double* node_t, double* xc, double* yc) {
    for (int k = 0; k < M; ++k) {
        xc[k] = lin_interp(node_t[k], nway, t, x);
        yc[k] = lin_interp(node_t[k], nway, t, y);
    }
}

/* ---- node functions on complex arguments ------------------------------------ */
static void dyn(int model, const double* p, const cplx* z, cplx* f) {
    if (model == M_POINTMASS) {
        /* dxdt returns u0, dydt returns u1 */
        f[0] = z[2];
        f[1] = z[3];
    } else if (model == M_DELAYDEMO) {
        /* build-defined test dynamics on delayed inputs (the reference ships no model with a horizon above 1) */
        f[0] = -p[0] * z[4] + z[2] + 0.1 * z[9] * z[1];
        f[1] = z[0] * z[7] - csin(z[5]) + z[3] * z[8];
    } else if (model == M_QUADROTOR) {
        const double m = p[0], Jy = p[1], g = p[2];
        f[0] = z[3];
        f[1] = z[4];
        f[2] = z[5];
        f[3] = -(z[6] / m) * csin(z[2]);
        f[4] = (z[6] / m) * ccos(z[2]) - g;
        f[5] = z[7] / Jy;
    } else {
        const double m = p[0], Ixx = p[1], Iyy = p[2], Izz = p[3], g = p[4], qS = p[5];
        const double CL0 = p[6], CLa = p[7], CD0 = p[8], CDk = p[9];
        const double Clda = p[10], Cmde = p[11], Cndr = p[12], V = p[13], damp = p[14];
        const cplx ph = z[3], th = z[4], ps = z[5], u = z[6], v = z[7], w = z[8];
        const cplx pr = z[9], qr = z[10], rr = z[11];
        const cplx sph = csin(ph), cph = ccos(ph), sth = csin(th), cth = ccos(th);
        const cplx sps = csin(ps), cps = ccos(ps);
        f[0] = cth * cps * u + (sph * sth * cps - cph * sps) * v + (cph * sth * cps + sph * sps) * w;
        f[1] = cth * sps * u + (sph * sth * sps + cph * cps) * v + (cph * sth * sps - sph * cps) * w;
        f[2] = -sth * u + sph * cth * v + cph * cth * w;
        f[3] = pr + (sth / cth) * (sph * qr + cph * rr);
        f[4] = cph * qr - sph * rr;
        f[5] = (sph * qr + cph * rr) / cth;
        const cplx al = w / V;
        const cplx CL = CL0 + CLa * al;
        const cplx CD = CD0 + CDk * CL * CL;
        f[6] = rr * v - qr * w - g * sth + (z[12] - qS * CD) / m;
        f[7] = pr * w - rr * u + g * sph * cth + (-damp * v) / m;
        f[8] = qr * u - pr * v + g * cph * cth + (-qS * CL) / m;
        f[9] = ((Iyy - Izz) * qr * rr + (qS * Clda * z[13] - damp * pr)) / Ixx;
        f[10] = ((Izz - Ixx) * pr * rr + (qS * Cmde * z[14] - damp * qr)) / Iyy;
        f[11] = ((Ixx - Iyy) * pr * qr + (qS * Cndr * z[15] - damp * rr)) / Izz;
    }
}

static cplx lag(int model, const double* p, const cplx* z) {
    if (model == M_POINTMASS) return z[2] * z[2] + z[3] * z[3]; /* objFunction :108 */
    if (model == M_QUADROTOR) return p[3] * z[6] * z[6] + p[4] * z[7] * z[7];
    if (model == M_DELAYDEMO) return z[2] * z[2] + z[3] * z[3] + p[1] * z[4] * z[6] + 0.05 * z[8] * z[8];
    return p[15] * (z[12] * z[12] + z[13] * z[13] + z[14] * z[14] + z[15] * z[15]);
}

/* one keep-out row; (xc,yc) of a track row is passed in */
static cplx keepout(const double* r, cplx x, cplx y, double txc, double tyc) {
    const int kind = (int)r[0];
    if (kind == P_ELLIPSE) { /* :174-182 */
        const double xc = r[1], yc = r[2], ct = r[3], st = r[4], asq = r[5], bsq = r[6];
        cplx dx = x - xc, dy = y - yc;
        cplx delx = ct * dx - st * dy;
        cplx dely = st * dx + ct * dy;
        return asq * bsq - (bsq * delx * delx + asq * dely * dely);
    }
    if (kind == P_DISC) { /* :243-247 with a fixed centre */
        cplx dx = x - r[1], dy = y - r[2];
        cplx dist = dx * dx + dy * dy;
        return dist * (-1.) + r[3];
    }
    { /* P_TRACK :243-247 */
        cplx dx = x - txc, dy = y - tyc;
        cplx dist = dx * dx + dy * dy;
        return dist * (-1.) + r[2];
    }
}

/* ---- one evaluation pass ------------------------------------------------------ */
/* Layouts as in include/emi355x.h. tau only fixes the node times.               */
int orc_eval(int model, const double* params, int maximize, int M, int B, const double* tau,
             const double* w, const double* D, double t0, double tf, int np, int path_sets,
             const double* recs, int px, int py, int ntracks, int track_sets, const double* trkx,
             const double* trky, const double* X, const double* U, double* RES, double* VALS,
             double* COST) {
    int ns, nc;
    if (model_dims(model, &ns, &nc)) return 1;
    const int nv = ns + nc, nres = ns + np, nvals = ns * nv + 2 * np + nv;
    const double h = (tf - t0) / 2.0, sgn = maximize ? -1.0 : 1.0, cs = 1e-30;
    (void)tau;
    for (int b = 0; b < B; ++b) {
        const double* Xb = X + (size_t)b * ns * M;
        const double* Ub = U + (size_t)b * nc * M;
        double* Rb = RES + (size_t)b * nres * M;
        double* Vb = VALS ? VALS + (size_t)b * nvals * M : 0;
        const double* rec = np ? recs + (size_t)(path_sets > 1 ? b : 0) * np * REC : 0;
        long double cost = 0;
        for (int k = 0; k < M; ++k) {
            cplx z[16], f[12];
            for (int i = 0; i < ns; ++i) z[i] = Xb[(size_t)i * M + k];
            for (int c = 0; c < nc; ++c) z[ns + c] = Ub[(size_t)c * M + k];
            /* dae: derivatives, then the path rows (ePSOPT.cpp:252-270) */
            dyn(model, params, z, f);
            for (int i = 0; i < ns; ++i) {
                long double dx = 0;
                for (int j = 0; j < M; ++j) dx += (long double)D[(size_t)k * M + j] * Xb[(size_t)i * M + j];
                Rb[(size_t)i * M + k] = (double)(dx - (long double)h * creal(f[i]));
            }
            for (int j = 0; j < np; ++j) {
                double txc = 0, tyc = 0;
                if ((int)rec[j * REC] == P_TRACK) {
                    size_t off = ((size_t)(track_sets > 1 ? b : 0) * ntracks + (int)rec[j * REC + 1]) * M + k;
                    txc = trkx[off]; tyc = trky[off];
                }
                Rb[(size_t)(ns + j) * M + k] = creal(keepout(rec + j * REC, z[px], z[py], txc, tyc));
            }
            /* integrand_cost (ePSOPT.cpp:199-213) */
            cost += (long double)w[k] * creal(lag(model, params, z));
            if (!Vb) continue;
            /* first derivatives by complex step, one variable at a time */
            for (int v = 0; v < nv; ++v) {
                cplx zz[16];
                memcpy(zz, z, sizeof(cplx) * nv);
                zz[v] += I * cs;
                dyn(model, params, zz, f);
                for (int i = 0; i < ns; ++i)
                    Vb[(size_t)(i * nv + v) * M + k] =
                        -h * (cimag(f[i]) / cs) + (v == i ? D[(size_t)k * M + k] : 0.0);
                Vb[(size_t)(ns * nv + 2 * np + v) * M + k] = sgn * h * w[k] * (cimag(lag(model, params, zz)) / cs);
            }
            for (int j = 0; j < np; ++j) {
                double txc = 0, tyc = 0;
                if ((int)rec[j * REC] == P_TRACK) {
                    size_t off = ((size_t)(track_sets > 1 ? b : 0) * ntracks + (int)rec[j * REC + 1]) * M + k;
                    txc = trkx[off]; tyc = trky[off];
                }
                Vb[(size_t)(ns * nv + 2 * j) * M + k] =
                    cimag(keepout(rec + j * REC, z[px] + I * cs, z[py], txc, tyc)) / cs;
                Vb[(size_t)(ns * nv + 2 * j + 1) * M + k] =
                    cimag(keepout(rec + j * REC, z[px], z[py] + I * cs, txc, tyc)) / cs;
            }
        }
        if (COST) COST[b] = (double)(sgn * h * cost);
    }
    return 0;
}

/* gradient of the node Lagrangian term by complex step */
static void lag_grad(int model, const double* p, int ns, int nc, int np, const double* rec, int px,
                     int py, const double* txc, const double* tyc, const double* z0, double cL,
                     const double* cf, const double* mu, double* g) {
    const int nv = ns + nc;
    const double cs = 1e-30;
    for (int v = 0; v < nv; ++v) {
        cplx z[16], f[12];
        for (int q = 0; q < nv; ++q) z[q] = z0[q];
        z[v] += I * cs;
        dyn(model, p, z, f);
        cplx s = cL * lag(model, p, z);
        for (int i = 0; i < ns; ++i) s += cf[i] * f[i];
        for (int j = 0; j < np; ++j) s += mu[j] * keepout(rec + j * REC, z[px], z[py], txc[j], tyc[j]);
        g[v] = cimag(s) / cs;
    }
}

/* H[B][nhess][M]: packed lower triangle of
 * sigma*sgn*h*w_k*L_zz - h*sum_i lamF f_i,zz + sum_j lamC c_j,zz  (central difference of
 * complex-step gradients, step 1e-5 scaled by max(1,|z_q|)).                     */
int orc_hess(int model, const double* params, int maximize, int M, int B, const double* w, double t0,
             double tf, int np, int path_sets, const double* recs, int px, int py, int ntracks,
             int track_sets, const double* trkx, const double* trky, const double* X, const double* U,
             const double* LamF, const double* LamC, double sigma, double* H) {
    int ns, nc;
    if (model_dims(model, &ns, &nc)) return 1;
    const int nv = ns + nc, nh = nv * (nv + 1) / 2;
    const double h = (tf - t0) / 2.0, sgn = maximize ? -1.0 : 1.0;
    double* txc = (double*)calloc(np + 1, sizeof(double));
    double* tyc = (double*)calloc(np + 1, sizeof(double));
    double* mu = (double*)calloc(np + 1, sizeof(double));
    for (int b = 0; b < B; ++b) {
        const double* rec = np ? recs + (size_t)(path_sets > 1 ? b : 0) * np * REC : 0;
        for (int k = 0; k < M; ++k) {
            double z[16], cf[12], gp[16], gm[16];
            for (int i = 0; i < ns; ++i) z[i] = X[((size_t)b * ns + i) * M + k];
            for (int c = 0; c < nc; ++c) z[ns + c] = U[((size_t)b * nc + c) * M + k];
            for (int i = 0; i < ns; ++i) cf[i] = -h * LamF[((size_t)b * ns + i) * M + k];
            for (int j = 0; j < np; ++j) {
                mu[j] = LamC[((size_t)b * np + j) * M + k];
                txc[j] = tyc[j] = 0;
                if ((int)rec[j * REC] == P_TRACK) {
                    size_t off = ((size_t)(track_sets > 1 ? b : 0) * ntracks + (int)rec[j * REC + 1]) * M + k;
                    txc[j] = trkx[off]; tyc[j] = trky[off];
                }
            }
            const double cL = sigma * sgn * h * w[k];
            for (int q = 0; q < nv; ++q) {
                const double d = 1e-5 * fmax(1.0, fabs(z[q]));
                const double zq = z[q];
                z[q] = zq + d;
                lag_grad(model, params, ns, nc, np, rec, px, py, txc, tyc, z, cL, cf, mu, gp);
                z[q] = zq - d;
                lag_grad(model, params, ns, nc, np, rec, px, py, txc, tyc, z, cL, cf, mu, gm);
                z[q] = zq;
                for (int v = q; v < nv; ++v)
                    H[((size_t)b * nh + v * (v + 1) / 2 + q) * M + k] = (gp[v] - gm[v]) / (2 * d);
            }
        }
    }
    free(txc); free(tyc); free(mu);
    return 0;
}
