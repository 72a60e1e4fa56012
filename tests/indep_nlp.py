"""Independent CPU optimiser for the solve-level parity checks (TEST INFRASTRUCTURE; numpy + scipy + the CPU oracle).

The NLP that ePSOPT::setup / addBounds define (reference src/ePSOPT/ePSOPT.cpp:40-81,125-155) restated on the
oracle's functions (oracle/emi_oracle.c through tests/oracle_lib.py: values, complex-step first derivatives,
long-double D.X), solved with scipy's SLSQP and polished to a KKT point by Newton's method on the active set (exact
first derivatives; Lagrangian Hessian from orc_hess).  Nothing of the product (etol_amd/) takes part.

    variables    z = [X (ns x M), U (nc x M)], node index fastest
    minimise     h sum_k w_k L(x_k, u_k)                                   (integrand_cost, :186-216)
    subject to   D.X - h f(X, U) = 0                                       (dae :252-260 + PSOPT's LGL defects)
                 -1000 <= c_j(x_k, t_k) <= 0                               (path rows :261-270, bounds :147-150)
                 x(t0) = x0,  xf - tol <= x(tf) <= xf + tol                (events :281-291, :137-141)
                 state / control boxes at every node                       (:134-135, :143-146)

Used by tests/golden/gen_solve_fixtures.py (fixtures) and by tests/test_gpu_solve.py (KKT check of a GPU solution).
"""
import numpy as np
from scipy.optimize import minimize

import oracle_lib as O


class Nlp:
    def __init__(self, model, params, M, t0, tf, recs, tracks, x0, xf, xtol, xlo, xup, ulo, uup, px=0, py=1):
        self.model, self.params, self.M, self.t0, self.tf = model, list(params), M, t0, tf
        self.ns, self.nc = O.MODEL_DIMS[model]
        self.nv = self.ns + self.nc
        self.mesh = O.lgl(M)
        self.tau, self.w, self.D = self.mesh
        self.h = (tf - t0) / 2
        self.recs = np.zeros((0, 8)) if recs is None else np.asarray(recs, dtype=float)
        self.np_ = self.recs.shape[0]
        self.tracks, self.px, self.py = tracks, px, py
        ns, nc, nv = self.ns, self.nc, self.nv
        lo = np.concatenate([np.repeat(xlo, M), np.repeat(ulo, M)]).astype(float)
        up = np.concatenate([np.repeat(xup, M), np.repeat(uup, M)]).astype(float)
        for i in range(ns):
            lo[i * M] = up[i * M] = x0[i]
            lo[i * M + M - 1] = max(lo[i * M + M - 1], xf[i] - xtol[i])
            up[i * M + M - 1] = min(up[i * M + M - 1], xf[i] + xtol[i])
        self.lo, self.up = lo, up
        self.n = nv * M

    def split(self, z):
        return z[: self.ns * self.M].reshape(1, self.ns, self.M), z[self.ns * self.M:].reshape(1, self.nc, self.M)

    def evaluate(self, z):
        X, U = self.split(z)
        RES, VALS, COST = O.evaluate(self.model, self.params, self.M, self.mesh, self.t0, self.tf, X, U,
                                     self.recs if self.np_ else None, self.tracks, px=self.px, py=self.py)
        return RES[0], VALS[0], COST[0]

    def cost(self, z):
        return self.evaluate(z)[2]

    def grad(self, z):
        V = self.evaluate(z)[1]
        return V[self.ns * self.nv + 2 * self.np_:].reshape(-1)

    def defect(self, z):
        return self.evaluate(z)[0][: self.ns].reshape(-1)

    def path(self, z):
        return self.evaluate(z)[0][self.ns:].reshape(-1)

    def jac_defect(self, z):
        ns, nv, M = self.ns, self.nv, self.M
        V = self.evaluate(z)[1]
        J = np.zeros((ns * M, nv * M))
        Doff = self.D - np.diag(np.diag(self.D))
        for i in range(ns):
            J[i * M:(i + 1) * M, i * M:(i + 1) * M] = Doff
            for v in range(nv):
                J[i * M + np.arange(M), v * M + np.arange(M)] += V[i * nv + v]      # -h df_i/dz_v (+ D_kk if v == i)
        return J

    def jac_path(self, z):
        ns, nv, M, np_ = self.ns, self.nv, self.M, self.np_
        V = self.evaluate(z)[1]
        J = np.zeros((np_ * M, nv * M))
        for j in range(np_):
            J[j * M + np.arange(M), self.px * M + np.arange(M)] = V[ns * nv + 2 * j]
            J[j * M + np.arange(M), self.py * M + np.arange(M)] = V[ns * nv + 2 * j + 1]
        return J

    def hess(self, z, lamF, lamC):
        """dense Hessian of cost + lamF.defect + lamC.path (node-block diagonal)"""
        ns, nv, M = self.ns, self.nv, self.M
        X, U = self.split(z)
        H = O.hessian(self.model, self.params, M, self.mesh, self.t0, self.tf, X, U, lamF[: ns * M].reshape(1, ns, M),
                      lamC.reshape(1, self.np_, M) if self.np_ else None, 1.0, self.recs if self.np_ else None, self.tracks,
                      px=self.px, py=self.py)[0]
        W = np.zeros((nv * M, nv * M))
        e = 0
        for a in range(nv):
            for b in range(a + 1):
                W[a * M + np.arange(M), b * M + np.arange(M)] = H[e]
                W[b * M + np.arange(M), a * M + np.arange(M)] = H[e]
                e += 1
        return W


class DelayedNlp(Nlp):
    """The delayed problem of tests/harness/etol_harness.cpp (harness_delay_demo; oracle model 3) in LIFTED form: the delayed
    values x(t - dt), x(t - 2 dt), u(t - dt) are the controls 2 .. 7 of their node and coupling rows  d - W(i dt) . source = 0
    tie them to the trajectory (W from the oracle's own orc_delay_matrix: plain Lagrange product formula, times before t0
    clamped to t0).  Reference: ePSOPT::dae appends get_delayed_state / get_delayed_control values at these places
    (src/ePSOPT/ePSOPT.cpp:231-248).  `defect` / `jac_defect` return the defect rows followed by the coupling rows."""

    def __init__(self, nsteps, dt, disc_r=0.5, xh=3, uh=1):
        M = nsteps + 1
        recs = np.array([[1.0, 2.0, 1.5, disc_r * disc_r, 0, 0, 0, 0]]) if disc_r > 0 else None
        big = 1e20
        super().__init__(3, [0.7, 0.3], M, 0.0, nsteps * dt, recs, None, x0=[1, 2], xf=[3, 1], xtol=[0.01, 0.01], xlo=[-10, -10],
                         xup=[10, 10], ulo=[-5, -5] + [-big] * 6, uup=[5, 5] + [big] * 6)
        ns, ncf = 2, 2
        self.links = []
        slot = ns + ncf
        for i in range(1, xh):
            for s_ in range(ns):
                self.links.append((slot, s_, O.delay_matrix(M, self.tau, self.t0, self.tf, i * dt)))
                slot += 1
        for i in range(1, uh + 1):
            for c in range(ncf):
                self.links.append((slot, ns + c, O.delay_matrix(M, self.tau, self.t0, self.tf, i * dt)))
                slot += 1
        assert slot == self.nv
        self.n_eq = (ns + len(self.links)) * M
        self.JL = np.zeros((len(self.links) * M, self.n))
        for l, (dst, src, W) in enumerate(self.links):
            self.JL[l * M:(l + 1) * M, dst * M:(dst + 1) * M] = np.eye(M)
            self.JL[l * M:(l + 1) * M, src * M:(src + 1) * M] -= W

    def lift(self, X, U):
        """z = [X | U | W . sources] from the free trajectory X (2 x M), U (2 x M)"""
        M = self.M
        z = np.zeros(self.n)
        z[: 2 * M] = np.asarray(X).ravel()
        z[2 * M:4 * M] = np.asarray(U).ravel()
        for dst, src, W in self.links:
            z[dst * M:(dst + 1) * M] = W @ z[src * M:(src + 1) * M]
        return z

    def defect(self, z):
        return np.concatenate([super().defect(z), self.JL @ z])

    def jac_defect(self, z):
        return np.vstack([super().jac_defect(z), self.JL])


def slsqp(P, z0):
    cons = [dict(type="eq", fun=P.defect, jac=P.jac_defect)]
    if P.np_:
        cons.append(dict(type="ineq", fun=lambda z: -P.path(z), jac=lambda z: -P.jac_path(z)))
    r = minimize(P.cost, z0, jac=P.grad, bounds=list(zip(P.lo, P.up)), constraints=cons, method="SLSQP",
                 options=dict(ftol=1e-10, maxiter=3000))
    return r.x, r


def kkt_ok(k):
    """the acceptance rule for a polished point (stationarity, feasibility, multiplier signs)"""
    return (k["stationarity"] < 1e-9 and k["defect"] < 1e-11 and k["path_violation"] < 1e-11 and
            k["min_path_multiplier"] > -1e-9 and k["wrong_sign_bound_multiplier"] < 1e-9)


def polish(P, z, iters=30):
    """Newton on the KKT conditions of the active set found at z; returns z, multipliers, residuals."""
    n, ns, M = P.n, P.ns, P.M
    z = np.clip(np.array(z, dtype=float), P.lo, P.up)
    fixed = P.lo == P.up
    act_lo = (~fixed) & (z - P.lo < 1e-7)
    act_up = (~fixed) & (P.up - z < 1e-7)
    z[act_lo] = P.lo[act_lo]
    z[act_up] = P.up[act_up]
    free = ~(fixed | act_lo | act_up)
    c = P.path(z) if P.np_ else np.zeros(0)
    act_c = c > -1e-7
    me = getattr(P, "n_eq", ns * M)          # equality rows: defects (+ coupling rows of a DelayedNlp)
    lamF = np.zeros(me)
    lamA = np.zeros(int(act_c.sum()))
    for it in range(iters):
        g, Jd = P.grad(z), P.jac_defect(z)
        Jp = P.jac_path(z)[act_c] if P.np_ else np.zeros((0, n))
        if it == 0:   # least-squares multipliers to start from
            A = np.vstack([Jd, Jp])[:, free]
            lam = np.linalg.lstsq(A.T, -g[free], rcond=None)[0]
            lamF, lamA = lam[:me], lam[me:]
        lamC = np.zeros(P.np_ * M)
        lamC[act_c] = lamA
        W = P.hess(z, lamF, lamC)
        A = np.vstack([Jd, Jp])[:, free]
        r1 = (g + Jd.T @ lamF + Jp.T @ lamA)[free]
        r2 = np.concatenate([P.defect(z), (P.path(z)[act_c] if P.np_ else np.zeros(0))])
        res = max(np.abs(r1).max(), np.abs(r2).max())
        if res < 1e-13:
            break
        nf, m = int(free.sum()), A.shape[0]
        K = np.block([[W[np.ix_(free, free)], A.T], [A, np.zeros((m, m))]])
        d = np.linalg.solve(K, -np.concatenate([r1, r2]))
        z[free] += d[:nf]
        lamF += d[nf:nf + me]
        lamA += d[nf + me:]
    # full KKT check at the polished point
    g, Jd = P.grad(z), P.jac_defect(z)
    c = P.path(z) if P.np_ else np.zeros(0)
    Jp = P.jac_path(z) if P.np_ else np.zeros((0, n))
    lamC = np.zeros(P.np_ * M)
    lamC[act_c] = lamA
    rz = g + Jd.T @ lamF + Jp.T @ lamC            # = bound multipliers on the non-free variables
    out = dict(stationarity=float(np.abs(rz[free]).max()), defect=float(np.abs(P.defect(z)).max()),
               path_violation=float(max(c.max(), 0.0)) if P.np_ else 0.0,
               bound_violation=float(max((P.lo - z).max(), (z - P.up).max(), 0.0)),
               min_path_multiplier=float(lamA.min()) if lamA.size else 0.0,
               wrong_sign_bound_multiplier=float(max(np.max(rz[act_lo] * -1, initial=0.0), np.max(rz[act_up], initial=0.0))),
               active_path_rows=int(act_c.sum()), active_bounds=int(act_lo.sum() + act_up.sum()),
               inactive_path_margin=float(-c[~act_c].max()) if P.np_ and (~act_c).any() else None)
    return z, lamF, lamC, out


def shipped_problem():
    import cases
    M, tf = 33, 16.0
    node_t = 0.5 * tf * (O.lgl(M)[0] + 1.0)
    recs, tx, ty = cases.ocp2d_tables(O.edge_ellipse, O.track_centres, node_t)
    P = Nlp(0, [], M, 0.0, tf, recs, (tx, ty), x0=[1, 2], xf=[5, 4], xtol=[0.01, 0.01], xlo=[0, 0], xup=[7, 7],
            ulo=[-0.5, -0.5], uup=[0.5, 0.5])
    return P, node_t


def quad_problem():
    M, tf = 41, 4.0
    discs = [(4.0, 3.2, 0.8), (6.3, 4.4, 0.7)]          # tests/harness/etol_harness.cpp configure_quadrotor, ndiscs = 2
    recs = np.zeros((2, 8))
    for j, (x, y, r) in enumerate(discs):
        recs[j, :4] = [1, x, y, r * r]                   # PATH_DISC
    P = Nlp(1, [1.0, 0.01, 9.81, 1.0, 1.0], M, 0.0, tf, recs, None, x0=[1, 1, 0, 0, 0, 0], xf=[8, 6, 0, 0, 0, 0],
            xtol=[0.01, 0.01, 0.01, 0.05, 0.05, 0.05], xlo=[0, 0, -1.2, -6, -6, -4], xup=[10, 10, 1.2, 6, 6, 4],
            ulo=[0, -1], uup=[25, 1])
    return P


def starts(P, bumps):
    """straight line between the boundary states, bent sideways by `bump` (several homotopy classes)"""
    M, ns, nc = P.M, P.ns, P.nc
    s = (P.tau + 1) / 2
    out = []
    for bump in bumps:
        z = np.zeros(P.n)
        a = np.array([P.lo[i * M] for i in range(ns)])
        b = np.array([0.5 * (P.lo[i * M + M - 1] + P.up[i * M + M - 1]) for i in range(ns)])
        for i in range(ns):
            z[i * M:(i + 1) * M] = a[i] + (b[i] - a[i]) * s
        d = b[:2] - a[:2]
        nrm = np.array([-d[1], d[0]]) / np.hypot(*d)
        z[0:M] += bump * nrm[0] * np.sin(np.pi * s)
        z[M:2 * M] += bump * nrm[1] * np.sin(np.pi * s)
        if P.model == 1:
            z[ns * M:(ns + 1) * M] = 9.81                # hover thrust
        out.append(np.clip(z, P.lo, P.up))
    return out


def solve_all(P, bumps, tag, extra_starts=()):
    found = []
    for z0 in list(starts(P, bumps)) + [np.clip(np.asarray(e, dtype=float), P.lo, P.up) for e in extra_starts]:
        z, r = slsqp(P, z0)
        viol = max(np.abs(P.defect(z)).max(), (P.path(z).max() if P.np_ else 0.0))
        if not r.success:      # SLSQP often stops on its line search within 1e-8 of the optimum: let the polish decide
            print(f"  {tag}: SLSQP exit '{r.message}' after {r.nit} iterations, constraint violation {viol:.2e}")
            if viol > 1e-4:
                continue
        z, lamF, lamC, k = polish(P, z)
        ok = kkt_ok(k)
        cost = P.cost(z)
        print(f"  {tag}: cost {cost:.10f}  KKT {k}  {'ok' if ok else 'REJECTED'}")
        if not ok:
            continue
        if any(np.abs(z - f["z"]).max() < 1e-7 for f in found):
            continue
        found.append(dict(z=z, cost=cost, kkt=k, lamF_max=float(np.abs(lamF).max())))
    return found


def mip_problem():
    """resource/configs/mip_2d_ex1.xml as the PSOPT example would pose it (container/singularity/ETOL-examples.def:131-132
    feeds it that file): 17 nodes, tf = 8, the same two polygons, the file's own two tracks; controls 2 and 3 of the file
    appear in no callback and are left out."""
    import cases
    M, tf = 17, 8.0
    node_t = 0.5 * tf * (O.lgl(M)[0] + 1.0)
    saved = cases.OCP2D["tracks"]
    cases.OCP2D["tracks"] = [dict(radius=0.5, t=[0.0, 32.0], x=[2.00, 2.50], y=[2.00, 2.00]),
                             dict(radius=0.5, t=[0.0, 32.0], x=[1.00, 1.00], y=[4.00, 3.00])]
    recs, tx, ty = cases.ocp2d_tables(O.edge_ellipse, O.track_centres, node_t)
    cases.OCP2D["tracks"] = saved
    return Nlp(0, [], M, 0.0, tf, recs, (tx, ty), x0=[1, 2], xf=[5, 4], xtol=[0.01, 0.01], xlo=[0, 0], xup=[7, 7],
               ulo=[-0.5, -0.5], uup=[0.5, 0.5])


def least_violation(P, bumps):
    """min sum max(0, c)^2 over the dynamics and the boxes: > 0 from every start = no feasible point found"""
    best = np.inf
    for z0 in starts(P, bumps):
        f = lambda z: float((np.maximum(P.path(z), 0.0) ** 2).sum())
        g = lambda z: 2.0 * P.jac_path(z).T @ np.maximum(P.path(z), 0.0)
        r = minimize(f, z0, jac=g, bounds=list(zip(P.lo, P.up)), constraints=[dict(type="eq", fun=P.defect, jac=P.jac_defect)],
                     method="SLSQP", options=dict(ftol=1e-14, maxiter=2000))
        if np.abs(P.defect(r.x)).max() < 1e-8:
            best = min(best, float(np.maximum(P.path(r.x), 0.0).max()))
    return best
