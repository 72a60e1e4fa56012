"""Delayed states and controls (ePSOPT::dae, reference src/ePSOPT/ePSOPT.cpp:231-248) on the device against the CPU oracle.
-m gpu

The reference appends get_delayed_state / get_delayed_control values to the callbacks' x and u.  Here they are extra inputs
of the node program (include/emi355x.h, emi_set_delays): W(i dt) . (node values) formed on the MFMA defect kernel, then the
ordinary node / defect / Hessian kernels on the extended input vector.  Oracle side (oracle/emi_oracle.c): the interpolation
matrix by the plain Lagrange product formula in long double (the product uses the barycentric form), and model 3, the same
node functions written on complex arguments (complex-step derivatives)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P3 = [0.7, 0.3]
DISC = np.array([[1.0, 2.0, 1.5, 0.25, 0, 0, 0, 0]])      # EMI_PATH_DISC, centre (2, 1.5), radius 0.5


def _harness():
    h = C.CDLL(os.path.join(ROOT, "tests", "harness", "libetol_harness.so"))
    h.harness_last_message.restype = C.c_char_p
    return h


def _extended(X, U, tau, t0, tf, dt, xh=3, uh=1):
    """[U | x(t - dt) | x(t - 2 dt) | u(t - dt)] with the ORACLE's interpolation matrices"""
    M = X.shape[-1]
    parts = [U]
    for i in range(1, xh):
        parts.append(np.einsum("kj,bij->bik", O.delay_matrix(M, tau, t0, tf, i * dt), X))
    for i in range(1, uh + 1):
        parts.append(np.einsum("kj,bij->bik", O.delay_matrix(M, tau, t0, tf, i * dt), U))
    return np.concatenate(parts, axis=1)


def _check(RES, VALS, COST, ref, X, D, tol=1e-12):
    ns = 2
    scale = np.einsum("kj,bij->bik", np.abs(D), np.abs(X)) + np.abs(ref[0][:, :ns]) + 1.0
    assert (np.abs(RES[:, :ns] - ref[0][:, :ns]) / scale).max() < tol
    if RES.shape[1] > ns:
        assert np.abs(RES[:, ns:] - ref[0][:, ns:]).max() / (np.abs(ref[0][:, ns:]).max() + 1.0) < tol
    assert VALS.shape == ref[1].shape
    for e in range(VALS.shape[1]):
        assert np.abs(VALS[:, e] - ref[1][:, e]).max() / (np.abs(ref[1][:, e]).max() + 1.0) < tol, e
    assert np.abs(COST - ref[2]).max() / (np.abs(ref[2]).max() + 1.0) < tol


@pytest.mark.parametrize("nsteps", [32, 127])
def test_delayed_problem_set_up_through_the_etol_api_matches_the_oracle(built, nsteps):
    """ETOL::eMI355X with setXrhorizon(3), setUrhorizon(1): callbacks traced with the delayed handles where ePSOPT::dae
    appends them, setup(), one evaluation on the device (eMI355X::evaluate) -- against oracle model 3 fed with delayed values
    from the oracle's own interpolation matrices.  33 nodes: the general kernels; 128 nodes: the MFMA ring path."""
    import etol_amd as E
    h = _harness()
    M, dt = nsteps + 1, 0.25
    t0, tf = 0.0, nsteps * dt
    rng = np.random.default_rng(nsteps)
    tau, w, D = E.lgl(M)
    t = t0 + (tf - t0) / 2 * (tau + 1)
    X = np.stack([1 + 0.5 * np.sin(0.7 * t) + 0.05 * rng.standard_normal(M), 2 - 0.1 * t + 0.05 * rng.standard_normal(M)])[None]
    U = np.stack([0.3 * np.cos(t), 0.2 + 0.05 * rng.standard_normal(M)])[None]
    z = np.ascontiguousarray(np.concatenate([X[0].ravel(), U[0].ravel()]))
    res, vals = np.zeros(64 * M), np.zeros(256 * M)
    cost, nres, nvals = C.c_double(), C.c_int(), C.c_int()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    st = h.harness_delay_demo(nsteps, C.c_double(dt), 3, 1, 1, dp(z), dp(res), res.size, dp(vals), vals.size, C.byref(cost),
                              C.byref(nres), C.byref(nvals))
    assert st == 0
    assert nres.value == 3 and nvals.value == 2 * 10 + 2 + 10          # 2 defect + 1 disc row; dynamics block, disc pair, gradient
    RES = res[:3 * M].reshape(1, 3, M)
    VALS = vals[:nvals.value * M].reshape(1, nvals.value, M)
    Uext = _extended(X, U, tau, t0, tf, dt)
    ref = O.evaluate(3, P3, M, (tau, w, D), t0, tf, X, Uext, DISC)
    _check(RES, VALS, np.array([cost.value]), ref, X, D)
    # the delayed values really enter: without them (history = present) the rows differ
    ref_nodelay = O.evaluate(3, P3, M, (tau, w, D), t0, tf, X, np.concatenate([U, X, X, U], axis=1), DISC)
    assert np.abs(ref_nodelay[0][:, :2] - ref[0][:, :2]).max() > 1e-3


def test_delay_interpolation_matrix_properties(built):
    """emi_delay_matrix (host) against the oracle's product formula; exact on polynomials up to the mesh degree; rows whose
    delayed time falls before t0 are the unit row of node 0 (the clamp this build assumes, include/emi355x.h)."""
    import etol_amd as E
    from etol_amd import _lib as L
    lib = L.load()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    for M, delay in ((9, 0.5), (33, 0.25), (33, 4.0), (128, 0.37), (200, 1.0)):
        t0, tf = 0.5, 8.5
        tau, w, _ = E.lgl(M)
        W = np.empty((M, M))
        assert lib.emi_delay_matrix(M, dp(tau), dp(w), t0, tf, delay, dp(W)) == 0
        Wo = O.delay_matrix(M, tau, t0, tf, delay)
        assert np.abs(W - Wo).max() < 1e-12 * max(1.0, np.abs(Wo).max())
        t = t0 + (tf - t0) / 2 * (tau + 1)
        early = t - delay < t0
        assert early.any() and np.array_equal(W[early], np.eye(M)[np.zeros(early.sum(), dtype=int)])
        p = np.polynomial.Polynomial(np.random.default_rng(M).standard_normal(min(M, 8)))
        s = lambda tt: 2 * (tt - t0) / (tf - t0) - 1
        assert np.abs(W @ p(s(t)) - p(s(np.maximum(t - delay, t0)))).max() < 1e-10
        assert np.abs(W.sum(1) - 1).max() < 1e-12


@pytest.mark.parametrize("M,B", [(128, 20), (33, 3), (256, 40)])
def test_batched_delayed_evaluation_and_hessian_against_the_oracle(built, M, B):
    """C-ABI level, batched: the traced delayed model (source as eMI355X::setup generated it) + emi_set_delays, device-pointer
    and host forms, every dispatch form the model program holds, and the Lagrangian Hessian blocks on the extended node
    variables (oracle: central differences of complex-step gradients, 1e-7)."""
    import torch
    import etol_amd as E
    h = _harness()
    z = np.zeros(4 * 9)
    res, vals = np.zeros(64 * 9), np.zeros(256 * 9)
    cost, nres, nvals = C.c_double(), C.c_int(), C.c_int()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    assert h.harness_delay_demo(8, C.c_double(0.5), 3, 1, 0, dp(z), dp(res), res.size, dp(vals), vals.size, C.byref(cost),
                                C.byref(nres), C.byref(nvals)) == 0
    source = h.harness_last_message().decode()
    assert "struct TracedModel" in source
    t0, tf, dt = 0.0, 6.0, 0.2
    ev = E.Evaluator(0)
    ev.set_mesh(M, t0, tf)
    ev.set_model_source("TracedModel", source, 2, 8)
    ev.set_delays(3, 1, dt)
    assert ev.n_delayed == 6 and ev.layout.nc == 8
    ev.set_batch(B)
    ev.set_path(DISC, 0, 1)
    rng = np.random.default_rng(M + B)
    t = ev.node_t
    X = np.stack([1 + 0.5 * np.sin(0.7 * t + rng.uniform(0, 3, (B, 1))), 2 - 0.1 * t + 0.3 * np.cos(t + rng.uniform(0, 3, (B, 1)))], axis=1)
    U = np.stack([0.3 * np.cos(t + rng.uniform(0, 3, (B, 1))), 0.2 + 0.1 * np.sin(2 * t + rng.uniform(0, 3, (B, 1)))], axis=1)
    X, U = np.ascontiguousarray(X), np.ascontiguousarray(U)
    Uext = _extended(X, U, ev.tau, t0, tf, dt)
    ref = O.evaluate(3, P3, M, (ev.tau, ev.w, ev.D), t0, tf, X, Uext, DISC)
    got = ev.eval_host(X, U)
    _check(*got, ref, X, ev.D)
    for mode in (1, 2, 3, 0):                 # one stream, two streams, one launch (where the shape allows), default
        ev.set_option("overlap_mode", mode)
        _check(*ev.eval_host(X, U), ref, X, ev.D)
    ev.set_option("overlap", 0)               # the sequential general path
    _check(*ev.eval_host(X, U), ref, X, ev.D)
    ev.set_option("overlap", 1)
    dX, dU = torch.from_numpy(X).cuda(), torch.from_numpy(U).cuda()
    outs = ev.alloc_outputs()
    ev.eval_dev(dX, dU, *outs)
    ev.synchronize()
    torch.cuda.synchronize()
    _check(*(o.cpu().numpy() for o in outs), ref, X, ev.D)
    lamF = rng.standard_normal((B, 2, M))
    lamC = rng.standard_normal((B, 1, M))
    H = ev.hess_host(X, U, lamF, lamC, sigma=0.8)
    Href = O.hessian(3, P3, M, (ev.tau, ev.w, ev.D), t0, tf, X, Uext, lamF, lamC, 0.8, DISC)
    assert H.shape == Href.shape and np.abs(H - Href).max() / (np.abs(Href).max() + 1.0) < 1e-7
    # a model change drops the delays (they belong to the model they were declared for)
    ev.set_model(E.MODEL_POINTMASS2D, [])
    assert ev.n_delayed == 0
    ev.close()


def test_delay_declarations_are_checked(built):
    import etol_amd as E
    from etol_amd import _lib as L
    lib = L.load()
    ev = E.Evaluator(0)
    ev.set_mesh(16, 0.0, 2.0)
    assert lib.emi_set_delays(ev.ctx, 3, 0, C.c_double(0.1)) == 2                 # before the model: EMI_ERR_STATE
    ev.set_model(E.MODEL_QUADROTOR2D, [1, 0.01, 9.81, 1, 1])                      # 6 states, 2 controls
    assert lib.emi_set_delays(ev.ctx, 2, 0, C.c_double(0.1)) == 1                 # 6 delayed states do not fit 2 controls
    assert b"controls" in lib.emi_last_error(ev.ctx)
    assert lib.emi_set_delays(ev.ctx, 1, 0, C.c_double(0.1)) == 0                 # horizon 1 / 0: nothing delayed, as the reference
    assert ev.n_delayed == 0
    assert lib.emi_set_delays(ev.ctx, 0, 1, C.c_double(0.0)) == 1                 # dt must be positive
    assert lib.emi_set_delays(ev.ctx, 0, 1, C.c_double(0.1)) == 0                 # 2 = 1 free control + its delayed copy
    ev._changed()                                                                 # (the C ABI was called behind the Evaluator's cached layout)
    assert ev.n_delayed == 1
    ev.close()


def test_mesh_change_after_set_delays_rebuilds_the_interpolation_matrices(built):
    """set_mesh(33) -> set_delays -> eval -> set_mesh(128) -> eval (a refinement loop's order of calls): W(delay) is built per
    mesh ([nd][M][M] on ITS nodes and horizon), so a new mesh must mark it stale -- left alone, the second evaluation reads a
    33 x 33 operator as 128 x 128 (past the allocation) or, with an unchanged M and a new horizon, returns wrong delayed values."""
    import etol_amd as E
    h = _harness()
    z = np.zeros(4 * 9)
    res, vals = np.zeros(64 * 9), np.zeros(256 * 9)
    cost, nres, nvals = C.c_double(), C.c_int(), C.c_int()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    assert h.harness_delay_demo(8, C.c_double(0.5), 3, 1, 0, dp(z), dp(res), res.size, dp(vals), vals.size, C.byref(cost),
                                C.byref(nres), C.byref(nvals)) == 0
    source = h.harness_last_message().decode()
    dt, B = 0.2, 3
    ev = E.Evaluator(0)
    ev.set_mesh(33, 0.0, 6.0)
    ev.set_model_source("TracedModel", source, 2, 8)
    ev.set_delays(3, 1, dt)
    ev.set_batch(B)
    ev.set_path(DISC, 0, 1)
    for M, t0, tf in ((33, 0.0, 6.0), (128, 0.0, 6.0), (128, 0.0, 9.0), (33, 1.0, 4.0)):      # larger mesh, new horizon, smaller mesh
        ev.set_mesh(M, t0, tf)
        ev.set_path(DISC, 0, 1)
        rng = np.random.default_rng(M + int(tf))
        t = ev.node_t
        X = np.ascontiguousarray(np.stack([1 + 0.5 * np.sin(0.7 * t + rng.uniform(0, 3, (B, 1))),
                                           2 - 0.1 * t + 0.3 * np.cos(t + rng.uniform(0, 3, (B, 1)))], axis=1))
        U = np.ascontiguousarray(np.stack([0.3 * np.cos(t + rng.uniform(0, 3, (B, 1))),
                                           0.2 + 0.1 * np.sin(2 * t + rng.uniform(0, 3, (B, 1)))], axis=1))
        ref = O.evaluate(3, P3, M, (ev.tau, ev.w, ev.D), t0, tf, X, _extended(X, U, ev.tau, t0, tf, dt), DISC)
        _check(*ev.eval_host(X, U), ref, X, ev.D)
    ev.close()


@pytest.mark.parametrize("nsteps,disc_r", [(24, 0.9), (32, 0.5), (40, 0.0)])
def test_delayed_problem_is_solved_through_the_etol_api(built, nsteps, disc_r):
    """solve() on a problem with setXrhorizon(3) / setUrhorizon(1) (reference src/ePSOPT/ePSOPT.cpp:231-248, where ePSOPT hands
    delayed values to the callbacks and the whole NLP to IPOPT): ETOL::eMI355X iterates on the delayed values as node variables
    tied to their sources by coupling rows with the mesh's interpolation operators; functions, Jacobian entries and Hessian
    blocks come from the DEVICE kernels on the extended node variables.  The trajectory returned must (a) satisfy the delayed
    dynamics as the device itself evaluates them (delays formed on the device again), and (b) be, after lifting, a KKT point of
    the independent restatement tests/indep_nlp.py::DelayedNlp (oracle functions + the oracle's own delay matrices): the
    Newton polish from it converges to a verified KKT point within 1e-6 relative (north_star's tolerance)."""
    import indep_nlp as N
    h = _harness()
    D = C.POINTER(C.c_double)
    h.harness_solve_delay_demo.argtypes = [C.c_int, C.c_double, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, D, D, D,
                                           C.POINTER(C.c_int), D]
    M, dt = nsteps + 1, 0.25
    X, U = np.zeros((2, M)), np.zeros((2, M))
    cost, iters, dmax = C.c_double(), C.c_int(), C.c_double()
    rc = h.harness_solve_delay_demo(nsteps, dt, 3, 1, disc_r, 1e-10, 0, C.byref(cost), X.ctypes.data_as(D), U.ctypes.data_as(D),
                                    C.byref(iters), C.byref(dmax))
    assert rc == 0, h.harness_last_message().decode()
    assert dmax.value < 1e-8                              # (a): evaluate() at the solution, delays formed by the device
    P = N.DelayedNlp(nsteps, dt, disc_r=disc_r)
    Z = P.lift(X, U)
    assert np.abs(P.defect(Z)).max() < 1e-8 and abs(P.cost(Z) - cost.value) < 1e-9 * max(1.0, abs(cost.value))
    zp, lamF, lamC, k = N.polish(P, Z)
    assert N.kkt_ok(k), k
    rel = np.abs(zp - Z).max() / np.abs(zp).max()
    print(f"delayed solve on the device evaluator: {M} nodes, {iters.value} iterations, cost {cost.value:.10f}, rel dist to the KKT point {rel:.2e}")
    assert rel < 1e-6
    if disc_r >= 0.9:
        assert k["active_path_rows"] > 0
