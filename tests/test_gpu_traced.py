"""Models compiled at run time (emi_set_model_source) on the GPU: a traced user model must give what
the hand-written kernel of the same model gives, and what the equations give.  -m gpu

ePSOPT evaluates such models by interpreting an ADOL-C tape per node on the CPU (reference
src/ePSOPT/ePSOPT.cpp:64-65, 218-276); here the trace is differentiated once on the host and the
generated struct runs in the library's node / Hessian / MFMA defect kernels."""
import ctypes as C
import os

import numpy as np
import pytest

import cases
import oracle_lib as O
from test_gpu_parity import TOL_DEFECT, TOL_NODE, check

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def traced_source(which):
    import torch  # noqa: F401
    lib = C.CDLL(os.path.join(HERE, "harness", "libetol_harness.so"))
    lib.harness_traced_model_source.restype = C.c_char_p
    return lib.harness_traced_model_source(which).decode()


def _quad_evaluators(name, f32=False):
    import etol_amd as E
    c = cases.case_inputs(name)
    evs = []
    for traced in (True, False):
        ev = E.Evaluator(0, f32=f32)
        ev.set_mesh(c["M"], c["t0"], c["tf"])
        if traced:
            ev.set_model_source("TracedModel", traced_source(0), 6, 2)
        else:
            ev.set_model(c["model"], c["params"])
        ev.set_batch(c["B"])
        if c.get("recs") is not None:
            ev.set_path(c["recs"], 0, 1)
        evs.append(ev)
    return c, evs[0], evs[1]


@pytest.mark.parametrize("name,ring", [("quad_256", True), ("quad_1024_obs", True), ("quad_ragged", False)])
def test_traced_quadrotor_matches_oracle_and_builtin_kernel(built, name, ring):
    import etol_amd as E
    c, tr, bi = _quad_evaluators(name)
    assert tr.layout.model == E.MODEL_SOURCE
    ref0 = O.evaluate(c["model"], c["params"], c["M"], (tr.tau, tr.w, tr.D), c["t0"], c["tf"], c["X"], c["U"], c.get("recs"))
    check(c, tr, tr.eval_host(c["X"], c["U"]), ref0)      # default dispatch
    if ring:                                              # ... is the one-launch pass (K sliced for few instances) for a traced model too
        assert "emi_pass_f64_kernel" in tr.last_defect_kernel, tr.last_defect_kernel
    tr.set_option("small_rows", 0)
    bi.set_option("small_rows", 0)
    assert tr.uses_fused_kernel == ring          # even/odd MFMA kernel instantiated for the traced struct
    got = tr.eval_host(c["X"], c["U"])
    ref = O.evaluate(c["model"], c["params"], c["M"], (tr.tau, tr.w, tr.D), c["t0"], c["tf"], c["X"], c["U"],
                     c.get("recs"))
    check(c, tr, got, ref)
    # against the hand-written kernel: same arithmetic up to the order of a few products
    hand = bi.eval_host(c["X"], c["U"])
    for a, b in zip(got, hand):
        assert np.abs(a - b).max() / (np.abs(b).max() + 1.0) < 1e-13
    # sequential general path gives the same rows
    tr.set_option("overlap", 0)
    seq = tr.eval_host(c["X"], c["U"])
    check(c, tr, seq, ref)
    # Lagrangian Hessian blocks
    rng = np.random.default_rng(5)
    lay = tr.layout
    lamF = rng.standard_normal((c["B"], lay.ns, c["M"]))
    lamC = rng.standard_normal((c["B"], lay.np, c["M"]))
    H = tr.hess_host(c["X"], c["U"], lamF, lamC, sigma=0.7)
    Hb = bi.hess_host(c["X"], c["U"], lamF, lamC, sigma=0.7)
    assert np.abs(H - Hb).max() / (np.abs(Hb).max() + 1.0) < 1e-13


def _custom_reference(X, U, tn, D, w, h):
    """numpy long-double restatement of the model of harness_traced_model_source(1)."""
    ld = np.longdouble
    a, b, c = X[:, 0].astype(ld), X[:, 1].astype(ld), U[:, 0].astype(ld)
    t = tn.astype(ld)[None, :]
    e3, q, sq = np.exp(a * ld(0.3)), 1 + b * b, np.sqrt(2 + a * a)
    f0 = e3 / q + np.tan(c) * sq - t
    lg, pw = np.log(3 + a * b + c * c), (2 + b) ** ld(1.5)
    s1 = np.sin(1 + b * ld(0.1))
    f1 = lg * pw - np.cos(a - c) / s1
    L = c * c * np.exp(-a) + b * np.sin(a * c)
    J = np.zeros((X.shape[0], 2, 3, X.shape[2]), dtype=ld)
    J[:, 0, 0] = ld(0.3) * e3 / q + np.tan(c) * a / sq
    J[:, 0, 1] = -e3 * 2 * b / q ** 2
    J[:, 0, 2] = sq / np.cos(c) ** 2
    inner = 3 + a * b + c * c
    J[:, 1, 0] = b / inner * pw + np.sin(a - c) / s1
    J[:, 1, 1] = a / inner * pw + lg * ld(1.5) * np.sqrt(2 + b) + np.cos(a - c) * np.cos(1 + b * ld(0.1)) * ld(0.1) / s1 ** 2
    J[:, 1, 2] = 2 * c / inner * pw - np.sin(a - c) / s1
    g = np.stack([-c * c * np.exp(-a) + b * c * np.cos(a * c), np.sin(a * c), 2 * c * np.exp(-a) + b * a * np.cos(a * c)], 1)
    F = np.stack([f0, f1], 1)
    DX = np.einsum("kj,bij->bik", D.astype(ld), X.astype(ld))
    RES = DX - h * F
    cost = h * np.einsum("k,bk->b", w.astype(ld), L)
    return RES.astype(np.float64), J.astype(np.float64), g.astype(np.float64), cost.astype(np.float64)


def test_traced_quadrotor_takes_the_one_launch_pass(built):
    """A run-time compiled model holds the pass kernel in the two forms the policy uses (SW = 1 with plain stores for small
    batches, SW = 2 with non-temporal stores for large ones) and the node kernel with non-temporal stores for the
    two-stream form: same results as the oracle and as the built-in instantiation on each of those paths."""
    import etol_amd as E
    M, B = 1024, 8
    X, U, recs = cases.W.quadrotor_batch(11, B, M, 3)
    evs = []
    for traced in (True, False):
        ev = E.Evaluator(0)
        ev.set_mesh(M, 0.0, cases.W.TF)
        if traced:
            ev.set_model_source("TracedModel", traced_source(0), 6, 2)
        else:
            ev.set_model(E.MODEL_QUADROTOR2D, cases.W.QUAD_PARAMS)
        ev.set_batch(B)
        ev.set_path(recs, 0, 1)
        evs.append(ev)
    tr, bi = evs
    ref = O.evaluate(E.MODEL_QUADROTOR2D, cases.W.QUAD_PARAMS, M, (tr.tau, tr.w, tr.D), 0.0, cases.W.TF, X, U, recs)
    c = dict(X=X)
    for opts, expect in ((dict(), "SW=1"), (dict(overlap_mode=3, node_store=2), "SW=2"), (dict(overlap_mode=2, node_store=2), "ring")):
        for ev in (tr, bi):
            ev.set_option("overlap_mode", 0)
            ev.set_option("node_store", -1)
            for k, v in opts.items():
                ev.set_option(k, v)
        got = tr.eval_host(X, U)
        assert expect in tr.last_defect_kernel, tr.last_defect_kernel
        assert ("one launch" in tr.last_defect_kernel) == (expect != "ring")
        check(c, tr, got, ref)
        hand = bi.eval_host(X, U)
        for a, b in zip(got, hand):
            assert np.abs(a - b).max() / (np.abs(b).max() + 1.0) < 1e-13
    tr.close()
    bi.close()


@pytest.mark.parametrize("M,B,ring", [(128, 20, True), (20, 3, False), (384, 33, True)])
def test_custom_traced_model_against_numpy(built, M, B, ring):
    """A model the library has no kernel for (2 states, 1 control, explicit time dependence, every
    traced elementary function)."""
    import etol_amd as E
    rng = np.random.default_rng(100 + M)
    X = rng.uniform(0.2, 1.2, (B, 2, M))
    U = rng.uniform(0.2, 1.2, (B, 1, M))
    t0, tf = 0.25, 3.0
    ev = E.Evaluator(0)
    ev.set_mesh(M, t0, tf)
    ev.set_model_source("TracedModel", traced_source(1), 2, 1)
    ev.set_batch(B)
    RES0 = ev.eval_host(X, U)[0]                 # default dispatch (skinny defect kernel for B * ns <= 24 rows)
    ev.set_option("small_rows", 0)
    assert ev.uses_fused_kernel == ring
    RES, VALS, COST = ev.eval_host(X, U)
    assert (np.abs(RES0 - RES) / (np.einsum("kj,bij->bik", np.abs(ev.D), np.abs(X)) + np.abs(RES) + 1.0)).max() < TOL_DEFECT
    h = (tf - t0) / 2
    rRES, J, g, rcost = _custom_reference(X, U, ev.node_t, ev.D, ev.w, h)
    scale = np.einsum("kj,bij->bik", np.abs(ev.D), np.abs(X)) + np.abs(rRES) + 1.0
    assert (np.abs(RES - rRES) / scale).max() < TOL_DEFECT
    dd = np.diag(ev.D)
    for i in range(2):
        for v in range(3):
            ref = -h * J[:, i, v] + (dd[None, :] if v == i else 0.0)
            got = VALS[:, i * 3 + v]
            assert np.abs(got - ref).max() / (np.abs(ref).max() + 1.0) < 10 * TOL_NODE, (i, v)
    for v in range(3):
        ref = h * ev.w[None, :] * g[:, v]
        assert np.abs(VALS[:, 6 + v] - ref).max() / (np.abs(ref).max() + 1.0) < 10 * TOL_NODE
    assert np.abs(COST - rcost).max() / (np.abs(rcost).max() + 1.0) < 1e-13
    # Hessian blocks: central differences of the device's own analytic Jacobian / gradient values
    lamF = rng.standard_normal((B, 2, M))
    H = ev.hess_host(X, U, lamF, None, sigma=0.9)
    eps = 1e-6
    Z = np.concatenate([X, U], 1)
    for q in range(3):
        Zp, Zm = Z.copy(), Z.copy()
        Zp[:, q] += eps
        Zm[:, q] -= eps
        _, Vp, _ = ev.eval_host(Zp[:, :2], Zp[:, 2:])
        _, Vm, _ = ev.eval_host(Zm[:, :2], Zm[:, 2:])
        dV = (Vp - Vm) / (2 * eps)
        for v in range(3):
            # d/dz_q of  0.9 * cost gradient_v + sum_i lamF_i * (defect Jacobian (i,v) without the D_kk term)
            ref = 0.9 * dV[:, 6 + v] + sum(lamF[:, i] * dV[:, i * 3 + v] for i in range(2))
            hi, lo = max(v, q), min(v, q)
            got = H[:, hi * (hi + 1) // 2 + lo]
            # the (i,i) Jacobian entries carry D_kk (up to N(N+1)/4): differencing them loses its ulp / eps
            tol = 1e-7 * (np.abs(ref).max() + 1.0) + 4 * np.finfo(float).eps * np.abs(dd)[None, :] / eps * np.abs(lamF).max()
            assert (np.abs(got - ref) < tol).all(), (v, q)


def test_traced_quadrotor_f32_context(built):
    c, tr, bi = _quad_evaluators("quad_256", f32=True)
    X = c["X"].astype(np.float32).astype(np.float64)
    U = c["U"].astype(np.float32).astype(np.float64)
    got = tr.eval_host(X, U)
    hand = bi.eval_host(X, U)
    for a, b in zip(got, hand):
        assert np.abs(a - b).max() / (np.abs(b).max() + 1.0) < 2e-6
    # a traced model takes the same fp32 MFMA defect kernel as the built-in ones (the kernel sees X and D only)
    assert tr.last_defect_kernel == bi.last_defect_kernel == "emi_defect_f32_ring_kernel"


def test_model_source_replaces_and_is_replaced(built):
    """emi_set_model after emi_set_model_source (and back) switches kernels cleanly; a text that does
    not compile leaves the previous model in place and reports the compiler log."""
    import etol_amd as E
    from etol_amd import _lib
    c = cases.case_inputs("quad_256")
    ev = E.Evaluator(0)
    ev.set_mesh(c["M"], c["t0"], c["tf"])
    ev.set_model(c["model"], c["params"])
    ev.set_batch(c["B"])
    a = ev.eval_host(c["X"], c["U"])
    ev.set_model_source("TracedModel", traced_source(0), 6, 2)
    b = ev.eval_host(c["X"], c["U"])
    with pytest.raises(_lib.EmiError) as ei:
        ev.set_model_source("TracedModel", "template <typename T> struct TracedModel { int x };", 6, 2)
    assert "error" in str(ei.value)
    b2 = ev.eval_host(c["X"], c["U"])
    ev.set_model(c["model"], c["params"])
    a2 = ev.eval_host(c["X"], c["U"])
    for p, q in zip(a, a2):
        assert np.array_equal(p, q)
    for p, q in zip(b, b2):
        assert np.array_equal(p, q)
    for p, q in zip(a, b):
        assert np.abs(p - q).max() / (np.abs(p).max() + 1.0) < 1e-13


def test_traced_path_rows_match_the_table_rows(built):
    """Constraint rows traced from callbacks (a disc and the ellipse of one polygon edge, written as Var
    arithmetic) against the same rows from the record table of the hand-written path: values, Jacobian entries,
    Hessian blocks; and mixed with a table row in front of them."""
    import etol_amd as E
    from etol_amd import _lib as L
    from etol_amd import workloads as W
    M, B = 256, 3
    X, U, _ = W.quadrotor_batch(5, B, M, 0)
    disc = np.zeros(L.PATH_REC); disc[:4] = [L.PATH_DISC, 4.0, 3.2, 0.64]
    ell = E.edge_ellipse(3.2, 2.5, 3.4, 2.6)
    extra = np.zeros(L.PATH_REC); extra[:4] = [L.PATH_DISC, 6.3, 4.4, 0.49]
    src = traced_source(2)
    rng = np.random.default_rng(9)
    for front in ([], [extra]):
        tr = E.Evaluator(0)
        tr.set_mesh(M, 0.0, W.TF)
        tr.set_model_source("TracedModel", src, 6, 2, npath=2, path_vars=(0, 1))
        tr.set_batch(B)
        tr.set_path(np.array(front).reshape(len(front), L.PATH_REC), 0, 1)
        bi = E.Evaluator(0)
        bi.set_mesh(M, 0.0, W.TF)
        bi.set_model(E.MODEL_QUADROTOR2D, W.QUAD_PARAMS)
        bi.set_batch(B)
        bi.set_path(np.array(front + [disc, ell]), 0, 1)
        assert tr.layout.np == bi.layout.np == len(front) + 2 and tr.layout.nvals == bi.layout.nvals
        a, b = tr.eval_host(X, U), bi.eval_host(X, U)
        for p, q in zip(a, b):
            assert p.shape == q.shape
            for e in range(p.shape[1]) if p.ndim == 3 else [None]:
                pe, qe = (p[:, e], q[:, e]) if e is not None else (p, q)
                assert np.abs(pe - qe).max() <= 1e-13 * (np.abs(qe).max() + 1e-300) + 1e-18, e
        ref = O.evaluate(E.MODEL_QUADROTOR2D, W.QUAD_PARAMS, M, (tr.tau, tr.w, tr.D), 0.0, W.TF, X, U, np.array(front + [disc, ell]))
        check(dict(X=X), tr, a, ref)
        lamF = rng.standard_normal((B, 6, M))
        lamC = rng.standard_normal((B, len(front) + 2, M))
        H, Hb = tr.hess_host(X, U, lamF, lamC, 0.8), bi.hess_host(X, U, lamF, lamC, 0.8)
        assert np.abs(H - Hb).max() / (np.abs(Hb).max() + 1.0) < 1e-13
        # structure of the Jacobian values is the same object for both
        assert all(np.array_equal(p, q) for p, q in zip(tr.jac_structure(), bi.jac_structure()))


def test_traced_moving_disc_rows_match_the_track_table(built):
    """Rows whose centres are waypoint tables interpolated at the node time inside the traced arithmetic
    (mx::interp1: piecewise linear through max/min) against the library's EMI_PATH_TRACK rows fed by
    emi_track_centres -- two-waypoint tracks of the shipped problem and a four-waypoint one."""
    import etol_amd as E
    from etol_amd import _lib as L
    from etol_amd import workloads as W
    tracks = [dict(r=0.5, t=[0.0, 32.0], x=[1.51, 2.00], y=[2.00, 2.00]),
              dict(r=0.5, t=[0.0, 32.0], x=[1.00, 1.00], y=[4.00, 3.00]),
              dict(r=0.4, t=[0.0, 4.0, 9.0, 16.0], x=[3.0, 3.5, 2.5, 4.0], y=[1.0, 2.5, 3.0, 2.0])]
    src = traced_source(4)
    assert "NPATH = 3" in src
    for M in (33, 128):
        B = 2
        X, U = W.pointmass_batch(3, B, M)
        tr = E.Evaluator(0)
        tr.set_mesh(M, 0.0, 16.0)
        tr.set_model_source("TracedModel", src, 2, 2, npath=3, path_vars=(0, 1))
        tr.set_batch(B)
        tr.set_path(np.zeros((0, L.PATH_REC)), 0, 1)
        bi = E.Evaluator(0)
        bi.set_mesh(M, 0.0, 16.0)
        bi.set_model(E.MODEL_POINTMASS2D, [])
        bi.set_batch(B)
        tx, ty, recs = [], [], []
        for i, k in enumerate(tracks):
            xc, yc = E.track_centres(k["t"], k["x"], k["y"], bi.node_t)
            tx.append(xc); ty.append(yc)
            r = np.zeros(L.PATH_REC); r[0], r[1], r[2] = L.PATH_TRACK, i, k["r"] ** 2
            recs.append(r)
        bi.set_tracks(np.array(tx), np.array(ty))
        bi.set_path(np.array(recs), 0, 1)
        a, b = tr.eval_host(X, U), bi.eval_host(X, U)
        for p, q in zip(a, b):
            assert np.abs(p - q).max() <= 1e-13 * (np.abs(q).max() + 1)
        lamF = np.random.default_rng(1).standard_normal((B, 2, M))
        lamC = np.random.default_rng(2).standard_normal((B, 3, M))
        assert np.abs(tr.hess_host(X, U, lamF, lamC, 0.6) - bi.hess_host(X, U, lamF, lamC, 0.6)).max() < 1e-13


def test_traced_rows_on_more_than_two_variables_on_the_device(built):
    """Rows traced from callbacks may depend on any states and controls of their node: a disc on (x, z), a speed limit
    on (vx, vz), a thrust-tilt coupling on (theta, thrust), a row of time and one state, behind one table row.  VALS
    holds PW = 6 partials per traced row (variables 0 1 2 3 4 6): values, partials, Hessian blocks and the structure
    query against numpy."""
    import etol_amd as E
    from etol_amd import _lib as L
    from etol_amd import workloads as W
    src = traced_source(3)
    assert "NPATH = 4, PW = 6" in src
    pv = [0, 1, 2, 3, 4, 6]
    M, B = 64, 3
    X, U, _ = W.quadrotor_batch(5, B, M, 0)
    extra = np.zeros(L.PATH_REC); extra[:4] = [L.PATH_DISC, 6.3, 4.4, 0.49]
    ev = E.Evaluator(0)
    ev.set_mesh(M, 0.0, 8.0)
    ev.set_model_source("TracedModel", src, 6, 2, npath=4, path_vars=pv)
    ev.set_batch(B)
    ev.set_path(extra[None], 0, 1)
    lay = ev.layout
    assert lay.np == 5 and lay.nvals == 6 * 8 + 2 * 1 + 6 * 4 + 8
    RES, VALS, COST = ev.eval_host(X, U)
    t = ev.node_t
    Z = np.concatenate([X, U], axis=1)
    rows = [0.64 - ((Z[:, 0] - 4.0) ** 2 + (Z[:, 1] - 3.2) ** 2), Z[:, 3] ** 2 + Z[:, 4] ** 2 - 9.0,
            Z[:, 6] * np.sin(Z[:, 2]) - 6.0, Z[:, 1] * np.cos(0.3 * t) - 9.5]
    for j, r in enumerate(rows):
        assert np.abs(RES[:, 6 + 1 + j] - r).max() < 1e-13 * (np.abs(r).max() + 1)
    grads = {(0, 0): -2 * (Z[:, 0] - 4.0), (0, 1): -2 * (Z[:, 1] - 3.2), (1, 3): 2 * Z[:, 3], (1, 4): 2 * Z[:, 4],
             (2, 2): Z[:, 6] * np.cos(Z[:, 2]), (2, 6): np.sin(Z[:, 2]), (3, 1): np.cos(0.3 * t) * np.ones_like(Z[:, 1])}
    base = 48 + 2
    for j in range(4):
        for q, v in enumerate(pv):
            ref = grads.get((j, v), np.zeros_like(Z[:, 0]))
            assert np.abs(VALS[:, base + j * 6 + q] - ref).max() < 1e-13 * (np.abs(ref).max() + 1), (j, v)
    # cost gradient sits behind the traced partials; the table row keeps its pair in front of them
    q8 = ev.eval_host(X, U)[1]
    assert np.array_equal(q8, VALS)
    bi = E.Evaluator(0)
    bi.set_mesh(M, 0.0, 8.0)
    bi.set_model(E.MODEL_QUADROTOR2D, W.QUAD_PARAMS)
    bi.set_batch(B)
    bi.set_path(extra[None], 0, 1)
    Rb, Vb, Cb = bi.eval_host(X, U)
    assert np.abs(VALS[:, :50] - Vb[:, :50]).max() < 1e-12 and np.abs(VALS[:, 74:] - Vb[:, 50:]).max() < 1e-12
    assert np.abs(COST - Cb).max() < 1e-12 * np.abs(Cb).max()
    # structure query: traced entries name their own columns
    r, c = ev.jac_structure()
    for j in range(4):
        for q, v in enumerate(pv):
            e = (base + j * 6 + q) * M
            assert c[e] == v * M and r[e] == 6 * M + 12 + (1 + j) * M
    # Hessian blocks: multiplier-weighted second derivatives of the traced rows land in the packed triangle
    rng = np.random.default_rng(8)
    lamF = np.zeros((B, 6, M))
    lamC = rng.standard_normal((B, 5, M))
    lamC[:, 0] = 0.0
    H = ev.hess_host(X, U, lamF, lamC, 0.0)
    pk = lambda a, b: a * (a + 1) // 2 + b
    ref = {pk(0, 0): -2 * lamC[:, 1], pk(1, 1): -2 * lamC[:, 1], pk(3, 3): 2 * lamC[:, 2], pk(4, 4): 2 * lamC[:, 2],
           pk(2, 2): -lamC[:, 3] * Z[:, 6] * np.sin(Z[:, 2]), pk(6, 2): lamC[:, 3] * np.cos(Z[:, 2])}
    for e in range(36):
        want = ref.get(e, np.zeros((B, M)))
        assert np.abs(H[:, e] - want).max() < 1e-12 * (np.abs(want).max() + 1), e
