"""Host C++ library (TrajectoryOptimizer base, loader, writers, KKT factorisation) on the CPU."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def H(built):
    import torch  # noqa: F401  (one HIP runtime per process: see etol_amd/_lib.py)
    lib = C.CDLL(os.path.join(ROOT, "tests", "harness", "libetol_harness.so"))
    lib.harness_load_configs.restype = C.c_char_p
    lib.harness_load_configs.argtypes = [C.c_char_p]
    lib.harness_roundtrip_configs.restype = C.c_char_p
    lib.harness_roundtrip_configs.argtypes = [C.c_char_p, C.c_char_p]
    lib.harness_save_csv.restype = C.c_char_p
    lib.harness_save_csv.argtypes = [C.c_char_p]
    D = C.POINTER(C.c_double)
    lib.harness_lin_interp.argtypes = [C.c_int, D, C.c_int, D, D, D]
    lib.harness_ldlt.argtypes = [C.c_int, D, D, C.POINTER(C.c_int)]
    return lib


def load(H, path):
    return json.loads(H.harness_load_configs(path.encode()).decode())


def test_load_shipped_ocp(H, xmls):
    c = load(H, xmls["ocp_2d_ex1.xml"])
    assert (c["nsteps"], c["dt"], c["nstates"], c["ncontrols"]) == (32, 0.5, 2, 2)
    assert c["xrhorizon"] == c["urhorizon"] == c["rhorizon"] == 0
    assert c["xlower"] == [0, 0] and c["xupper"] == [7, 7] and c["x0"] == [1, 2] and c["xf"] == [5, 4]
    assert c["xtol"] == [0.01, 0.01] and c["ulower"] == [-0.5, -0.5] and c["uupper"] == [0.5, 0.5]
    assert c["xvartype"] == [0, 0] and c["uvartype"] == [0, 0]
    assert [len(z) for z in c["zones"]] == [5, 4]
    assert c["zones"][0][2] == [3.5, 3.4, 0.0] and c["zones"][1][3] == [2.1, 3.5, 0.0]
    assert c["nexclzones"] == 0          # convex partition (CGAL) is not part of this build
    assert [t["radius"] for t in c["tracks"]] == [0.5, 0.5]
    assert c["tracks"][0]["waypoints"] == [[0, 1.51, 2.0], [32, 2.0, 2.0]]
    assert c["tracks"][1]["waypoints"] == [[0, 1.0, 4.0], [32, 1.0, 3.0]]


def test_load_shipped_mip(H, xmls):
    c = load(H, xmls["mip_2d_ex1.xml"])
    assert (c["nsteps"], c["nstates"], c["ncontrols"], c["xrhorizon"], c["rhorizon"]) == (16, 2, 4, 1, 1)
    assert c["tracks"][0]["waypoints"][0] == [0, 2.0, 2.0]


def test_loader_caps_and_unknown_nodes(H, xmls):
    c = load(H, xmls["edge_caps.xml"])
    assert c["nstates"] == 2 and len(c["xlower"]) == 2          # third <state> beyond nstates is skipped
    assert c["ncontrols"] == 3 and len(c["ulower"]) == 3
    assert (c["xrhorizon"], c["urhorizon"], c["rhorizon"]) == (2, 3, 3)
    assert len(c["zones"]) == 1                                  # nzones=1
    assert len(c["zones"][0]) == 3                               # ncorners=2 admits size<=2 before push: 3 corners
    assert len(c["tracks"]) == 1 and c["tracks"][0]["waypoints"] == [[0, 1.51]]   # nwaypoints=1, ndatums=1
    c = load(H, xmls["edge_no_mex_count.xml"])
    assert c["tracks"] == [] and c["zones"] == []                # <mexzones> without nzones loads nothing


def test_save_load_roundtrip(H, tmp_path, xmls):
    a = load(H, xmls["ocp_2d_ex1.xml"])
    b = json.loads(H.harness_roundtrip_configs(xmls["ocp_2d_ex1.xml"].encode(),
                                               str(tmp_path / "rt.xml").encode()).decode())
    assert a == b


def test_csv_writer_format_and_no_overwrite(H, tmp_path):
    p = str(tmp_path / "traj.csv")
    n1 = H.harness_save_csv(p.encode()).decode()
    n2 = H.harness_save_csv(p.encode()).decode()
    n3 = H.harness_save_csv(p.encode()).decode()
    assert n1 == p and n2.endswith("traj1.csv") and n3.endswith("traj2.csv")
    txt = open(n1).read()
    assert txt == "time,traj0,traj1\n0.000000,1.000000,2.500000\n0.500000,1.250000,-3.000000\n1.000000,0.000000,4.000000"


def test_linear_interpolation_template(H):
    tv = np.array([0.0, 4.0, 10.0]); ref = np.array([1.0, 3.0, -3.0])
    q = np.array([-2.0, 0.0, 1.0, 4.0, 9.0, 10.0, 11.0])
    out = np.zeros(len(q))
    D = C.POINTER(C.c_double)
    H.harness_lin_interp(len(q), q.ctypes.data_as(D), 3, tv.ctypes.data_as(D), ref.ctypes.data_as(D), out.ctypes.data_as(D))
    oxc, _ = O.track_centres(tv, ref, ref, q)
    assert np.array_equal(out, oxc)
    assert np.allclose(out[1:6], np.interp(q[1:6], tv, ref)) and np.isclose(out[0], 0.0) and np.isclose(out[6], -4.0)


def test_dense_ldlt_inertia_and_solve(H):
    rng = np.random.default_rng(7)
    D = C.POINTER(C.c_double)
    for n, npos in ((1, 1), (2, 1), (7, 3), (40, 25), (133, 60)):
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        ev = np.concatenate([rng.uniform(0.5, 3, npos), -rng.uniform(0.5, 3, n - npos)])
        A = (Q * ev) @ Q.T
        A = (A + A.T) / 2
        b = rng.standard_normal(n)
        x = b.copy()
        inertia = (C.c_int * 3)()
        assert H.harness_ldlt(n, np.ascontiguousarray(A).ctypes.data_as(D), x.ctypes.data_as(D), inertia) == 0
        assert list(inertia) == [npos, n - npos, 0]
        assert np.abs(A @ x - b).max() < 1e-10
    # KKT-shaped matrix with a zero (2,2) block
    n, m = 30, 12
    Hm = rng.standard_normal((n, n)); Hm = Hm @ Hm.T + np.eye(n)
    J = rng.standard_normal((m, n))
    K = np.block([[Hm, J.T], [J, np.zeros((m, m))]])
    b = rng.standard_normal(n + m); x = b.copy()
    inertia = (C.c_int * 3)()
    assert H.harness_ldlt(n + m, np.ascontiguousarray(K).ctypes.data_as(D), x.ctypes.data_as(D), inertia) == 0
    assert list(inertia) == [n, m, 0] and np.abs(K @ x - b).max() < 1e-9


# ---- NLP iteration logic, driven by the CPU oracle as evaluator (test-only plumbing) -----------
def _solve_with_oracle(H, xml, with_obstacles, tol=1e-9, max_iter=400):
    D = C.POINTER(C.c_double)
    H.harness_solve_example1_oracle.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_double, C.c_int, C.c_int, D,
                                                C.POINTER(C.c_int), D, D, C.c_int, C.POINTER(C.c_int)]
    H.harness_last_message.restype = C.c_char_p
    X, U = np.zeros((2, 64)), np.zeros((2, 64))
    cost, M, it = C.c_double(), C.c_int(), C.c_int()
    rc = H.harness_solve_example1_oracle(xml.encode(),
                                         os.path.join(ROOT, "oracle", "liboracle.so").encode(), with_obstacles, tol, 0,
                                         max_iter, C.byref(cost), C.byref(M), X.ctypes.data_as(D), U.ctypes.data_as(D),
                                         64, C.byref(it))
    assert rc == 0, H.harness_last_message().decode()
    m = M.value
    return cost.value, X.reshape(-1)[:2 * m].reshape(2, m), U.reshape(-1)[:2 * m].reshape(2, m), it.value


def test_nlp_iteration_reaches_analytic_optimum(H, xmls):
    g = json.load(open(os.path.join(GOLD, "ocp2d.json")))
    cost, X, U, iters = _solve_with_oracle(H, xmls["ocp_2d_ex1.xml"], 0)
    assert iters < 40 and abs(cost - g["cost"]) < 1e-6 * g["cost"]
    assert np.abs(U[0] - g["u"][0]).max() < 1e-6 and np.abs(U[1] - g["u"][1]).max() < 1e-6


def test_nlp_iteration_with_keepouts_is_feasible_and_stationary(H, xmls):
    import cases
    g = json.load(open(os.path.join(GOLD, "ocp2d.json")))
    cost, X, U, iters = _solve_with_oracle(H, xmls["ocp_2d_ex1.xml"], 1)
    assert iters < 200 and g["cost"] < cost < 2 * g["cost"]
    M = 33
    mesh = O.lgl(M)
    recs, tx, ty = cases.ocp2d_tables(O.edge_ellipse, O.track_centres, 8.0 * (mesh[0] + 1))
    RES, _, COST = O.evaluate(0, [], M, mesh, 0.0, 16.0, X[None], U[None], recs, (tx, ty))
    assert np.abs(RES[0, :2]).max() < 1e-8 and RES[0, 2:].max() < 1e-8 and abs(COST[0] - cost) < 1e-10
    assert np.all(np.abs(U) <= 0.5 + 1e-9) and abs(X[0, -1] - 5) <= 0.01 + 1e-9 and abs(X[1, -1] - 4) <= 0.01 + 1e-9


def test_nlp_iteration_quadrotor_vgp(H):
    """The headline model as an NLP: 6-state quadrotor, 25 LGL nodes, two disc keep-outs."""
    D = C.POINTER(C.c_double)
    H.harness_solve_quadrotor_oracle.argtypes = [C.c_char_p, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, D,
                                                 C.POINTER(C.c_int), D, D, C.c_int, C.POINTER(C.c_int)]
    H.harness_last_message.restype = C.c_char_p
    X, U = np.zeros(6 * 64), np.zeros(2 * 64)
    cost, M, it = C.c_double(), C.c_int(), C.c_int()
    rc = H.harness_solve_quadrotor_oracle(os.path.join(ROOT, "oracle", "liboracle.so").encode(), 24, 0.16, 2, 1e-8, 0,
                                          C.byref(cost), C.byref(M), X.ctypes.data_as(D), U.ctypes.data_as(D), 64,
                                          C.byref(it))
    assert rc == 0, H.harness_last_message().decode()
    m = M.value
    X, U = X[:6 * m].reshape(6, m), U[:2 * m].reshape(2, m)
    mesh = O.lgl(m)
    recs = np.array([[1, 4.0, 3.2, 0.64, 0, 0, 0, 0], [1, 6.3, 4.4, 0.49, 0, 0, 0, 0]], dtype=float)
    RES, _, COST = O.evaluate(1, [1.0, 0.01, 9.81, 1.0, 1.0], m, mesh, 0.0, 24 * 0.16, X[None], U[None], recs)
    assert np.abs(RES[0, :6]).max() < 1e-7 and RES[0, 6:].max() < 1e-7 and abs(COST[0] - cost.value) < 1e-8
    assert np.allclose(X[:, 0], [1, 1, 0, 0, 0, 0]) and np.all(np.abs(X[:3, -1] - [8, 6, 0]) <= 0.01 + 1e-9)
    assert U[0].min() >= -1e-9 and U[0].max() <= 25 + 1e-9 and np.abs(U[1]).max() <= 1 + 1e-9


def test_newton_step_backend_interface_keeps_the_inertia_right(H):
    """solve_nlp hands a quasi-definite matrix (node blocks convexified) to its KktBackend and recovers
    the exact step by a Woodbury correction.  Driven here with a host stand-in for the device backend
    (same matrix as etol_amd/csrc/emi_kkt.hip, dense LDL^T so that the inertia can be counted): the
    factorised matrix must have exactly nz positive / md negative eigenvalues at EVERY iteration, and
    the iteration must end where the built-in host backend ends."""
    D = C.POINTER(C.c_double)
    H.harness_solve_quadrotor_oracle.argtypes = [C.c_char_p, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, D,
                                                 C.POINTER(C.c_int), D, D, C.c_int, C.POINTER(C.c_int)]
    H.harness_set_linear_solver.argtypes = [C.c_char_p]
    H.harness_last_message.restype = C.c_char_p
    out = {}
    try:
        for name in ("host", "device"):
            H.harness_set_linear_solver(name.encode())
            X, U = np.zeros(6 * 64), np.zeros(2 * 64)
            cost, M, it = C.c_double(), C.c_int(), C.c_int()
            rc = H.harness_solve_quadrotor_oracle(os.path.join(ROOT, "oracle", "liboracle.so").encode(), 24, 0.16, 2, 1e-8,
                                                  0, C.byref(cost), C.byref(M), X.ctypes.data_as(D), U.ctypes.data_as(D),
                                                  64, C.byref(it))
            assert rc == 0, H.harness_last_message().decode()
            out[name] = (cost.value, X.copy(), it.value)
        assert H.harness_kkt_standin_wrong_inertia() == 0
    finally:
        H.harness_set_linear_solver(b"auto")
    assert out["host"][2] < 150 and out["device"][2] < 150
    assert abs(out["host"][0] - out["device"][0]) < 1e-8 * out["host"][0]
    assert np.abs(out["host"][1] - out["device"][1]).max() < 1e-6


def test_scaled_iteration_reaches_the_unscaled_optimum(H, xmls):
    """Alg::scaling = "automatic" (ePSOPT.cpp:63): the iteration on bound-scaled variables / state-scaled defect rows (quadrotor:
    scales 10, 10, 1.2, 6, 6, 4 | 25, 1) ends at the trajectory of the unscaled one where the problem has ONE optimum (no
    keep-outs), in the caller's units; with keep-outs it may settle in another homotopy class (here: 398.4 against 389.3, the
    other side of the second disc), so that case is checked as what it claims to be: feasible to 1e-8 against the oracle."""
    D = C.POINTER(C.c_double)
    H.harness_solve_quadrotor_oracle.argtypes = [C.c_char_p, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, D,
                                                 C.POINTER(C.c_int), D, D, C.c_int, C.POINTER(C.c_int)]
    H.harness_last_message.restype = C.c_char_p
    H.harness_set_scaling.argtypes = [C.c_int]
    out = {}
    try:
        for nd in (0, 2):
            for mode in (0, 1):
                H.harness_set_scaling(mode)
                X, U = np.zeros(6 * 64), np.zeros(2 * 64)
                cost, M, it = C.c_double(), C.c_int(), C.c_int()
                rc = H.harness_solve_quadrotor_oracle(os.path.join(ROOT, "oracle", "liboracle.so").encode(), 24, 0.16, nd, 1e-9,
                                                      0, C.byref(cost), C.byref(M), X.ctypes.data_as(D), U.ctypes.data_as(D), 64,
                                                      C.byref(it))
                assert rc == 0, H.harness_last_message().decode()
                m = M.value
                out[nd, mode] = (cost.value, X[:6 * m].reshape(6, m).copy(), U[:2 * m].reshape(2, m).copy(), it.value)
    finally:
        H.harness_set_scaling(-1)
    a, b = out[0, 0], out[0, 1]
    assert b[3] <= a[3] and abs(a[0] - b[0]) < 1e-10 * a[0]
    assert np.abs(a[1] - b[1]).max() < 1e-6 and np.abs(a[2] - b[2]).max() < 1e-5
    cost, X, U, iters = out[2, 1]
    m = X.shape[1]
    recs = np.array([[1, 4.0, 3.2, 0.64, 0, 0, 0, 0], [1, 6.3, 4.4, 0.49, 0, 0, 0, 0]], dtype=float)
    RES, _, COST = O.evaluate(1, [1.0, 0.01, 9.81, 1.0, 1.0], m, O.lgl(m), 0.0, 24 * 0.16, X[None], U[None], recs)
    assert iters < 150 and np.abs(RES[0, :6]).max() < 1e-8 and RES[0, 6:].max() < 1e-8 and abs(COST[0] - cost) < 1e-8
    assert U[0].min() >= -1e-9 and U[0].max() <= 25 + 1e-9 and np.abs(U[1]).max() <= 1 + 1e-9


def test_nlp_iteration_fixedwing_lateral_offset(H):
    """The 12-state fixed-wing model as an NLP: trimmed flight at 25 m/s, 8 s, end 10 m to the side.
    Coupled, strongly nonconvex node blocks -- the case that exercises the inertia handling and the
    residual-based step acceptance of solve_nlp."""
    from etol_amd import workloads as W
    D = C.POINTER(C.c_double)
    H.harness_solve_fixedwing_oracle.argtypes = [C.c_char_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, D,
                                                 C.POINTER(C.c_int), D, D, C.c_int, C.POINTER(C.c_int)]
    H.harness_last_message.restype = C.c_char_p
    X, U = np.zeros(12 * 32), np.zeros(4 * 32)
    cost, M, it = C.c_double(), C.c_int(), C.c_int()
    rc = H.harness_solve_fixedwing_oracle(os.path.join(ROOT, "oracle", "liboracle.so").encode(), 24, 8.0, 10.0, 1e-7, 0,
                                          C.byref(cost), C.byref(M), X.ctypes.data_as(D), U.ctypes.data_as(D), 32,
                                          C.byref(it))
    assert rc == 0, H.harness_last_message().decode()
    m = M.value
    assert m == 25 and it.value < 250
    X, U = X[:12 * m].reshape(12, m), U[:4 * m].reshape(4, m)
    RES, _, COST = O.evaluate(2, W.FW_PARAMS, m, O.lgl(m), 0.0, 8.0, X[None], U[None])
    assert np.abs(RES[0, :12]).max() < 1e-6 and abs(COST[0] - cost.value) < 1e-8 * abs(cost.value)
    assert abs(X[1, -1] - 10.0) <= 0.5 + 1e-9 and abs(X[1, 0]) < 1e-12          # east offset reached within xtol
    assert U[0].min() >= -1e-9 and U[0].max() <= 60 + 1e-9 and np.abs(U[1:]).max() <= 0.5 + 1e-9
    assert np.abs(X[3]).max() > 0.02                                             # it banks to get there


@pytest.mark.skipif(not os.path.isdir("/root/reference/include/ETOL"), reason="reference tree not present (GPU box)")
def test_host_sources_compile_against_the_reference_headers():
    """Drop-in check: the eSolver, its NLP/trace code and the example compile with the REFERENCE's
    TrajectoryOptimizer.hpp / ETOL_Types.hpp first on the include path (ours only supplies eMI355X*.hpp, emi355x.h)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    srcs = ["etol_amd/host/eMI355X.cpp", "etol_amd/host/emi_nlp.cpp", "etol_amd/host/emi_trace.cpp",
            "etol_amd/examples/etol_mi355x_example1.cpp", "etol_amd/examples/etol_mi355x_montecarlo.cpp"]
    for src in srcs:
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I/root/reference/include", "-Iinclude", "-Ietol_amd/host", src],
                           cwd=root, capture_output=True, text=True)
        assert r.returncode == 0, src + "\n" + r.stderr[-3000:]


@pytest.mark.parametrize("disc_r,scaling", [(0.9, 0), (0.9, 1), (0.5, 1), (0.0, 0)])
def test_delayed_problem_is_solved_through_coupling_rows(H, disc_r, scaling):
    """Delayed states / controls (reference src/ePSOPT/ePSOPT.cpp:231-248) in solve(): the delayed values are node variables
    tied to their sources by coupling rows with the interpolation operators (mi355x::NlpLink, make_nlp of a lifted Prob).
    Here on the CPU: the 2-state demo problem (state horizon 3, control horizon 1) with the oracle's model 3 as evaluator and
    the dense host LDL^T; the result must be a KKT point of the same NLP restated in tests/indep_nlp.py (DelayedNlp: the
    oracle's functions and the oracle's own delay matrices), which an independent Newton polish confirms within 1e-6."""
    import indep_nlp as N
    nsteps, dt = 24, 0.25
    M = nsteps + 1
    D = C.POINTER(C.c_double)
    H.harness_solve_delay_demo_oracle.argtypes = [C.c_char_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, D, D,
                                                  C.POINTER(C.c_int)]
    H.harness_last_message.restype = C.c_char_p
    Z = np.zeros(10 * M)
    cost, iters = C.c_double(), C.c_int()
    rc = H.harness_solve_delay_demo_oracle(os.path.join(ROOT, "oracle", "liboracle.so").encode(), nsteps, dt, disc_r, 1e-10, 0, scaling, C.byref(cost),
                                           Z.ctypes.data_as(D), C.byref(iters))
    assert rc == 0, H.harness_last_message().decode()
    P = N.DelayedNlp(nsteps, dt, disc_r=disc_r)
    # the iteration's answer satisfies the delayed dynamics and the coupling rows
    assert np.abs(P.defect(Z)).max() < 1e-8
    assert abs(P.cost(Z) - cost.value) < 1e-9 * max(1.0, abs(cost.value))
    zp, lamF, lamC, k = N.polish(P, Z)
    assert N.kkt_ok(k), k
    rel = np.abs(zp - Z).max() / np.abs(zp).max()
    print(f"delayed solve: {iters.value} iterations, cost {cost.value:.10f}, polished {P.cost(zp):.10f}, rel dist {rel:.2e}, KKT {k}")
    assert rel < 1e-6
    if disc_r >= 0.9:
        assert k["active_path_rows"] > 0          # the keep-out binds at this size
    # the delays matter: the same trajectory is not feasible for the problem without them (history = present)
    Zn = Z.copy()
    for dst, src, W in P.links:
        Zn[dst * M:(dst + 1) * M] = Z[src * M:(src + 1) * M]
    assert np.abs(N.Nlp.defect(P, Zn)).max() > 1e-3


def test_planned_cold_start_guess_finds_the_gap_in_a_wall_of_keepouts(built):
    """mi355x::planned_path_guess (the last cold-start guess of solve(): a route through the free space of the static keep-outs, the node
    positions spread along it).  The layout is Monte-Carlo scenario 938 of config 4 (profiles/r04_notes.md section 22): the start sits
    under a wall of overlapping discs that the straight line to the goal enters at once; the only way out is the gap below the first
    disc.  The planned positions start and end where they must, stay outside every disc, and leave through that gap; with the gap closed
    and the start walled in there is no route and the function says so."""
    import ctypes as C
    H = C.CDLL(os.path.join(ROOT, "tests", "harness", "libetol_harness.so"))
    D = C.POINTER(C.c_double)
    H.harness_planned_path.argtypes = [C.c_int, D, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, D, D]
    discs = np.array([(1.10, 2.09, 0.44), (1.27, 4.12, 0.23), (8.37, 4.69, 0.38), (1.96, 2.52, 0.58), (7.62, 6.90, 0.52), (1.23, 3.84, 0.53),
                      (1.41, 7.11, 0.37), (7.74, 3.61, 0.50), (1.49, 7.12, 0.48), (7.21, 5.93, 0.21), (6.95, 6.54, 0.25), (1.23, 2.72, 0.33),
                      (4.74, 5.63, 0.22), (4.04, 3.13, 0.30), (8.46, 4.16, 0.33), (1.75, 1.64, 0.41), (6.23, 3.88, 0.32), (7.08, 4.23, 0.52),
                      (4.08, 2.70, 0.29), (6.04, 3.69, 0.44)])
    M = 65
    xs, ys = np.zeros(M), np.zeros(M)
    dp = lambda a: a.ctypes.data_as(D)
    flat = np.ascontiguousarray(discs.ravel())
    assert H.harness_planned_path(len(discs), dp(flat), 1.0, 1.0, 8.0, 6.0, 0.0, 10.0, M, dp(xs), dp(ys)) == 0
    assert abs(xs[0] - 1) < 1e-12 and abs(ys[0] - 1) < 1e-12 and abs(xs[-1] - 8) < 1e-12 and abs(ys[-1] - 6) < 1e-12
    clear = np.min(np.hypot(xs[:, None] - discs[None, :, 0], ys[:, None] - discs[None, :, 1]) - discs[None, :, 2], axis=1)
    assert clear.min() > 0.0, clear.min()                                # no node inside a disc
    k = np.argmin(np.abs(xs - 1.75))
    assert ys[k] < 1.64 - 0.41                                            # out through the gap BELOW the first disc
    assert np.all(np.diff(np.hypot(np.diff(xs), np.diff(ys)) / np.diff(np.arange(M))) < 10)     # (finite steps: sanity)
    # a ring of discs round the start: no route
    ring = np.array([(1 + 0.9 * np.cos(a), 1 + 0.9 * np.sin(a), 0.5) for a in np.linspace(0, 2 * np.pi, 16, endpoint=False)])
    flat = np.ascontiguousarray(ring.ravel())
    assert H.harness_planned_path(len(ring), dp(flat), 1.0, 1.0, 8.0, 6.0, -5.0, 10.0, M, dp(xs), dp(ys)) == 1
