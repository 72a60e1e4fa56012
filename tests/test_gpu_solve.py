"""eMI355X::setup()/solve() on the GPU through the C++ host library.  -m gpu"""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

import cases
import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def H(built):
    import torch  # noqa: F401
    lib = C.CDLL(os.path.join(ROOT, "tests", "harness", "libetol_harness.so"))
    D = C.POINTER(C.c_double)
    lib.harness_solve_example1.argtypes = [C.c_char_p, C.c_int, C.c_double, C.c_int, D, C.POINTER(C.c_int), D, D, D,
                                           C.c_int, C.POINTER(C.c_int)]
    lib.harness_last_message.restype = C.c_char_p
    lib.harness_last_linear_solver.restype = C.c_char_p
    lib.harness_set_linear_solver.argtypes = [C.c_char_p]
    return lib


def solve(H, xml, with_obstacles, tol=1e-9):
    cap = 160
    X, U, T = np.zeros((2, cap)), np.zeros((2, cap)), np.zeros(cap)
    cost, M, iters = C.c_double(), C.c_int(), C.c_int()
    D = C.POINTER(C.c_double)
    rc = H.harness_solve_example1(xml.encode(), with_obstacles, tol, 0, C.byref(cost), C.byref(M), X.ctypes.data_as(D),
                                  U.ctypes.data_as(D), T.ctypes.data_as(D), cap, C.byref(iters))
    assert rc == 0, H.harness_last_message().decode()
    m = M.value
    return cost.value, X.reshape(-1)[:2 * m].reshape(2, m), U.reshape(-1)[:2 * m].reshape(2, m), T[:m], iters.value


def test_obstacle_free_problem_reaches_the_analytic_optimum(H, xmls):
    """north_star tolerance: trajectories within 1e-6 relative of the known optimum."""
    g = json.load(open(os.path.join(GOLD, "ocp2d.json")))
    cost, X, U, T, iters = solve(H, xmls["ocp_2d_ex1.xml"], 0)
    assert X.shape[1] == 33 and iters < 60
    assert abs(cost - g["cost"]) < 1e-6 * g["cost"]
    assert np.abs(U[0] - g["u"][0]).max() < 1e-6 and np.abs(U[1] - g["u"][1]).max() < 1e-6
    tau = O.lgl(33)[0]
    assert np.abs(T - 8.0 * (tau + 1)).max() < 1e-13            # LGL times, not k*dt (ePSOPT.cpp:157-182)
    assert np.abs(X[0] - (1 + g["u"][0] * T)).max() < 1e-6 * 5 and np.abs(X[1] - (2 + g["u"][1] * T)).max() < 1e-6 * 4
    assert X[0, 0] == 1.0 and X[1, 0] == 2.0                      # hard initial state


def test_shipped_problem_with_keepouts_solves_and_is_feasible(H, xmls):
    g = json.load(open(os.path.join(GOLD, "ocp2d.json")))
    cost, X, U, T, iters = solve(H, xmls["ocp_2d_ex1.xml"], 1)
    assert cost > g["cost"]                                       # the straight line crosses exz1
    assert cost < 2.0 * g["cost"]
    # independent feasibility check with the CPU oracle, on whatever mesh the refinement ended on
    # (controls that ride their bounds make the interpolant overshoot, so ePSOPT-style automatic
    # refinement may add nodes here even though xdot = u is integrated exactly)
    M = X.shape[1]
    assert M >= 33
    mesh = O.lgl(M)
    assert np.abs(T - 8.0 * (mesh[0] + 1)).max() < 1e-12
    recs, tx, ty = cases.ocp2d_tables(O.edge_ellipse, O.track_centres, 8.0 * (mesh[0] + 1))
    RES, _, COST = O.evaluate(0, [], M, mesh, 0.0, 16.0, X[None], U[None], recs, (tx, ty))
    assert np.abs(RES[0, :2]).max() < 1e-7                        # defects
    assert RES[0, 2:].max() < 1e-7 and RES[0, 2:].min() > -1000    # path rows within [-1000, 0]
    assert abs(COST[0] - cost) < 1e-9
    assert np.all(np.abs(U) <= 0.5 + 1e-9) and np.all(X >= -1e-9) and np.all(X <= 7 + 1e-9)
    assert abs(X[0, -1] - 5) <= 0.01 + 1e-9 and abs(X[1, -1] - 4) <= 0.01 + 1e-9


def _independent_optimum(name, cost):
    """the local optimum of tests/golden/solve_optima.json (scipy SLSQP + active-set Newton on the oracle's functions,
    tests/golden/gen_solve_fixtures.py) whose cost is closest to `cost`"""
    prob = json.load(open(os.path.join(GOLD, "solve_optima.json")))["problems"][name]
    o = min(prob["optima"], key=lambda q: abs(q["cost"] - cost))
    return o["cost"], np.array(o["X"]), np.array(o["U"]), [q["cost"] for q in prob["optima"]]


def _assert_trajectory_parity(tag, cost, X, U, name):
    """north_star's tolerance: solved trajectories within 1e-6 relative of the CPU result on the same problem.
    Keep-outs make these problems non-convex with many local optima (constraints act at the nodes only), so the CPU
    result is (a) a stored optimum of the independent optimiser when the solve landed in one of those basins, else
    (b) the KKT point the independent Newton polish (tests/indep_nlp.py, oracle functions only) converges to from the
    GPU trajectory: the GPU result must be within 1e-6 of an exact local solution of the oracle-defined NLP."""
    import indep_nlp as N
    c, Xs, Us, all_costs = _independent_optimum(name, cost)
    nc = Us.shape[0]
    stored = X.shape == Xs.shape and abs(cost - c) < 1e-6 * abs(c)
    if not stored:
        P = {"ocp_2d_ex1": lambda: N.shipped_problem()[0], "quadrotor_41": N.quad_problem, "mip_2d_ex1": N.mip_problem}[name]()
        assert X.shape == (P.ns, P.M), (X.shape, P.M)
        z, lamF, lamC, k = N.polish(P, np.concatenate([X.ravel(), U[:nc].ravel()]))
        assert N.kkt_ok(k), k
        Xs, Us = P.split(z)[0][0], P.split(z)[1][0]
        c = P.cost(z)
        out = os.environ.get("EMI_TEST_WRITE_STARTS", "")      # a test writes nothing unless asked: a directory here
        if out and os.path.isdir(out):                         # collects starts for tests/golden/gen_solve_fixtures.py (rounded: a start, not an answer)
            path = os.path.join(out, "solve_starts.json")
            d = json.load(open(path)) if os.path.exists(path) else {}
            d.setdefault(name, []).append(dict(X=np.round(X, 3).tolist(), U=np.round(U[:nc], 3).tolist()))
            json.dump(d, open(path, "w"))
    ex = np.abs(X - Xs).max() / np.abs(Xs).max()
    eu = np.abs(U[:nc] - Us).max() / np.abs(Us).max()
    print(f"{tag}: cost {cost:.10f} vs {'stored optimum' if stored else 'KKT point polished from it'} {c:.10f} "
          f"(stored: {all_costs}); rel err states {ex:.2e}, controls {eu:.2e}")
    assert abs(cost - c) < 1e-6 * abs(c), (cost, c)
    assert ex < 1e-6 and eu < 1e-6, (ex, eu)
    return stored


def test_shipped_problem_with_all_keepouts_matches_an_independent_optimiser(H, xmls):
    """resource/configs/ocp_2d_ex1.xml with its 9 ellipse rows and 2 moving-disc rows active, 33 nodes: the trajectory
    ETOL::eMI355X::solve() returns is a local optimum an independent CPU optimiser also finds, to 1e-6."""
    H.harness_set_refine.argtypes = [C.c_int]
    H.harness_set_refine(0)                  # the fixture is the optimum ON the 33-node mesh
    try:
        cost, X, U, T, iters = solve(H, xmls["ocp_2d_ex1.xml"], 1, tol=1e-10)
    finally:
        H.harness_set_refine(-1)
    _assert_trajectory_parity("ocp_2d_ex1", cost, X, U, "ocp_2d_ex1")


def test_quadrotor_41_nodes_matches_an_independent_optimiser(H):
    cost, X, U, iters, mesh_iters, _ = _solve_quadrotor(H, 40, 0.1, 2, refine=0, tol=1e-10)
    _assert_trajectory_parity("quadrotor_41", cost, X, U, "quadrotor_41")


def test_jacobian_based_defect_scaling_reaches_the_stored_optima(H, xmls):
    """Alg::defect_scaling = "jacobian-based" (the shipped example asks PSOPT for it, reference
    src/Examples/PSOPT/etol_psopt_example1.cpp:90-91): defect rows weighted by the reciprocal of their Jacobian row norm in the merit
    function of the NLP iteration.  The shipped problem with all 11 rows (33 nodes, host LDL^T) and the 41-node quadrotor (device
    Newton step) must still end within 1e-6 of an optimum the independent CPU optimiser holds or confirms."""
    H.harness_set_refine.argtypes = [C.c_int]
    H.harness_set_defect_scaling.argtypes = [C.c_int]
    H.harness_set_defect_scaling(1)
    H.harness_set_refine(0)
    try:
        cost, X, U, T, iters = solve(H, xmls["ocp_2d_ex1.xml"], 1, tol=1e-10)
        _assert_trajectory_parity("ocp_2d_ex1 (jacobian-based defect scaling)", cost, X, U, "ocp_2d_ex1")
        cost, X, U, iters, mesh_iters, _ = _solve_quadrotor(H, 40, 0.1, 2, refine=0, tol=1e-10)
        _assert_trajectory_parity("quadrotor_41 (jacobian-based defect scaling)", cost, X, U, "quadrotor_41")
        print(f"quadrotor_41 with jacobian-based defect scaling: {iters} iterations")
    finally:
        H.harness_set_refine(-1)
        H.harness_set_defect_scaling(0)


def test_shipped_mip_configuration_is_solved_from_a_bent_start(H, xmls):
    """mip_2d_ex1.xml as the reference's container feeds it to the PSOPT example: 17 nodes, tf = 8, four controls of
    which the callbacks read two.  The straight-line start ends locally infeasible; solve() retries from bent lines
    (IPOPT's restoration phase does that job for ePSOPT) and lands on the optimum the independent optimiser finds."""
    H.harness_set_traced.argtypes = [C.c_int]
    H.harness_set_refine.argtypes = [C.c_int]
    H.harness_set_traced(1)
    H.harness_set_refine(0)
    try:
        cost, X, U, T, iters = solve(H, xmls["mip_2d_ex1.xml"], 1, tol=1e-10)
    finally:
        H.harness_set_traced(0)
        H.harness_set_refine(-1)
    _assert_trajectory_parity("mip_2d_ex1", cost, X, U, "mip_2d_ex1")


def test_ode_error_estimate_matches_an_independent_evaluation(H):
    """The mesh-refinement criterion (PSOPT's relative local error: the ODE residual of the interpolated solution
    integrated between consecutive nodes, ePSOPT.cpp:69-71) computed by ETOL::eMI355X::odeError -- f at the
    quadrature points on the device -- against the same quantity from numpy + the CPU oracle, on the stored optimum
    of the 41-node quadrotor problem and on a rough trajectory."""
    prob = json.load(open(os.path.join(GOLD, "solve_optima.json")))["problems"]["quadrotor_41"]
    M, tf = prob["M"], prob["tf"]
    h = tf / 2
    tau, w, D = O.lgl(M)
    gx = np.polynomial.legendre.leggauss(5)
    D_ = C.POINTER(C.c_double)
    H.harness_ode_error_quadrotor.argtypes = [C.c_int, C.c_double, C.c_int, D_, D_, D_]
    rng = np.random.default_rng(3)
    for label in ("optimum", "perturbed"):
        X, U = np.array(prob["optima"][0]["X"]), np.array(prob["optima"][0]["U"])
        if label == "perturbed":
            X = X + 0.05 * rng.standard_normal(X.shape) * np.sin(np.pi * (tau + 1) / 2)
            U = np.clip(U + 0.3 * rng.standard_normal(U.shape), [[0.0], [-1.0]], [[25.0], [1.0]])
        got = C.c_double()
        assert H.harness_ode_error_quadrotor(M - 1, tf / (M - 1), 2, np.ascontiguousarray(X).ctypes.data_as(D_),
                                             np.ascontiguousarray(U).ctypes.data_as(D_), C.byref(got)) == 0
        # independent evaluation: Lagrange interpolant by polynomial fitting in the Legendre basis, oracle f
        Z = np.vstack([X, U])
        coef = [np.polynomial.legendre.Legendre.fit(tau, Z[v], M - 1, domain=[-1, 1]) for v in range(8)]
        pts = np.concatenate([tau[k] + 0.5 * (tau[k + 1] - tau[k]) * (gx[0] + 1) for k in range(M - 1)])
        Zq = np.array([c(pts) for c in coef])
        Zq[6] = np.clip(Zq[6], 0, 25); Zq[7] = np.clip(Zq[7], -1, 1)
        dZq = np.array([coef[i].deriv()(pts) for i in range(6)]) / h
        P = len(pts)
        RES, _, _ = O.evaluate(1, [1.0, 0.01, 9.81, 1.0, 1.0], P, (pts, np.zeros(P), np.zeros((P, P))), 0.0, tf,
                               Zq[None, :6], Zq[None, 6:])
        f = -RES[0, :6] / h
        resid = np.abs(dZq - f).reshape(6, M - 1, 5)
        eta = (resid * gx[1]).sum(axis=2) * 0.5 * np.diff(tau) * h
        wi = np.maximum(np.abs(X).max(axis=1), np.abs(X @ D.T).max(axis=1) / h)
        ref = (eta / (wi[:, None] + 1)).max()
        print(f"ODE error estimate, {label}: device-assisted {got.value:.6e}, independent {ref:.6e}")
        assert abs(got.value - ref) < 1e-7 * ref + 1e-13
    assert got.value > 1e-3          # the rough trajectory is far from satisfying the ODE between its nodes


def test_example_program_runs(built, tmp_path, xmls):
    exe = os.path.join(ROOT, "etol_amd", "lib", "etol_mi355x_example1")
    r = subprocess.run([exe, xmls["ocp_2d_ex1.xml"]], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Minimization Score" in r.stdout and "Graceful Exit" in r.stdout
    rows = open(tmp_path / "state_mi355x1.csv").read().split("\n")
    assert rows[0] == "time,traj0,traj1" and len(rows) >= 34


def test_example_program_takes_the_shipped_mip_configuration(built, tmp_path, xmls):
    """resource/configs/mip_2d_ex1.xml (states rhorizon="1", four controls) is what the reference's container feeds
    its PSOPT example (container/singularity/ETOL-examples.def:131-132): a state horizon of 1 adds no delayed
    state (ePSOPT.cpp:231), the two extra controls are free variables the callbacks never read."""
    exe = os.path.join(ROOT, "etol_amd", "lib", "etol_mi355x_example1")
    r = subprocess.run([exe, xmls["mip_2d_ex1.xml"]], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "not supported" not in r.stdout + r.stderr                   # setup() takes the configuration
    assert "Minimization Score" in r.stdout and "Graceful Exit" in r.stdout
    rows = open(tmp_path / "control_mi355x1.csv").read().split("\n")
    assert rows[0] == "time,traj0,traj1,traj2,traj3" and len(rows) >= 18


def _solve_quadrotor(H, nsteps, dt, ndiscs, refine, ode_tol=1e-4, tol=1e-8):
    D = C.POINTER(C.c_double)
    H.harness_solve_quadrotor.argtypes = [C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, C.c_int, C.c_double, D,
                                          C.POINTER(C.c_int), D, D, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), D]
    X, U = np.zeros(6 * 160), np.zeros(2 * 160)
    cost, M, it, mit, oerr = C.c_double(), C.c_int(), C.c_int(), C.c_int(), C.c_double()
    rc = H.harness_solve_quadrotor(nsteps, dt, ndiscs, tol, 0, refine, ode_tol, C.byref(cost), C.byref(M),
                                   X.ctypes.data_as(D), U.ctypes.data_as(D), 160, C.byref(it), C.byref(mit), C.byref(oerr))
    assert rc == 0, H.harness_last_message().decode()
    m = M.value
    return cost.value, X[:6 * m].reshape(6, m), U[:2 * m].reshape(2, m), it.value, mit.value, oerr.value


def test_quadrotor_vgp_solves_on_the_gpu(H):
    """6-state quadrotor VGP (the headline model), 41 LGL nodes, two disc keep-outs, through
    ETOL::eMI355X setup()/solve(); feasibility checked independently with the CPU oracle."""
    cost, X, U, iters, mesh_iters, _ = _solve_quadrotor(H, 40, 0.1, 2, refine=0)
    m = X.shape[1]
    assert m == 41 and iters < 300 and mesh_iters == 1
    mesh = O.lgl(m)
    recs = np.array([[1, 4.0, 3.2, 0.64, 0, 0, 0, 0], [1, 6.3, 4.4, 0.49, 0, 0, 0, 0]], dtype=float)
    RES, _, COST = O.evaluate(1, [1.0, 0.01, 9.81, 1.0, 1.0], m, mesh, 0.0, 4.0, X[None], U[None], recs)
    assert np.abs(RES[0, :6]).max() < 1e-7 and RES[0, 6:].max() < 1e-7 and abs(COST[0] - cost) < 1e-8
    assert np.allclose(X[:, 0], [1, 1, 0, 0, 0, 0]) and np.all(np.abs(X[:3, -1] - [8, 6, 0]) <= 0.01 + 1e-9)
    assert 380 < cost < 450      # ~ hover effort g^2 * tf plus the manoeuvre


def test_mesh_refinement_adds_nodes_until_the_ode_tolerance_is_met(H, xmls):
    """mesh_refinement="automatic" (ePSOPT's default, ePSOPT.cpp:69-71): a coarse quadrotor mesh is
    refined; the linear point-mass problem is exact on any mesh and is left alone."""
    H.harness_set_quad_tau_max.argtypes = [C.c_double]
    H.harness_set_quad_tau_max(20.0)       # torque never saturates: smooth solution, spectral convergence
    try:
        coarse = _solve_quadrotor(H, 12, 4.0 / 12, 1, refine=0)
        fine = _solve_quadrotor(H, 12, 4.0 / 12, 1, refine=1, ode_tol=1e-4)
        ref = _solve_quadrotor(H, 64, 4.0 / 64, 1, refine=0)      # dense-mesh answer
    finally:
        H.harness_set_quad_tau_max(1.0)
    assert coarse[1].shape[1] == 13 and coarse[4] == 1
    assert 13 < fine[1].shape[1] < 129 and fine[4] > 1 and fine[5] <= 1e-4
    # same problem on a dense mesh, cold-started: the nonconvex keep-out admits several nearby
    # local solutions, so the costs are compared loosely
    assert abs(fine[0] - ref[0]) < 1e-3 * ref[0]
    m = fine[1].shape[1]
    RES, _, COST = O.evaluate(1, [1.0, 0.01, 9.81, 1.0, 1.0], m, O.lgl(m), 0.0, 4.0, fine[1][None], fine[2][None],
                              np.array([[1, 4.0, 3.2, 0.64, 0, 0, 0, 0]], dtype=float))
    assert np.abs(RES[0, :6]).max() < 1e-7 and RES[0, 6:].max() < 1e-7 and abs(COST[0] - fine[0]) < 1e-8
    # shipped problem: dynamics xdot = u are integrated exactly by the collocation scheme
    cost, X, U, T, iters = solve(H, xmls["ocp_2d_ex1.xml"], 0)
    assert X.shape[1] == 33


def test_traced_callbacks_solve_like_the_builtin_models(H, xmls):
    """Callbacks written as arithmetic on the solver's scalar type (the ePSOPT way, reference
    etol_psopt_example1.cpp:101-138; here mi355x::Var): setup() traces them, differentiates the trace
    and compiles the model for the device.  Same problems, same answers as the hand-written kernels,
    including across mesh refinement (one compilation, several meshes)."""
    base1 = solve(H, xmls["ocp_2d_ex1.xml"], 1)
    baseq = _solve_quadrotor(H, 40, 0.1, 2, refine=0)
    H.harness_set_quad_tau_max.argtypes = [C.c_double]
    H.harness_set_traced.argtypes = [C.c_int]
    H.harness_set_traced(1)
    try:
        tr1 = solve(H, xmls["ocp_2d_ex1.xml"], 1)
        trq = _solve_quadrotor(H, 40, 0.1, 2, refine=0)
        H.harness_set_quad_tau_max(20.0)
        trr = _solve_quadrotor(H, 12, 4.0 / 12, 1, refine=1, ode_tol=1e-4)
    finally:
        H.harness_set_traced(0)
        H.harness_set_quad_tau_max(1.0)
    assert tr1[1].shape == base1[1].shape and abs(tr1[0] - base1[0]) < 1e-9 * base1[0]
    assert np.abs(tr1[1] - base1[1]).max() < 1e-7 and np.abs(tr1[2] - base1[2]).max() < 1e-7
    assert trq[1].shape == baseq[1].shape
    if abs(trq[0] - baseq[0]) < 1e-8 * baseq[0]:
        # both runs stop at a KKT error of 1e-8; the minimiser is determined to about the square root of that
        assert np.abs(trq[1] - baseq[1]).max() < 2e-5 and np.abs(trq[2] - baseq[2]).max() < 2e-4
    else:
        # The two keep-outs make this problem non-convex with several local optima (tests/golden/solve_optima.json holds 400.340 and
        # 402.807 among them), and the traced model agrees with the hand-written one to 1e-13, not to the bit: the iteration may part
        # ways at a step where two branches are within rounding.  Each run must then sit on an optimum the independent optimiser holds.
        for cost, X, U in ((trq[0], trq[1], trq[2]), (baseq[0], baseq[1], baseq[2])):
            c, Xs, Us, all_costs = _independent_optimum("quadrotor_41", cost)
            assert abs(cost - c) < 1e-6 * abs(c), (cost, all_costs)
            assert np.abs(X - Xs).max() / np.abs(Xs).max() < 1e-5 and np.abs(U - Us).max() / np.abs(Us).max() < 1e-4
    # refinement with the traced model: converged, and feasible for the oracle's hand-written equations
    m = trr[1].shape[1]
    assert m > 13 and trr[4] > 1 and trr[5] <= 1e-4
    RES, _, COST = O.evaluate(1, [1.0, 0.01, 9.81, 1.0, 1.0], m, O.lgl(m), 0.0, 4.0, trr[1][None], trr[2][None],
                              np.array([[1, 4.0, 3.2, 0.64, 0, 0, 0, 0]], dtype=float))
    assert np.abs(RES[0, :6]).max() < 1e-7 and RES[0, 6:].max() < 1e-7 and abs(COST[0] - trr[0]) < 1e-8


def test_solve_with_traced_rows_on_velocities_and_a_control(H):
    """A speed limit (vx^2 + vz^2 <= v^2) and a thrust-tilt coupling (T sin(theta) <= 0.6 v^2) written with the handles
    in a constraint callback, next to the two disc keep-outs of the record table: rows on four variables that are not
    the keep-outs' positions, one of them a control.  The solve must respect them (they are active) and cost more than
    the same problem without them."""
    H.harness_set_traced.argtypes = [C.c_int]
    H.harness_set_speed_limit.argtypes = [C.c_double]
    H.harness_set_traced(1)
    try:
        free = _solve_quadrotor(H, 40, 0.1, 2, refine=0)
        vmax_free = np.sqrt(free[1][3] ** 2 + free[1][4] ** 2).max()
        H.harness_set_speed_limit(0.8 * vmax_free)
        lim = _solve_quadrotor(H, 40, 0.1, 2, refine=0)
    finally:
        H.harness_set_speed_limit(0.0)
        H.harness_set_traced(0)
    X, U = lim[1], lim[2]
    v2 = (0.8 * vmax_free) ** 2
    speed = X[3] ** 2 + X[4] ** 2 - v2
    tilt = U[0] * np.sin(X[2]) - 0.6 * v2
    print(f"speed-limited solve: cost {lim[0]:.6f} (free {free[0]:.6f}), max speed row {speed.max():.2e}, max tilt row {tilt.max():.2e}")
    assert speed.max() < 1e-6 and tilt.max() < 1e-6            # feasible at every node
    assert speed.max() > -1e-4                                   # and the limit binds
    assert lim[0] > free[0] * (1 + 1e-4)
    m = X.shape[1]
    RES, _, COST = O.evaluate(1, [1.0, 0.01, 9.81, 1.0, 1.0], m, O.lgl(m), 0.0, 4.0, X[None], U[None],
                              np.array([[1, 4.0, 3.2, 0.64, 0, 0, 0, 0], [1, 6.3, 4.4, 0.49, 0, 0, 0, 0]], dtype=float))
    assert np.abs(RES[0, :6]).max() < 1e-7 and RES[0, 6:].max() < 1e-7 and abs(COST[0] - lim[0]) < 1e-7 * lim[0]


def _set_linear_solver(H, name):
    H.harness_set_linear_solver.argtypes = [C.c_char_p]
    H.harness_last_linear_solver.restype = C.c_char_p
    H.harness_set_linear_solver(name.encode())


def test_device_newton_step_gives_the_host_solution(H, xmls):
    """Same NLP iteration, Newton step once through the dense host backend and once through the device
    backend (Schur complement + Cholesky, low-rank correction on the device): same solutions."""
    try:
        _set_linear_solver(H, "host")
        h1 = solve(H, xmls["ocp_2d_ex1.xml"], 1)
        hq = _solve_quadrotor(H, 40, 0.1, 2, refine=0)
        assert H.harness_last_linear_solver().decode().startswith("host")
        _set_linear_solver(H, "device")
        d1 = solve(H, xmls["ocp_2d_ex1.xml"], 1)
        dq = _solve_quadrotor(H, 40, 0.1, 2, refine=0)
        assert H.harness_last_linear_solver().decode().startswith("device")
    finally:
        _set_linear_solver(H, "auto")
    assert abs(d1[0] - h1[0]) < 1e-8 * h1[0] and np.abs(d1[1] - h1[1]).max() < 1e-6
    assert abs(dq[0] - hq[0]) < 1e-8 * hq[0] and np.abs(dq[1] - hq[1]).max() < 1e-5 and np.abs(dq[2] - hq[2]).max() < 1e-4


def test_quadrotor_256_nodes_end_to_end_on_the_device(H):
    """Config-2 mesh (256 LGL nodes, 6 states): evaluation, derivatives AND the Newton step on the GPU
    ("auto" picks the device above 400 KKT rows: here 3584).  Feasibility by the CPU oracle."""
    import time
    t0 = time.time()
    cost, X, U, iters, mesh_iters, _ = _solve_quadrotor_cap(H, 255, 4.0 / 255, 2, cap=300)
    dt = time.time() - t0
    assert H.harness_last_linear_solver().decode().startswith("device")
    m = X.shape[1]
    assert m == 256 and iters < 400
    recs = np.array([[1, 4.0, 3.2, 0.64, 0, 0, 0, 0], [1, 6.3, 4.4, 0.49, 0, 0, 0, 0]], dtype=float)
    RES, _, COST = O.evaluate(1, [1.0, 0.01, 9.81, 1.0, 1.0], m, O.lgl(m), 0.0, 4.0, X[None], U[None], recs)
    # Alg::scaling = "automatic": the tolerance applies to defect rows scaled like their states (bounds 10, 10, 1.2, 6, 6, 4)
    scaled = np.abs(RES[0, :6]) / np.array([10, 10, 1.2, 6, 6, 4.0])[:, None]
    assert scaled.max() < 1e-7 and RES[0, 6:].max() < 1e-7 and abs(COST[0] - cost) < 1e-8
    assert 380 < cost < 450
    print(f"quadrotor M=256: {iters} iterations, {dt:.2f} s wall")


def _solve_quadrotor_cap(H, nsteps, dt, ndiscs, cap):
    D = C.POINTER(C.c_double)
    H.harness_solve_quadrotor.argtypes = [C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, C.c_int, C.c_double, D,
                                          C.POINTER(C.c_int), D, D, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), D]
    X, U = np.zeros(6 * cap), np.zeros(2 * cap)
    cost, M, it, mit, oerr = C.c_double(), C.c_int(), C.c_int(), C.c_int(), C.c_double()
    rc = H.harness_solve_quadrotor(nsteps, dt, ndiscs, 1e-8, 0, 0, 1e-4, C.byref(cost), C.byref(M),
                                   X.ctypes.data_as(D), U.ctypes.data_as(D), cap, C.byref(it), C.byref(mit), C.byref(oerr))
    assert rc == 0, H.harness_last_message().decode()
    m = M.value
    return cost.value, X[:6 * m].reshape(6, m), U[:2 * m].reshape(2, m), it.value, mit.value, oerr.value


def test_config3_sized_problem_end_to_end(H):
    """Config 3 of the scope table as ONE optimisation: 6-state quadrotor, 1024 LGL nodes, 20 disc
    keep-outs (8192 variables, 6144 defect + 20480 path rows).  Mesh sequencing 33 -> ... -> 513 -> 1024,
    every evaluation, derivative and Newton step (KKT matrix of 14336 rows) on the GPU."""
    import time
    t0 = time.time()
    cost, X, U, iters, mesh_iters, _ = _solve_quadrotor_cap(H, 1023, 4.0 / 1023, 20, cap=1100)
    dt = time.time() - t0
    assert H.harness_last_linear_solver().decode().startswith("device")
    m = X.shape[1]
    assert m == 1024 and mesh_iters >= 6 and iters < 200
    # regression guard (round 3 shipped Alg::scaling = "automatic" as the default: 58 iterations on the last mesh and a worse local
    # optimum, 416.19; profiles/r04_regress_probe.json): the 1024-node solve of this problem takes 12 iterations from the 513-node
    # interpolant and ends at cost 400.468
    assert iters <= 30 and abs(cost - 400.468) < 0.5, (iters, cost)
    discs = [(4.0, 3.2, 0.8), (6.3, 4.4, 0.7), (2.5, 1.2, 0.4), (1.6, 3.4, 0.35), (3.1, 5.2, 0.30), (5.2, 1.4, 0.35),
             (7.4, 2.6, 0.30), (8.6, 4.2, 0.25), (5.0, 6.3, 0.35), (2.2, 7.1, 0.30), (6.9, 7.4, 0.35), (8.9, 7.9, 0.30),
             (0.9, 5.6, 0.25), (3.9, 8.4, 0.30), (9.2, 1.3, 0.30), (7.0, 0.8, 0.25), (4.6, 4.9, 0.20), (2.9, 2.9, 0.20),
             (5.6, 3.0, 0.20), (7.6, 5.4, 0.20)]
    recs = np.array([[1, x, y, r * r, 0, 0, 0, 0] for x, y, r in discs], dtype=float)
    RES, _, COST = O.evaluate(1, [1.0, 0.01, 9.81, 1.0, 1.0], m, O.lgl(m), 0.0, 4.0, X[None], U[None], recs)
    assert np.abs(RES[0, :6]).max() < 1e-6 and RES[0, 6:].max() < 1e-6 and abs(COST[0] - cost) < 1e-7
    assert np.allclose(X[:, 0], [1, 1, 0, 0, 0, 0]) and np.all(np.abs(X[:3, -1] - [8, 6, 0]) <= 0.01 + 1e-9)
    assert 380 < cost < 460
    print(f"quadrotor M=1024, 20 keep-outs: {mesh_iters} meshes, last solve {iters} iterations, {dt:.1f} s wall")


def test_device_newton_step_without_path_rows(H):
    """No keep-outs at all (np = 0): the device backend must cope with an empty path table."""
    try:
        _set_linear_solver(H, "device")
        cost, X, U, iters, _, _ = _solve_quadrotor(H, 32, 0.125, 0, refine=0)
    finally:
        _set_linear_solver(H, "auto")
    m = X.shape[1]
    RES, _, COST = O.evaluate(1, [1.0, 0.01, 9.81, 1.0, 1.0], m, O.lgl(m), 0.0, 4.0, X[None], U[None], None)
    assert np.abs(RES[0, :6]).max() < 1e-7 and abs(COST[0] - cost) < 1e-8 and iters < 100


def test_fixedwing_lateral_offset_solves_on_the_gpu(H):
    """12-state fixed-wing model (config 5's dynamics) through ETOL::eMI355X: generated Hessian, coupled nonconvex
    16x16 node blocks, Newton step on the device.  Checked against the oracle's residuals at the returned point."""
    from etol_amd import workloads as W
    D = C.POINTER(C.c_double)
    H.harness_solve_fixedwing.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, D, C.POINTER(C.c_int), D, D,
                                          C.c_int, C.POINTER(C.c_int)]
    _set_linear_solver(H, "device")
    try:
        n = 48
        X, U = np.zeros(12 * (n + 1)), np.zeros(4 * (n + 1))
        cost, M, it = C.c_double(), C.c_int(), C.c_int()
        rc = H.harness_solve_fixedwing(n, 8.0, 10.0, 1e-7, 0, C.byref(cost), C.byref(M), X.ctypes.data_as(D),
                                       U.ctypes.data_as(D), n + 1, C.byref(it))
        assert rc == 0, H.harness_last_message().decode()
    finally:
        _set_linear_solver(H, "auto")
    m = M.value
    assert m == n + 1 and it.value < 150
    X, U = X.reshape(12, m), U.reshape(4, m)
    RES, _, COST = O.evaluate(2, W.FW_PARAMS, m, O.lgl(m), 0.0, 8.0, X[None], U[None])
    # (defect rows scaled like their states, Alg::scaling = "automatic": bounds of the harness problem)
    scaled = np.abs(RES[0, :12]) / np.array([2000, 200, 200, 1.0, 0.6, 1.5, 40, 10, 10, 2, 2, 2.0])[:, None]
    assert scaled.max() < 1e-6 and abs(COST[0] - cost.value) < 1e-8 * abs(cost.value)
    assert abs(X[1, -1] - 10.0) <= 0.5 + 1e-9 and np.abs(X[3]).max() > 0.02
    assert U[0].min() >= -1e-9 and U[0].max() <= 60 + 1e-9 and np.abs(U[1:]).max() <= 0.5 + 1e-9


def test_fixedwing_129_nodes_iteration_count_regression_guard(H):
    """The 129-node fixed-wing lateral offset (tools/solve_times.py): 11 iterations on the last mesh in round 2, 103 in round 3
    (Alg::scaling = "automatic" as the default, together with the primal regularisation ladder: profiles/r04_regress_probe.json), 11
    again with the default back at "none".  Guard on the count and on the optimum."""
    D = C.POINTER(C.c_double)
    H.harness_solve_fixedwing.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, D, C.POINTER(C.c_int), D, D,
                                          C.c_int, C.POINTER(C.c_int)]
    n = 128
    X, U = np.zeros(12 * (n + 1)), np.zeros(4 * (n + 1))
    cost, M, it = C.c_double(), C.c_int(), C.c_int()
    rc = H.harness_solve_fixedwing(n, 12.0, 20.0, 1e-7, 0, C.byref(cost), C.byref(M), X.ctypes.data_as(D), U.ctypes.data_as(D), n + 1,
                                   C.byref(it))
    assert rc == 0, H.harness_last_message().decode()
    assert M.value == n + 1 and it.value <= 30, it.value
    assert abs(cost.value - 3101.9036) < 1e-2, cost.value


def test_montecarlo_example_shards_scenarios_over_ranks_and_threads(built):
    """etol_mi355x_montecarlo: independent scenarios, static block partition over ranks (RANK / WORLD_SIZE /
    LOCAL_RANK), host threads inside a rank.  Two 'ranks' on the one GPU of the test box must solve, between
    them, exactly the scenarios a single rank solves, with the same costs."""
    exe = os.path.join(ROOT, "etol_amd", "lib", "etol_mi355x_montecarlo")

    def run(rank, world, threads, **extra):
        # the two 'ranks' of this test run one after the other on one GPU: no communicator can form, gather off
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", EMI_MC_GATHER="0" if world > 1 else "1", **extra)
        r = subprocess.run([exe, "6", "40", "4", str(threads)], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        lines = r.stdout.strip().split("\n")
        summary = json.loads([l for l in lines if "solves_per_s" in l][-1])
        if world == 1:
            summary["gather"] = json.loads([l for l in lines if "gathered" in l][-1])
        costs = {int(l.split()[1]): float(l.split()[l.split().index("cost") + 1]) for l in lines if l.startswith("scenario")}
        iters = {int(l.split()[1]): int(l.split()[l.split().index("iterations") + 1]) for l in lines if l.startswith("scenario")}
        return summary, costs, iters

    import tempfile
    with tempfile.TemporaryDirectory() as d:
        s_all, c_all, i_all = run(0, 1, 3, EMI_MC_SAVE=d)
        # the RCCL gather (emi_comm_gather through the C ABI; world 1 here) hands rank 0 every solved trajectory
        g = s_all["gather"]
        assert g["gathered"] == 6 and g["gathered_solved"] == 6 and abs(g["cost_sum"] - sum(c_all.values())) < 1e-6
        for sc in range(6):
            rows = open(os.path.join(d, f"scenario{sc}_state.csv")).read().split("\n")
            assert rows[0] == "time,traj0,traj1,traj2,traj3,traj4,traj5" and len(rows) == 42
            assert rows[1].startswith("0.000000,1.000000,1.000000,")            # x(t0) of the scenario set
    assert s_all["solved"] == 6 and sorted(c_all) == list(range(6))
    s0, c0, i0 = run(0, 2, 2)
    s1, c1, i1 = run(1, 2, 1)
    assert sorted(c0) == [0, 1, 2] and sorted(c1) == [3, 4, 5] and s0["solved"] == 3 and s1["solved"] == 3
    # a scenario's iterates do not depend on which rank or thread solved it, nor on what runs beside it: same iteration
    # count and the same cost to the printed digit (the factorisation is reproducible under concurrency)
    for s, c in {**c0, **c1}.items():
        assert c == c_all[s], (s, c, c_all[s])
    assert {**i0, **i1} == i_all
    assert len(set(round(c, 3) for c in c_all.values())) > 1          # the scenarios really differ


def test_montecarlo_with_shared_newton_steps_solves_the_same_scenarios(built):
    """EMI_MC_BATCH: the worker threads share a KktBatcher -- their factorisations and single-right-hand-side solves go out as
    batched launches (emi_kkt_factor_batch / emi_kkt_solve_batch).  The iteration of each scenario is the same algorithm on the same
    matrices; only the factorisation's blocking differs (two-level form at every size, batched rocBLAS kernels), so every scenario
    must be solved and end within 1e-6 relative of the cost its own-launch run reaches (129 nodes: device Newton step on every
    mesh of the ladder above 33 nodes; 10 scenarios on 5 threads in 1 and in 2 groups)."""
    exe = os.path.join(ROOT, "etol_amd", "lib", "etol_mi355x_montecarlo")

    def run(threads, **extra):
        env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", EMI_MC_GATHER="0", **extra)
        r = subprocess.run([exe, "10", "128", "8", str(threads)], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        lines = r.stdout.strip().split("\n")
        summary = json.loads([l for l in lines if "solves_per_s" in l][-1])
        costs = {int(l.split()[1]): float(l.split()[l.split().index("cost") + 1]) for l in lines if l.startswith("scenario")}
        batch = [l for l in lines if l.startswith("batcher")]
        return summary, costs, batch

    s_ref, c_ref, _ = run(5)
    assert s_ref["solved"] == 10 and s_ref["kkt_batch_groups"] == 0
    for groups in ("1", "2"):
        s_b, c_b, lines = run(5, EMI_MC_BATCH=groups)
        assert s_b["solved"] == 10 and s_b["kkt_batch_groups"] == int(groups) and len(lines) == int(groups)
        largest = max(int(l.split("largest batch")[1]) for l in lines)
        assert largest >= 2, lines                      # launches really were shared
        for sc, c in c_ref.items():
            assert abs(c_b[sc] - c) < 1e-6 * abs(c), (groups, sc, c_b[sc], c)


def test_montecarlo_scenario_without_a_feasible_path_on_its_side_ends_early(built):
    """Scenario 27 of the 257-node / 10 keep-out set cannot clear its keep-outs from the side the coarse meshes chose:
    the largest elastic variable stays at 0.15 whatever the penalty weight.  The solve must say so (or, should a later
    change find a way round, solve it) instead of raising the weight to 1e12 over a thousand iterations."""
    exe = os.path.join(ROOT, "etol_amd", "lib", "etol_mi355x_montecarlo")
    env = dict(os.environ, EMI_MC_ONLY="27", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([exe, "32", "256", "10", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode in (0, 2), r.stdout[-2000:] + r.stderr[-2000:]      # 2: not every scenario solved
    line = [l for l in r.stdout.split("\n") if l.startswith("scenario")][0]
    f = line.split()
    rc, iters = int(f[f.index("rc") + 1]), int(f[f.index("iterations") + 1])
    assert rc == 0 or "locally infeasible" in line, line
    assert iters < 700, line


def test_a_failed_rung_restarts_the_ladder_from_a_bent_guess(built):
    """Scenario 27 of the 513-node / 20 keep-out set: its 257-node rung ends locally infeasible.  The ladder must then start
    again on 33 nodes from another guess (the route planned through the free space of the keep-outs, then the straight line bent to one
    side) -- not cold-start the 513-node mesh, which took four attempts of 400 iterations (41 s) before -- and solve the scenario."""
    exe = os.path.join(ROOT, "etol_amd", "lib", "etol_mi355x_montecarlo")
    env = dict(os.environ, EMI_MC_ONLY="27", EMI_MC_PRINT_LEVEL="5", EMI_MC_GATHER="0", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([exe, "32", "512", "20", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.split("\n") if l.startswith("scenario")][0]
    f = line.split()
    rc, iters, nodes = int(f[f.index("rc") + 1]), int(f[f.index("iterations") + 1]), int(f[f.index("nodes") + 1])
    assert rc == 0 and nodes == 513, line
    rung_failed = any(l.startswith("mesh sequencing:") and "converged" not in l and "nodes," in l for l in r.stdout.split("\n"))
    # (should the rung converge one day: fine; the restart is from the route planned through the free space first, then from the bent lines)
    assert not rung_failed or "ladder restarted from the route planned" in r.stdout or "ladder restarted from the line bent by" in r.stdout
    assert "cold start on 513 nodes" not in r.stdout
    assert iters < 1200, line


def test_shipped_example_with_traced_obstacle_rows(H, xmls):
    """The obstacle rows of src/Examples/PSOPT/etol_psopt_example1.cpp:153-190 computed with mi355x::Var
    arithmetic in the callback (nine ellipse rows), traced and compiled into the kernels, next to the
    moving-disc rows from the record table: a feasible solution of the same problem."""
    base = solve(H, xmls["ocp_2d_ex1.xml"], 1)
    H.harness_set_traced.argtypes = [C.c_int]
    H.harness_set_traced(2)
    try:
        tr = solve(H, xmls["ocp_2d_ex1.xml"], 1)
        H.harness_set_traced(3)           # the moving-disc rows traced too (waypoint tables through mx::interp1)
        tr3 = solve(H, xmls["ocp_2d_ex1.xml"], 1)
    finally:
        H.harness_set_traced(0)
    print(f"all rows traced: cost {tr3[0]:.8f} on {tr3[1].shape[1]} nodes")
    M3 = tr3[1].shape[1]
    mesh3 = O.lgl(M3)
    recs3, tx3, ty3 = cases.ocp2d_tables(O.edge_ellipse, O.track_centres, 8.0 * (mesh3[0] + 1))
    RES3, _, COST3 = O.evaluate(0, [], M3, mesh3, 0.0, 16.0, tr3[1][None], tr3[2][None], recs3, (tx3, ty3))
    assert np.abs(RES3[0, :2]).max() < 1e-7 and RES3[0, 2:].max() < 1e-7 and abs(COST3[0] - tr3[0]) < 1e-9
    # the traced rows are normalised differently inside the iteration, so the two runs may pass through different
    # iterates and stop the mesh refinement at different node counts: compare what is mesh-independent
    print(f"table rows: cost {base[0]:.8f} on {base[1].shape[1]} nodes; traced rows: cost {tr[0]:.8f} on {tr[1].shape[1]} nodes")
    g = json.load(open(os.path.join(GOLD, "ocp2d.json")))
    assert g["cost"] < tr[0] < 1.05 * base[0]          # above the obstacle-free optimum, not worse than the other local solution
    M = tr[1].shape[1]
    mesh = O.lgl(M)
    recs, tx, ty = cases.ocp2d_tables(O.edge_ellipse, O.track_centres, 8.0 * (mesh[0] + 1))
    RES, _, COST = O.evaluate(0, [], M, mesh, 0.0, 16.0, tr[1][None], tr[2][None], recs, (tx, ty))
    assert np.abs(RES[0, :2]).max() < 1e-7 and RES[0, 2:].max() < 1e-7 and abs(COST[0] - tr[0]) < 1e-9
