"""Newton-step linear algebra on the device (emi_kkt_factor / emi_kkt_solve) against numpy.  -m gpu

The reference reaches this step inside IPOPT (src/ePSOPT/ePSOPT.cpp:62, 84)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def dense_kkt(D, Qblk, Jblk, fixed, dc, M, ns, nv):
    nz, N = nv * M, (nv + ns) * M
    K = np.zeros((N, N))
    for v in range(nv):
        for q in range(v + 1):
            blk = Qblk[v * (v + 1) // 2 + q]
            for k in range(M):
                K[v * M + k, q * M + k] = blk[k]
                K[q * M + k, v * M + k] = blk[k]
    for i in range(ns):
        rows = nz + i * M + np.arange(M)
        K[np.ix_(rows, i * M + np.arange(M))] = D
        for v in range(nv):
            K[rows, v * M + np.arange(M)] = Jblk[i * nv + v]
        K[rows, rows] = -dc
    K[nz:, :nz][:, :] = K[nz:, :nz]
    K[:nz, nz:] = K[nz:, :nz].T
    for q in np.nonzero(fixed)[0]:
        K[q, :] = 0
        K[:, q] = 0
        K[q, q] = 1
    return K


@pytest.mark.parametrize("method", [1, 0])       # 1: Schur complement + Cholesky, 0: LU of the full matrix
@pytest.mark.parametrize("M,model", [(9, 0), (33, 1), (64, 1), (200, 1), (256, 1)])
def test_kkt_factor_solve_matches_numpy(built, M, model, method):
    import etol_amd as E
    from etol_amd import workloads as W
    ns, nc, _ = E.model_dims(model)
    nv, nh = ns + nc, (ns + nc) * (ns + nc + 1) // 2
    ev = E.Evaluator(0)
    ev.set_mesh(M, 0.0, 4.0)
    ev.set_model(model, W.QUAD_PARAMS if model == 1 else [])
    ev.set_batch(1)
    ev.set_option("kkt_method", method)
    rng = np.random.default_rng(M)
    # SPD node blocks (as after inertia correction), arbitrary Jacobian node entries
    Qblk = np.zeros((nh, M))
    for k in range(M):
        A = rng.standard_normal((nv, nv))
        Qk = A @ A.T + nv * np.eye(nv)
        for v in range(nv):
            for q in range(v + 1):
                Qblk[v * (v + 1) // 2 + q, k] = Qk[v, q]
    Jblk = rng.standard_normal((ns * nv, M))
    for i in range(ns):
        Jblk[i * nv + i] += np.diag(ev.D)
    fixed = np.zeros(nv * M, dtype=np.uint8)
    fixed[np.arange(ns) * M] = 1            # initial state, as eMI355X::addBounds fixes it
    dc = 1e-9
    assert ev.kkt_factor(Qblk, Jblk, fixed, dc) == 0
    K = dense_kkt(ev.D, Qblk, Jblk, fixed, dc, M, ns, nv)
    assert np.allclose(K, K.T)
    many = rng.standard_normal((5, (nv + ns) * M))          # several right-hand sides in one call
    sols = ev.kkt_solve(many)
    for b, x in zip(many, sols):
        b = b.copy()
        b[np.nonzero(fixed)[0]] = 0
        assert np.abs(K @ x - b).max() < 1e-10 * (np.abs(K).max() * np.abs(x).max() + 1)
    for _ in range(2):                       # one factorisation, several calls
        rhs = rng.standard_normal((nv + ns) * M)
        sol = ev.kkt_solve(rhs)
        ref_rhs = rhs.copy()
        ref_rhs[np.nonzero(fixed)[0]] = 0
        ref = np.linalg.solve(K, ref_rhs)
        assert np.all(sol[np.nonzero(fixed)[0]] == 0)
        # backward error of the device solution in the numpy matrix
        res = K @ sol - ref_rhs
        assert np.abs(res).max() < 1e-10 * (np.abs(K).max() * np.abs(sol).max() + 1)
        assert np.abs(sol - ref).max() < 1e-7 * (np.abs(ref).max() + 1)


@pytest.mark.parametrize("M", [171, 200, 256, 300])
def test_block_inverse_triangular_solves_match_rocblas_trsv(built, M):
    """Single right-hand sides go through the library's own block-inverse triangular solves once the Schur complement has
    1024 rows or more (emi_trsv_fwd / bwd kernels: 256-row blocks, diagonal blocks inverted once per factorisation, a partial
    last block at 171 / 200 / 300 nodes); "kkt_block_trsv" 0 takes rocsolver_dpotrs (rocBLAS trsv).  Same solutions to
    rounding, both with a small backward error in the numpy matrix."""
    import etol_amd as E
    from etol_amd import workloads as W
    ns, nv, nh = 6, 8, 36
    ev = E.Evaluator(0)
    ev.set_mesh(M, 0.0, 4.0)
    ev.set_model(1, W.QUAD_PARAMS)
    ev.set_batch(1)
    rng = np.random.default_rng(1000 + M)
    Qblk = np.zeros((nh, M))
    for k in range(M):
        A = rng.standard_normal((nv, nv))
        Qk = A @ A.T + nv * np.eye(nv)
        for v in range(nv):
            for q in range(v + 1):
                Qblk[v * (v + 1) // 2 + q, k] = Qk[v, q]
    Jblk = rng.standard_normal((ns * nv, M))
    for i in range(ns):
        Jblk[i * nv + i] += np.diag(ev.D)
    fixed = np.zeros(nv * M, dtype=np.uint8)
    fixed[np.arange(ns) * M] = 1
    K = dense_kkt(ev.D, Qblk, Jblk, fixed, 1e-9, M, ns, nv)
    rhs = rng.standard_normal((3, (nv + ns) * M))
    sols = {}
    for mode in (1, 2, 0):                                                  # gemv form (default), own diagonal-block kernel, rocBLAS trsv
        ev.set_option("kkt_block_trsv", mode)
        assert ev.kkt_factor(Qblk, Jblk, fixed, 1e-9) == 0
        sols[mode] = np.array([ev.kkt_solve(b) for b in rhs])           # one right-hand side per call
        for b, x in zip(rhs, sols[mode]):
            b = b.copy()
            b[np.nonzero(fixed)[0]] = 0
            assert np.abs(K @ x - b).max() < 1e-10 * (np.abs(K).max() * np.abs(x).max() + 1), mode
        again = np.array([ev.kkt_solve(b) for b in rhs])
        assert np.array_equal(again, sols[mode])                        # fixed summation order: bitwise reproducible
    ev.set_option("kkt_block_trsv", 1)
    assert np.abs(sols[1] - sols[0]).max() < 1e-9 * (np.abs(sols[0]).max() + 1)
    assert np.abs(sols[2] - sols[0]).max() < 1e-9 * (np.abs(sols[0]).max() + 1)


@pytest.mark.parametrize("M", [171, 256, 300])
def test_two_level_cholesky_matches_the_one_level_form(built, M):
    """"kkt_cholesky" 2 (default from 1024 rows of the Schur complement: 171 nodes and up for 6 states): 64-column steps that update
    only the rest of their outer panel, one rank-768 dsyrk per outer panel for everything behind it -- against the one-level form
    (1) and the numpy matrix, with outer panels of 128 / 768 / 2048 columns (partial last panels; one panel = the whole matrix)."""
    import etol_amd as E
    from etol_amd import workloads as W
    ns, nv, nh = 6, 8, 36
    ev = E.Evaluator(0)
    ev.set_mesh(M, 0.0, 4.0)
    ev.set_model(1, W.QUAD_PARAMS)
    ev.set_batch(1)
    rng = np.random.default_rng(2000 + M)
    Qblk = np.zeros((nh, M))
    for k in range(M):
        A = rng.standard_normal((nv, nv))
        Qk = A @ A.T + nv * np.eye(nv)
        for v in range(nv):
            for q in range(v + 1):
                Qblk[v * (v + 1) // 2 + q, k] = Qk[v, q]
    Jblk = rng.standard_normal((ns * nv, M))
    for i in range(ns):
        Jblk[i * nv + i] += np.diag(ev.D)
    fixed = np.zeros(nv * M, dtype=np.uint8)
    fixed[np.arange(ns) * M] = 1
    K = dense_kkt(ev.D, Qblk, Jblk, fixed, 1e-9, M, ns, nv)
    rhs = rng.standard_normal((nv + ns) * M)
    b = rhs.copy()
    b[np.nonzero(fixed)[0]] = 0
    sols = {}
    try:
        # (last entry of a tuple: "kkt_chol_panel" 2 = the panel solve on the matrix pipe (default), 1 its scalar form, 0 rocBLAS dtrsm)
        # ("kkt_chol_diag" 2 = the 64 x 64 diagonal block by one wave with matrix-pipe updates (default), 1 = column by column by 256 threads)
        for mode, outer, panel, diag in ((1, 768, 1, 1), (2, 128, 2, 2), (2, 768, 2, 2), (2, 2048, 2, 2), (1, 768, 2, 2), (2, 768, 1, 2), (2, 768, 0, 2),
                                         (2, 768, 2, 1), (1, 768, 1, 2)):
            ev.set_option("kkt_cholesky", mode)
            ev.set_option("kkt_chol_outer", outer)
            ev.set_option("kkt_chol_panel", panel)
            ev.set_option("kkt_chol_diag", diag)
            assert ev.kkt_factor(Qblk, Jblk, fixed, 1e-9) == 0
            x = ev.kkt_solve(rhs)
            assert np.abs(K @ x - b).max() < 1e-10 * (np.abs(K).max() * np.abs(x).max() + 1), (mode, outer, panel, diag)
            sols[mode, outer, panel, diag] = x
            # the low-rank path factorises a second, small matrix with the same routine and solves many right-hand sides (dpotrs)
            X3 = ev.kkt_solve(np.stack([rhs, 2 * rhs, -rhs]))
            assert np.abs(X3[0] - x).max() < 1e-9 * (np.abs(x).max() + 1)
    finally:
        ev.set_option("kkt_cholesky", 2)
        ev.set_option("kkt_chol_outer", 768)
        ev.set_option("kkt_chol_panel", 2)
        ev.set_option("kkt_chol_diag", 2)
    for k, x in sols.items():
        assert np.abs(x - sols[1, 768, 1, 1]).max() < 1e-9 * (np.abs(x).max() + 1), k
    ev.close()


def test_schur_method_falls_back_to_lu_for_indefinite_blocks(built):
    """Method 1 needs positive definite node blocks; anything else must silently take the general LU."""
    import etol_amd as E
    from etol_amd import workloads as W
    M, ns, nv, nh = 17, 6, 8, 36
    ev = E.Evaluator(0)
    ev.set_mesh(M, 0.0, 2.0)
    ev.set_model(1, W.QUAD_PARAMS)
    ev.set_batch(1)
    rng = np.random.default_rng(3)
    Qblk = np.zeros((nh, M))
    for k in range(M):
        A = rng.standard_normal((nv, nv))
        Qk = A + A.T                                  # indefinite
        for v in range(nv):
            for q in range(v + 1):
                Qblk[v * (v + 1) // 2 + q, k] = Qk[v, q]
    Jblk = rng.standard_normal((ns * nv, M))
    for i in range(ns):
        Jblk[i * nv + i] += np.diag(ev.D)
    fixed = np.zeros(nv * M, dtype=np.uint8)
    fixed[np.arange(ns) * M] = 1
    assert ev.kkt_factor(Qblk, Jblk, fixed, 1e-9) == 0
    K = dense_kkt(ev.D, Qblk, Jblk, fixed, 1e-9, M, ns, nv)
    rhs = rng.standard_normal((nv + ns) * M)
    sol = ev.kkt_solve(rhs)
    rhs[np.nonzero(fixed)[0]] = 0
    assert np.abs(K @ sol - rhs).max() < 1e-9 * (np.abs(K).max() * np.abs(sol).max() + 1)


def test_kkt_reports_singular_matrix_and_state_errors(built):
    import etol_amd as E
    from etol_amd import _lib
    ev = E.Evaluator(0)
    ev.set_mesh(5, 0.0, 1.0)
    ev.set_model(0, [])
    ev.set_batch(1)
    with pytest.raises(_lib.EmiError):
        ev.kkt_solve(np.zeros(6 * 5))                         # nothing factorised yet
    # Q = 0, J = 0 and every variable free: structurally singular
    info = ev.kkt_factor(np.zeros((10, 5)), np.zeros((8, 5)), np.zeros(20, dtype=np.uint8), 0.0)
    assert info > 0
    with pytest.raises(_lib.EmiError):
        ev.kkt_solve(np.zeros(6 * 5))


@pytest.mark.parametrize("method", [1, 0])
def test_lowrank_correction_gives_exact_solves_and_the_inertia_verdict(built, method):
    """K = K~ - U Delta U^T: after emi_kkt_lowrank the solves answer for K, and `exact` says whether K has
    the inertia of K~ (checked against numpy eigenvalues)."""
    import etol_amd as E
    from etol_amd import workloads as W
    M, ns, nv, nh = 24, 6, 8, 36
    ev = E.Evaluator(0)
    ev.set_mesh(M, 0.0, 3.0)
    ev.set_model(1, W.QUAD_PARAMS)
    ev.set_batch(1)
    ev.set_option("kkt_method", method)
    rng = np.random.default_rng(7)
    fixed = np.zeros(nv * M, dtype=np.uint8)
    fixed[np.arange(ns) * M] = 1
    Jblk = rng.standard_normal((ns * nv, M))
    for i in range(ns):
        Jblk[i * nv + i] += np.diag(ev.D)
    for scale, expect in ((0.05, True), (50.0, False)):
        # Q~ positive definite blocks; the "true" Q differs by reflected eigen-directions at some nodes
        Qt = np.zeros((nh, M))
        node, vec, delta = [], [], []
        for k in range(M):
            A = rng.standard_normal((nv, nv))
            Qk = A @ A.T + nv * np.eye(nv)
            for v in range(nv):
                for q in range(v + 1):
                    Qt[v * (v + 1) // 2 + q, k] = Qk[v, q]
            if k % 3 == 1:
                u = rng.standard_normal(nv)
                u[fixed[np.arange(nv) * M + k] != 0] = 0
                node.append(k); vec.append(u); delta.append(scale * (1 + rng.random()))
        assert ev.kkt_factor(Qt, Jblk, fixed, 1e-9) == 0
        Kt = dense_kkt(ev.D, Qt, Jblk, fixed, 1e-9, M, ns, nv)
        K = Kt.copy()
        for k, u, d in zip(node, vec, delta):
            idx = np.arange(nv) * M + k
            K[np.ix_(idx, idx)] -= d * np.outer(u, u)
        eig = np.linalg.eigvalsh(K)
        true_ok = (eig > 0).sum() == nv * M and (eig < 0).sum() == ns * M
        assert true_ok == expect                      # the two scales were chosen to land on either side
        exact = ev.kkt_lowrank(node, np.array(vec), delta)
        assert exact == true_ok
        rhs = rng.standard_normal((nv + ns) * M)
        sol = ev.kkt_solve(rhs)
        b = rhs.copy()
        b[np.nonzero(fixed)[0]] = 0
        Kref = K if exact else Kt
        assert np.abs(Kref @ sol - b).max() < 1e-9 * (np.abs(Kref).max() * np.abs(sol).max() + 1)
        assert ev.kkt_lowrank([], np.zeros((0, nv)), []) is True     # cleared: back to K~
        sol2 = ev.kkt_solve(rhs)
        assert np.abs(Kt @ sol2 - b).max() < 1e-9 * (np.abs(Kt).max() * np.abs(sol2).max() + 1)


def test_concurrent_contexts_give_reproducible_factorisations(built):
    """Monte-Carlo runs factorise from several host threads at once, one context (stream, rocBLAS handle) each.
    rocsolver_dpotrf under such load now and then reports a non-positive pivot for a matrix that is positive
    definite (tools/race_probe.py found ~1 % of the calls); emi_kkt_factor confirms a failure on a kept
    copy before it raises the regularisation.  Here: 6 threads, the same data, every repeat of
    factor / low-rank verdict / solve must be bit-identical to the thread's first."""
    import threading

    import etol_amd as E
    from etol_amd import workloads as W
    M, reps, nthreads = 65, 120, 6
    ns, nc, _ = E.model_dims(1)
    nv, nh = ns + nc, (ns + nc) * (ns + nc + 1) // 2
    rng = np.random.default_rng(7)
    Qblk = np.zeros((nh, M))
    for k in range(M):
        A = rng.standard_normal((nv, nv))
        Qk = A @ A.T + nv * np.eye(nv)
        for v in range(nv):
            for q in range(v + 1):
                Qblk[v * (v + 1) // 2 + q, k] = Qk[v, q]
    Jnode = rng.standard_normal((ns * nv, M))
    fixed = np.zeros(nv * M, dtype=np.uint8)
    fixed[np.arange(ns) * M] = 1
    r = 12
    node = rng.integers(0, M, r).astype(np.int32)
    vec = rng.standard_normal((r, nv))
    delta = np.abs(rng.standard_normal(r)) * 0.1 + 0.01
    rhs = rng.standard_normal((3, (nv + ns) * M))
    bad, errors, firsts = [], [], [None] * nthreads

    def work(tid):
        try:
            ev = E.Evaluator(0)
            ev.set_mesh(M, 0.0, 4.0)
            ev.set_model(1, W.QUAD_PARAMS)
            ev.set_batch(1)
            Jblk = Jnode.copy()
            for i in range(ns):
                Jblk[i * nv + i] += np.diag(ev.D)
            for rep in range(reps):
                assert ev.kkt_factor(Qblk, Jblk, fixed, 1e-9) == 0
                exact = ev.kkt_lowrank(node, vec, delta)
                sol = ev.kkt_solve(rhs)
                if firsts[tid] is None:
                    firsts[tid] = (exact, sol.copy())
                elif exact != firsts[tid][0] or not np.array_equal(sol, firsts[tid][1]):
                    bad.append((tid, rep, float(np.abs(sol - firsts[tid][1]).max())))
            ev.close()
        except Exception as e:      # noqa: BLE001 -- reported by the main thread
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert not bad, bad[:5]
    for tid in range(1, nthreads):           # and the threads agree with each other
        assert firsts[tid][0] == firsts[0][0] and np.array_equal(firsts[tid][1], firsts[0][1])


def _random_kkt(ev, M, ns, nv, rng, weak_nodes=0):
    nh = nv * (nv + 1) // 2
    Qblk = np.zeros((nh, M))
    for k in range(M):
        A = rng.standard_normal((nv, nv))
        Qk = A @ A.T + nv * np.eye(nv)
        for v in range(nv):
            for q in range(v + 1):
                Qblk[v * (v + 1) // 2 + q, k] = Qk[v, q]
    Jblk = rng.standard_normal((ns * nv, M))
    for i in range(ns):
        Jblk[i * nv + i] += np.diag(ev.D)
    fixed = np.zeros(nv * M, dtype=np.uint8)
    fixed[np.arange(ns) * M] = 1
    return Qblk, Jblk, fixed


@pytest.mark.parametrize("M,model,n", [(33, 1, 5), (65, 1, 3), (171, 1, 4), (256, 1, 6), (9, 0, 2), (1024, 1, 3)])
def test_batched_factor_and_solve_match_numpy_and_the_single_path(built, M, model, n):
    """emi_kkt_factor_batch / emi_kkt_solve_batch: n scenarios (n contexts) on one mesh, every launch of the factorisation carrying
    the whole batch (pointer-table kernels + rocBLAS *_batched).  Each scenario's solution must solve ITS matrix (numpy, backward
    error) and agree with what the single entry points return for the same matrix; low-rank corrections of individual scenarios
    survive the batched solve; a scenario with an indefinite node block leaves the batch through the single path (LU) inside the
    call.  Meshes: one-level sizes with a partial last Cholesky block (33, 65, 171 nodes), block-inverse solves with a partial last
    block (171, 256), the 1024-node shape of config 4 (96 full steps, 8 outer panels, 12 diagonal blocks)."""
    import ctypes as C
    import etol_amd as E
    from etol_amd import _lib as L
    from etol_amd import workloads as W
    lib = L.load()
    ns, nc, _ = E.model_dims(model)
    nv, N = ns + nc, (2 * ns + nc) * M
    rng = np.random.default_rng(7000 + M + n)
    evs = []
    for b in range(n):
        ev = E.Evaluator(0)
        ev.set_mesh(M, 0.0, 4.0)
        ev.set_model(model, W.QUAD_PARAMS if model == 1 else [])
        ev.set_batch(1)
        evs.append(ev)
    probs = [_random_kkt(evs[0], M, ns, nv, rng) for _ in range(n)]
    bad = n - 1 if n >= 4 else -1                  # one scenario whose node block 3 is indefinite: not the quasi-definite case
    if bad >= 0:
        Q = probs[bad][0]
        Q[0, min(3, M - 1)] = -5.0
    dc = np.full(n, 1e-9)
    D = C.POINTER(C.c_double)
    dp = lambda a: a.ctypes.data_as(D)
    ctxs = (C.c_void_p * n)(*[ev.ctx for ev in evs])
    Qp = (D * n)(*[dp(p[0]) for p in probs])
    Jp = (D * n)(*[dp(p[1]) for p in probs])
    Fp = (C.POINTER(C.c_ubyte) * n)(*[p[2].ctypes.data_as(C.POINTER(C.c_ubyte)) for p in probs])
    info = np.full(n, -7, dtype=np.int32)
    st = lib.emi_kkt_factor_batch(n, ctxs, Qp, Jp, Fp, dp(dc), info.ctypes.data_as(C.POINTER(C.c_int)))
    assert st == 0, lib.emi_last_error(evs[0].ctx)
    assert np.all(info == 0), info
    Ks = [dense_kkt(evs[0].D, p[0], p[1], p[2], 1e-9, M, ns, nv) for p in probs] if M <= 256 else None
    # a low-rank correction on scenario 1 (as the interior-point iteration adds after reflecting a node block): K = K~ - d u u^T
    lr = None
    if M <= 256 and n >= 3:
        node, vec, delta = np.array([M // 2], dtype=np.int32), rng.standard_normal((1, nv)) * 0.3, np.array([0.7])
        lr = (node, vec, delta, evs[1].kkt_lowrank(node, vec, delta))
    for rep in range(2):                          # one factorisation, several batched solves
        rhs = [rng.standard_normal(N) for _ in range(n)]
        work = [r.copy() for r in rhs]
        Rp = (D * n)(*[dp(w) for w in work])
        assert lib.emi_kkt_solve_batch(n, ctxs, Rp) == 0, lib.emi_last_error(evs[0].ctx)
        for b in range(n):
            single = evs[b].kkt_solve(rhs[b])     # the same factors through the single entry point
            scale = np.abs(single).max() + 1
            assert np.abs(work[b] - single).max() < 1e-9 * scale, (b, np.abs(work[b] - single).max())
            if Ks is not None:
                K = Ks[b].copy()
                if lr is not None and b == 1 and lr[3]:
                    u = np.zeros(N)
                    u[np.arange(nv) * M + lr[0][0]] = lr[1][0]
                    u[np.nonzero(probs[b][2])[0]] = 0
                    K -= lr[2][0] * np.outer(u, u)
                ref = rhs[b].copy()
                ref[np.nonzero(probs[b][2])[0]] = 0
                assert np.abs(K @ work[b] - ref).max() < 1e-9 * (np.abs(K).max() * np.abs(work[b]).max() + 1), b
    # the batch against factorisations made one by one (fresh contexts): same solutions to rounding
    for b in range(min(n, 2)):
        ev = E.Evaluator(0)
        ev.set_mesh(M, 0.0, 4.0)
        ev.set_model(model, W.QUAD_PARAMS if model == 1 else [])
        ev.set_batch(1)
        if b == 1 and lr is not None:
            continue
        assert ev.kkt_factor(probs[b][0], probs[b][1], probs[b][2], 1e-9) == 0
        x1 = ev.kkt_solve(rhs[b])
        assert np.abs(x1 - work[b]).max() < 1e-8 * (np.abs(x1).max() + 1)
        ev.close()
    for ev in evs:
        ev.close()


@pytest.mark.parametrize("M,n", [(65, 1), (200, 3), (256, 4)])
def test_refined_solves_on_the_device_reach_round_off_in_the_nominal_matrix(built, M, n):
    """emi_kkt_solve_refined(_batch): the Newton step with its iterative refinement done on the device (residual r = b - K x with the
    nominal node blocks, the D operator as two batched GEMMs, the low-rank term of the scenarios whose correction is active; only
    the residual norms cross to the host).  Against numpy: the returned x solves the NOMINAL matrix to round-off although the
    factorisation holds dc = 1e-9 more than the nominal 0 in one scenario; rel reports the residual honestly."""
    import ctypes as C
    import etol_amd as E
    from etol_amd import _lib as L
    from etol_amd import workloads as W
    lib = L.load()
    ns, nv = 6, 8
    N = (2 * ns + 2) * M
    rng = np.random.default_rng(9100 + M)
    evs, probs = [], []
    for b in range(n):
        ev = E.Evaluator(0)
        ev.set_mesh(M, 0.0, 4.0)
        ev.set_model(1, W.QUAD_PARAMS)
        ev.set_batch(1)
        evs.append(ev)
        probs.append(_random_kkt(ev, M, ns, nv, rng))
    dcs = np.array([1e-9, 0.0, 1e-8, 1e-9][:n])
    D = C.POINTER(C.c_double)
    dp = lambda a: a.ctypes.data_as(D)
    ctxs = (C.c_void_p * n)(*[ev.ctx for ev in evs])
    for b in range(n):
        assert evs[b].kkt_factor(*probs[b], dc=max(dcs[b], 0.0)) == 0
    evs[0].set_option("kkt_refine_exp", 14)      # (process-wide) refine to round-off here; the default stops at 1e-10 of the right-hand side
    lr = None
    if n >= 3:          # scenario 2: an active low-rank correction (K = K~ - d u u^T)
        node, vec, delta = np.array([M // 3], dtype=np.int32), rng.standard_normal((1, nv)) * 0.3, np.array([0.5])
        assert evs[2].kkt_lowrank(node, vec, delta)
        lr = (node, vec, delta)
    rhs = [rng.standard_normal(N) for _ in range(n)]
    work = [r.copy() for r in rhs]
    Rp = (D * n)(*[dp(w) for w in work])
    rel = np.zeros(n)
    nsv, rev, stat = (np.zeros(n, dtype=np.int32) for _ in range(3))
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    st = lib.emi_kkt_solve_refined_batch(n, ctxs, Rp, dp(dcs), 8, dp(rel), ip(nsv), ip(rev), ip(stat))
    assert st == 0, lib.emi_last_error(evs[0].ctx)
    assert np.all(stat == 0) and np.all(nsv >= 1)
    for b in range(n):
        K = dense_kkt(evs[b].D, probs[b][0], probs[b][1], probs[b][2], dcs[b], M, ns, nv)      # the NOMINAL matrix
        if lr is not None and b == 2:
            u = np.zeros(N)
            u[np.arange(nv) * M + lr[0][0]] = lr[1][0]
            u[np.nonzero(probs[b][2])[0]] = 0
            K -= lr[2][0] * np.outer(u, u)
        ref = rhs[b].copy()
        ref[np.nonzero(probs[b][2])[0]] = 0
        res = np.abs(K @ work[b] - ref).max() / max(1.0, np.abs(ref).max())
        assert res < 1e-11, (b, res, rel[b], nsv[b])
        assert abs(res - rel[b]) < 1e-11            # what the device reports is what numpy sees
        x = np.linalg.solve(K, ref)
        assert np.abs(x - work[b]).max() < 1e-8 * (np.abs(x).max() + 1)
    # the single entry point
    w1 = rhs[0].copy()
    r1, n1, v1, s1 = C.c_double(), C.c_int(), C.c_int(), C.c_int()
    assert lib.emi_kkt_solve_refined(evs[0].ctx, dp(w1), float(dcs[0]), 8, C.byref(r1), C.byref(n1), C.byref(v1), C.byref(s1)) == 0
    assert np.abs(w1 - work[0]).max() < 1e-12 * (np.abs(w1).max() + 1)
    # the default depth (IPOPT's residual_ratio_max): no more solves than before, and the residual it reports is below 1e-10
    evs[0].set_option("kkt_refine_exp", 10)
    w2 = rhs[0].copy()
    r2, n2 = C.c_double(), C.c_int()
    assert lib.emi_kkt_solve_refined(evs[0].ctx, dp(w2), float(dcs[0]), 8, C.byref(r2), C.byref(n2), C.byref(v1), C.byref(s1)) == 0
    assert n2.value <= n1.value and r2.value < 1e-10, (n2.value, n1.value, r2.value)
    # an LU factorisation is not offered: EMI_ERR_UNSUPPORTED (5)
    evs[0].set_option("kkt_method", 0)
    assert evs[0].kkt_factor(*probs[0], dc=1e-9) == 0
    assert lib.emi_kkt_is_schur(evs[0].ctx) == 0
    assert lib.emi_kkt_solve_refined(evs[0].ctx, dp(w1), 1e-9, 8, C.byref(r1), C.byref(n1), C.byref(v1), C.byref(s1)) == 5
    for ev in evs:
        ev.close()


def test_primal_regularisation_levels_are_reported_and_refined_out(built):
    """The Schur path's ladder beyond its dual levels (csrc/emi_kkt.hip: dc x 1e3 + dw 1e-7 | 1e-5 | dc x 1e6 + dw 1e-3 on the free STATE
    diagonals): a matrix whose state variables carry next to no curvature (P = Q^-1 ~ 1 / curv, S = J P J^T beyond what a Cholesky
    in double precision resolves on a 512-node mesh) must (a) still be factorised, (b) REPORT what was really factorised
    (emi_kkt_last_regularisation: the iteration treats the shift like its own delta_w), and (c) give, through the refinement against
    the NOMINAL matrix, a step whose reported residual is the residual numpy sees -- small when the refinement contracts, honestly
    large when it does not (the advisor's round-3 finding: the shifted factorisation used to be handed back as if it were exact)."""
    import ctypes as C
    import etol_amd as E
    from etol_amd import _lib as L
    from etol_amd import workloads as W
    lib = L.load()
    M, ns, nv = 512, 6, 8
    nh, N = nv * (nv + 1) // 2, (2 * ns + 2) * M
    ev = E.Evaluator(0)
    ev.set_mesh(M, 0.0, 4.0)
    ev.set_model(1, W.QUAD_PARAMS)
    ev.set_batch(1)
    rng = np.random.default_rng(77)
    D = C.POINTER(C.c_double)
    dp = lambda a: a.ctypes.data_as(D)
    Jblk = 0.01 * rng.standard_normal((ns * nv, M))
    for i in range(ns):
        Jblk[i * nv + i] += np.diag(ev.D)
    fixed = np.zeros(nv * M, dtype=np.uint8)
    fixed[np.arange(ns) * M] = 1
    reached = None
    for curv in (1e-8, 1e-10, 1e-12, 1e-14):
        Qblk = np.zeros((nh, M))
        for v in range(nv):
            Qblk[v * (v + 1) // 2 + v] = curv if v < ns else 1.0          # diagonal node blocks: states ~ flat, controls curved
        assert ev.kkt_factor(Qblk, Jblk, fixed, 0.0) == 0
        dc, dw = C.c_double(), C.c_double()
        assert lib.emi_kkt_last_regularisation(ev.ctx, C.byref(dc), C.byref(dw)) == 0
        assert dc.value >= 1e-9 and dw.value >= 0.0
        print(f"state curvature {curv:.0e}: factorised with dc {dc.value:.1e}, dw {dw.value:.1e}, Schur path {lib.emi_kkt_is_schur(ev.ctx)}")
        if dw.value > 0.0 and lib.emi_kkt_is_schur(ev.ctx):
            reached = (curv, Qblk, dc.value, dw.value)
            break
    if reached is None:
        pytest.skip("no primal level of the ladder was needed on this mesh")
    curv, Qblk, dca, dwa = reached
    K = dense_kkt(ev.D, Qblk, Jblk, fixed, 0.0, M, ns, nv)                 # the NOMINAL matrix (dc = 0, no shift)
    b = rng.standard_normal(N)
    ref = b.copy()
    ref[np.nonzero(fixed)[0]] = 0
    x = b.copy()
    rel, nsv, rev, stat = C.c_double(), C.c_int(), C.c_int(), C.c_int()
    assert lib.emi_kkt_solve_refined(ev.ctx, dp(x), 0.0, 8, C.byref(rel), C.byref(nsv), C.byref(rev), C.byref(stat)) == 0
    assert stat.value == 0
    res = np.abs(K @ x - ref).max() / max(1.0, np.abs(ref).max())
    print(f"  shift dw {dwa:.1e}: {nsv.value} solves, reverted {rev.value}, residual reported {rel.value:.2e}, numpy {res:.2e}")
    assert abs(res - rel.value) <= 1e-6 * max(res, rel.value) + 1e-13         # the report is what numpy sees
    plain = ev.kkt_solve(b)                                                    # one solve with the shifted factors, no refinement
    res0 = np.abs(K @ plain - ref).max() / max(1.0, np.abs(ref).max())
    assert res <= res0 * (1 + 1e-9)                                             # refinement never hands back something worse
    ev.close()
