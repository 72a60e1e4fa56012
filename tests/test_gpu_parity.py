"""Parity of the HIP path (through the C ABI) against the CPU oracle.  -m gpu"""
import os

import numpy as np
import pytest

import cases
import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# fp64 tolerances, relative to the row scale (written here on purpose):
#   node functions / Jacobian values: a handful of roundings          -> 1e-13
#   defect rows: an M-term dot product with |D| up to N(N+1)/4; the GPU sums in
#   a different order than the oracle's long-double sum               -> 5e-13 of sum|D_kj||x_j|
TOL_NODE = 1e-13
TOL_DEFECT = 5e-13


def run_case(name, maximize=False):
    import etol_amd as E
    c = cases.case_inputs(name)
    ev = E.Evaluator(0)
    ev.set_mesh(c["M"], c["t0"], c["tf"])
    ev.set_model(c["model"], c["params"], maximize=maximize)
    ev.set_batch(c["B"])
    recs = c.get("recs")
    tracks = None
    if c.get("ocp2d"):
        recs, tx, ty = cases.ocp2d_tables(E.edge_ellipse, E.track_centres, ev.node_t)
        tracks = (tx, ty)
        ev.set_tracks(tx, ty)
    if recs is not None:
        ev.set_path(recs, 0, 1)
    mesh = (ev.tau, ev.w, ev.D)
    got = ev.eval_host(c["X"], c["U"])
    # the oracle gets its OWN tables for the shipped problem (independent restatement)
    orecs, otracks = recs, tracks
    if c.get("ocp2d"):
        orecs, otx, oty = cases.ocp2d_tables(O.edge_ellipse, O.track_centres, ev.node_t)
        otracks = (otx, oty)
    ref = O.evaluate(c["model"], c["params"], c["M"], mesh, c["t0"], c["tf"], c["X"], c["U"], orecs, otracks,
                     maximize=maximize)
    return c, ev, got, ref


def check(c, ev, got, ref):
    ns = c["X"].shape[1]
    RES, VALS, COST = got
    rRES, rVALS, rCOST = ref
    # defect rows against sum_j |D_kj||x_j| + h|f|
    absdx = np.einsum("kj,bij->bik", np.abs(ev.D), np.abs(c["X"]))
    scale = absdx + np.abs(rRES[:, :ns]) + 1.0
    err = np.abs(RES[:, :ns] - rRES[:, :ns]) / scale
    assert err.max() < TOL_DEFECT, f"defect rel err {err.max():.3e}"
    if RES.shape[1] > ns:
        s = np.abs(rRES[:, ns:]).max() + 1.0
        assert np.abs(RES[:, ns:] - rRES[:, ns:]).max() / s < TOL_NODE
    for e in range(VALS.shape[1]):
        s = np.abs(rVALS[:, e]).max() + 1.0
        d = np.abs(VALS[:, e] - rVALS[:, e]).max() / s
        assert d < TOL_NODE, f"VALS entry {e}: rel err {d:.3e}"
    assert np.abs(COST - rCOST).max() / (np.abs(rCOST).max() + 1.0) < 1e-13


@pytest.mark.parametrize("name", cases.CASES)
def test_eval_matches_oracle(built, name):
    check(*run_case(name))


def test_maximize_flips_cost_sign(built):
    c, ev, got, ref = run_case("quad_256", maximize=True)
    check(c, ev, got, ref)
    assert (got[2] < 0).all()


def test_epsopt_style_port_agrees(built):
    """third opinion: the std::any / dual-number port of ePSOPT::dae"""
    c, ev, got, _ = run_case("quad_1024_obs")
    ref = O.evaluate(c["model"], c["params"], c["M"], (ev.tau, ev.w, ev.D), c["t0"], c["tf"], c["X"], c["U"],
                     c["recs"], style="epsopt")
    # the port sums D.X in plain double in yet another order: same tolerance class
    check(c, ev, got, ref)


def test_nojac_and_split_passes(built):
    import etol_amd as E
    c, ev, got, ref = run_case("quad_ragged")
    # values-only pass leaves the same RES / COST
    RES2, _, COST2 = ev.eval_host(c["X"], c["U"], flags=E.EVAL_ALL | E.EVAL_NOJAC)
    assert np.array_equal(RES2, got[0]) and np.array_equal(COST2, got[2])
    # node pass then defect-accumulate pass == fused call
    RESn, VALSn, COSTn = ev.eval_host(c["X"], c["U"], flags=E.EVAL_NODES)
    RESd, _, _ = ev.eval_host(c["X"], c["U"], flags=E.EVAL_DEFECT, res_in=RESn)
    assert np.array_equal(RESd, got[0]) and np.array_equal(VALSn, got[1])


def test_linearity_of_defect_operator(built):
    """size-independent property at the full C3 size: K4 is linear in X."""
    import etol_amd as E
    M, B = 1024, 4
    ev = E.Evaluator(0)
    ev.set_mesh(M, 0.0, 16.0)
    ev.set_model(E.MODEL_QUADROTOR2D, [1, 0.01, 9.81, 1, 1])
    ev.set_batch(B)
    rng = np.random.default_rng(5)
    Xa, Xb = rng.standard_normal((2, B, 6, M))
    U = np.zeros((B, 2, M))
    z = np.zeros((B, 6, M))
    Ra, _, _ = ev.eval_host(Xa, U, flags=E.EVAL_DEFECT, res_in=z)
    Rb, _, _ = ev.eval_host(Xb, U, flags=E.EVAL_DEFECT, res_in=z)
    Rab, _, _ = ev.eval_host(2.0 * Xa - 3.0 * Xb, U, flags=E.EVAL_DEFECT, res_in=z)
    scale = np.einsum("kj,bij->bik", np.abs(ev.D), 2 * np.abs(Xa) + 3 * np.abs(Xb))
    assert (np.abs(Rab - (2.0 * Ra - 3.0 * Rb)) / scale).max() < 1e-14
    # D annihilates constants and differentiates tau exactly
    ones = np.ones((B, 6, M))
    R1, _, _ = ev.eval_host(ones, U, flags=E.EVAL_DEFECT, res_in=z)
    assert np.abs(R1).max() < 1e-9
    Rt, _, _ = ev.eval_host(ones * ev.tau, U, flags=E.EVAL_DEFECT, res_in=z)
    assert np.abs(Rt - 1.0).max() < 1e-9


def test_hessian_blocks(built):
    for name in ("pointmass_xml", "quad_ragged", "fixedwing_64"):
        c, ev, _, _ = run_case(name)
        lay = ev.layout
        rng = np.random.default_rng(11)
        lamF = rng.standard_normal((c["B"], lay.ns, c["M"]))
        lamC = rng.standard_normal((c["B"], lay.np, c["M"]))
        H = ev.hess_host(c["X"], c["U"], lamF, lamC, sigma=0.7)
        recs, tracks = c.get("recs"), None
        if c.get("ocp2d"):
            recs, tx, ty = cases.ocp2d_tables(O.edge_ellipse, O.track_centres, ev.node_t)
            tracks = (tx, ty)
        Href = O.hessian(c["model"], c["params"], c["M"], (ev.tau, ev.w, ev.D), c["t0"], c["tf"], c["X"], c["U"],
                         lamF, lamC, 0.7, recs, tracks)
        # oracle Hessian = central difference of complex-step gradients: ~1e-9 relative
        assert np.abs(H - Href).max() / (np.abs(Href).max() + 1.0) < 1e-7


def test_jac_structure_matches_dense_jacobian(built):
    """Assemble the full NLP Jacobian from VALS + structure + (I (x) D) and compare with
    a finite-difference Jacobian of the oracle's constraint vector (small case)."""
    import etol_amd as E
    M, B = 7, 1
    P = [1, 0.01, 9.81, 1, 1]
    ev = E.Evaluator(0)
    ev.set_mesh(M, 0.0, 3.0)
    ev.set_model(E.MODEL_QUADROTOR2D, P)
    ev.set_batch(B)
    recs = np.zeros((2, 8))
    recs[:, 0] = E.PATH_DISC
    recs[:, 1:4] = [[1, 2, 0.3], [4, 1, 0.2]]
    ev.set_path(recs, 0, 1)
    rng = np.random.default_rng(3)
    X = rng.uniform(0, 5, (B, 6, M))
    U = rng.uniform(0, 5, (B, 2, M))
    RES, VALS, COST = ev.eval_host(X, U)
    rows, cols = ev.jac_structure()
    ns, nv, np_ = 6, 8, 2
    n, m = nv * M, ns * M + 2 * ns + np_ * M
    J = np.zeros((m, n))
    g = np.zeros(n)
    v = VALS[0].reshape(-1)
    for e in range(len(v)):
        if rows[e] < 0:
            g[cols[e]] += v[e]
        else:
            J[rows[e], cols[e]] += v[e]
    for i in range(ns):  # off-diagonal part of I (x) D (the diagonal is already in VALS)
        J[i * M:(i + 1) * M, i * M:(i + 1) * M] += ev.D - np.diag(np.diag(ev.D))
    for i in range(ns):  # events (ePSOPT.cpp:281-291): x(t0), x(tf)
        J[ns * M + i, i * M] = 1.0
        J[ns * M + ns + i, i * M + M - 1] = 1.0

    def gfun(zv):
        Xz = zv[:ns * M].reshape(1, ns, M)
        Uz = zv[ns * M:].reshape(1, 2, M)
        R, _, C = O.evaluate(1, P, M, (ev.tau, ev.w, ev.D), 0.0, 3.0, Xz, Uz, recs)
        ev_ = np.concatenate([Xz[0, :, 0], Xz[0, :, -1]])
        return np.concatenate([R[0, :ns].reshape(-1), ev_, R[0, ns:].reshape(-1)]), C[0]

    z0 = np.concatenate([X.reshape(-1), U.reshape(-1)])
    Jfd = np.zeros_like(J)
    gfd = np.zeros(n)
    for q in range(n):
        d = 1e-6
        zp, zm = z0.copy(), z0.copy()
        zp[q] += d
        zm[q] -= d
        (gp, cp), (gm, cm) = gfun(zp), gfun(zm)
        Jfd[:, q] = (gp - gm) / (2 * d)
        gfd[q] = (cp - cm) / (2 * d)
    assert np.abs(J - Jfd).max() < 1e-6 * (np.abs(Jfd).max() + 1)
    assert np.abs(g - gfd).max() < 1e-6 * (np.abs(gfd).max() + 1)


def test_fused_and_general_kernels_agree(built):
    """The fused kernel (even/odd MFMA split + interleaved node work) and the general two-kernel
    path are two implementations of the same pass; also B off the 16-instance tile."""
    import etol_amd as E
    for model, M, B, nobs in ((E.MODEL_QUADROTOR2D, 256, 21, 3), (E.MODEL_QUADROTOR2D, 1024, 3, 20),
                              (E.MODEL_POINTMASS2D, 512, 33, 2)):
        ev = E.Evaluator(0)
        ev.set_mesh(M, 0.0, 12.0)
        if model == E.MODEL_QUADROTOR2D:
            from etol_amd import workloads as W
            ev.set_model(model, W.QUAD_PARAMS)
            X, U, recs = W.quadrotor_batch(9, B, M, nobs)
        else:
            from etol_amd import workloads as W
            ev.set_model(model, [])
            X, U = W.pointmass_batch(9, B, M)
            recs = np.zeros((B, nobs, 8)); recs[:, :, 0] = E.PATH_DISC
            recs[:, :, 1:4] = np.random.default_rng(2).uniform(0.5, 3, (B, nobs, 3))
        ev.set_batch(B)
        ev.set_path(recs, 0, 1)
        assert ev.uses_fused_kernel                # (few instances too: the one-launch pass with a sliced K range, not the skinny kernel)
        by_default = ev.eval_host(X, U)
        assert "emi_pass_f64_kernel" in ev.last_defect_kernel
        ev.set_option("small_rows", 0)             # the MFMA paths, whatever the batch
        assert ev.uses_fused_kernel
        fused = ev.eval_host(X, U)
        fused_nojac = ev.eval_host(X, U, flags=E.EVAL_ALL | E.EVAL_NOJAC)
        # every variant of the even/odd MFMA kernel: register-staged 64- and 128-column tiles (1, 2),
        # LDS-DMA ring (3); and both launch modes
        for ct in (1, 2, 3):
            if ct == 2 and M % 256:
                continue
            for mode in (2, 1):
                ev.set_option("sym_ct", ct)
                ev.set_option("overlap_mode", mode)
                assert ev.uses_fused_kernel
                other = ev.eval_host(X, U)
                assert np.array_equal(other[1], fused[1]) and np.array_equal(other[2], fused[2])
                s = np.einsum("kj,bij->bik", np.abs(ev.D), np.abs(X)) + 1.0
                assert (np.abs(other[0][:, :X.shape[1]] - fused[0][:, :X.shape[1]]) / s).max() < 1e-13
        ev.set_option("sym_ct", 3)
        ev.set_option("overlap_mode", 2)
        ev.set_option("overlap", 0)
        assert not ev.uses_fused_kernel
        general = ev.eval_host(X, U)
        ref = O.evaluate(model, W.QUAD_PARAMS if model == E.MODEL_QUADROTOR2D else [], M, (ev.tau, ev.w, ev.D), 0.0,
                         12.0, X, U, recs)
        c = dict(X=X, U=U)
        check(c, ev, fused, ref)
        check(c, ev, general, ref)
        check(c, ev, by_default, ref)
        assert np.array_equal(by_default[1], fused[1]) and np.array_equal(by_default[2], fused[2])
        assert np.array_equal(fused[1], general[1])            # node work is the same arithmetic
        # (without the Jacobian the pass goes as two launches with an unsplit K range: the defect rows of a small batch agree
        # with the K-sliced one-launch sum to rounding, everything else bit for bit)
        ns = X.shape[1]
        s = np.einsum("kj,bij->bik", np.abs(ev.D), np.abs(X)) + 1.0
        assert (np.abs(fused_nojac[0][:, :ns] - fused[0][:, :ns]) / s).max() < 1e-13
        assert np.array_equal(fused_nojac[0][:, ns:], fused[0][:, ns:]) and np.array_equal(fused_nojac[2], fused[2])


def test_f32_context_fixedwing(built):
    """Config-5 arithmetic type: f32 storage and node arithmetic, D.X accumulated in f64.
    Tolerances: 2e-6 of the row scale (f32 rounding of the inputs), defect rows against
    sum|D_kj||x_j| (an f32 accumulator would lose every digit: |D| reaches N(N+1)/4)."""
    import etol_amd as E
    from etol_amd import workloads as W
    M, B = 256, 3
    X, U = W.fixedwing_batch(4, B, M)
    X = X.astype(np.float32).astype(np.float64)       # what the device will see
    U = U.astype(np.float32).astype(np.float64)
    ev = E.Evaluator(0, f32=True)
    ev.set_mesh(M, 0.0, 20.0)
    ev.set_model(E.MODEL_FIXEDWING12, W.FW_PARAMS)
    ev.set_batch(B)
    assert ev.layout.real_bytes == 4 and not ev.uses_fused_kernel
    RES, VALS, COST = ev.eval_host(X, U)
    rRES, rVALS, rCOST = O.evaluate(E.MODEL_FIXEDWING12, W.FW_PARAMS, M, (ev.tau, ev.w, ev.D), 0.0, 20.0, X, U)
    scale = np.einsum("kj,bij->bik", np.abs(ev.D), np.abs(X)) + np.abs(rRES) + 1.0
    assert (np.abs(RES - rRES) / scale).max() < 2e-6
    for e in range(VALS.shape[1]):
        assert np.abs(VALS[:, e] - rVALS[:, e]).max() / (np.abs(rVALS[:, e]).max() + 1.0) < 2e-6
    assert np.abs(COST - rCOST).max() / np.abs(rCOST).max() < 2e-6
    # device-pointer form with f32 torch tensors
    import torch
    dX = torch.from_numpy(X.astype(np.float32)).cuda()
    dU = torch.from_numpy(U.astype(np.float32)).cuda()
    dRES, dVALS, dCOST = ev.alloc_outputs()
    ev.eval_dev(dX, dU, dRES, dVALS, dCOST)
    ev.synchronize()
    assert np.array_equal(dRES.cpu().numpy().astype(np.float64), RES)


@pytest.mark.parametrize("M,B", [(512, 64), (256, 128), (4096, 16)])
def test_f32_pass_as_one_launch_matches_the_sequential_pair_and_the_oracle(built, M, B):
    """Config-5 arithmetic, the pass as ONE launch (emi_pass_f32_kernel: MFMA-role and node-role workgroups in one grid; the
    defect rows are zeroed and both roles ADD their share with float atomics).  Two contributions per element, 0 + a + b =
    0 + b + a exactly: the rows must equal those of the node kernel followed by the MFMA kernel bit for bit (signed zeros
    aside), VALS and COST too; and all of it the oracle's values to f32 accuracy."""
    import torch
    import etol_amd as E
    from etol_amd import workloads as W
    gen = min(B, 16)
    X, U = W.fixedwing_batch(4, gen, M)
    X, U = np.tile(X, (B // gen, 1, 1)), np.tile(U, (B // gen, 1, 1))
    X = X.astype(np.float32).astype(np.float64) + 0.0
    U = U.astype(np.float32).astype(np.float64)
    ev = E.Evaluator(0, f32=True)
    ev.set_mesh(M, 0.0, 20.0)
    ev.set_model(E.MODEL_FIXEDWING12, W.FW_PARAMS)
    ev.set_batch(B)
    dX, dU = torch.from_numpy(X.astype(np.float32)).cuda(), torch.from_numpy(U.astype(np.float32)).cuda()
    ev.set_option("f32_ring", 0)            # the one-launch kernel holds the register-staged MFMA body: compare like with like
    res = {}
    for mode, name in ((0, "default, one launch allowed"), (1, "sequential"), (3, "one launch")):
        ev.set_option("f32_one_launch", 1 if mode == 0 else 0)          # (by itself the library keeps the sequential pair: it is faster)
        ev.set_option("overlap_mode", mode)
        outs = ev.alloc_outputs()
        for t in outs:
            t.fill_(float("nan"))
        torch.cuda.synchronize()
        for _ in range(2):                       # twice on the same buffers: the rows are zeroed per pass, the ticket resets itself
            ev.eval_dev(dX, dU, *outs)
        ev.synchronize()
        torch.cuda.synchronize()
        assert ("emi_pass_f32_kernel" in ev.last_defect_kernel) == (mode != 1), (name, ev.last_defect_kernel)
        res[mode] = [o.cpu().numpy().astype(np.float64) for o in outs]
        assert not any(np.isnan(a).any() for a in res[mode])
    for q in range(3):
        assert np.array_equal(res[3][q], res[1][q]), q
        assert np.array_equal(res[0][q], res[3][q]), q
    sub = slice(0, 4)
    rRES, rVALS, rCOST = O.evaluate(E.MODEL_FIXEDWING12, W.FW_PARAMS, M, (ev.tau, ev.w, ev.D), 0.0, 20.0, X[sub], U[sub])
    scale = np.einsum("kj,bij->bik", np.abs(ev.D), np.abs(X[sub])) + np.abs(rRES) + 1.0
    assert (np.abs(res[3][0][sub] - rRES) / scale).max() < (2e-6 if M <= 512 else 5e-6)
    for e in range(rVALS.shape[1]):
        assert np.abs(res[3][1][sub][:, e] - rVALS[:, e]).max() / (np.abs(rVALS[:, e]).max() + 1.0) < 2e-6
    assert np.abs(res[3][2][sub] - rCOST).max() / np.abs(rCOST).max() < 2e-6
    # the ring form of the MFMA kernel (LDS-DMA operands, software-pipelined K loop; the default): another pairing of the k
    # terms, same values to f32 accuracy
    ev.set_option("f32_ring", 1)
    ev.set_option("overlap_mode", 0)
    ev.set_option("f32_one_launch", 0)
    outs = ev.alloc_outputs()
    for t in outs:
        t.fill_(float("nan"))
    torch.cuda.synchronize()
    ev.eval_dev(dX, dU, *outs)
    ev.synchronize()
    torch.cuda.synchronize()
    assert "emi_defect_f32_ring_kernel" in ev.last_defect_kernel
    ring = [o.cpu().numpy().astype(np.float64) for o in outs]
    assert np.array_equal(ring[1], res[1][1]) and np.array_equal(ring[2], res[1][2])          # node kernel outputs: identical
    assert (np.abs(ring[0][sub] - rRES) / scale).max() < (2e-6 if M <= 512 else 5e-6)
    full_scale = np.einsum("kj,bij->bik", np.abs(ev.D), np.abs(X)) + np.abs(res[1][0]) + 1.0
    assert (np.abs(ring[0] - res[1][0]) / full_scale).max() < 2e-6                            # every instance, against the staged form
    ev.close()


def test_device_pointer_form_matches_host_form(built):
    import etol_amd as E
    import torch
    c, ev, got, _ = run_case("quad_1024_obs")
    dX, dU = torch.from_numpy(c["X"]).cuda(), torch.from_numpy(c["U"]).cuda()
    RES, VALS, COST = ev.alloc_outputs()
    for _ in range(3):      # repeated passes on the same buffers give the same bits
        ev.eval_dev(dX, dU, RES, VALS, COST)
    ev.synchronize()
    torch.cuda.synchronize()
    assert np.array_equal(RES.cpu().numpy(), got[0]) and np.array_equal(VALS.cpu().numpy(), got[1])
    assert np.array_equal(COST.cpu().numpy(), got[2])
    with pytest.raises(ValueError):
        ev.eval_dev(dX[:1], dU, RES, VALS, COST)


def test_random_shapes_against_oracle(built):
    """Seeded sweep over ragged shapes: node counts on and off every tile size, batch sizes on and
    off the 16-instance tile, every keep-out kind, shared and per-instance tables, both models
    with second derivatives, sign flip."""
    import etol_amd as E
    from etol_amd import workloads as W
    rng = np.random.default_rng(20251003)
    for trial in range(14):
        model = [E.MODEL_POINTMASS2D, E.MODEL_QUADROTOR2D][trial % 2]
        M = int(rng.choice([2, 3, 17, 64, 127, 128, 130, 256, 384]))
        B = int(rng.choice([1, 2, 15, 16, 17, 40]))
        npth = int(rng.choice([0, 1, 4, 7]))
        per_inst = bool(rng.integers(2))
        maximize = bool(rng.integers(2))
        t0, tf = float(rng.uniform(-1, 1)), float(rng.uniform(2, 20))
        ev = E.Evaluator(0)
        ev.set_mesh(M, t0, tf)
        params = W.QUAD_PARAMS if model == E.MODEL_QUADROTOR2D else []
        ev.set_model(model, params, maximize=maximize)
        ev.set_batch(B)
        ns = 6 if model == E.MODEL_QUADROTOR2D else 2
        X = rng.uniform(-2, 8, (B, ns, M))
        U = rng.uniform(-1, 12, (B, 2, M))
        nsets = B if per_inst else 1
        ntracks = 0
        recs = np.zeros((nsets, npth, 8))
        for s in range(nsets):
            for j in range(npth):
                kind = int(rng.integers(3))
                if kind == E.PATH_ELLIPSE:
                    tt = rng.uniform(-3, 3)
                    a2 = rng.uniform(0.05, 2.0)
                    recs[s, j] = [kind, rng.uniform(0, 6), rng.uniform(0, 6), np.cos(tt), np.sin(tt), a2, 0.2 * a2, 0]
                elif kind == E.PATH_DISC:
                    recs[s, j] = [kind, rng.uniform(0, 6), rng.uniform(0, 6), rng.uniform(0.1, 2), 0, 0, 0, 0]
                else:
                    recs[s, j] = [kind, 0, rng.uniform(0.1, 2), 0, 0, 0, 0, 0]   # track index set below
        # tracks: give every TRACK row of set 0 its own track index; other sets reuse the indices in order
        if npth:
            is_trk = recs[:, :, 0] == E.PATH_TRACK
            ntracks = int(is_trk.sum(1).max())
            for s in range(nsets):
                recs[s, is_trk[s], 1] = np.arange(int(is_trk[s].sum()))
        tracks = None
        if ntracks:
            tsets = nsets
            tx = rng.uniform(0, 6, (tsets, ntracks, M))
            ty = rng.uniform(0, 6, (tsets, ntracks, M))
            ev.set_tracks(tx, ty)
            tracks = (tx, ty)
        if npth:
            ev.set_path(recs, 0, 1)
        got = ev.eval_host(X, U)
        ref = O.evaluate(model, params, M, (ev.tau, ev.w, ev.D), t0, tf, X, U, recs if npth else None, tracks,
                         maximize=maximize)
        check(dict(X=X, U=U), ev, got, ref)
        lamF = rng.standard_normal((B, ns, M))
        lamC = rng.standard_normal((B, npth, M))
        H = ev.hess_host(X, U, lamF, lamC, sigma=1.3)
        Href = O.hessian(model, params, M, (ev.tau, ev.w, ev.D), t0, tf, X, U, lamF, lamC, 1.3, recs if npth else None,
                         tracks, maximize=maximize)
        assert np.abs(H - Href).max() / (np.abs(Href).max() + 1.0) < 1e-6, f"trial {trial}"
        ev.close()


def test_abi_error_paths(built):
    """Wrong call order and bad arguments come back as status codes with a message, never a crash."""
    import ctypes as C
    import etol_amd as E
    from etol_amd import _lib as L
    lib = L.load()
    ctx = C.c_void_p()
    assert lib.emi_create(99, C.byref(ctx)) == 1                     # no such device: EMI_ERR_ARG
    assert lib.emi_create(0, C.byref(ctx)) == 0
    z = (C.c_double * 4)()
    assert lib.emi_set_batch(ctx, 4) == 2                            # before the mesh: EMI_ERR_STATE
    assert b"emi_set_mesh" in lib.emi_last_error(ctx)
    assert lib.emi_eval_dev(ctx, None, None, None, None, None, 3) == 2
    tau, w, D = E.lgl(8)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    assert lib.emi_set_mesh(ctx, 8, dp(tau), dp(w), dp(D), 1.0, 1.0) == 1      # tf must exceed t0
    assert lib.emi_set_mesh(ctx, 8, dp(tau), dp(w), dp(D), 0.0, 2.0) == 0
    assert lib.emi_set_path(ctx, 1, 1, z, 0, 1) == 2                 # before the model
    assert lib.emi_set_model(ctx, 1, None, 0, 0) == 1                # quadrotor wants 5 parameters
    assert lib.emi_set_model(ctx, 0, None, 0, 0) == 0
    bad = np.zeros(8); bad[0] = 9
    assert lib.emi_set_path(ctx, 1, 1, dp(bad), 0, 1) == 1           # unknown row kind
    assert lib.emi_set_path(ctx, 1, 1, dp(np.zeros(8)), 0, 0) == 1   # px == py
    assert lib.emi_set_batch(ctx, 0) == 1
    assert lib.emi_set_batch(ctx, 3) == 0
    assert lib.emi_set_path(ctx, 1, 2, dp(np.zeros(16)), 0, 1) == 0  # 2 sets for a batch of 3 ...
    assert lib.emi_eval_dev(ctx, C.c_void_p(8), C.c_void_p(8), C.c_void_p(8), C.c_void_p(8), C.c_void_p(8), 3) == 2
    assert b"sets" in lib.emi_last_error(ctx)                        # ... is caught before any launch
    assert lib.emi_set_option(ctx, b"no_such_option", 1) == 1
    assert lib.emi_eval_dev(ctx, None, None, None, None, None, 0) in (1, 2)
    assert lib.emi_hess_dev(ctx, None, None, None, None, 1.0, None) in (1, 2)
    assert lib.emi_destroy(ctx) == 0
    assert lib.emi_destroy(None) == 1


@pytest.mark.parametrize("inputs", ["smooth", "bench"])
def test_f32_defect_survives_single_precision_at_4096_nodes(built, inputs):
    """Config 5 shape: M = 4096, 12 states, f32, against the CPU ORACLE (oracle/emi_oracle.c: D.X accumulated in
    long double from the f64 matrix).  The f32 MFMA kernel contracts D with x_j - s (s = x at the centre of each
    32-column output tile); its error is measured relative to sum_j |D_kj||x_j - x_k| -- the scale of the shifted
    terms -- on (a) smooth trajectories, what an NLP iterate looks like, and (b) the seeded random inputs of the
    benchmark.  Without the shift (the f64-accumulating fallback on the same f32 operands) the
    smooth case is more than a digit worse."""
    import etol_amd as E
    from etol_amd import workloads as W
    M, B = 4096, 2
    ev = E.Evaluator(0, f32=True)
    ev.set_mesh(M, 0.0, 20.0)
    ev.set_model(E.MODEL_FIXEDWING12, W.FW_PARAMS)
    ev.set_batch(B)
    if inputs == "smooth":
        rng = np.random.default_rng(99)
        basis = np.stack([ev.tau ** p for p in range(6)])                    # low-order polynomials in tau
        sx = np.array([100, 100, 50, 0.4, 0.3, 3.0, 5, 2, 2, 0.5, 0.5, 0.5])
        X = (rng.standard_normal((B, 12, 6)) * sx[None, :, None] / 3) @ basis
        X[:, 6] += 25.0
        U = (rng.standard_normal((B, 4, 6)) * 0.1) @ basis + np.array([30, 0, 0, 0.0])[None, :, None]
    else:
        X, U = W.fixedwing_batch(4, B, M)
    X = X.astype(np.float32).astype(np.float64)                              # what the f32 context sees
    U = U.astype(np.float32).astype(np.float64)
    RES, _, _ = ev.eval_host(X, U)
    ev.set_option("overlap", 0)                 # forces the unshifted fallback kernel
    RES_plain, _, _ = ev.eval_host(X, U)
    ref = O.evaluate(E.MODEL_FIXEDWING12, W.FW_PARAMS, M, (ev.tau, ev.w, ev.D), 0.0, 20.0, X, U)[0]
    A = np.abs(ev.D)
    scale = np.zeros_like(ref)
    for k0 in range(0, M, 256):
        diff = np.abs(X[:, :, None, :] - X[:, :, k0:k0 + 256, None])          # [B][12][256][M]
        scale[:, :, k0:k0 + 256] = np.einsum("kj,bskj->bsk", A[k0:k0 + 256], diff)
    err = np.abs(RES - ref) / (scale + np.abs(ref) + 1.0)
    err_plain = np.abs(RES_plain - ref) / (scale + np.abs(ref) + 1.0)
    print(f"f32 defect, {inputs}: max err {err.max():.3e} of sum|D||x_j - x_k| (unshifted: {err_plain.max():.3e})")
    # measured: smooth 6.6e-8 (f32 epsilon is 6e-8; unshifted 2.2e-6); bench inputs 1.4e-6 -- with random node values
    # x_j - s is as large as x_j, the shift buys nothing and 4096 f32 accumulations cost ~20 epsilon
    assert err.max() < (2e-6 if inputs == "smooth" else 5e-6), err.max()
    if inputs == "smooth":
        assert err_plain.max() > 10 * err.max()  # the unshifted form loses more than a digit on top


@pytest.mark.parametrize("which", [0, 1])
def test_rows_match_reference_executed_callbacks(built, which):
    """HIP path against vectors the REFERENCE's own code produced (tests/golden/ref_dymos_ex1.json:
    src/Examples/Dymos/etol_dymos_example1.cpp callbacks run by oracle/_ref/ref_vectors on the shipped
    ocp_2d_ex1.xml data and on a synthetic table set): the 9 ellipse rows and 2 moving-disc rows of config 1,
    their partials, dynamics partials, cost and cost gradient; track centres bit for bit."""
    import etol_amd as E
    import test_ref_vectors as R
    c = R._load("ref_dymos_ex1.json")["cases"][which]
    M, node_t = c["M"], np.array(c["node_t"])
    t0, tf = node_t[0], node_t[-1]
    h = (tf - t0) / 2
    ev = E.Evaluator(0)
    ev.set_mesh(M, t0, tf)
    assert np.abs(ev.node_t - node_t).max() < 1e-14 * tf
    ev.set_model(E.MODEL_POINTMASS2D, [])
    X, U = np.array(c["X"]), np.array(c["U"])
    ev.set_batch(X.shape[0])
    recs, tx, ty = R.dymos_tables(c, E.edge_ellipse, E.track_centres)
    ev.set_tracks(tx, ty)
    ev.set_path(recs, 0, 1)
    RES, VALS, COST = ev.eval_host(X, U)
    ref = R.compare_with_reference(c, RES, VALS, h)
    assert np.array_equal(tx, ref["centres_hdr"][:, 0]) and np.array_equal(ty, ref["centres_hdr"][:, 1])
    assert np.abs(COST - h * (ref["L"] * ev.w).sum(axis=1)).max() < 1e-13 * np.abs(COST).max()
    DX = np.einsum("kj,bij->bik", ev.D, X)
    assert np.abs((DX - RES[:, :2]) / h - ref["F"]).max() < 1e-11
    for v in range(4):
        assert np.abs(VALS[:, 8 + 2 * recs.shape[0] + v] - h * ev.w * ref["L_p"][:, v]).max() < R.TOL * h
    ev.close()


def test_setters_in_any_order_leave_no_stale_sizes(built):
    """emi_set_mesh after emi_set_batch re-sizes the per-block cost partials (64 -> 1025 nodes is 1 -> 5 blocks per
    instance), and emi_set_model drops a path table whose (px, py) name states of the previous model."""
    import etol_amd as E
    from etol_amd import _lib
    ev = E.Evaluator(0)
    B = 3
    ev.set_mesh(64, 0.0, 4.0)
    ev.set_model(E.MODEL_QUADROTOR2D, cases.W.QUAD_PARAMS)
    ev.set_batch(B)
    X, U, recs = cases.W.quadrotor_batch(11, B, 1025, 2)
    ev.set_path(recs[:1], 3, 4)                       # rows on states 3 and 4
    ev.set_mesh(1025, 0.0, 4.0)                       # no emi_set_batch after it
    got = ev.eval_host(X, U)
    ref = O.evaluate(E.MODEL_QUADROTOR2D, cases.W.QUAD_PARAMS, 1025, (ev.tau, ev.w, ev.D), 0.0, 4.0, X, U, recs[:1], px=3, py=4)
    assert np.abs(got[2] - ref[2]).max() / np.abs(ref[2]).max() < 1e-13
    assert np.abs(got[0][:, 6:] - ref[0][:, 6:]).max() < 1e-12
    ev.set_model(E.MODEL_POINTMASS2D, [])             # 2 states: a table on states (3, 4) must not survive
    Xp, Up = cases.W.pointmass_batch(3, B, 1025)
    RES, VALS, COST = ev.eval_host(Xp, Up)
    assert RES.shape[1] == 2
    refp = O.evaluate(E.MODEL_POINTMASS2D, [], 1025, (ev.tau, ev.w, ev.D), 0.0, 4.0, Xp, Up)
    assert np.abs(COST - refp[2]).max() / np.abs(refp[2]).max() < 1e-13
    ev.close()


def test_rccl_gather_entry_points_world_of_one(built):
    """emi_comm_* (the RCCL gather of include/emi355x.h) with a one-rank communicator on the test box: id, create,
    gather to self (the root's block is a device copy), error paths.  More ranks need more GPUs: the gloo tests
    cover the partition, the driver's multi-GPU run the transfer."""
    import ctypes as C
    import torch
    from etol_amd import _lib as L
    lib = L.load()
    ident = (C.c_char * 128)()
    assert lib.emi_comm_unique_id(ident) == 0, lib.emi_comm_last_error(None)
    comm = C.c_void_p()
    assert lib.emi_comm_create(0, 1, 1, ident, C.byref(comm)) == 1                    # rank out of range
    assert lib.emi_comm_create(0, 1, 0, ident, C.byref(comm)) == 0, lib.emi_comm_last_error(None)
    src = torch.arange(4096, dtype=torch.float64, device="cuda:0")
    dst = torch.zeros(4096, dtype=torch.float64, device="cuda:0")
    assert lib.emi_comm_gather(comm, src.data_ptr(), None, src.numel() * 8, 0, None) == 1  # root without a buffer
    assert b"receive buffer" in lib.emi_comm_last_error(comm)
    assert lib.emi_comm_gather(comm, src.data_ptr(), dst.data_ptr(), src.numel() * 8, 0, None) == 0, lib.emi_comm_last_error(comm)
    assert torch.equal(src, dst)
    assert lib.emi_comm_destroy(comm) == 0


@pytest.mark.parametrize("sym_ct", [3, 5, 6, 7, 8, 0])
@pytest.mark.parametrize("shape", [(1024, 8), (256, 19), (128, 32)])
def test_every_mfma_defect_kernel_variant_matches_the_oracle(built, sym_ct, shape):
    """The even/odd MFMA defect kernels (one-workgroup ring; state-split rings with SW = 6 / 2 / 1 / 3; the choice by
    batch size), with the streaming kernel beside them and its three store policies: same results as the oracle at
    several mesh and batch sizes (B not a multiple of the 16-instance tile, M = 128 = one tile)."""
    import etol_amd as E
    M, B = shape
    ev = E.Evaluator(0)
    ev.set_mesh(M, 0.0, 9.0)
    ev.set_model(E.MODEL_QUADROTOR2D, cases.W.QUAD_PARAMS)
    ev.set_batch(B)
    X, U, recs = cases.W.quadrotor_batch(21, B, M, 3)
    ev.set_path(recs[:1], 0, 1)
    ev.set_option("sym_ct", sym_ct)
    ref = O.evaluate(E.MODEL_QUADROTOR2D, cases.W.QUAD_PARAMS, M, (ev.tau, ev.w, ev.D), 0.0, 9.0, X, U, recs[:1])
    c = dict(X=X)
    for store in (0, 1, 2):
        ev.set_option("node_store", store)
        got = ev.eval_host(X, U)
        assert ev.uses_fused_kernel
        check(c, ev, got, ref)
    ev.set_option("overlap_mode", 3)      # the pass as one launch (MFMA-role and node-role workgroups, COST by ticket)
    for store in (1, 3):                  # ... with write-through (sc1) and nt sc1 result stores (compiler-issued buffer stores)
        ev.set_option("node_store", store)
        check(c, ev, ev.eval_host(X, U), ref)
    ev.set_option("node_store", -1)
    one = ev.eval_host(X, U)
    if sym_ct != 3 and M == 1024:      # (the other shapes do not fill whole XCD shares and fall back to two streams)
        assert "one launch" in ev.last_defect_kernel, ev.last_defect_kernel
    check(c, ev, one, ref)
    assert np.array_equal(ev.eval_host(X, U)[2], one[2])              # COST: fixed summation order
    ev.set_option("overlap_mode", 2)
    if sym_ct in (5, 6, 7, 8):
        # split-K: partial sums through the slab, combined in slice order -- by the workgroup that draws a tile's last
        # ticket (default) or by a second launch ("sym_combine" 0): bitwise the same, and reproducible
        for ks in (2, 4, 8):
            ev.set_option("sym_ksplit", ks)
            ev.set_option("sym_combine", 1)
            got2 = ev.eval_host(X, U)
            split = "K slices" in ev.last_defect_kernel
            assert split or M // 16 // ks < 2 or (M // 16 // ks) % 2
            assert not split or "in-kernel" in ev.last_defect_kernel
            check(c, ev, got2, ref)
            again = ev.eval_host(X, U)
            assert np.array_equal(again[0], got2[0])          # fixed summation order: bitwise reproducible
            ev.set_option("sym_combine", 0)
            got3 = ev.eval_host(X, U)
            assert not split or "combine_kernel" in ev.last_defect_kernel
            assert np.array_equal(got3[0], got2[0])
            ev.set_option("sym_combine", 1)
            ev.set_option("overlap_mode", 3)                  # ... and as one launch
            got4 = ev.eval_host(X, U)
            if M == 1024:
                assert "one launch" in ev.last_defect_kernel and "K slices" in ev.last_defect_kernel, ev.last_defect_kernel
                assert np.array_equal(got4[0], got2[0])
            check(c, ev, got4, ref)
            ev.set_option("overlap_mode", 2)
        ev.set_option("sym_ksplit", 0)
    ev.close()


@pytest.mark.parametrize("shape", [(1024, 128), (1024, 19), (128, 40), (384, 33), (2048, 16), (1024, 512), (256, 48), (768, 256)])
def test_deep_k_tiles_of_the_mfma_role_match_the_oracle(built, shape):
    """"sym_bk" 16: K tiles of 16 instead of 8 in the MFMA role of the one-launch pass (half as many barriers and counted waits per
    flop; 128-byte operand rows, eight chunks, their own swizzle; two ds_read_b128 per operand row and tile; mirrored chunks 7 - kq
    and 3 - kq).  SW = 1 and SW = 2, every store flavour, tile orders, against the oracle and -- same sums in another order of
    the k-steps? no: the k terms of a row are added in the SAME order, tile after tile, so the defect rows must be BITWISE those of
    the 8-deep form."""
    import etol_amd as E
    M, B = shape
    ev = E.Evaluator(0)
    ev.set_mesh(M, 0.0, 9.0)
    ev.set_model(E.MODEL_QUADROTOR2D, cases.W.QUAD_PARAMS)
    ev.set_batch(B)
    X, U, recs = cases.W.quadrotor_batch(29, B, M, 3)
    ev.set_path(recs[:1], 0, 1)
    ref = O.evaluate(E.MODEL_QUADROTOR2D, cases.W.QUAD_PARAMS, M, (ev.tau, ev.w, ev.D), 0.0, 9.0, X, U, recs[:1])
    c = dict(X=X)
    ev.set_option("overlap_mode", 3)
    for sym_ct in (7, 6):
        ev.set_option("sym_ct", sym_ct)
        ev.set_option("sym_ksplit", 1)
        ev.set_option("sym_bk", 8)
        base = ev.eval_host(X, U)
        assert "one launch" in ev.last_defect_kernel and "K tiles of 16" not in ev.last_defect_kernel
        check(c, ev, base, ref)
        ev.set_option("sym_bk", 16)
        assert ev.plan(B)["k_tile"] == 16
        for store in (0, 2, 1, 3):
            ev.set_option("node_store", store)
            for cpart, gblk in ((0, 0), (-1, 0), (2, 0), (0, 2)):
                ev.set_option("sym_cpart", cpart)
                ev.set_option("sym_gblk", gblk)
                poison = ev.eval_host(X + 1.0, U)
                got = ev.eval_host(X, U)
                assert "K tiles of 16" in ev.last_defect_kernel, ev.last_defect_kernel
                check(c, ev, got, ref)
                assert not np.array_equal(poison[0], got[0])
                # k-steps in another grouping: (0,2,4,6 | 1,3,5,7) per 8 against (0,2,4,6 | 1,3,5,7 | 8,.. | 9,..) per 16 -- the same
                # sequence of additions per accumulator, hence the same bits
                assert np.array_equal(got[0], base[0]), (sym_ct, store, cpart, gblk)
        ev.set_option("node_store", -1)
        ev.set_option("sym_cpart", 0)
        ev.set_option("sym_gblk", 0)
        if sym_ct == 6 and M % 256 == 0:
            # "sym_ctc" 2: two 64-column sub-tiles per MFMA workgroup (an X tile read by half as many workgroups, the x fragments shared by
            # the MFMAs of both sub-tiles), with 8- and 16-deep K tiles, every store flavour and tile order: the same additions per
            # output element in the same order, so again the bits of the base form
            ev.set_option("sym_ctc", 2)
            for bk in (8, 16):
                ev.set_option("sym_bk", bk)
                assert ev.plan(B)["column_tiles"] == 2 and ev.plan(B)["k_tile"] == bk
                for store in (0, 2):
                    ev.set_option("node_store", store)
                    for cpart, gblk in ((0, 0), (-1, 0), (2, 0), (0, 2), (0, 1)):
                        ev.set_option("sym_cpart", cpart)
                        ev.set_option("sym_gblk", gblk)
                        poison = ev.eval_host(X + 1.0, U)
                        got = ev.eval_host(X, U)
                        assert "128-column tiles" in ev.last_defect_kernel, ev.last_defect_kernel
                        check(c, ev, got, ref)
                        assert not np.array_equal(poison[0], got[0])
                        assert np.array_equal(got[0], base[0]), (bk, store, cpart, gblk)
            ev.set_option("sym_ctc", 0)
            ev.set_option("sym_bk", 0)
            ev.set_option("node_store", -1)
            ev.set_option("sym_cpart", 0)
            ev.set_option("sym_gblk", 0)
        # "sym_hs" 2: the K range of a tile in two halves inside a 512-thread workgroup, partial sums of the second half through LDS: the two
        # halves are added once, so the values (not the bits) of the base form; reproducible from call to call
        ev.set_option("sym_hs", 2)
        for bk in (8, 16):
            if (M // 2 // bk) % 4:
                continue
            ev.set_option("sym_bk", bk)
            assert ev.plan(B)["k_halves"] == 2, ev.plan(B)
            for store in (0, 2):
                ev.set_option("node_store", store)
                for cpart in (0, -1, 2):
                    ev.set_option("sym_cpart", cpart)
                    poison = ev.eval_host(X + 1.0, U)
                    got = ev.eval_host(X, U)
                    assert "two halves" in ev.last_defect_kernel, ev.last_defect_kernel
                    check(c, ev, got, ref)
                    assert not np.array_equal(poison[0], got[0])
                    assert np.array_equal(ev.eval_host(X, U)[0], got[0])
        ev.set_option("sym_hs", 0)
        ev.set_option("sym_bk", 0)
        ev.set_option("node_store", -1)
        ev.set_option("sym_cpart", 0)
    ev.close()


@pytest.mark.parametrize("shape", [(1024, 40), (2048, 24), (512, 72), (256, 128), (1280, 32), (768, 48), (1024, 256), (512, 384)])
def test_partitioned_tile_orders_of_the_mfma_role(built, shape):
    """"sym_cpart": the column tiles of the state-split ring cut into 1 / 2 / 4 / 8 partitions over the XCDs (the default
    for large batches), against the plain order and the oracle -- for meshes with 2 .. 16 column tiles (10 and 6 among them: partitions of 5 and 3 columns), batch sizes whose
    group counts do or do not divide over the XCD groups (the plan then falls back to the plain order), SW = 6 / 2 / 1,
    two streams and one launch.  Every tile is visited exactly once or rows of the result stay unwritten / stale."""
    import etol_amd as E
    M, B = shape
    ev = E.Evaluator(0)
    ev.set_mesh(M, 0.0, 9.0)
    ev.set_model(E.MODEL_QUADROTOR2D, cases.W.QUAD_PARAMS)
    ev.set_batch(B)
    X, U, recs = cases.W.quadrotor_batch(23, B, M, 2)
    ev.set_path(recs[:1], 0, 1)
    ref = O.evaluate(E.MODEL_QUADROTOR2D, cases.W.QUAD_PARAMS, M, (ev.tau, ev.w, ev.D), 0.0, 9.0, X, U, recs[:1])
    c = dict(X=X)
    for sym_ct in (5, 6, 7):
        ev.set_option("sym_ct", sym_ct)
        base = None
        for mode in (2, 3):
            ev.set_option("overlap_mode", mode)
            for cpart, gblk, cxo in ((-1, 0, 0), (1, 0, 0), (2, 0, 0), (4, 0, 0), (8, 0, 0), (0, 0, 0), (0, 1, 1), (0, 1, 2), (0, 2, 2), (0, 2, 4), (0, 4, 1)):
                ev.set_option("sym_cpart", cpart)
                ev.set_option("sym_gblk", gblk)                # grouped order: super-blocks of gblk instance groups, column blocks of cxo
                ev.set_option("sym_cx", cxo)
                poison = ev.eval_host(X + 1.0, U)              # another input first: a tile left out would keep these rows
                got = ev.eval_host(X, U)
                assert ev.uses_fused_kernel
                check(c, ev, got, ref)
                if base is None:
                    base = got
                assert np.array_equal(got[0], base[0]), (sym_ct, mode, cpart, gblk, cxo)      # the order of the tiles changes no bit
                assert not np.array_equal(poison[0], got[0])
            ev.set_option("sym_gblk", 0)
            ev.set_option("sym_cx", 0)
    ev.close()


def test_default_dispatch_on_random_shapes_matches_the_general_path(built):
    """tools/gpu_stress.py with a fixed seed: 30 random (mesh, batch, keep-out count, model) shapes through whatever the
    policy of emi_eval_dev picks, against the sequential general path (which the tests above pin to the oracle).  (This is
    the check that found the 5-column partition bug of the tile order.)"""
    import argparse
    import importlib.util
    spec = importlib.util.spec_from_file_location("gpu_stress", os.path.join(ROOT, "tools", "gpu_stress.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    worst, kinds = mod.run(argparse.Namespace(cases=30, seed=7, big=False))
    assert worst < 1e-11 and len(kinds) >= 2, (worst, kinds)      # (since the grid is padded, every shape goes as the one-launch pass: SW = 1 / 2)


def test_very_large_batches_are_evaluated_in_slices_with_the_same_results(built):
    """emi_eval_dev evaluates a batch above 2048 instances as one launch over its multiple of 256 instances plus a second
    launch for the remainder (per-instance keep-out tables, cost partials and outputs offset per piece), or in pieces of
    "slice" instances when that option is set; with per-kernel profiling on it evaluates the batch in one piece: same results."""
    import torch
    import etol_amd as E
    M, B = 128, 2100
    ev = E.Evaluator(0)
    ev.set_mesh(M, 0.0, 5.0)
    ev.set_model(E.MODEL_QUADROTOR2D, cases.W.QUAD_PARAMS)
    ev.set_batch(B)
    X, U, recs = cases.W.quadrotor_batch(31, 60, M, 2)
    reps = (B + 59) // 60
    X, U, recs = np.tile(X, (reps, 1, 1))[:B], np.tile(U, (reps, 1, 1))[:B], np.tile(recs, (reps, 1, 1))[:B]
    X = X + 1e-3 * np.arange(B)[:, None, None]               # every instance different
    ev.set_path(recs, 0, 1)
    dX, dU = torch.from_numpy(X).cuda(), torch.from_numpy(U).cuda()
    a, b = ev.alloc_outputs(), ev.alloc_outputs()
    ev.eval_dev(dX, dU, *a)                                   # 2048 + 52 instances
    c3 = ev.alloc_outputs()
    ev.set_option("slice", 512)
    ev.eval_dev(dX, dU, *c3)                                  # 512 + 512 + 512 + 512 + 52
    ev.set_option("slice", 0)
    torch.cuda.synchronize()
    for p, q in zip(a, c3):
        assert (p - q).abs().max().item() <= 1e-12 * q.abs().max().item()
    ev.profile(1)
    ev.eval_dev(dX, dU, *b)                                   # one piece
    ev.profile_read()
    ev.profile(0)
    torch.cuda.synchronize()
    # same values; not always the same bits: the defect kernel variant is chosen from the batch size of the piece it
    # works on, and the split-K variants add the K slices in another (fixed) order
    for p, q in zip(a, b):
        assert (p - q).abs().max().item() <= 1e-12 * q.abs().max().item()
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])      # node kernel outputs: identical
    ref = O.evaluate(E.MODEL_QUADROTOR2D, cases.W.QUAD_PARAMS, M, (ev.tau, ev.w, ev.D), 0.0, 5.0, X[2090:2094], U[2090:2094], recs[2090:2094])
    assert np.abs(a[2][2090:2094].cpu().numpy() - ref[2]).max() / np.abs(ref[2]).max() < 1e-13
    ev.close()


@pytest.mark.parametrize("sym_ct", [0, 3, 5, 7])
def test_mfma_defect_kernel_variants_for_the_two_state_model(built, sym_ct):
    """The 2-state point mass (the reference example's model) through the even/odd MFMA kernels: SW = 2 (all states),
    SW = 1, the round-1 ring and the default choice, two streams and one launch, with ellipse and moving-disc rows."""
    import etol_amd as E
    M, B = 256, 40
    ev = E.Evaluator(0)
    ev.set_mesh(M, 0.0, 16.0)
    ev.set_model(E.MODEL_POINTMASS2D, [])
    ev.set_batch(B)
    X, U = cases.W.pointmass_batch(5, B, M)
    recs, tx, ty = cases.ocp2d_tables(E.edge_ellipse, E.track_centres, ev.node_t)
    ev.set_tracks(tx, ty)
    ev.set_path(recs, 0, 1)
    ev.set_option("sym_ct", sym_ct)
    ref = O.evaluate(E.MODEL_POINTMASS2D, [], M, (ev.tau, ev.w, ev.D), 0.0, 16.0, X, U, recs, (tx, ty))
    for mode in (2, 3, 0):
        ev.set_option("overlap_mode", mode)
        got = ev.eval_host(X, U)
        assert ev.uses_fused_kernel
        check(dict(X=X), ev, got, ref)
    ev.close()


def _expected_default_form(ev, B):
    """The name emi_eval_dev reports for the LAST launch of a batch of B instances, from the library's own statement of its
    launch policy (emi_plan_pass: csrc/emi_api.hip plan_pass / plan_piece are the one definition; nothing is restated here)."""
    p = ev.plan(B)
    last = p["tail"] if p["tail"] else (p["piece"] if p["piece"] else B)
    q = ev.plan(last)
    assert q["one_launch"] == 1 and q["piece"] == 0
    return (f"emi_pass_f64_kernel<SW={q['sw']}> (MFMA + node roles, one launch" + (f", {q['ksplit']} K slices" if q["ksplit"] > 1 else ")")), q


# the shapes the benchmark lines and the profiles quote, pinned: a policy change has to be made here as well as in plan_pass
PINNED_PLANS = {128: dict(sw=2, ksplit=1, k_tile=16, store_mode=1, block_order=1, piece=0), 1024: dict(sw=2, ksplit=1, k_tile=16, store_mode=2, piece=0),
                16: dict(sw=1, ksplit=4), 64: dict(sw=1, ksplit=2), 80: dict(sw=1, ksplit=1, k_tile=16), 112: dict(sw=2, ksplit=1, k_tile=16),
                320: dict(sw=2, ksplit=1, k_tile=16, block_order=150), 4096: dict(sw=2, k_tile=8, column_tiles=2, store_mode=2, piece=0), 256: dict(sw=2, k_tile=8), 512: dict(sw=2, k_tile=16)}


@pytest.mark.parametrize("B", [1, 3, 5, 16, 64, 80, 112, 128, 256, 320, 512, 1024, 2064, 2560, 4096])
def test_default_dispatch_at_the_benchmarked_shapes_matches_the_oracle(built, B):
    """The shapes bench.py and config 4 actually run -- M = 1024, 20 PER-INSTANCE keep-outs, B = 1 / 3 / 5 (role counts that are no
    multiple of 8: the grid is padded with workgroups that return at once), 16 / 64 (SW = 1 with 4 / 2 K slices per tile, combined
    in-kernel by ticket), 80 / 112 (no slices any more above 64 instances: SW = 1 / SW = 2 with 16-deep K tiles), 320 (MFMA workgroups at 1.5 x the
    even density, 16-deep), 128 (the shard of config 4:
    one launch, SW = 2 with 16-deep K tiles, MFMA workgroups first, write-through stores), 256 (SW = 2, MFMA workgroups first, non-temporal stores), 512 (SW = 2, MFMA
    workgroups at 1.25 x the even density), 1024 (the headline: SW = 2, MFMA workgroups at 1.1 x the even density, 2 column partitions), 2064 (one launch of
    2048 instances + a 16-instance tail), 2560 (one launch in the grouped tile order) -- through the DEFAULT dispatch (no option
    set), device-pointer form as bench.py calls it, against the CPU oracle on sampled instances that sit on every tile
    edge: 0, 15, 16, B/2, B-1, first / last of every slice.  Outputs are poisoned first: a tile or a role left out shows."""
    import torch
    import etol_amd as E
    M = 1024
    ev = E.Evaluator(0)
    ev.set_mesh(M, 0.0, cases.W.TF)
    ev.set_model(E.MODEL_QUADROTOR2D, cases.W.QUAD_PARAMS)
    ev.set_batch(B)
    X, U, recs = cases.W.quadrotor_batch(3, B, M, 20)
    ev.set_path(recs, 0, 1)
    dX, dU = torch.from_numpy(X).cuda(), torch.from_numpy(U).cuda()
    outs = ev.alloc_outputs()
    for t in outs:
        t.fill_(float("nan"))
    torch.cuda.synchronize()                 # (the fills run on torch's stream, the evaluator launches on its own)
    for _ in range(2):                       # twice: self-resetting tickets, same bits
        ev.eval_dev(dX, dU, *outs)
    ev.synchronize()
    torch.cuda.synchronize()
    expected, plan = _expected_default_form(ev, B)
    assert expected in ev.last_defect_kernel, (ev.last_defect_kernel, plan)
    for k, v in PINNED_PLANS.get(B, {}).items():
        assert plan[k] == v, (B, k, plan)
    for t in outs:
        assert not torch.isnan(t).any().item()          # every row of every instance was written
    sample = [0, 15, 16, 17, B // 2 - 1, B // 2, B - 16, B - 1, 1023, 1024, 2047, 2048, 2063, 2303, 2304, 4080]
    e = O.sampled_errors(E.MODEL_QUADROTOR2D, cases.W.QUAD_PARAMS, M, (ev.tau, ev.w, ev.D), 0.0, cases.W.TF, X, U, recs, outs, sample)
    print(f"B={B}: {ev.last_defect_kernel}: {e}")
    assert e["defect"] < TOL_DEFECT and e["path"] < TOL_NODE and e["vals"] < TOL_NODE and e["cost"] < 1e-13, e
    ev.close()


def test_full_report_size_16384_instances_against_the_oracle_and_by_reordering(built):
    """SURVEY section 8(d)'s largest report size, B = 16384 at M = 1024 with 20 per-instance keep-outs (17 GB of results: inputs of
    this size never stay in the Infinity Cache between the roles, a different regime from 1024 instances), through the default
    dispatch.  Two checks: (1) sampled instances on every tile / slice edge against the CPU oracle; (2) a size-independent property
    over ALL instances -- the same batch in REVERSED instance order must give the same bits per instance (an instance's rows depend on
    nothing but its own inputs; every instance then sits in another tile, another XCD share and another launch position)."""
    import torch
    import etol_amd as E
    M, B = 1024, 16384
    if torch.cuda.get_device_properties(0).total_memory < 60e9:
        pytest.skip("needs 40 GB of device memory")
    ev = E.Evaluator(0)
    ev.set_mesh(M, 0.0, cases.W.TF)
    ev.set_model(E.MODEL_QUADROTOR2D, cases.W.QUAD_PARAMS)
    ev.set_batch(B)
    X, U, recs = cases.W.quadrotor_batch(11, B, M, 20)
    recs = np.asarray(recs)
    assert recs.ndim == 3 and recs.shape[0] == B
    ev.set_path(recs, 0, 1)
    dX, dU = torch.from_numpy(X).cuda(), torch.from_numpy(U).cuda()
    outs = ev.alloc_outputs()
    for t in outs:
        t.fill_(float("nan"))
    torch.cuda.synchronize()
    ev.eval_dev(dX, dU, *outs)
    ev.synchronize()
    expected, plan = _expected_default_form(ev, B)
    assert expected in ev.last_defect_kernel, (ev.last_defect_kernel, plan)
    assert plan["column_tiles"] == 2 and plan["piece"] == 0, plan
    for t in outs:
        assert not torch.isnan(t).any().item()
    sample = [0, 15, 16, 255, 256, 2047, 2048, 4095, 4096, 8191, 8192, 12287, 16367, 16368, 16383]
    e = O.sampled_errors(E.MODEL_QUADROTOR2D, cases.W.QUAD_PARAMS, M, (ev.tau, ev.w, ev.D), 0.0, cases.W.TF, X, U, recs, outs, sample)
    print(f"B={B}: {ev.last_defect_kernel}: {e}")
    assert e["defect"] < TOL_DEFECT and e["path"] < TOL_NODE and e["vals"] < TOL_NODE and e["cost"] < 1e-13, e
    # the same instances in reversed order
    ev.set_path(np.ascontiguousarray(recs[::-1]), 0, 1)
    rX, rU = dX.flip(0).contiguous(), dU.flip(0).contiguous()
    outs2 = ev.alloc_outputs()
    for t in outs2:
        t.fill_(float("nan"))
    torch.cuda.synchronize()
    ev.eval_dev(rX, rU, *outs2)
    ev.synchronize()
    torch.cuda.synchronize()
    for a, b in zip(outs, outs2):
        # compared in slabs: a flipped copy of the 13 GB VALS array beside both result sets is not needed
        for i0 in range(0, B, 2048):
            assert torch.equal(a[i0:i0 + 2048], b[B - i0 - 2048:B - i0].flip(0)), i0
    ev.close()


@pytest.mark.parametrize("B", [256, 1024])
def test_front_loaded_pass_orders_match_the_oracle(built, B):
    """"pass_order" >= 100 (MFMA workgroups at 1.25 x / 4 x / far beyond the even density, node workgroups at the tail: the
    block -> role map whose first version sent node workgroups out of range).  pass_role_of clamps the density itself; every
    order must give the bits of the default order and the oracle's values."""
    import torch
    import etol_amd as E
    M = 1024
    ev = E.Evaluator(0)
    ev.set_mesh(M, 0.0, cases.W.TF)
    ev.set_model(E.MODEL_QUADROTOR2D, cases.W.QUAD_PARAMS)
    ev.set_batch(B)
    X, U, recs = cases.W.quadrotor_batch(3, B, M, 20)
    ev.set_path(recs, 0, 1)
    ev.set_option("overlap_mode", 3)
    dX, dU = torch.from_numpy(X).cuda(), torch.from_numpy(U).cuda()
    base = None
    for order in (-1, 0, 1, 125, 400, 100000):
        ev.set_option("pass_order", order)
        outs = ev.alloc_outputs()
        for t in outs:
            t.fill_(float("nan"))
        torch.cuda.synchronize()             # (the fills run on torch's stream, the evaluator launches on its own)
        ev.eval_dev(dX, dU, *outs)
        ev.synchronize()
        torch.cuda.synchronize()
        assert "one launch" in ev.last_defect_kernel
        for t in outs:
            assert not torch.isnan(t).any().item(), order
        if base is None:
            base = outs
            e = O.sampled_errors(E.MODEL_QUADROTOR2D, cases.W.QUAD_PARAMS, M, (ev.tau, ev.w, ev.D), 0.0, cases.W.TF, X, U, recs, outs,
                                 [0, 15, 16, B // 2, B - 1])
            assert e["defect"] < TOL_DEFECT and e["path"] < TOL_NODE and e["vals"] < TOL_NODE and e["cost"] < 1e-13, e
        else:
            for p, q in zip(outs, base):
                assert torch.equal(p, q), order
    ev.close()
