"""The CPU oracle against the golden fixtures (tests/golden, made by gen_golden.py).  No GPU needed."""
import json
import os

import numpy as np

import oracle_lib as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODELS = json.load(open(os.path.join(GOLD, "models.json")))
NS = {0: 2, 1: 6, 2: 12}


def _one_node_eval(model, pt, params, recs=None, style="oracle"):
    """Evaluate at a single state/control point: put it at every node of a 3-node mesh with D = 0,
    w = 1, h = 1 so that RES = -f, VALS = -J (+0), cost gradient = dL."""
    M = 3
    ns = NS[model]
    z = np.array(pt["z"])
    X = np.repeat(z[:ns, None], M, 1)[None]
    U = np.repeat(z[ns:, None], M, 1)[None]
    mesh = (np.array([-1.0, 0, 1]), np.ones(M), np.zeros((M, M)))
    return O.evaluate(model, params, M, mesh, 0.0, 2.0, X, U, recs, style=style)


def test_lgl_oracle_against_golden():
    g = json.load(open(os.path.join(GOLD, "lgl.json")))
    for M in (3, 4, 5, 9, 33):
        tau, w, D = O.lgl(M)
        assert np.abs(tau - g[str(M)]["tau"]).max() < 2e-16
        assert np.abs(w - g[str(M)]["w"]).max() < 1e-15
        assert np.abs(D - np.array(g[str(M)]["D"])).max() < 1e-12 * M * M
    c4 = g["closed"]["4"]
    tau, w, _ = O.lgl(4)
    assert np.abs(tau - c4["tau"]).max() < 2e-16 and np.abs(w - c4["w"]).max() < 1e-15


def test_models_against_sympy():
    for style in ("oracle", "epsopt"):
        for model in (0, 1, 2):
            params = MODELS[str(model)]["params"]
            ns = NS[model]
            for pt in MODELS[str(model)]["points"]:
                RES, VALS, COST = _one_node_eval(model, pt, params, style=style)
                nv = len(pt["z"])
                f = -RES[0, :ns, 1]
                J = -VALS[0, :ns * nv, 1].reshape(ns, nv)
                gL = VALS[0, ns * nv:, 1]
                assert np.abs(f - pt["f"]).max() < 1e-13 * (np.abs(pt["f"]).max() + 1)
                assert np.abs(J - np.array(pt["J"])).max() < 1e-13 * (np.abs(pt["J"]).max() + 1)
                assert np.abs(gL - np.array(pt["gL"])).max() < 1e-13 * (np.abs(pt["gL"]).max() + 1)
                assert abs(COST[0] - 3 * pt["L"]) < 1e-12 * (abs(pt["L"]) + 1)


def test_keepout_rows_against_sympy_and_reference_partials():
    for r in MODELS["keepout"]:
        recs = np.zeros((2, 8))
        recs[0] = [0, r["xc"], r["yc"], r["ct"], r["st"], r["asq"], r["bsq"], 0]
        recs[1] = [1, r["xc"], r["yc"], r["rsq"], 0, 0, 0, 0]
        pt = dict(z=[r["x"], r["y"], 0.1, 0.2])
        RES, VALS, _ = _one_node_eval(0, pt, [], recs)
        ell = [RES[0, 2, 1], VALS[0, 8, 1], VALS[0, 9, 1]]
        disc = [RES[0, 3, 1], VALS[0, 10, 1], VALS[0, 11, 1]]
        assert np.abs(np.array(ell) - r["ell"][:3]).max() < 1e-14 * (np.abs(r["ell"][:3]).max() + 1)
        assert np.abs(np.array(disc) - r["disc"][:3]).max() < 1e-14 * (np.abs(r["disc"][:3]).max() + 1)
        # the reference's own analytic partials of the ellipse row
        # (src/Examples/Dymos/etol_dymos_example1.cpp:239-240)
        dx, dy = r["x"] - r["xc"], r["y"] - r["yc"]
        delx, dely = r["ct"] * dx - r["st"] * dy, r["st"] * dx + r["ct"] * dy
        ref_dx = -2.0 * (r["bsq"] * delx * r["ct"] + r["asq"] * dely * r["st"])
        ref_dy = -2.0 * (-r["bsq"] * delx * r["st"] + r["asq"] * dely * r["ct"])
        assert abs(ell[1] - ref_dx) < 1e-14 * (abs(ref_dx) + 1) and abs(ell[2] - ref_dy) < 1e-14 * (abs(ref_dy) + 1)


def test_oracle_hessian_against_sympy():
    for model in (0, 1, 2):
        params = MODELS[str(model)]["params"]
        ns = NS[model]
        for pt in MODELS[str(model)]["points"][:3]:
            M = 3
            z = np.array(pt["z"])
            X = np.repeat(z[:ns, None], M, 1)[None]
            U = np.repeat(z[ns:, None], M, 1)[None]
            mesh = (np.array([-1.0, 0, 1]), np.ones(M), np.zeros((M, M)))
            # orc_hess: cL = sigma*h*w = sigma (h = w = 1), cf_i = -h*lamF_i = -lamF_i
            lamF = -np.repeat(np.array(pt["cf"])[:, None], M, 1)[None]
            H = O.hessian(model, params, M, mesh, 0.0, 2.0, X, U, lamF, None, sigma=pt["cL"])
            ref = np.array(pt["H"])
            assert np.abs(H[0, :, 1] - ref).max() < 2e-7 * (np.abs(ref).max() + 1)


def test_analytic_optimum_of_obstacle_free_problem():
    """Config 1 without keep-outs is a convex QP with a closed-form optimum: the oracle's defect is
    zero there and its quadrature returns the closed-form cost (pins defect + cost conventions)."""
    g = json.load(open(os.path.join(GOLD, "ocp2d.json")))
    M = 33
    mesh = O.lgl(M)
    t = g["tf"] / 2 * (mesh[0] + 1)
    X = np.array([g["x0"][0] + g["u"][0] * t, g["x0"][1] + g["u"][1] * t])[None]
    U = np.array([np.full(M, g["u"][0]), np.full(M, g["u"][1])])[None]
    for style in ("oracle", "epsopt"):
        RES, VALS, COST = O.evaluate(0, [], M, mesh, 0.0, g["tf"], X, U, style=style)
        assert np.abs(RES).max() < 1e-11
        assert abs(COST[0] - g["cost"]) < 1e-13
        assert abs(X[0, 0, -1] - g["xf"][0]) < 1e-14
    # maximise flips the sign of cost and cost gradient only
    _, Vm, Cm = O.evaluate(0, [], M, mesh, 0.0, g["tf"], X, U, maximize=True)
    assert abs(Cm[0] + g["cost"]) < 1e-13 and np.allclose(Vm[0, 8:], -VALS[0, 8:]) and np.allclose(Vm[0, :8], VALS[0, :8])
