"""The oracle and the host functions against REFERENCE-EXECUTED vectors (tests/golden/ref_*.json, made by
tests/golden/gen_ref_vectors.py from oracle/_ref/ref_vectors: the reference's own
src/Examples/Dymos/etol_dymos_example1.cpp callbacks and include/ETOL/TrajectoryOptimizer.hpp templates,
compiled where they lie).  CPU only; the GPU side of the same vectors is in test_gpu_parity.py."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-14   # relative to the scale of the compared quantity (a handful of roundings, same formulas)


def _load(name):
    return json.load(open(os.path.join(HERE, "golden", name)))


def wrapped(g):
    """what etol_dymos_example1.cpp wraps around a constraint value g: exp(g) - 1 (:219, :275)"""
    return np.expm1(g)


def dymos_tables(c, edge_ellipse, track_centres):
    """path records + track centres of a fixture case, rows in the reference's order: every polygon
    edge, then every track (etol_dymos_example1.cpp:66-75)"""
    node_t = np.array(c["node_t"])
    recs = [edge_ellipse(*e) for e in c["edges"]]
    tx, ty = [], []
    for ti, trk in enumerate(c["tracks"]):
        xc, yc = track_centres(trk["t"], trk["x"], trk["y"], node_t)
        tx.append(xc)
        ty.append(yc)
        r = np.zeros(8)
        r[0], r[1], r[2] = 2, ti, trk["radius"] ** 2       # PATH_TRACK
        recs.append(r)
    return np.array(recs), np.array(tx), np.array(ty)


def compare_with_reference(c, RES, VALS, h, ns=2, nc=2):
    """RES / VALS of an evaluation pass (layouts of include/emi355x.h) against the reference's outputs"""
    ref = {k: np.array(v) for k, v in c["ref"].items()}
    ne, nt = len(c["edges"]), len(c["tracks"])
    nv = ns + nc
    X, U = np.array(c["X"]), np.array(c["U"])
    # ellipse rows: value exp(g)-1, partials exp(exp(g)) dg (:244-246)
    g = RES[:, ns:ns + ne]
    assert np.abs(wrapped(g) - ref["obs"]).max() < TOL
    for d in range(2):
        ours = np.exp(np.exp(g)) * VALS[:, ns * nv + 2 * np.arange(ne) + d]
        assert np.abs(ours - ref["obs_p"][:, :, d]).max() < TOL * (1 + np.abs(ref["obs_p"]).max())
    # moving discs: value exp(g)-1, partials exp(g) dg (:297-298)
    g = RES[:, ns + ne:ns + ne + nt]
    assert np.abs(wrapped(g) - ref["saa"]).max() < TOL
    for d in range(2):
        ours = np.exp(g) * VALS[:, ns * nv + 2 * (ne + np.arange(nt)) + d]
        assert np.abs(ours - ref["saa_p"][:, :, d]).max() < TOL * (1 + np.abs(ref["saa_p"]).max())
    # dynamics partials: VALS[i*nv+v] = -h df_i/dz_v (+ D_kk on the diagonal, a state column: skip those)
    for i in range(ns):
        for v in range(ns, nv):
            assert np.abs(VALS[:, i * nv + v] + h * ref["F_p"][:, i, v]).max() < TOL * h
    return ref


@pytest.mark.parametrize("which", [0, 1])
def test_oracle_rows_match_reference_callbacks(which):
    c = _load("ref_dymos_ex1.json")["cases"][which]
    M, node_t = c["M"], np.array(c["node_t"])
    t0, tf = node_t[0], node_t[-1]
    h = (tf - t0) / 2
    tau, w, D = O.lgl(M)
    assert np.abs(t0 + h * (tau + 1) - node_t).max() < 1e-14 * tf
    recs, tx, ty = dymos_tables(c, O.edge_ellipse, O.track_centres)
    X, U = np.array(c["X"]), np.array(c["U"])
    RES, VALS, COST = O.evaluate(0, [], M, (tau, w, D), t0, tf, X, U, recs, (tx, ty))
    ref = compare_with_reference(c, RES, VALS, h)
    # track centres: bit for bit what the reference's header template and the example's own function return
    assert np.array_equal(tx, ref["centres_hdr"][:, 0]) and np.array_equal(ty, ref["centres_hdr"][:, 1])
    assert np.array_equal(tx, ref["centres_ex"][:, 0]) and np.array_equal(ty, ref["centres_ex"][:, 1])
    # integrand and dynamics values: cost = h sum w L;  defect = D.X - h F
    assert np.abs(COST - h * (ref["L"] * w).sum(axis=1)).max() < 1e-13 * np.abs(COST).max()
    DX = np.einsum("kj,bij->bik", D, X)
    assert np.abs((DX - RES[:, :2]) / h - ref["F"]).max() < 1e-11
    # cost gradient entries: h w dL/dz
    for v in range(4):
        assert np.abs(VALS[:, 8 + 2 * recs.shape[0] + v] - h * w * ref["L_p"][:, v]).max() < TOL * h


def test_edge_constants_match_the_tuples_the_reference_consumed():
    """orc_edge_ellipse / emi_edge_ellipse against the (xc, yc, radsq, tt) tuples fed to the reference run"""
    import etol_amd as E
    for c in _load("ref_dymos_ex1.json")["cases"]:
        for e, (xc, yc, radsq, tt) in zip(c["edges"], c["exz"]):
            for fn in (O.edge_ellipse, E.edge_ellipse):
                r = fn(*e)
                assert r[1] == xc and r[2] == yc and r[5] == radsq
                assert abs(r[3] - np.cos(tt)) < 4e-16 and abs(r[4] - np.sin(tt)) < 4e-16 and abs(r[6] - 0.2 * radsq) < 1e-17


def test_track_centres_bit_exact_with_reference_interpolation():
    import etol_amd as E
    for c in _load("ref_interp.json")["cases"]:
        q = np.array(c["query"])
        hdr, ex = np.array(c["header"]), np.array(c["example"])
        assert np.array_equal(hdr, ex)     # the reference's two implementations agree with each other
        for fn in (O.track_centres, E.track_centres):
            xc, yc = fn(c["t"], c["x"], c["y"], q)
            assert np.array_equal(xc, hdr[:, 0]) and np.array_equal(yc, hdr[:, 1])


def test_header_templates_match_reference(built):
    H = C.CDLL(os.path.join(os.path.dirname(HERE), "tests", "harness", "libetol_harness.so"))
    D, I = C.POINTER(C.c_double), C.POINTER(C.c_int)
    c = _load("ref_traj.json")
    tab = np.array(c["traj"])
    R, Cc = tab.shape[0], tab.shape[1] - 1
    idxs = np.array(c["idxs"], dtype=np.int32)
    sc, of = np.array(c["scale"]), np.array(c["offset"])
    ext, s2, o2 = np.zeros((R, 1 + len(idxs))), np.zeros((R, 1 + Cc)), np.zeros((R, 1 + Cc))
    H.harness_traj_templates(R, Cc, tab.ctypes.data_as(D), len(idxs), idxs.ctypes.data_as(I), len(sc), sc.ctypes.data_as(D),
                             len(of), of.ctypes.data_as(D), ext.ctypes.data_as(D), s2.ctypes.data_as(D), o2.ctypes.data_as(D))
    assert np.array_equal(ext, np.array(c["extract"]))
    assert np.array_equal(s2, np.array(c["scaled"]))
    assert np.array_equal(o2, np.array(c["offsetted"]))
    # and linear_interpolation of OUR header (include/ETOL/TrajectoryOptimizer.hpp) on the reference's vectors
    for ci in _load("ref_interp.json")["cases"]:
        q, tv, xv = np.array(ci["query"]), np.array(ci["t"]), np.array(ci["x"])
        out = np.zeros(len(q))
        H.harness_lin_interp(len(q), q.ctypes.data_as(D), len(tv), tv.ctypes.data_as(D), xv.ctypes.data_as(D), out.ctypes.data_as(D))
        assert np.array_equal(out, np.array(ci["header"])[:, 0])
