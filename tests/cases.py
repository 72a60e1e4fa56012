"""Seeded parity cases shared by the GPU tests, smoke() and the golden generator."""
import numpy as np

from etol_amd import _lib as L
from etol_amd import workloads as W

# the shipped 2-D problem: resource/configs/ocp_2d_ex1.xml (values re-typed by hand)
OCP2D = dict(
    nsteps=32, dt=0.5,
    exz=[[(3.20, 2.50), (3.40, 2.60), (3.50, 3.40), (3.30, 3.00), (3.10, 3.50)],
         [(2.20, 2.50), (2.40, 2.60), (2.50, 3.40), (2.10, 3.50)]],
    tracks=[dict(radius=0.5, t=[0.0, 32.0], x=[1.51, 2.00], y=[2.00, 2.00]),
            dict(radius=0.5, t=[0.0, 32.0], x=[1.00, 1.00], y=[4.00, 3.00])],
)


def ocp2d_tables(edge_ellipse, track_centres, node_t):
    """Path records + track tables of the shipped problem, in the reference's row order:
    every polygon edge (etol_psopt_example1.cpp:159-186), then every track (:232-249)."""
    recs = []
    for poly in OCP2D["exz"]:
        n = len(poly)
        for i in range(n):
            (xa, ya), (xb, yb) = poly[i], poly[(i + 1) % n]
            recs.append(edge_ellipse(xa, ya, xb, yb))
    tx, ty = [], []
    for ti, trk in enumerate(OCP2D["tracks"]):
        xc, yc = track_centres(trk["t"], trk["x"], trk["y"], node_t)
        tx.append(xc)
        ty.append(yc)
        r = np.zeros(L.PATH_REC)
        r[0], r[1], r[2] = L.PATH_TRACK, ti, trk["radius"] ** 2
        recs.append(r)
    return np.array(recs), np.array(tx), np.array(ty)


def case_inputs(name):
    """-> dict(model, params, M, B, t0, tf, X, U, recs?, ocp2d?)"""
    if name == "pointmass_xml":       # C1 shape: M=33 (odd), 9 ellipse rows + 2 track rows
        M, B = 33, 2
        X, U = W.pointmass_batch(0, B, M)
        return dict(model=L.MODEL_POINTMASS2D, params=[], M=M, B=B, t0=0.0, tf=16.0, X=X, U=U, ocp2d=True)
    if name == "quad_256":            # C2
        M, B = 256, 3
        X, U, _ = W.quadrotor_batch(1, B, M, 0)
        return dict(model=L.MODEL_QUADROTOR2D, params=W.QUAD_PARAMS, M=M, B=B, t0=0.0, tf=W.TF, X=X, U=U)
    if name == "quad_1024_obs":       # C3, per-instance obstacle fields (C4 layout)
        M, B = 1024, 2
        X, U, recs = W.quadrotor_batch(2, B, M, 20)
        return dict(model=L.MODEL_QUADROTOR2D, params=W.QUAD_PARAMS, M=M, B=B, t0=0.0, tf=W.TF, X=X, U=U, recs=recs)
    if name == "quad_ragged":         # M and B off every tile size, one shared obstacle set
        M, B = 50, 19
        X, U, recs = W.quadrotor_batch(7, B, M, 3)
        return dict(model=L.MODEL_QUADROTOR2D, params=W.QUAD_PARAMS, M=M, B=B, t0=0.5, tf=9.0, X=X, U=U, recs=recs[:1])
    if name == "quad_tiny":           # smallest mesh
        M, B = 2, 1
        X, U, _ = W.quadrotor_batch(8, B, M, 0)
        return dict(model=L.MODEL_QUADROTOR2D, params=W.QUAD_PARAMS, M=M, B=B, t0=0.0, tf=1.0, X=X, U=U)
    if name == "fixedwing_64":        # C5 model in f64
        M, B = 64, 2
        X, U = W.fixedwing_batch(4, B, M)
        return dict(model=L.MODEL_FIXEDWING12, params=W.FW_PARAMS, M=M, B=B, t0=0.0, tf=20.0, X=X, U=U)
    raise KeyError(name)


CASES = ["pointmass_xml", "quad_256", "quad_1024_obs", "quad_ragged", "quad_tiny", "fixedwing_64"]
