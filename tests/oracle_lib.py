"""ctypes access to the CPU oracle (oracle/liboracle.so, oracle/libepsopt_style.so).

Test infrastructure: imported only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.
"""
import ctypes as C
import os

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_D = C.POINTER(C.c_double)


def _dp(a):
    return a.ctypes.data_as(_D) if a is not None and a.size else None


_orc = None
_eps = None


def orc():
    global _orc
    if _orc is None:
        path = os.path.join(_ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make oracle`")
        _orc = C.CDLL(path)
        _orc.orc_lgl.argtypes = [C.c_int, _D, _D, _D]
        _orc.orc_edge_ellipse.argtypes = [C.c_double] * 4 + [_D]
        _orc.orc_edge_ellipse.restype = None
        _orc.orc_track_centres.argtypes = [C.c_int, _D, _D, _D, C.c_int, _D, _D, _D]
        _orc.orc_track_centres.restype = None
        _orc.orc_eval.argtypes = ([C.c_int, _D, C.c_int, C.c_int, C.c_int, _D, _D, _D, C.c_double, C.c_double,
                                   C.c_int, C.c_int, _D, C.c_int, C.c_int, C.c_int, C.c_int, _D, _D, _D, _D, _D, _D, _D])
        _orc.orc_hess.argtypes = ([C.c_int, _D, C.c_int, C.c_int, C.c_int, _D, C.c_double, C.c_double, C.c_int, C.c_int,
                                   _D, C.c_int, C.c_int, C.c_int, C.c_int, _D, _D, _D, _D, _D, _D, C.c_double, _D])
    return _orc


def eps():
    global _eps
    if _eps is None:
        path = os.path.join(_ROOT, "oracle", "libepsopt_style.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make oracle`")
        _eps = C.CDLL(path)
        _eps.eps_eval.argtypes = ([C.c_int, _D, C.c_int, C.c_int, C.c_int, _D, _D, _D, C.c_double, C.c_double,
                                   C.c_int, C.c_int, _D, C.c_int, C.c_int, C.c_int, C.c_int, _D, _D, _D, _D, _D, _D, _D, C.c_int])
    return _eps


def lgl(M):
    tau, w, D = np.empty(M), np.empty(M), np.empty((M, M))
    assert orc().orc_lgl(M, _dp(tau), _dp(w), _dp(D)) == 0
    return tau, w, D


def delay_matrix(M, tau, t0, tf, delay):
    """W[k][j]: value at t_k - delay (clamped to t0) of the Lagrange basis polynomial j of the nodes (emi_oracle.c)"""
    tau = np.ascontiguousarray(tau, dtype=np.float64)
    W = np.empty((M, M))
    orc().orc_delay_matrix.argtypes = [C.c_int, _D, C.c_double, C.c_double, C.c_double, _D]
    assert orc().orc_delay_matrix(M, _dp(tau), t0, tf, delay, _dp(W)) == 0
    return W


def edge_ellipse(xa, ya, xb, yb):
    rec = np.zeros(8)
    orc().orc_edge_ellipse(xa, ya, xb, yb, _dp(rec))
    return rec


def track_centres(t, x, y, node_t):
    t, x, y, node_t = (np.ascontiguousarray(v, dtype=np.float64) for v in (t, x, y, node_t))
    xc, yc = np.empty(len(node_t)), np.empty(len(node_t))
    orc().orc_track_centres(len(t), _dp(t), _dp(x), _dp(y), len(node_t), _dp(node_t), _dp(xc), _dp(yc))
    return xc, yc


MODEL_DIMS = {0: (2, 2), 1: (6, 2), 2: (12, 4), 3: (2, 8)}      # 3: delay demo (2 controls + 6 delayed values)


def _prep(model, params, M, recs, tracks, X, U):
    ns, nc = MODEL_DIMS[model]
    X = np.ascontiguousarray(X, dtype=np.float64)
    U = np.ascontiguousarray(U, dtype=np.float64)
    B = X.shape[0]
    assert X.shape == (B, ns, M) and U.shape == (B, nc, M)
    params = np.ascontiguousarray(params if params is not None else [], dtype=np.float64)
    if recs is None or np.size(recs) == 0:
        recs = np.zeros((1, 0, 8))
    recs = np.ascontiguousarray(recs, dtype=np.float64)
    if recs.ndim == 2:
        recs = recs[None]
    if tracks is None:
        tx = ty = np.zeros((1, 0, M))
    else:
        tx, ty = (np.ascontiguousarray(a, dtype=np.float64) for a in tracks)
        if tx.ndim == 2:
            tx, ty = tx[None], ty[None]
    return ns, nc, B, X, U, params, recs, tx, ty


def evaluate(model, params, M, mesh, t0, tf, X, U, recs=None, tracks=None, px=0, py=1, maximize=False,
             style="oracle", nthreads=0):
    """One evaluation pass on the CPU. style: 'oracle' (emi_oracle.c) or 'epsopt' (epsopt_style.cpp)."""
    tau, w, D = (np.ascontiguousarray(a, dtype=np.float64) for a in mesh)
    ns, nc, B, X, U, params, recs, tx, ty = _prep(model, params, M, recs, tracks, X, U)
    np_ = recs.shape[1]
    nv = ns + nc
    RES = np.zeros((B, ns + np_, M))
    VALS = np.zeros((B, ns * nv + 2 * np_ + nv, M))
    COST = np.zeros(B)
    args = [model, _dp(params), int(maximize), M, B, _dp(tau), _dp(w), _dp(D), t0, tf, np_, recs.shape[0],
            _dp(recs), px, py, tx.shape[1], tx.shape[0], _dp(tx), _dp(ty), _dp(X), _dp(U), _dp(RES), _dp(VALS), _dp(COST)]
    if style == "oracle":
        assert orc().orc_eval(*args) == 0
    else:
        assert eps().eps_eval(*args, int(nthreads)) == 0
    return RES, VALS, COST


def hessian(model, params, M, mesh, t0, tf, X, U, lamF, lamC, sigma=1.0, recs=None, tracks=None, px=0, py=1,
            maximize=False):
    tau, w, D = (np.ascontiguousarray(a, dtype=np.float64) for a in mesh)
    ns, nc, B, X, U, params, recs, tx, ty = _prep(model, params, M, recs, tracks, X, U)
    np_ = recs.shape[1]
    nv = ns + nc
    lamF = np.ascontiguousarray(lamF, dtype=np.float64)
    lamC = np.ascontiguousarray(lamC if lamC is not None else np.zeros((B, 0, M)), dtype=np.float64)
    H = np.zeros((B, nv * (nv + 1) // 2, M))
    assert orc().orc_hess(model, _dp(params), int(maximize), M, B, _dp(w), t0, tf, np_, recs.shape[0], _dp(recs),
                          px, py, tx.shape[1], tx.shape[0], _dp(tx), _dp(ty), _dp(X), _dp(U), _dp(lamF), _dp(lamC),
                          float(sigma), _dp(H)) == 0
    return H


def sampled_errors(model, params, M, mesh, t0, tf, X, U, recs, got, instances, maximize=False):
    """Compare device results of a LARGE batch with the oracle on a few sampled instances (the oracle takes ~10 ms per
    1024-node instance).  X, U, recs: host arrays of the whole batch (recs per instance or one shared set); got = (RES,
    VALS, COST) as arrays or torch tensors of the whole batch.  Returns the worst errors in the units of the parity
    tolerances of tests/test_gpu_parity.py: defect rows relative to sum_j |D_kj||x_j| + |row| + 1, node rows / VALS
    entries relative to the entry's largest magnitude + 1, COST relative."""
    ns = MODEL_DIMS[model][0]
    idx = sorted({int(i) for i in instances if 0 <= int(i) < X.shape[0]})
    Xs, Us = np.ascontiguousarray(X[idx]), np.ascontiguousarray(U[idx])
    rs = None
    if recs is not None and np.size(recs):
        recs = np.asarray(recs)
        rs = recs[idx] if recs.ndim == 3 and recs.shape[0] == X.shape[0] and recs.shape[0] > 1 else recs
    ref = evaluate(model, params, M, mesh, t0, tf, Xs, Us, rs, maximize=maximize)
    pick = lambda a: (a[idx].cpu().numpy() if hasattr(a, "cpu") else np.asarray(a)[idx])
    RES, VALS, COST = (pick(a) for a in got)
    absD = np.abs(np.asarray(mesh[2]))
    scale = np.einsum("kj,bij->bik", absD, np.abs(Xs)) + np.abs(ref[0][:, :ns]) + 1.0
    out = {"instances": idx,
           "defect": float((np.abs(RES[:, :ns] - ref[0][:, :ns]) / scale).max()),
           "path": float(np.abs(RES[:, ns:] - ref[0][:, ns:]).max() / (np.abs(ref[0][:, ns:]).max() + 1.0)) if RES.shape[1] > ns else 0.0,
           "vals": float(max(np.abs(VALS[:, e] - ref[1][:, e]).max() / (np.abs(ref[1][:, e]).max() + 1.0) for e in range(VALS.shape[1]))),
           "cost": float(np.abs(COST - ref[2]).max() / (np.abs(ref[2]).max() + 1.0))}
    out["max_rel_err"] = max(out["defect"], out["path"], out["vals"], out["cost"])
    return out
