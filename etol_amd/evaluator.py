"""Host-side Python mirror of the eMI355X evaluator (thin wrapper over the C ABI).

PyTorch is used only as plumbing: device allocations (torch tensors whose
data_ptr() is handed to the C ABI) and torch.distributed.  All arithmetic runs
in the HIP kernels of libemi355x.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def lgl(M):
    """LGL nodes, weights and differentiation matrix (host, emi_lgl)."""
    lib = L.load()
    tau = np.empty(M)
    w = np.empty(M)
    D = np.empty((M, M))
    L.check(lib.emi_lgl(M, _dp(tau), _dp(w), _dp(D)), what="emi_lgl")
    return tau, w, D


def model_dims(model):
    lib = L.load()
    ns, nc, npar = C.c_int(), C.c_int(), C.c_int()
    L.check(lib.emi_model_dims(model, C.byref(ns), C.byref(nc), C.byref(npar)), what="emi_model_dims")
    return ns.value, nc.value, npar.value


def edge_ellipse(xa, ya, xb, yb):
    lib = L.load()
    rec = np.zeros(L.PATH_REC)
    L.check(lib.emi_edge_ellipse(xa, ya, xb, yb, _dp(rec)), what="emi_edge_ellipse")
    return rec


def track_centres(t, x, y, node_t):
    lib = L.load()
    t, x, y, node_t = (np.ascontiguousarray(v, dtype=np.float64) for v in (t, x, y, node_t))
    xc, yc = np.empty(len(node_t)), np.empty(len(node_t))
    L.check(lib.emi_track_centres(len(t), _dp(t), _dp(x), _dp(y), len(node_t), _dp(node_t), _dp(xc), _dp(yc)),
            what="emi_track_centres")
    return xc, yc


class Evaluator:
    """One libemi355x context: mesh + model + batch, evaluated on one GPU."""

    def __init__(self, device=0, f32=False):
        self.lib = L.load()
        self.ctx = C.c_void_p()
        self.f32 = bool(f32)
        create = self.lib.emi_create_f32 if f32 else self.lib.emi_create
        L.check(create(int(device), C.byref(self.ctx)), what="emi_create")
        self.device = torch.device("cuda", int(device))
        self.dtype = torch.float32 if f32 else torch.float64
        self._keep = []
        self._lay = None        # cached emi_get_layout / emi_get_delays answers: eval_dev is the timed call of bench.py and of any
        self._nd = None         # batched user, and two ctypes round trips per pass show at 14 - 20 us passes; reset by every set_*

    def _changed(self):
        self._lay = None
        self._nd = None

    def close(self):
        if self.ctx:
            self.lib.emi_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, st, what):
        L.check(st, self.ctx, what)

    # ---- problem definition -------------------------------------------------
    def set_mesh(self, M, t0, tf, mesh=None):
        tau, w, D = mesh if mesh is not None else lgl(M)
        self.tau, self.w, self.D = (np.ascontiguousarray(a, dtype=np.float64) for a in (tau, w, D))
        self._ck(self.lib.emi_set_mesh(self.ctx, M, _dp(self.tau), _dp(self.w), _dp(self.D), t0, tf), "emi_set_mesh")
        self._changed()
        self.node_t = t0 + (tf - t0) / 2.0 * (self.tau + 1.0)

    def set_model(self, model, params=(), maximize=False):
        p = np.ascontiguousarray(params, dtype=np.float64)
        self._ck(self.lib.emi_set_model(self.ctx, model, _dp(p) if p.size else None, p.size, int(maximize)),
                 "emi_set_model")
        self._changed()

    def set_model_source(self, struct_name, source, ns, nc, params=(), maximize=False, npath=0, path_vars=()):
        """Install a model given as the text of a model struct (compiled for gfx950 here).  npath rows traced from
        constraint callbacks depend on the node variables path_vars (ascending; states first, then controls)."""
        p = np.ascontiguousarray(params, dtype=np.float64)
        pv = np.ascontiguousarray(path_vars, dtype=np.int32)
        self._ck(self.lib.emi_set_model_source(self.ctx, struct_name.encode(), source.encode(), ns, nc, npath,
                                               pv.ctypes.data_as(C.POINTER(C.c_int)) if pv.size else None, pv.size,
                                               _dp(p) if p.size else None, p.size, int(maximize)),
                 "emi_set_model_source")
        self._changed()

    def set_batch(self, B):
        self._ck(self.lib.emi_set_batch(self.ctx, B), "emi_set_batch")
        self._changed()

    def set_delays(self, x_horizon, u_horizon, dt):
        """Delayed states / controls as extra inputs of the node functions (include/emi355x.h, emi_set_delays): the model is
        written on nc + (x_horizon - 1) ns + u_horizon nc controls, evaluations keep taking U[B][nc][M]."""
        self._ck(self.lib.emi_set_delays(self.ctx, int(x_horizon), int(u_horizon), float(dt)), "emi_set_delays")
        self._changed()

    @property
    def n_delayed(self):
        if self._nd is None:
            n = C.c_int()
            self._ck(self.lib.emi_get_delays(self.ctx, None, None, C.byref(n)), "emi_get_delays")
            self._nd = n.value
        return self._nd

    def set_path(self, recs, px=0, py=1):
        recs = np.ascontiguousarray(recs, dtype=np.float64)
        if recs.ndim == 2:
            recs = recs[None]
        nsets, npth = recs.shape[0], recs.shape[1]
        self._ck(self.lib.emi_set_path(self.ctx, npth, nsets, _dp(recs) if recs.size else None, px, py), "emi_set_path")
        self._changed()

    def set_tracks(self, xc, yc):
        xc = np.ascontiguousarray(xc, dtype=np.float64)
        yc = np.ascontiguousarray(yc, dtype=np.float64)
        if xc.ndim == 2:
            xc, yc = xc[None], yc[None]
        self._ck(self.lib.emi_set_tracks(self.ctx, xc.shape[1], xc.shape[0], _dp(xc), _dp(yc)), "emi_set_tracks")
        self._changed()

    def use_stream(self, stream_ptr):
        self._ck(self.lib.emi_set_stream(self.ctx, C.c_void_p(stream_ptr)), "emi_set_stream")

    @property
    def layout(self):
        if self._lay is None:
            lay = L.Layout()
            self._ck(self.lib.emi_get_layout(self.ctx, C.byref(lay)), "emi_get_layout")
            self._lay = lay
        return self._lay

    def jac_structure(self):
        lay = self.layout
        n = lay.nvals * lay.M
        rows, cols = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32)
        ip = C.POINTER(C.c_int)
        self._ck(self.lib.emi_jac_structure(self.ctx, rows.ctypes.data_as(ip), cols.ctypes.data_as(ip)), "emi_jac_structure")
        return rows, cols

    # ---- device-resident evaluation -----------------------------------------
    def alloc_outputs(self):
        lay = self.layout
        kw = dict(dtype=self.dtype, device=self.device)
        res = torch.empty((lay.B, lay.nres, lay.M), **kw)
        vals = torch.empty((lay.B, lay.nvals, lay.M), **kw)
        cost = torch.empty((lay.B,), **kw)
        return res, vals, cost

    def eval_dev(self, X, U, RES, VALS, COST, flags=L.EVAL_ALL):
        """X,U,RES,VALS,COST: contiguous torch tensors on this evaluator's device."""
        lay = self.layout
        for t, shape in ((X, (lay.B, lay.ns, lay.M)), (U, (lay.B, lay.nc - self.n_delayed, lay.M)),
                         (RES, (lay.B, lay.nres, lay.M)), (VALS, (lay.B, lay.nvals, lay.M)), (COST, (lay.B,))):
            if t is None:
                continue
            if tuple(t.shape) != shape or t.dtype != self.dtype or not t.is_contiguous() or t.device != self.device:
                raise ValueError(f"tensor {tuple(t.shape)} {t.dtype} {t.device} does not match layout {shape} {self.dtype}")
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        self._ck(self.lib.emi_eval_dev(self.ctx, ptr(X), ptr(U), ptr(RES), ptr(VALS), ptr(COST), flags), "emi_eval_dev")

    def hess_dev(self, X, U, lamF, lamC, sigma, H):
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        self._ck(self.lib.emi_hess_dev(self.ctx, ptr(X), ptr(U), ptr(lamF), ptr(lamC), float(sigma), ptr(H)), "emi_hess_dev")

    def synchronize(self):
        self._ck(self.lib.emi_synchronize(self.ctx), "emi_synchronize")

    # ---- host-buffer evaluation (numpy in / numpy out) ------------------------
    def eval_host(self, X, U, flags=L.EVAL_ALL, res_in=None):
        lay = self.layout
        X = np.ascontiguousarray(X, dtype=np.float64)
        U = np.ascontiguousarray(U, dtype=np.float64)
        assert X.shape == (lay.B, lay.ns, lay.M) and U.shape == (lay.B, lay.nc - self.n_delayed, lay.M)
        RES = np.zeros((lay.B, lay.nres, lay.M)) if res_in is None else np.ascontiguousarray(res_in, dtype=np.float64).copy()
        VALS = np.zeros((lay.B, lay.nvals, lay.M))
        COST = np.zeros(lay.B)
        self._ck(self.lib.emi_eval_host(self.ctx, _dp(X), _dp(U), _dp(RES), _dp(VALS), _dp(COST), flags), "emi_eval_host")
        return RES, VALS, COST

    def hess_host(self, X, U, lamF, lamC, sigma=1.0):
        lay = self.layout
        X, U, lamF = (np.ascontiguousarray(a, dtype=np.float64) for a in (X, U, lamF))
        lamC = np.ascontiguousarray(lamC if lamC is not None else np.zeros((lay.B, 0, lay.M)), dtype=np.float64)
        H = np.zeros((lay.B, lay.nhess, lay.M))
        self._ck(self.lib.emi_hess_host(self.ctx, _dp(X), _dp(U), _dp(lamF), _dp(lamC) if lamC.size else None,
                                         float(sigma), _dp(H)), "emi_hess_host")
        return H

    # ---- Newton step (KKT solve) on the device ----------------------------------
    def kkt_factor(self, Qblk, Jblk, fixed, dc=0.0):
        """Assemble and LU-factorise the KKT matrix of one instance; returns rocSOLVER's info (0 = ok)."""
        lay = self.layout
        nv = lay.ns + lay.nc
        Qblk = np.ascontiguousarray(Qblk, dtype=np.float64)
        Jblk = np.ascontiguousarray(Jblk, dtype=np.float64)
        fixed = np.ascontiguousarray(fixed, dtype=np.uint8)
        assert Qblk.shape == (lay.nhess, lay.M) and Jblk.shape == (lay.ns * nv, lay.M) and fixed.shape == (nv * lay.M,)
        info = C.c_int(-1)
        self._ck(self.lib.emi_kkt_factor(self.ctx, _dp(Qblk), _dp(Jblk), fixed.ctypes.data_as(C.POINTER(C.c_ubyte)),
                                         float(dc), C.byref(info)), "emi_kkt_factor")
        return info.value

    def kkt_lowrank(self, node, vec, delta):
        """K = K~ - sum delta_c u_c u_c^T; returns True iff K has the inertia of K~ (solves are then with K)."""
        node = np.ascontiguousarray(node, dtype=np.int32)
        vec = np.ascontiguousarray(vec, dtype=np.float64)
        delta = np.ascontiguousarray(delta, dtype=np.float64)
        exact = C.c_int(0)
        r = node.size
        self._ck(self.lib.emi_kkt_lowrank(self.ctx, r, node.ctypes.data_as(C.POINTER(C.c_int)) if r else None,
                                          _dp(vec) if r else None, _dp(delta) if r else None, C.byref(exact)), "emi_kkt_lowrank")
        return bool(exact.value)

    def kkt_solve(self, rhs):
        """rhs: [N] or [nrhs][N]; returns the solution(s) in the same shape."""
        rhs = np.ascontiguousarray(rhs, dtype=np.float64).copy()
        nrhs = 1 if rhs.ndim == 1 else rhs.shape[0]
        self._ck(self.lib.emi_kkt_solve(self.ctx, _dp(rhs), nrhs), "emi_kkt_solve")
        return rhs

    # ---- measurement -----------------------------------------------------------
    def timer_start(self):
        self._ck(self.lib.emi_timer_start(self.ctx), "emi_timer_start")

    def timer_stop(self):
        ms = C.c_float()
        self._ck(self.lib.emi_timer_stop(self.ctx, C.byref(ms)), "emi_timer_stop")
        return ms.value

    def profile(self, level):
        """0 off, 1 every bracket, 2 the defect (MFMA) kernel only, 3 the node kernel only (include/emi355x.h)"""
        self._ck(self.lib.emi_profile_enable(self.ctx, int(level)), "emi_profile_enable")

    def profile_read(self):
        nm, dm, fm = C.c_float(), C.c_float(), C.c_float()
        nl, dl, fl = C.c_int(), C.c_int(), C.c_int()
        self._ck(self.lib.emi_profile_read(self.ctx, C.byref(nm), C.byref(nl), C.byref(dm), C.byref(dl), C.byref(fm),
                                           C.byref(fl)), "emi_profile_read")
        return dict(node_ms=nm.value, node_launches=nl.value, defect_ms=dm.value, defect_launches=dl.value,
                    pass_ms=fm.value, overlapped_passes=fl.value)

    def set_option(self, name, value):
        self._ck(self.lib.emi_set_option(self.ctx, name.encode(), int(value)), "emi_set_option")

    def plan(self, B=None):
        """What the default dispatch does with a batch of B instances (default: the batch set): dict of emi_pass_plan_t."""
        p = L.PassPlan()
        self._ck(self.lib.emi_plan_pass(self.ctx, int(B if B is not None else self.layout.B), C.byref(p)), "emi_plan_pass")
        return {n: getattr(p, n) for n, _ in L.PassPlan._fields_}

    @property
    def last_defect_kernel(self):
        return self.lib.emi_last_defect_kernel(self.ctx).decode()

    @property
    def uses_fused_kernel(self):
        f = C.c_int()
        self._ck(self.lib.emi_last_path(self.ctx, C.byref(f)), "emi_last_path")
        return bool(f.value)
