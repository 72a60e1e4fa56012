#!/usr/bin/env python3
"""ONE piece of the device Newton step repeated, for a per-kernel anatomy under rocprofv3:
   rocprofv3 --kernel-trace --stats -d out -- python tools/factor_anatomy.py --nodes 1024 --what factor --reps 20
   (what: factor | lowrank | refined).  Prints the wall time per call; the kernel statistics divided by reps (+1 warm-up call, +1
   factorisation for lowrank / refined) are the kernels of one call."""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch  # noqa: F401

import etol_amd as E
from etol_amd import _lib as L
from etol_amd import workloads as W
from kkt_times import problem


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=1024)
    ap.add_argument("--what", default="factor")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--lowrank", type=int, default=300)
    a = ap.parse_args()
    lib = L.load()
    M, ns, nv = a.nodes, 6, 8
    N = (2 * ns + 2) * M
    rng = np.random.default_rng(5)
    D = C.POINTER(C.c_double)
    ev = E.Evaluator(0)
    ev.set_mesh(M, 0.0, 4.0)
    ev.set_model(1, W.QUAD_PARAMS)
    ev.set_batch(1)
    prob = problem(ev, M, ns, nv, rng)
    r = min(a.lowrank, M)
    node = np.sort(rng.choice(M, size=r, replace=False)).astype(np.int32)
    vec = rng.standard_normal((r, nv)) * 0.2
    delta = np.full(r, 0.5)
    rhs = rng.standard_normal(N)
    w = rhs.copy()
    rel, nsv, rev, stat = C.c_double(), C.c_int(), C.c_int(), C.c_int()

    def refined():
        w[:] = rhs
        lib.emi_kkt_solve_refined(ev.ctx, w.ctypes.data_as(D), 1e-9, 8, C.byref(rel), C.byref(nsv), C.byref(rev), C.byref(stat))

    fn = {"factor": lambda: ev.kkt_factor(*prob, dc=1e-9), "lowrank": lambda: ev.kkt_lowrank(node, vec, delta), "refined": refined}[a.what]
    ev.kkt_factor(*prob, dc=1e-9)
    fn()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        fn()
    print(f"{a.what} at {M} nodes: {1e3 * (time.perf_counter() - t0) / a.reps:.3f} ms per call over {a.reps} calls"
          + (f" ({nsv.value} solves)" if a.what == "refined" else ""), flush=True)
    ev.close()


if __name__ == "__main__":
    main()
