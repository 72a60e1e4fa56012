"""what the FIRST emi_kkt_factor of a context costs (rocBLAS handle, workspace allocations) against the following ones, for
three contexts created one after the other in one process"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import etol_amd as E
from etol_amd import workloads as W
ns, nv, nh = 6, 8, 36
for rep in range(3):
    t0 = time.perf_counter()
    ev = E.Evaluator(0)
    t_create = time.perf_counter() - t0
    for M in (33, 65, 129):
        ev.set_mesh(M, 0.0, 4.0); ev.set_model(1, W.QUAD_PARAMS); ev.set_batch(1)
        rng = np.random.default_rng(1)
        A = rng.standard_normal((M, nv, nv)); Q = A @ A.transpose(0, 2, 1) + nv * np.eye(nv)
        Qblk = np.zeros((nh, M))
        for v in range(nv):
            for q in range(v + 1): Qblk[v * (v + 1) // 2 + q] = Q[:, v, q]
        Jblk = rng.standard_normal((ns * nv, M))
        for i in range(ns): Jblk[i * nv + i] += np.diag(ev.D)
        fixed = np.zeros(nv * M, dtype=np.uint8); fixed[np.arange(ns) * M] = 1
        ts = []
        for k in range(4):
            t0 = time.perf_counter(); ev.kkt_factor(Qblk, Jblk, fixed, 1e-9); ts.append(1e3 * (time.perf_counter() - t0))
        print(f"context {rep} (create {1e3*t_create:.1f} ms) M={M}: factor calls {ts[0]:.1f} {ts[1]:.1f} {ts[2]:.1f} {ts[3]:.1f} ms", flush=True)
    t0 = time.perf_counter(); ev.close(); print(f"   close {1e3*(time.perf_counter()-t0):.1f} ms")
