"""factor time of emi_kkt_factor at 1024 / 513 / 257 nodes with the blocked Cholesky (kkt_cholesky 1) and its two-level form (2); checks
that both give the same solution of one right-hand side"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import etol_amd as E
from etol_amd import workloads as W
ns, nv, nh = 6, 8, 36
for M in (1024, 512, 256):
    ev = E.Evaluator(0); ev.set_mesh(M, 0.0, 4.0); ev.set_model(1, W.QUAD_PARAMS); ev.set_batch(1)
    rng = np.random.default_rng(1)
    A = rng.standard_normal((M, nv, nv)); Q = A @ A.transpose(0, 2, 1) + nv * np.eye(nv)
    Qblk = np.zeros((nh, M))
    for v in range(nv):
        for q in range(v + 1): Qblk[v * (v + 1) // 2 + q] = Q[:, v, q]
    Jblk = rng.standard_normal((ns * nv, M))
    for i in range(ns): Jblk[i * nv + i] += np.diag(ev.D)
    fixed = np.zeros(nv * M, dtype=np.uint8); fixed[np.arange(ns) * M] = 1
    rhs = rng.standard_normal((nv + ns) * M)
    sol = {}
    for mode, outer in ((1, 512), (2, 256), (2, 512), (2, 768), (2, 1024), (2, 512)):
        ev.set_option("kkt_cholesky", mode)
        ev.set_option("kkt_chol_outer", outer)
        ev.kkt_factor(Qblk, Jblk, fixed, 1e-9)
        t0 = time.perf_counter()
        for _ in range(5): ev.kkt_factor(Qblk, Jblk, fixed, 1e-9)
        tf = (time.perf_counter() - t0) / 5
        sol[mode] = ev.kkt_solve(rhs)
        print(f"M={M} kkt_cholesky={mode} outer={outer}: factor {1e3*tf:.2f} ms", flush=True)
    d = np.abs(sol[1] - sol[2]).max() / np.abs(sol[1]).max()
    print(f"M={M}: solutions differ by {d:.2e} relative", flush=True)
    ev.set_option("kkt_cholesky", 1)
    ev.close()
