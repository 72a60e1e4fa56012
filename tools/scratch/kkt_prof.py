"""10 factorisations + 30 single-right-hand-side solves of the 1024-node KKT system: run under rocprofv3 --kernel-trace --stats"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import etol_amd as E
from etol_amd import workloads as W
ns, nv, nh, M = 6, 8, 36, 1024
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
for _ in range(10): ev.kkt_factor(Qblk, Jblk, fixed, 1e-9)
for _ in range(30): ev.kkt_solve(rhs)
print("done", flush=True)
