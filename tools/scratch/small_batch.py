import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import etol_amd as E
from etol_amd import workloads as W
M = 1024
for B in (1, 2, 4, 8, 16):
    for small in (96, 0):
        ev = E.Evaluator(0); ev.set_mesh(M, 0.0, W.TF); ev.set_model(E.MODEL_QUADROTOR2D, W.QUAD_PARAMS); ev.set_batch(B)
        ev.set_option("small_rows", small)
        X, U, recs = W.quadrotor_batch(3, B, M, 20); ev.set_path(recs, 0, 1)
        dX = torch.from_numpy(X).cuda(); dU = torch.from_numpy(U).cuda(); RES, VALS, COST = ev.alloc_outputs()
        for _ in range(20): ev.eval_dev(dX, dU, RES, VALS, COST)
        ev.synchronize(); t0 = time.perf_counter(); n = 3000
        for _ in range(n): ev.eval_dev(dX, dU, RES, VALS, COST)
        ev.synchronize(); el = time.perf_counter() - t0
        print(f"B={B:3d} small_rows={small:3d} fused={ev.uses_fused_kernel}: {1e6*el/n:7.1f} us/pass  {B*M*n/el:.3e} node-evals/s")
        ev.close()
