import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import etol_amd as E
from etol_amd import workloads as W
M = 1024
for B in (1, 4):
    ev = E.Evaluator(0); ev.set_mesh(M, 0.0, W.TF); ev.set_model(E.MODEL_QUADROTOR2D, W.QUAD_PARAMS); ev.set_batch(B)
    X, U, recs = W.quadrotor_batch(3, B, M, 20); ev.set_path(recs, 0, 1)
    dX = torch.from_numpy(X).cuda(); dU = torch.from_numpy(U).cuda(); RES, VALS, COST = ev.alloc_outputs()
    for _ in range(20): ev.eval_dev(dX, dU, RES, VALS, COST)
    ev.synchronize()
    ev.profile(True)
    n = 200
    for _ in range(n): ev.eval_dev(dX, dU, RES, VALS, COST)
    p = ev.profile_read()
    print(B, {k: (v / n if k.endswith("_ms") else v) for k, v in p.items()})
    ev.close()
