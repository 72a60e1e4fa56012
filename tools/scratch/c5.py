import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import etol_amd as E
from etol_amd import workloads as W
M = 4096
for B in (16, 64, 256):
    ev = E.Evaluator(0, f32=True); ev.set_mesh(M, 0.0, 20.0); ev.set_model(E.MODEL_FIXEDWING12, W.FW_PARAMS); ev.set_batch(B)
    X, U = W.fixedwing_batch(4, min(B, 16), M)
    X = np.tile(X, (B // min(B, 16), 1, 1)); U = np.tile(U, (B // min(B, 16), 1, 1))
    dX = torch.from_numpy(X).to("cuda", torch.float32); dU = torch.from_numpy(U).to("cuda", torch.float32)
    RES, VALS, COST = ev.alloc_outputs()
    for _ in range(5): ev.eval_dev(dX, dU, RES, VALS, COST)
    ev.synchronize()
    for rep in range(3):
        t0 = time.perf_counter(); n = 20
        for _ in range(n): ev.eval_dev(dX, dU, RES, VALS, COST)
        ev.synchronize(); el = time.perf_counter() - t0
        print(f"c5 B={B}: {1e3*el/n:.3f} ms/pass  {B*M*n/el:.3e} node-evals/s")
    ev.close(); del dX, dU, RES, VALS, COST; torch.cuda.empty_cache()
