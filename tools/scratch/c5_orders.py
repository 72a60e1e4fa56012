import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import etol_amd as E
from etol_amd import workloads as W
M, B = 4096, 256
ev = E.Evaluator(0, f32=True); ev.set_mesh(M, 0.0, 20.0); ev.set_model(E.MODEL_FIXEDWING12, W.FW_PARAMS); ev.set_batch(B)
X, U = W.fixedwing_batch(4, 16, M)
X, U = np.tile(X, (16, 1, 1)), np.tile(U, (16, 1, 1))
dX, dU = torch.from_numpy(X).to("cuda", torch.float32), torch.from_numpy(U).to("cuda", torch.float32)
outs = ev.alloc_outputs(); torch.cuda.synchronize()
forms = [("sequential", dict(overlap_mode=1)), ("one interleaved", dict(overlap_mode=3, pass_order=0)), ("one mfma first", dict(overlap_mode=3, pass_order=1)),
         ("one 150", dict(overlap_mode=3, pass_order=150)), ("one 200", dict(overlap_mode=3, pass_order=200)), ("two streams", dict(overlap_mode=2))]
for r in range(3):
    for name, opts in forms:
        for k, v in {**dict(overlap_mode=0, pass_order=-1), **opts}.items(): ev.set_option(k, v)
        for _ in range(5): ev.eval_dev(dX, dU, *outs)
        ev.synchronize(); t0 = time.perf_counter()
        for _ in range(40): ev.eval_dev(dX, dU, *outs)
        ev.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / 40
        print(f"round {r} {name:18s} {ms:.4f} ms  {ev.last_defect_kernel[:40]}", flush=True)
