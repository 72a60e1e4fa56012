"""Why does the pass slow down from 4096 instances on (per instance)?  B = 1024 evaluator; rotate over k input sets and / or k
output sets per pass: inputs 67 MB each, outputs 1.09 GB each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import etol_amd as E
from etol_amd import workloads as W
M, B = 1024, 1024
ev = E.Evaluator(0); ev.set_mesh(M, 0.0, W.TF); ev.set_model(E.MODEL_QUADROTOR2D, W.QUAD_PARAMS); ev.set_batch(B)
X, U, recs = W.quadrotor_batch(3, 64, M, 20)
X, U, recs = np.tile(X, (16, 1, 1)), np.tile(U, (16, 1, 1)), np.tile(recs, (16, 1, 1))
ev.set_path(recs, 0, 1)
ins = [(torch.from_numpy(X).cuda() + i, torch.from_numpy(U).cuda()) for i in range(8)]
outs = [ev.alloc_outputs() for _ in range(8)]
torch.cuda.synchronize()
def run(ki, ko, steps=160):
    for s in range(16): ev.eval_dev(*ins[s % ki], *outs[s % ko])
    ev.synchronize(); t0 = time.perf_counter()
    for s in range(steps): ev.eval_dev(*ins[s % ki], *outs[s % ko])
    ev.synchronize(); return 1e3 * (time.perf_counter() - t0) / steps
for r in range(2):
    for ki, ko in ((1, 1), (2, 1), (4, 1), (8, 1), (1, 2), (1, 4), (1, 8), (4, 4), (8, 8)):
        print(f"inputs x{ki} ({67*ki} MB)  outputs x{ko} ({1.09*ko:.1f} GB): {run(ki, ko):.4f} ms per pass", flush=True)
