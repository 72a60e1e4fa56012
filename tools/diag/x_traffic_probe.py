import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import etol_amd as E
from etol_amd import workloads as W
M, B = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ev = E.Evaluator(0); ev.set_mesh(M, 0.0, W.TF); ev.set_model(E.MODEL_QUADROTOR2D, W.QUAD_PARAMS); ev.set_batch(B)
X, U, recs = W.quadrotor_batch(3, 64, M, 20)
reps = (B + 63) // 64
X, U = np.tile(X, (reps, 1, 1))[:B], np.tile(U, (reps, 1, 1))[:B]
ev.set_path(np.tile(recs, (reps, 1, 1))[:B], 0, 1)
dX, dU = torch.from_numpy(X).cuda(), torch.from_numpy(U).cuda()
outs = ev.alloc_outputs()
def run(opts, n=300):
    for k, v in opts.items(): ev.set_option(k, v)
    for _ in range(30): ev.eval_dev(dX, dU, *outs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): ev.eval_dev(dX, dU, *outs)
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
for r in range(3):
    for name, o in [("one launch sw2", dict(overlap_mode=3, sym_ct=6, sym_ablate=0)),
                    ("  X -> 16 instances", dict(overlap_mode=3, sym_ct=6, sym_ablate=8)),
                    ("  De/Do -> first 64 rows", dict(overlap_mode=3, sym_ct=6, sym_ablate=64)),
                    ("  both", dict(overlap_mode=3, sym_ct=6, sym_ablate=72)),
                    ("  no epilogue stores", dict(overlap_mode=3, sym_ct=6, sym_ablate=4)),
                    ("  node role only", dict(overlap_mode=3, sym_ct=6, sym_ablate=16)),
                    ("  MFMA role only", dict(overlap_mode=3, sym_ct=6, sym_ablate=32)),
                    ("  MFMA role only, no epilogue stores", dict(overlap_mode=3, sym_ct=6, sym_ablate=36)),
                    ("  neither role (launch + dispatch)", dict(overlap_mode=3, sym_ct=6, sym_ablate=48)),
                    ("two streams sw2", dict(overlap_mode=2, sym_ct=6, sym_ablate=0)),
                    ("sequential sw2", dict(overlap_mode=1, sym_ct=6, sym_ablate=0))]:
        print(f"B={B} {name:40s} {run(o):.4f} ms")
