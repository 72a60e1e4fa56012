import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import etol_amd as E
from etol_amd import workloads as W
M = 1024
for B in (256, 1024):
    ev = E.Evaluator(0)
    ev.set_mesh(M, 0.0, W.TF); ev.set_model(E.MODEL_QUADROTOR2D, W.QUAD_PARAMS); ev.set_batch(B)
    X, U, recs = W.quadrotor_batch(3, B, M, 20)
    ev.set_path(recs, 0, 1)
    ev.set_option("overlap_mode", 3)
    dX, dU = torch.from_numpy(X).cuda(), torch.from_numpy(U).cuda()
    base = None
    for order in (-1, 0, -1, 0, 1, 125):
        ev.set_option("pass_order", order)
        outs = ev.alloc_outputs()
        for t in outs: t.fill_(float("nan"))
        ev.eval_dev(dX, dU, *outs); ev.synchronize(); torch.cuda.synchronize()
        nans = [int(torch.isnan(t).sum().item()) for t in outs]
        if base is None: base = outs
        diffs = []
        for p, q in zip(outs, base):
            d = (p != q) & ~(torch.isnan(p) & torch.isnan(q))
            idx = d.nonzero()
            diffs.append((int(d.sum().item()), idx[:3].tolist(), [(p[tuple(i)].item(), q[tuple(i)].item()) for i in idx[:3]]))
        print(B, order, ev.last_defect_kernel, "nans", nans, "diffs", diffs, flush=True)
    ev.close()
