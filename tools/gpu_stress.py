#!/usr/bin/env python3
"""Random shapes through the default dispatch of emi_eval_dev against the sequential general path ("overlap" 0, itself
checked against the oracle in tests/): whatever the policy picks -- one launch or two streams, state split, tile order,
slices of very large batches -- must give the same rows.  Inputs differ per instance, outputs are poisoned first.

  python tools/gpu_stress.py [--cases 60] [--seed 1]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import etol_amd as E
from etol_amd import workloads as W


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--big", action="store_true", help="a fixed list of large shapes (the one-launch pass for large batches, slices)")
    ap.add_argument("--traced", action="store_true", help="the quadrotor as a run-time compiled (traced) model instead of the built-in one")
    a = ap.parse_args()
    run(a)


def run(a):
    """a.cases, a.seed, a.big as the command line gives them; returns (worst relative difference, {path: count})"""
    rng = np.random.default_rng(a.seed)
    Ms = [128, 256, 384, 512, 640, 768, 896, 1024, 1152, 1280, 1536, 1792, 2048]
    worst = 0.0
    kinds = {}
    big = [(1024, 768), (1024, 1000), (1024, 1024), (1024, 1100), (1024, 2048), (1024, 2064), (1024, 2600), (512, 1536), (512, 2048),
           (2048, 400), (2048, 512), (256, 3072), (640, 1232), (1280, 640)]
    for case in range(len(big) if a.big else a.cases):
        M = int(rng.choice(Ms))
        cap = max(8, min(2600, int(1.2e6 // M)))                 # keep a case under ~2.4 GB of outputs (two sets)
        B = int(rng.integers(1, cap + 1)) if rng.random() < 0.7 else int(rng.choice([16, 128, 256, 384, 512, 768, 1024, 2048, 2064]))
        B = min(B, cap)
        if a.big:
            M, B = big[case]
        nobs = int(rng.choice([0, 3, 20]))
        model = E.MODEL_QUADROTOR2D if rng.random() < 0.8 else E.MODEL_POINTMASS2D
        ev = E.Evaluator(0)
        ev.set_mesh(M, 0.0, W.TF)
        if getattr(a, "traced", False):
            model = E.MODEL_QUADROTOR2D
        if model == E.MODEL_QUADROTOR2D and getattr(a, "traced", False):
            import ctypes as C
            h = C.CDLL(os.path.join(ROOT, "tests", "harness", "libetol_harness.so"))
            h.harness_traced_model_source.restype = C.c_char_p
            ev.set_model_source("TracedModel", h.harness_traced_model_source(0).decode(), 6, 2)
            X, U, recs = W.quadrotor_batch(int(rng.integers(1, 200)), min(B, 32), M, max(nobs, 1))
        elif model == E.MODEL_QUADROTOR2D:
            ev.set_model(model, W.QUAD_PARAMS)
            X, U, recs = W.quadrotor_batch(int(rng.integers(1, 200)), min(B, 32), M, max(nobs, 1))
        else:
            ev.set_model(model, [])
            X, U = W.pointmass_batch(int(rng.integers(1, 200)), min(B, 32), M)
            _, _, recs = W.quadrotor_batch(3, min(B, 32), M, max(nobs, 1))        # disc keep-outs on (x, y) of the point mass
        ev.set_batch(B)
        reps = (B + X.shape[0] - 1) // X.shape[0]
        X = np.tile(X, (reps, 1, 1))[:B] + 1e-3 * rng.standard_normal((B, 1, 1))
        U = np.tile(U, (reps, 1, 1))[:B]
        if nobs:
            ev.set_path(np.tile(recs, (reps, 1, 1))[:B][:, :nobs], 0, 1)
        dX, dU = torch.from_numpy(X).cuda(), torch.from_numpy(U).cuda()
        a1, a2 = ev.alloc_outputs(), ev.alloc_outputs()
        for t in a1 + a2:
            t.fill_(float("nan"))
        torch.cuda.synchronize()
        ev.eval_dev(dX, dU, *a1)
        kind = ev.last_defect_kernel if ev.uses_fused_kernel else "general path"
        ev.set_option("overlap", 0)
        ev.eval_dev(dX, dU, *a2)
        torch.cuda.synchronize()
        err = 0.0
        for p, q in zip(a1, a2):
            assert torch.isfinite(p).all() and torch.isfinite(q).all(), (case, M, B, kind)
            err = max(err, ((p - q).abs().max() / (q.abs().max() + 1.0)).item())
        worst = max(worst, err)
        kinds[kind.split("(")[0].strip()] = kinds.get(kind.split("(")[0].strip(), 0) + 1
        print(f"case {case:3d}  M {M:5d}  B {B:5d}  rows {nobs:2d}  model {model}  {kind[:70]:70s}  err {err:.2e}", flush=True)
        assert err < 1e-11, (case, M, B, kind, err)
        ev.close()
        del dX, dU, a1, a2
        torch.cuda.empty_cache()
    print("worst relative difference", worst)
    print("paths taken:", kinds)
    return worst, kinds


if __name__ == "__main__":
    main()
