#!/usr/bin/env python3
"""Where does the MFMA defect kernel spend its time on a small batch (the 128-instance shard of config 4)?
Stand-alone kernel time (HIP events around the kernel, overlap_mode 1 = back to back on one stream) of the ring2 forms
with parts switched off through the diagnostics option sym_ablate (results invalid then): 1 no MFMAs / fragment reads,
2 no operand DMA after the prologue, 4 no epilogue (f at the output nodes, stores); 7 = the loop skeleton alone.

  python tools/small_batch_anatomy.py [--batch 128] [--out gpurun_out/small_batch_anatomy.json]"""
import argparse
import json
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
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--nodes", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "small_batch_anatomy.json"))
    a = ap.parse_args()
    M, B = a.nodes, a.batch
    ev = E.Evaluator(0)
    ev.set_mesh(M, 0.0, W.TF)
    ev.set_model(E.MODEL_QUADROTOR2D, W.QUAD_PARAMS)
    ev.set_batch(B)
    gen = min(B, 64)
    X, U, recs = W.quadrotor_batch(3, gen, M, 20)
    reps = (B + gen - 1) // gen
    X, U = np.tile(X, (reps, 1, 1))[:B], np.tile(U, (reps, 1, 1))[:B]
    ev.set_path(np.tile(recs, (reps, 1, 1))[:B], 0, 1)
    dX, dU = torch.from_numpy(X).cuda(), torch.from_numpy(U).cuda()
    RES, VALS, COST = ev.alloc_outputs()
    rows = []
    forms = [("sw1", 7, 1), ("sw2", 6, 1), ("sw6", 5, 1), ("sw6_ks4", 5, 4), ("sw2_ks4", 6, 4), ("sw1_ks4", 7, 4)]
    for name, ct, ks in forms:
        for ab in (0, 1, 2, 3, 4, 5, 7):
            ev.set_option("overlap_mode", 1)
            ev.set_option("sym_ct", ct)
            ev.set_option("sym_ksplit", ks)
            ev.set_option("sym_ablate", ab)
            for _ in range(20):
                ev.eval_dev(dX, dU, RES, VALS, COST)
            torch.cuda.synchronize()
            ev.profile(2)
            for _ in range(a.steps):
                ev.eval_dev(dX, dU, RES, VALS, COST)
            torch.cuda.synchronize()
            p = ev.profile_read()
            ev.profile(False)
            us = 1e3 * p["defect_ms"] / max(p["defect_launches"], 1)
            rows.append(dict(form=name, B=B, M=M, ablate=ab, kernel=ev.last_defect_kernel, launches=p["defect_launches"], us_per_launch=us))
            print(f"{name:8s} ablate {ab}  {us:7.2f} us  ({p['defect_launches']} launches)  {ev.last_defect_kernel[:60]}")
    ev.set_option("sym_ablate", 0)
    json.dump(rows, open(a.out, "w"), indent=1)
    ev.close()


if __name__ == "__main__":
    main()
