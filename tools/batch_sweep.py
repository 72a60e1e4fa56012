#!/usr/bin/env python3
"""Throughput of one evaluation pass over batch size and configuration (SURVEY.md section 8d:
B in {1, 64, 1024, 16384}; configs 2, 3 and the f32 arithmetic of config 5).  Writes
gpurun_out/batch_sweep.json.  Run on the GPU box: python tools/batch_sweep.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import etol_amd as E
from etol_amd import workloads as W


def run(name, model, params, M, B, n_obs, f32=False, overlap=True, steps=None):
    ev = E.Evaluator(0, f32=f32)
    ev.set_mesh(M, 0.0, W.TF)
    ev.set_model(model, params)
    ev.set_batch(B)
    if not overlap:
        ev.set_option("overlap", 0)
    gen = min(B, 64)      # generate 64 distinct instances and tile them: the kernels do not care
    if model == E.MODEL_QUADROTOR2D:
        X, U, recs = W.quadrotor_batch(3, gen, M, n_obs)
    else:
        X, U = W.fixedwing_batch(4, gen, M)
        recs = None
    reps = (B + gen - 1) // gen
    X = np.tile(X, (reps, 1, 1))[:B]
    U = np.tile(U, (reps, 1, 1))[:B]
    if n_obs:
        ev.set_path(np.tile(recs, (reps, 1, 1))[:B], 0, 1)
    dt = torch.float32 if f32 else torch.float64
    dX = torch.from_numpy(X).to("cuda", dt)
    dU = torch.from_numpy(U).to("cuda", dt)
    RES, VALS, COST = ev.alloc_outputs()
    work = B * M
    steps = steps or max(20, min(2000, int(2e8 / work)))
    for _ in range(15):
        ev.eval_dev(dX, dU, RES, VALS, COST)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ev.eval_dev(dX, dU, RES, VALS, COST)
    ev.synchronize()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out = dict(config=name, M=M, B=B, path_rows=n_obs, dtype="f32" if f32 else "f64", steps=steps,
               path="overlapped" if (overlap and ev.uses_fused_kernel) else "sequential",
               ms_per_pass=1e3 * el / steps, node_evals_per_s=work * steps / el)
    ev.close()
    del dX, dU, RES, VALS, COST
    torch.cuda.empty_cache()
    return out


def main():
    rows = []
    for B in (1, 64, 128, 256, 512, 1024, 16384):
        rows.append(run("c3 quadrotor N=1024 +20 keep-outs", E.MODEL_QUADROTOR2D, W.QUAD_PARAMS, 1024, B, 20))
        rows.append(run("c3 quadrotor N=1024 +20 keep-outs", E.MODEL_QUADROTOR2D, W.QUAD_PARAMS, 1024, B, 20, overlap=False))
        print(json.dumps(rows[-2]), flush=True)
        print(json.dumps(rows[-1]), flush=True)
    for B in (64, 1024, 16384):
        rows.append(run("c2 quadrotor N=256", E.MODEL_QUADROTOR2D, W.QUAD_PARAMS, 256, B, 0))
        print(json.dumps(rows[-1]), flush=True)
    rows.append(run("c4 shard: 128 scenarios of c3 per GPU", E.MODEL_QUADROTOR2D, W.QUAD_PARAMS, 1024, 128, 20))
    print(json.dumps(rows[-1]), flush=True)
    for B in (16, 256):
        rows.append(run("c5 fixed wing N=4096 (f32, shifted-difference D.X on f32 MFMA)", E.MODEL_FIXEDWING12,
                        W.FW_PARAMS, 4096, B, 0, f32=True, steps=40))
        print(json.dumps(rows[-1]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "batch_sweep.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
