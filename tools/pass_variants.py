#!/usr/bin/env python3
"""A/B of kernel variants of the evaluation pass in ONE process on one device (interleaved rounds; rule 24 of the
HIP guide: never compare timings taken in different processes or on different devices).

  python tools/pass_variants.py [--batch 1024] [--rounds 5] [--steps 200] [--out gpurun_out/pass_variants.jsonl]

Each variant is a set of emi_set_option values; per variant and round: wall ms per pass over `steps` passes, and the
per-kernel HIP-event times (overlap_mode 1 runs the two kernels back to back on one stream = stand-alone times;
overlap_mode 2 runs them concurrently).  Correctness of every variant is checked once against variant 0."""
import argparse
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

VARIANTS = {
    "ring_bk16_concurrent": dict(sym_ct=3, overlap_mode=2, node_store=0),
    "ring_bk16_conc_nt": dict(sym_ct=3, overlap_mode=2, node_store=2),
    "ring2_sw6_conc_nt": dict(sym_ct=5, overlap_mode=2, node_store=2),
    "ring2_sw2_conc_nt": dict(sym_ct=6, overlap_mode=2, node_store=2),
    "ring2_sw1_conc_nt": dict(sym_ct=7, overlap_mode=2, node_store=2),
    "ring2_sw3_conc_nt": dict(sym_ct=8, overlap_mode=2, node_store=2),
    "sw6_ks2_conc": dict(sym_ct=5, sym_ksplit=2, overlap_mode=2, node_store=-1),
    "sw6_ks4_conc": dict(sym_ct=5, sym_ksplit=4, overlap_mode=2, node_store=-1),
    "sw6_ks8_conc": dict(sym_ct=5, sym_ksplit=8, overlap_mode=2, node_store=-1),
    "sw6_ks4_seq": dict(sym_ct=5, sym_ksplit=4, overlap_mode=1, node_store=-1),
    "sw1_conc": dict(sym_ct=7, overlap_mode=2, node_store=-1),
    "auto": dict(sym_ct=0, overlap_mode=2, node_store=-1),
    "default": dict(sym_ct=0, overlap_mode=0, node_store=-1),
    "default_cost_kernel": dict(sym_ct=0, overlap_mode=0, node_store=-1, cost_in_kernel=0),
    "one_launch_auto": dict(sym_ct=0, overlap_mode=3, node_store=-1),
    "one_launch_sw6": dict(sym_ct=5, overlap_mode=3, node_store=-1),
    "one_launch_sw3": dict(sym_ct=8, overlap_mode=3, node_store=-1),
    "one_launch_sw2": dict(sym_ct=6, overlap_mode=3, node_store=-1),
    "one_launch_sw1": dict(sym_ct=7, overlap_mode=3, node_store=-1),
    "one_launch_sw2_plain": dict(sym_ct=6, overlap_mode=3, node_store=0),
    "one_launch_sw1_nt": dict(sym_ct=7, overlap_mode=3, node_store=2),
    "one_launch_sw1_ks2": dict(sym_ct=7, overlap_mode=3, node_store=-1, sym_ksplit=2),
    "one_launch_sw2_ks2": dict(sym_ct=6, overlap_mode=3, node_store=-1, sym_ksplit=2),
    "one_launch_sw2_ks4": dict(sym_ct=6, overlap_mode=3, node_store=-1, sym_ksplit=4),
    "one_launch_sw3_ks2": dict(sym_ct=8, overlap_mode=3, node_store=-1, sym_ksplit=2),
    "one_launch_sw6_ks4": dict(sym_ct=5, overlap_mode=3, node_store=-1, sym_ksplit=4),
    "one_launch_sw1_ks1": dict(sym_ct=7, overlap_mode=3, node_store=-1, sym_ksplit=1),
    "sw2_ks2_conc_ticket": dict(sym_ct=6, sym_ksplit=2, overlap_mode=2, node_store=-1),
    "sw6_ks4_conc_kernel": dict(sym_ct=5, sym_ksplit=4, overlap_mode=2, node_store=-1, sym_combine=0),
    "one_launch_sw2_plain_order": dict(sym_ct=6, overlap_mode=3, node_store=-1, sym_cpart=-1),
    "one_launch_sw2_cpart2": dict(sym_ct=6, overlap_mode=3, node_store=-1, sym_cpart=2),
    "one_launch_sw2_cpart4": dict(sym_ct=6, overlap_mode=3, node_store=-1, sym_cpart=4),
    "one_launch_sw2_cpart8": dict(sym_ct=6, overlap_mode=3, node_store=-1, sym_cpart=8),
    "one_launch_sw3_cpart4": dict(sym_ct=8, overlap_mode=3, node_store=-1, sym_cpart=4),
    "one_launch_sw1_plain_order": dict(sym_ct=7, overlap_mode=3, node_store=-1, sym_cpart=-1),
    "one_launch_sw1_cpart4": dict(sym_ct=7, overlap_mode=3, node_store=-1, sym_cpart=4),
    "one_launch_sw1_cpart8": dict(sym_ct=7, overlap_mode=3, node_store=-1, sym_cpart=8),
    "sw2_conc_plain_order": dict(sym_ct=6, overlap_mode=2, node_store=2, sym_cpart=-1),
    "sw2_conc_cpart4": dict(sym_ct=6, overlap_mode=2, node_store=2, sym_cpart=4),
    "sw6_conc_cpart4": dict(sym_ct=5, overlap_mode=2, node_store=2, sym_cpart=4),
    "one_launch_sw2_front105": dict(sym_ct=6, overlap_mode=3, node_store=-1, pass_order=105),
    "one_launch_sw2_front110": dict(sym_ct=6, overlap_mode=3, node_store=-1, pass_order=110),
    "one_launch_sw2_front115": dict(sym_ct=6, overlap_mode=3, node_store=-1, pass_order=115),
    "one_launch_sw2_front125": dict(sym_ct=6, overlap_mode=3, node_store=-1, pass_order=125),
    "one_launch_sw1_interleaved": dict(sym_ct=7, overlap_mode=3, node_store=-1, pass_order=0),
    "one_launch_sw1_mfma_first": dict(sym_ct=7, overlap_mode=3, node_store=-1, pass_order=1),
    "one_launch_sw2_mfma_first": dict(sym_ct=6, overlap_mode=3, node_store=-1, pass_order=1),
    "one_launch_sw1_nst4": dict(sym_ct=7, overlap_mode=3, node_store=-1, sym_nst=4),
    "one_launch_sw2_nst4": dict(sym_ct=6, overlap_mode=3, node_store=-1, sym_nst=4),
    "one_launch_sw1_nst4_nt": dict(sym_ct=7, overlap_mode=3, node_store=2, sym_nst=4),
    "ring2_sw2_conc_plain": dict(sym_ct=6, overlap_mode=2, node_store=0),
    "ring2_sw1_conc_plain": dict(sym_ct=7, overlap_mode=2, node_store=0),
    "ring2_auto_conc_nt": dict(sym_ct=4, overlap_mode=2, node_store=2),
    "ring_bk16_seq_nt": dict(sym_ct=3, overlap_mode=1, node_store=2),
    "ring_bk16_sequential": dict(sym_ct=3, overlap_mode=1),
    "ring2_auto_seq_nt": dict(sym_ct=4, overlap_mode=1, node_store=2),
    "ring2_auto_sequential": dict(sym_ct=4, overlap_mode=1),
    "ring2_sw6_sequential": dict(sym_ct=5, overlap_mode=1),
    "ring2_sw2_sequential": dict(sym_ct=6, overlap_mode=1),
    "ring2_sw1_sequential": dict(sym_ct=7, overlap_mode=1),
    "ring2_sw3_sequential": dict(sym_ct=8, overlap_mode=1),
    "general_sequential": dict(overlap=0),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--nodes", type=int, default=1024)
    ap.add_argument("--obstacles", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--only", default="")
    ap.add_argument("--no-profile", action="store_true", help="no per-kernel events in the timed loop (wall time only)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "pass_variants.jsonl"))
    a = ap.parse_args()
    M, B = a.nodes, a.batch
    names = [n for n in VARIANTS if not a.only or n in a.only.split(",")]
    ev = E.Evaluator(0)
    ev.set_mesh(M, 0.0, W.TF)
    ev.set_model(E.MODEL_QUADROTOR2D, W.QUAD_PARAMS)
    ev.set_batch(B)
    gen = min(B, 64)
    X, U, recs = W.quadrotor_batch(3, gen, M, a.obstacles)
    reps = (B + gen - 1) // gen
    X, U = np.tile(X, (reps, 1, 1))[:B], np.tile(U, (reps, 1, 1))[:B]
    if a.obstacles:
        ev.set_path(np.tile(recs, (reps, 1, 1))[:B], 0, 1)
    dX, dU = torch.from_numpy(X).cuda(), torch.from_numpy(U).cuda()
    RES, VALS, COST = ev.alloc_outputs()

    def apply(opts):
        ev.set_option("overlap", 1)
        ev.set_option("overlap_mode", 2)
        ev.set_option("node_store", 0)
        ev.set_option("sym_ksplit", 0)
        ev.set_option("sym_nst", 3)
        ev.set_option("cost_in_kernel", 1)
        ev.set_option("sym_combine", 1)
        ev.set_option("sym_cpart", 0)
        ev.set_option("pass_order", -1)
        for k, v in opts.items():
            ev.set_option(k, v)

    ref = None
    for n in names:                                     # correctness of every variant against the first
        apply(VARIANTS[n])
        RES.zero_(); VALS.zero_(); COST.zero_()
        torch.cuda.synchronize()      # torch's stream and the evaluator's stream are different streams
        ev.eval_dev(dX, dU, RES, VALS, COST)
        torch.cuda.synchronize()
        got = (RES.clone(), VALS.clone(), COST.clone())
        if ref is None:
            ref = got
        else:
            scale = ref[0].abs().max().item()
            err = max((got[i] - ref[i]).abs().max().item() for i in range(3))
            assert err <= 1e-9 * scale, (n, err, scale)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    out = open(a.out, "a")
    for n in names:                                     # warm-up of every variant
        apply(VARIANTS[n])
        for _ in range(20):
            ev.eval_dev(dX, dU, RES, VALS, COST)
    torch.cuda.synchronize()
    for r in range(a.rounds):
        for n in names:
            apply(VARIANTS[n])
            for _ in range(5):
                ev.eval_dev(dX, dU, RES, VALS, COST)
            torch.cuda.synchronize()
            ev.profile(not a.no_profile)
            t0 = time.perf_counter()
            for _ in range(a.steps):
                ev.eval_dev(dX, dU, RES, VALS, COST)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            p = ev.profile_read()
            ev.profile(False)
            rec = dict(variant=n, B=B, M=M, round=r, ms_per_pass=1e3 * el / a.steps,
                       node_evals_per_s=B * M * a.steps / el,
                       node_ms=p["node_ms"] / max(p["node_launches"], 1), mfma_ms=p["defect_ms"] / max(p["defect_launches"], 1),
                       pass_ms=p["pass_ms"] / max(p["overlapped_passes"], 1))
            out.write(json.dumps(rec) + "\n")
            out.flush()
    out.close()
    # summary: median over rounds
    rows = [json.loads(l) for l in open(a.out) if l.strip()]
    print(f"B={B} M={M}")
    for n in names:
        rs = [x for x in rows if x["variant"] == n and x["B"] == B and x["M"] == M]
        med = lambda k: float(np.median([x[k] for x in rs]))
        print(f"  {n:26s} pass {med('ms_per_pass'):.4f} ms  ({med('node_evals_per_s'):.3e}/s)  node {med('node_ms'):.4f}  mfma {med('mfma_ms'):.4f}")
    ev.close()


if __name__ == "__main__":
    main()
