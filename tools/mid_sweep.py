#!/usr/bin/env python3
"""Pass time over the batch range for the forms the dispatch policy chooses between, in ONE process per call (interleaved
rounds, medians): python tools/mid_sweep.py --batches 192,256,... [--forms default,one_sw1,one_sw2,two_sw2,two_ring,two_sw6]
   writes gpurun_out/mid_sweep.jsonl (one line per batch and form)"""
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

FORMS = {
    "default": dict(),
    "one_sw1": dict(overlap_mode=3, sym_ct=7),
    "one_sw2": dict(overlap_mode=3, sym_ct=6),
    "one_sw2_first": dict(overlap_mode=3, sym_ct=6, pass_order=1),
    "one_sw1_inter": dict(overlap_mode=3, sym_ct=7, pass_order=0),
    "one_sw1_first": dict(overlap_mode=3, sym_ct=7, pass_order=1),
    "one_sw3": dict(overlap_mode=3, sym_ct=8),
    "one_sw1_first_plain": dict(overlap_mode=3, sym_ct=7, pass_order=1, node_store=0),
    "one_sw1_first_nt": dict(overlap_mode=3, sym_ct=7, pass_order=1, node_store=2),
    "one_sw1_inter_nt": dict(overlap_mode=3, sym_ct=7, pass_order=0, node_store=2),
    "one_sw2_inter_nt": dict(overlap_mode=3, sym_ct=6, pass_order=0, node_store=2),
    "one_sw2_f150_nt": dict(overlap_mode=3, sym_ct=6, pass_order=150, node_store=2),
    "one_sw3_first_nt": dict(overlap_mode=3, sym_ct=8, pass_order=1, node_store=2),
    "one_sw3_first": dict(overlap_mode=3, sym_ct=8, pass_order=1),
    "one_sw3_inter": dict(overlap_mode=3, sym_ct=8, pass_order=0),
    "one_sw1_ks2_first": dict(overlap_mode=3, sym_ct=7, pass_order=1, sym_ksplit=2),
    "one_sw1_ks4_first": dict(overlap_mode=3, sym_ct=7, pass_order=1, sym_ksplit=4),
    "one_sw2_ks4_first": dict(overlap_mode=3, sym_ct=6, pass_order=1, sym_ksplit=4),
    "one_sw1_ks2_inter": dict(overlap_mode=3, sym_ct=7, pass_order=0, sym_ksplit=2),
    "one_sw2_ks2_first": dict(overlap_mode=3, sym_ct=6, pass_order=1, sym_ksplit=2),
    "one_sw3_ks2_first": dict(overlap_mode=3, sym_ct=8, pass_order=1, sym_ksplit=2),
    "one_sw2_front150": dict(overlap_mode=3, sym_ct=6, pass_order=150),
    "one_sw2_front200": dict(overlap_mode=3, sym_ct=6, pass_order=200),
    "one_sw2_first_plain": dict(overlap_mode=3, sym_ct=6, pass_order=1, node_store=0),
    "one_sw2_first_nt": dict(overlap_mode=3, sym_ct=6, pass_order=1, node_store=2),
    "two_sw2": dict(overlap_mode=2, sym_ct=6),
    "two_sw1": dict(overlap_mode=2, sym_ct=7),
    "two_ring": dict(overlap_mode=2, sym_ct=3),
    "two_sw6": dict(overlap_mode=2, sym_ct=5),
    "noslice": dict(slice=0),
    "slice1024": dict(slice=1024),
    "slice2048": dict(slice=2048),
    "slice4096": dict(slice=4096),
    "one_sw2_plain": dict(overlap_mode=3, sym_ct=6, node_store=0),
    "one_sw2_nt": dict(overlap_mode=3, sym_ct=6, node_store=2),
    "one_sw1_nt": dict(overlap_mode=3, sym_ct=7, node_store=2),
    "one_sw2_plainorder": dict(overlap_mode=3, sym_ct=6, sym_cpart=-1),
    "one_sw2_cp1": dict(overlap_mode=3, sym_ct=6, sym_cpart=1),
    "one_sw2_cp2": dict(overlap_mode=3, sym_ct=6, sym_cpart=2),
    "one_sw2_cp4": dict(overlap_mode=3, sym_ct=6, sym_cpart=4),
    "one_sw2_cp8": dict(overlap_mode=3, sym_ct=6, sym_cpart=8),
    "one_sw6": dict(overlap_mode=3, sym_ct=5),
    "g2c2": dict(sym_gblk=2, sym_cx=2),
    "g2c2_noslice": dict(sym_gblk=2, sym_cx=2, slice=0),
    "g2c2_slice2048": dict(sym_gblk=2, sym_cx=2, slice=2048),
    "g2c2_slice512": dict(sym_gblk=2, sym_cx=2, slice=512),
    "g4c2": dict(sym_gblk=4, sym_cx=2),
    "g1c2": dict(sym_gblk=1, sym_cx=2),
    "g2c4": dict(sym_gblk=2, sym_cx=4),
    "slice512": dict(slice=512),
    "one_sw2_ntsc1": dict(overlap_mode=3, sym_ct=6, node_store=3),
    "one_sw2_sc1_cp1": dict(overlap_mode=3, sym_ct=6, node_store=1, sym_cpart=1),
    "one_sw2_ntsc1_cp1": dict(overlap_mode=3, sym_ct=6, node_store=3, sym_cpart=1),
    "one_sw1_sc1": dict(overlap_mode=3, sym_ct=7, node_store=1),
    "one_sw1_first_sc1": dict(overlap_mode=3, sym_ct=7, node_store=1, pass_order=1),
    "one_sw2_first_sc1": dict(overlap_mode=3, sym_ct=6, node_store=1, pass_order=1),
    "abl1_mfma_off": dict(overlap_mode=3, sym_ct=7, pass_order=1, sym_ablate=16),
    "abl1_node_off": dict(overlap_mode=3, sym_ct=7, pass_order=1, sym_ablate=32),
    "abl1_epi_off": dict(overlap_mode=3, sym_ct=7, pass_order=1, sym_ablate=4),
    "abl1_node_off_epi_off": dict(overlap_mode=3, sym_ct=7, pass_order=1, sym_ablate=36),
    "sw3_node_off": dict(overlap_mode=3, sym_ct=8, sym_ablate=32),
    "sw6_node_off": dict(overlap_mode=3, sym_ct=5, sym_ablate=32),
    "sw2_node_off_epi_off": dict(overlap_mode=3, sym_ct=6, sym_ablate=36),
    "sw3_node_off_epi_off": dict(overlap_mode=3, sym_ct=8, sym_ablate=36),
    "sw6_node_off_epi_off": dict(overlap_mode=3, sym_ct=5, sym_ablate=36),
    "sw1_node_off_epi_off": dict(overlap_mode=3, sym_ct=7, sym_ablate=36),
    "no_skinny": dict(small_rows=0),
    "po0": dict(pass_order=0),
    "po1": dict(pass_order=1),
    "po110": dict(pass_order=110),
    "po125": dict(pass_order=125),
    "po150": dict(pass_order=150),
    "po200": dict(pass_order=200),
    "po300": dict(pass_order=300),
    "po150_cp1": dict(pass_order=150, sym_cpart=1),
    "po150_cp2": dict(pass_order=150, sym_cpart=2),
    "po150_cp4": dict(pass_order=150, sym_cpart=4),
    "po0_cp2": dict(pass_order=0, sym_cpart=2),
    "abl_x": dict(overlap_mode=3, sym_ct=6, sym_ablate=8),
    "abl_panels": dict(overlap_mode=3, sym_ct=6, sym_ablate=64),
    "abl_epi": dict(overlap_mode=3, sym_ct=6, sym_ablate=4),
    "abl_x_panels": dict(overlap_mode=3, sym_ct=6, sym_ablate=72),
    "abl_x_panels_epi": dict(overlap_mode=3, sym_ct=6, sym_ablate=76),
    "abl_mfma_off": dict(overlap_mode=3, sym_ct=6, sym_ablate=16),
    "abl_node_off": dict(overlap_mode=3, sym_ct=6, sym_ablate=32),
    "one_sw2_nst4": dict(overlap_mode=3, sym_ct=6, sym_nst=4),
    "bk16": dict(sym_bk=16),
    "o100": dict(pass_order=100, sym_bk=16), "o105": dict(pass_order=105, sym_bk=16), "o110": dict(pass_order=110, sym_bk=16), "o115": dict(pass_order=115, sym_bk=16), "o120": dict(pass_order=120, sym_bk=16), "o130": dict(pass_order=130, sym_bk=16),
    "nt": dict(node_store=2), "plain": dict(node_store=0), "sc1": dict(node_store=1), "ntsc1": dict(node_store=3), "cp1": dict(sym_cpart=1), "cpm1": dict(sym_cpart=-1), "cp4": dict(sym_cpart=4),
    "nt_order150": dict(node_store=2, pass_order=150), "order150_only": dict(pass_order=150, sym_bk=16), "order0_only": dict(pass_order=0, sym_bk=16),
    "sw1_ks1": dict(overlap_mode=3, sym_ct=7, sym_ksplit=1), "sw1_ks2": dict(overlap_mode=3, sym_ct=7, sym_ksplit=2), "sw2_ks1": dict(overlap_mode=3, sym_ct=6, sym_ksplit=1),
    "slice1280": dict(slice=1280), "slice1536": dict(slice=1536), "ct1": dict(sym_ctc=1), "cp2_ct1": dict(sym_cpart=2, sym_ctc=1), "cp2": dict(sym_cpart=2),
    "order0": dict(pass_order=0), "order110": dict(pass_order=110), "order125": dict(pass_order=125), "order150": dict(pass_order=150), "order200": dict(pass_order=200),
    "order0_bk16": dict(pass_order=0, sym_bk=16), "order125_bk16": dict(pass_order=125, sym_bk=16), "order150_bk16": dict(pass_order=150, sym_bk=16),
    "order1_bk16": dict(pass_order=1, sym_bk=16),
    "hs2": dict(sym_hs=2),
    "sw1_hs2": dict(overlap_mode=3, sym_ct=7, sym_ksplit=1, sym_hs=2, sym_bk=8),
    "sw1_hs2_bk16": dict(overlap_mode=3, sym_ct=7, sym_ksplit=1, sym_hs=2, sym_bk=16),
    "sw2_hs2": dict(overlap_mode=3, sym_ct=6, sym_ksplit=1, sym_hs=2, sym_bk=8),
    "sw2_hs2_bk16": dict(overlap_mode=3, sym_ct=6, sym_ksplit=1, sym_hs=2, sym_bk=16),
    "sw2_hs2_bk16_node_off": dict(overlap_mode=3, sym_ct=6, sym_ksplit=1, sym_hs=2, sym_bk=16, sym_ablate=32),
    "sw1_hs2_node_off": dict(overlap_mode=3, sym_ct=7, sym_ksplit=1, sym_hs=2, sym_bk=8, sym_ablate=32),
    "sw2_hs2_node_off": dict(overlap_mode=3, sym_ct=6, sym_ksplit=1, sym_hs=2, sym_bk=8, sym_ablate=32),
    "ct2": dict(sym_ctc=2),
    "ct2_bk16": dict(sym_ctc=2, sym_bk=16),
    "ct2_bk8": dict(sym_ctc=2, sym_bk=8),
    "sw2_ct2_node_off": dict(overlap_mode=3, sym_ct=6, sym_ksplit=1, sym_ctc=2, sym_ablate=32),
    "sw2_ct2_bk16_node_off": dict(overlap_mode=3, sym_ct=6, sym_ksplit=1, sym_ctc=2, sym_bk=16, sym_ablate=32),
    "ct2_g2c2": dict(sym_ctc=2, sym_gblk=2, sym_cx=2),
    "ct2_g2c1": dict(sym_ctc=2, sym_gblk=2, sym_cx=1),
    "ct2_g4c1": dict(sym_ctc=2, sym_gblk=4, sym_cx=1),
    "sw1_bk16": dict(overlap_mode=3, sym_ct=7, sym_ksplit=1, sym_bk=16),
    "sw2_bk16": dict(overlap_mode=3, sym_ct=6, sym_ksplit=1, sym_bk=16),
    "sw1_bk16_first": dict(overlap_mode=3, sym_ct=7, sym_ksplit=1, sym_bk=16, pass_order=1),
    "sw2_bk16_first": dict(overlap_mode=3, sym_ct=6, sym_ksplit=1, sym_bk=16, pass_order=1),
    "sw1_bk16_inter": dict(overlap_mode=3, sym_ct=7, sym_ksplit=1, sym_bk=16, pass_order=0),
    "sw2_bk16_inter": dict(overlap_mode=3, sym_ct=6, sym_ksplit=1, sym_bk=16, pass_order=0),
    "sw1_bk16_node_off": dict(overlap_mode=3, sym_ct=7, sym_ksplit=1, sym_bk=16, sym_ablate=32),
    "sw1_bk8_node_off": dict(overlap_mode=3, sym_ct=7, sym_ksplit=1, sym_bk=8, sym_ablate=32),
    "sw2_bk16_node_off": dict(overlap_mode=3, sym_ct=6, sym_ksplit=1, sym_bk=16, sym_ablate=32),
    "sw2_bk8_node_off": dict(overlap_mode=3, sym_ct=6, sym_ksplit=1, sym_bk=8, sym_ablate=32),
    "sw2_bk16_g2c2": dict(overlap_mode=3, sym_ct=6, sym_bk=16, sym_gblk=2, sym_cx=2),
    "one_sw2_g2c2_nst4": dict(overlap_mode=3, sym_ct=6, sym_nst=4, sym_gblk=2, sym_cx=2),
    "one_sw2_sc1": dict(overlap_mode=3, sym_ct=6, node_store=1),
    "one_sw2_sc1_plainorder": dict(overlap_mode=3, sym_ct=6, node_store=1, sym_cpart=-1),
    "one_sw2_sc1_g2c2": dict(overlap_mode=3, sym_ct=6, node_store=1, sym_gblk=2, sym_cx=2),
    "one_sw2_sc1_g4c1": dict(overlap_mode=3, sym_ct=6, node_store=1, sym_gblk=4, sym_cx=1),
    "one_sw2_sc1_g4c2": dict(overlap_mode=3, sym_ct=6, node_store=1, sym_gblk=4, sym_cx=2),
    "one_sw2_sc1_g2c4": dict(overlap_mode=3, sym_ct=6, node_store=1, sym_gblk=2, sym_cx=4),
    "one_sw2_sc1_g8c2": dict(overlap_mode=3, sym_ct=6, node_store=1, sym_gblk=8, sym_cx=2),
    "one_sw2_g1c1": dict(overlap_mode=3, sym_ct=6, sym_gblk=1, sym_cx=1),
    "one_sw2_g1c2": dict(overlap_mode=3, sym_ct=6, sym_gblk=1, sym_cx=2),
    "one_sw2_g1c4": dict(overlap_mode=3, sym_ct=6, sym_gblk=1, sym_cx=4),
    "one_sw2_g1c8": dict(overlap_mode=3, sym_ct=6, sym_gblk=1, sym_cx=8),
    "one_sw2_g2c1": dict(overlap_mode=3, sym_ct=6, sym_gblk=2, sym_cx=1),
    "one_sw2_g2c2": dict(overlap_mode=3, sym_ct=6, sym_gblk=2, sym_cx=2),
    "one_sw2_g2c4": dict(overlap_mode=3, sym_ct=6, sym_gblk=2, sym_cx=4),
    "one_sw2_g2c8": dict(overlap_mode=3, sym_ct=6, sym_gblk=2, sym_cx=8),
    "one_sw2_g4c1": dict(overlap_mode=3, sym_ct=6, sym_gblk=4, sym_cx=1),
    "one_sw2_g4c2": dict(overlap_mode=3, sym_ct=6, sym_gblk=4, sym_cx=2),
    "one_sw2_g4c4": dict(overlap_mode=3, sym_ct=6, sym_gblk=4, sym_cx=4),
    "one_sw2_g4c8": dict(overlap_mode=3, sym_ct=6, sym_gblk=4, sym_cx=8),
    "one_sw2_g8c1": dict(overlap_mode=3, sym_ct=6, sym_gblk=8, sym_cx=1),
    "one_sw2_g8c2": dict(overlap_mode=3, sym_ct=6, sym_gblk=8, sym_cx=2),
    "one_sw2_g8c4": dict(overlap_mode=3, sym_ct=6, sym_gblk=8, sym_cx=4),
    "one_sw2_g8c8": dict(overlap_mode=3, sym_ct=6, sym_gblk=8, sym_cx=8),
    "one_sw1_g2c2": dict(overlap_mode=3, sym_ct=7, sym_gblk=2, sym_cx=2),
    "one_sw1_g2c4": dict(overlap_mode=3, sym_ct=7, sym_gblk=2, sym_cx=4),
    "one_sw1_g4c2": dict(overlap_mode=3, sym_ct=7, sym_gblk=4, sym_cx=2),
    "one_sw1_g4c4": dict(overlap_mode=3, sym_ct=7, sym_gblk=4, sym_cx=4),
    "one_sw3_cp4": dict(overlap_mode=3, sym_ct=8, sym_cpart=4),
}
RESET = dict(small_rows=24, overlap_mode=0, sym_ct=0, pass_order=-1, slice=0, node_store=-1, sym_cpart=0, sym_gblk=0, sym_cx=0, sym_nst=3, sym_ksplit=0, sym_ablate=0, sym_bk=0, sym_ctc=0, sym_hs=0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="256,512")
    ap.add_argument("--forms", default="default,one_sw1,one_sw2,two_sw2,two_ring")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--ms", type=float, default=60.0, help="timed region per form and round, milliseconds")
    ap.add_argument("--nodes", type=int, default=1024)
    ap.add_argument("--touch", action="store_true", help="read the inputs of a pass with a torch reduction just before it (same stream)")
    ap.add_argument("--rotate", type=int, default=1, help="cycle over this many copies of the inputs (67 MB each at 1024 instances): "
                    "from 4 copies on they no longer stay in the 256 MB Infinity Cache from one pass to the next")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "mid_sweep.jsonl"))
    a = ap.parse_args()
    M = a.nodes
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    for B in [int(x) for x in a.batches.split(",")]:
        ev = E.Evaluator(0)
        ev.set_mesh(M, 0.0, W.TF)
        ev.set_model(E.MODEL_QUADROTOR2D, W.QUAD_PARAMS)
        ev.set_batch(B)
        gen = min(B, 64)
        X, U, recs = W.quadrotor_batch(3, gen, M, 20)
        reps = (B + gen - 1) // gen
        X, U, recs = np.tile(X, (reps, 1, 1))[:B], np.tile(U, (reps, 1, 1))[:B], np.tile(recs, (reps, 1, 1))[:B]
        ev.set_path(recs, 0, 1)
        ins = [(torch.from_numpy(X).cuda(), torch.from_numpy(U).cuda()) for _ in range(max(1, a.rotate))]
        dX, dU = ins[0]
        outs = ev.alloc_outputs()
        if a.touch:
            ev.use_stream(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        forms = [f for f in a.forms.split(",") if f]
        times = {f: [] for f in forms}
        names = {}
        t_end = time.perf_counter() + 0.3
        while time.perf_counter() < t_end:          # clocks up
            ev.eval_dev(dX, dU, *outs)
        ev.synchronize()
        for r in range(a.rounds):
            for f in forms:
                for k, v in {**RESET, **FORMS[f]}.items():
                    ev.set_option(k, v)
                for _ in range(5):
                    ev.eval_dev(dX, dU, *outs)
                ev.synchronize()
                est = times[f][-1] if times[f] else 0.3e-3 * max(B, 64) / 1024
                steps = max(10, int(a.ms * 1e-3 / est))
                t0 = time.perf_counter()
                for q in range(steps):
                    if a.touch:
                        x_, u_ = ins[q % len(ins)]
                        x_.sum(); u_.sum()
                    ev.eval_dev(*ins[q % len(ins)], *outs)
                ev.synchronize()
                times[f].append((time.perf_counter() - t0) / steps)
                names[f] = ev.last_defect_kernel
        with open(a.out, "a") as fo:
            for f in forms:
                ms = 1e3 * float(np.median(times[f]))
                rec = dict(B=B, M=M, form=f, rotate=a.rotate, touch=a.touch, ms_per_pass=round(ms, 5), node_evals_per_s=float("%.4g" % (B * M / (ms * 1e-3))),
                           kernel=names[f], rounds=a.rounds)
                fo.write(json.dumps(rec) + "\n")
                print(f"B={B:6d} {f:16s} {ms:9.4f} ms  {rec['node_evals_per_s']:.3e}/s  {names[f][:60]}", flush=True)
        ev.close()
        del dX, dU, outs, ins
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
