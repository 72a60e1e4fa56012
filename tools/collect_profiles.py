#!/usr/bin/env python3
"""Copy what tools/gpu_round.sh (and the A/B scripts) left in gpurun_out/ into profiles/ under this round's names.
  python tools/collect_profiles.py r02"""
import collections
import glob
import json
import os
import shutil
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(R, "gpurun_out")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"


def med(path, with_kernels=False):
    rows = [json.loads(l) for l in open(path) if l.strip()]
    d = collections.OrderedDict()
    for r in rows:
        d.setdefault((r["B"], r["variant"]), []).append(r)
    out = []
    for (B, v), rs in d.items():
        e = dict(B=B, variant=v, rounds=len(rs), ms_per_pass=round(float(np.median([x["ms_per_pass"] for x in rs])), 5))
        e["node_evals_per_s"] = float("%.4g" % (B * rs[0]["M"] / (e["ms_per_pass"] * 1e-3)))
        if with_kernels:
            e["node_ms"] = round(float(np.median([x["node_ms"] for x in rs])), 5)
            e["mfma_ms"] = round(float(np.median([x["mfma_ms"] for x in rs])), 5)
        out.append(e)
    return out


pv = {}
for key, name, wk in (("with_kernel_events", "pass_variants.jsonl", True), ("wall_only", "pass_variants_noprof.jsonl", False),
                      ("large_batches_one_launch_vs_two_streams", "pv_large.jsonl", False),
                      ("tile_order_column_partitions", "pv_cpart.jsonl", False), ("mid_batches", "pv_mid.jsonl", False),
                      ("ring_depth", "pv_nst.jsonl", False)):
    p = os.path.join(G, name)
    if os.path.exists(p):
        pv[key] = med(p, wk)
pv["note"] = ("tools/pass_variants.py on one MI355X per section (sections come from different boxes: compare within a section only), one "
              "process per batch size, interleaved rounds, medians; 6-state quadrotor, 1024 nodes, 20 keep-outs.  with_kernel_events: "
              "level-1 profiling (per-kernel HIP events; 'mfma_ms' of a one-launch variant is the pass kernel); wall_only: no events.  "
              "Variants are emi_set_option settings (tools/pass_variants.py VARIANTS); 'default' is what the library chooses by itself.  "
              "large_batches / tile_order / mid_batches / ring_depth: tools/gpu_large.sh, gpu_cpart.sh, gpu_mid.sh, gpu_nst.sh, measured "
              "while the kernels were being changed (large_batches before the tile order was partitioned and before the epilogue was "
              "specialised; tile_order before small batches went back to the plain order).")
json.dump(pv, open(os.path.join(R, "profiles", f"{tag}_pass_variants.json"), "w"), indent=1)
for src, dst in (("pmc_traffic.json", "pmc_traffic.json"), ("pmc_busy.json", "pmc_busy.json"), ("batch_sweep.json", "batch_sweep.json"),
                 ("probes.jsonl", "probes.jsonl"), ("small_batch_anatomy.json", "small_batch_anatomy.json")):
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(R, "profiles", f"{tag}_{dst}"))
for d, dst in (("prof", "kernel_stats.csv"), ("prof_c5", "c5_kernel_stats.csv")):
    ks = sorted(glob.glob(os.path.join(G, d, "*", "*kernel_stats.csv")), key=os.path.getmtime)
    if ks:
        shutil.copy(ks[-1], os.path.join(R, "profiles", f"{tag}_{dst}"))
with open(os.path.join(R, "profiles", f"{tag}_bench_lines.jsonl"), "w") as f:
    for n in ("bench.log", "bench_c5.log", "bench_b128.log", "bench_b256.log", "bench_b512.log", "bench2.log"):
        p = os.path.join(G, n)
        if os.path.exists(p):
            l = [x for x in open(p) if x.startswith('{"metric"')]
            if l:
                f.write(l[-1])
print(open(os.path.join(R, "profiles", f"{tag}_kernel_stats.csv")).read()[:500])
