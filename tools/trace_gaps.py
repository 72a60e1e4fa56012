#!/usr/bin/env python3
"""Timeline of one evaluation pass from a rocprofv3 --kernel-trace CSV: per pass (one emi_nodes_kernel each) the span
from the first kernel's start to the last kernel's end, each kernel's duration, and how much of the span no kernel
was running.  usage: trace_gaps.py <kernel_trace.csv> [skip_passes]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows if "emi_" in r["Kernel_Name"]))
short = lambda n: next((k for k in ("emi_nodes_kernel", "emi_symdefect_ring2", "emi_symdefect_ring_f64", "emi_symdefect_combine",
                                    "emi_cost_finish", "emi_defect_f32_mfma", "emi_defect_f64") if k in n), n[:30])
# a pass = kernels between consecutive emi_cost_finish ends
passes, cur = [], []
for s, e, n in ev:
    cur.append((s, e, short(n)))
    if "emi_cost_finish" in n:
        passes.append(cur)
        cur = []
passes = passes[skip:]
if not passes:
    sys.exit("no passes found")
tot = defaultdict(float)
span_sum = busy_sum = gap_prev = 0.0
prev_end = None
for p in passes:
    s0, e1 = min(x[0] for x in p), max(x[1] for x in p)
    span_sum += e1 - s0
    # union of busy intervals
    iv = sorted((x[0], x[1]) for x in p)
    b, cs, ce = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce:
            b += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    b += ce - cs
    busy_sum += b
    for s, e, n in p:
        tot[n] += e - s
    if prev_end is not None:
        gap_prev += max(0, s0 - prev_end)
    prev_end = e1
n = len(passes)
print(f"{n} passes: span {span_sum / n / 1e3:.2f} us, some kernel running {busy_sum / n / 1e3:.2f} us, "
      f"idle inside the span {(span_sum - busy_sum) / n / 1e3:.2f} us, gap to the next pass {gap_prev / max(n - 1, 1) / 1e3:.2f} us")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"  {k:28s} {v / n / 1e3:8.2f} us per pass")
