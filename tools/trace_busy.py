#!/usr/bin/env python3
"""How busy the GPU was during a traced run: python tools/trace_busy.py <kernel_trace.csv> [--top 20] [--out summary.json]

Reads a rocprofv3 --kernel-trace CSV (streamed: such a file of a Monte-Carlo run has a million rows) and prints
  * the span of the trace, the UNION of the kernel intervals (time at least one kernel was running), the sum of the kernel
    durations and their ratio (average number of kernels in flight while the GPU was busy),
  * launches per second, hardware queues seen, busy time per queue,
  * the kernels with the largest total duration (count, total, average).
"""
import argparse
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace("emi::(anonymous namespace)::", "").replace("emi::", "")
    name = re.sub(r"\(.*", "", name)
    if name.startswith("Cijk_"):
        m = re.search(r"MT(\d+x\d+x\d+)", name)
        return "Tensile GEMM " + (m.group(1) if m else "")
    return name[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--top", type=int, default=20)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    iv = []
    per_kernel = defaultdict(lambda: [0, 0])
    per_queue = defaultdict(list)
    with open(a.trace, newline="") as f:
        for row in csv.DictReader(f):
            s, e = int(row["Start_Timestamp"]), int(row["End_Timestamp"])
            iv.append((s, e))
            k = per_kernel[short(row["Kernel_Name"])]
            k[0] += 1
            k[1] += e - s
            per_queue[row["Queue_Id"]].append((s, e))
    if not iv:
        sys.exit("no kernel rows")

    def union(v):
        v.sort()
        tot, cs, ce = 0, v[0][0], v[0][1]
        for s, e in v[1:]:
            if s > ce:
                tot += ce - cs
                cs, ce = s, e
            elif e > ce:
                ce = e
        return tot + ce - cs

    span = max(e for _, e in iv) - min(s for s, _ in iv)
    busy = union(iv)
    total = sum(e - s for s, e in iv)
    out = {
        "launches": len(iv), "span_s": span / 1e9, "busy_s": busy / 1e9, "busy_fraction": busy / span,
        "kernel_seconds": total / 1e9, "kernels_in_flight_while_busy": total / busy,
        "launches_per_s": len(iv) / (span / 1e9), "queues": len(per_queue),
        "queue_busy_s": {q: union(v) / 1e9 for q, v in sorted(per_queue.items())},
        "top": [{"kernel": n, "calls": c, "total_s": t / 1e9, "avg_us": t / c / 1e3, "share": t / total}
                for n, (c, t) in sorted(per_kernel.items(), key=lambda kv: -kv[1][1])[:a.top]],
    }
    print(f"launches {out['launches']}  span {out['span_s']:.3f} s  busy {out['busy_s']:.3f} s ({out['busy_fraction']:.2f})  "
          f"kernel seconds {out['kernel_seconds']:.3f}  in flight while busy {out['kernels_in_flight_while_busy']:.2f}  "
          f"{out['launches_per_s']:.0f} launches/s  {out['queues']} queues")
    print("queue busy s:", " ".join(f"{q}:{b:.2f}" for q, b in out["queue_busy_s"].items()))
    for t in out["top"]:
        print(f"  {t['calls']:8d}  {t['total_s']:8.3f} s  {t['avg_us']:8.1f} us  {100 * t['share']:5.1f} %  {t['kernel']}")
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
