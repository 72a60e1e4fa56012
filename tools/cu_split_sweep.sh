#!/bin/bash
# A/B of the overlapped pass: both kernels on every CU (0) vs disjoint CU sets (n CUs to the MFMA kernel)
mkdir -p gpurun_out
for n in 0 96 128 160 176 192 208 224 0; do
  EMI_CU_SPLIT=$n timeout -k 10 120 python bench.py --no-cpu-baseline --steps 300 --warmup 30 > gpurun_out/cusplit_$n.log 2>&1 || { echo "n=$n failed"; tail -3 gpurun_out/cusplit_$n.log; continue; }
  python - "$n" <<'PY'
import json, sys
n = sys.argv[1]
b = json.loads(open(f"gpurun_out/cusplit_{n}.log").read().strip().split("\n")[-1])
r = b["roofline"]
print(f"cu_split {n:>3}: {b['ms_per_step']:.4f} ms/pass  value {b['value']:.3e}  kernels {r.get('kernels_ms')}")
PY
done
