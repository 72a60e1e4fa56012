#!/bin/bash
mkdir -p gpurun_out
for ab in 0 1 2 3; do
  echo "== ablate $ab"
  EMI_ABLATE=$ab timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_ms'])" || exit 1
done
