#!/bin/bash
# shader clock and power of the config-3 one-launch pass with one role switched off (sym_ablate 16: MFMA role returns at once,
# 32: node role returns at once): which role draws the package to its power limit?
mkdir -p gpurun_out
for ab in 0 16 32; do
  EMI_SYM_ABLATE=$ab timeout -k 10 120 python bench.py --no-cpu-baseline --steps 60000 --warmup 5 > gpurun_out/clock_role_$ab.log 2>&1 &
  PID=$!
  sleep 11
  for i in 1 2 3; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power \(W\)" | tr -s ' ' | tr '\n' '|'; echo " [ablate $ab]"
    sleep 1
  done
  wait $PID
  tail -1 gpurun_out/clock_role_$ab.log | python -c "
import sys,json; d=json.loads(sys.stdin.read(),strict=False); print('ablate $ab ms_per_step', round(d['ms_per_step'],4))"
done
