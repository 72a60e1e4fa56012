#!/bin/bash
# shader clock and power while the config-5 pass (f32 MFMA defect kernel) and the config-3 pass loop: what is the matrix peak
# at the clock the chip actually holds?
mkdir -p gpurun_out
for cfg in c5 c3; do
  timeout -k 10 120 python bench.py --config $cfg --no-cpu-baseline --steps $([ $cfg = c5 ] && echo 12000 || echo 60000) --warmup 5 > gpurun_out/clock_$cfg.log 2>&1 &
  PID=$!
  sleep 12
  for i in 1 2 3 4 5; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr -s ' ' | tr '\n' '|'; echo " [$cfg]"
    sleep 1
  done
  wait $PID
  tail -1 gpurun_out/clock_$cfg.log | cut -c1-200
done
