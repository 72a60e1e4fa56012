#!/bin/bash
# samples the shader clock / power while the bench pass loops (diagnostics: is the overlapped pass power-bound?)
# mode "overlap": node kernel || MFMA defect kernel (default pass); "node": the MFMA kernel emptied (EMI_SYM_ABLATE=7)
mkdir -p gpurun_out
for mode in overlap node; do
  if [ $mode = node ]; then export EMI_SYM_ABLATE=7; else export EMI_SYM_ABLATE=0; fi
  timeout -k 10 120 python bench.py --no-cpu-baseline --steps 40000 --warmup 50 > gpurun_out/clock_$mode.log 2>&1 &
  PID=$!
  sleep 9
  for i in 1 2 3 4; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk|fclk" | tr -s ' ' | tr '\n' '|'; echo " [$mode]"
    sleep 1.5
  done
  wait $PID
  tail -1 gpurun_out/clock_$mode.log | cut -c1-170
done
