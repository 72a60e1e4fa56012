#!/bin/bash
# kernel timeline of bench passes at a given per-GPU batch: where does a pass spend its time between kernels?
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for b in "$@"; do
  rm -rf $OUT/trace_b$b
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_b$b -- python $GRAFT_REPO_ROOT/bench.py --batch $b --steps 100 --warmup 10 --no-cpu-baseline > $OUT/trace_b$b.log 2>&1
  echo "B=$b rc=$?"; python $GRAFT_REPO_ROOT/tools/trace_gaps.py $OUT/trace_b$b/*/*kernel_trace.csv 40
done
