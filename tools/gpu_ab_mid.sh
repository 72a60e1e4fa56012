#!/bin/bash
# the default pass over a few batch sizes with the main build and every build under _alt/ on this box (tools/ab_build.sh)
# usage: bash tools/gpu_ab_mid.sh "<batches, comma separated>" [forms]
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
for d in . $(ls -d _alt/*/ 2>/dev/null); do
  echo "== build $d"
  (cd $R/$d && timeout -k 10 300 python tools/mid_sweep.py --batches $1 --forms ${2:-default} --rounds 5 --ms 50 2>&1 | grep "^B=" | cut -c1-100) || exit 1
done
