#!/bin/bash
# time the same variants with the main build and with every build under _alt/ on this box (see tools/ab_build.sh)
# usage: bash tools/gpu_ab.sh "<batches>" "<variants>"
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
for b in $1; do
  for d in . $(ls -d _alt/*/ 2>/dev/null); do
    echo "== build $d"
    (cd $R/$d && timeout -k 10 300 python tools/pass_variants.py --batch $b --rounds 5 --steps 200 --no-profile --out $R/gpurun_out/ab_$(basename $d)_$b.jsonl --only "$2" 2>&1 | grep -v amdgpu.ids) || exit 1
  done
done
