#!/bin/bash
# round 4, call 16: kernels of ONE factorisation / low-rank correction / refined solve at 1024 and 257 nodes
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for M in 1024 257; do
  for what in factor lowrank refined; do
    rm -rf $R/gpurun_out/anat
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/anat -- python3 $R/tools/factor_anatomy.py --nodes $M --what $what --reps 20 > $R/gpurun_out/anat_${what}_$M.log 2>&1
    echo "rc=$? $(grep 'ms per call' $R/gpurun_out/anat_${what}_$M.log)"
    f=$(ls $R/gpurun_out/anat/*/*kernel_stats.csv 2>/dev/null | head -1)
    [ -n "$f" ] && cp $f $R/gpurun_out/anat_${what}_${M}_kernel_stats.csv
  done
done
rm -rf $R/gpurun_out/anat
