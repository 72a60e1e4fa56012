#!/bin/bash
# tools/gpu_kkt_prof.sh with the main build and every build under _alt/ (tools/ab_build.sh): the Cholesky kernels' average durations
R=$GRAFT_REPO_ROOT
for d in . $(ls -d _alt/*/ 2>/dev/null); do
  echo "== build $d"
  (cd $R/$d && GRAFT_REPO_ROOT=$R/$d bash tools/gpu_kkt_prof.sh 2>&1 | grep "sum of\|chol_\|rc=")
done
