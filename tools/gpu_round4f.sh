#!/bin/bash
# final tree (band / small-batch policies, write-through stores, refinement to 1e-10, one stream per context): the whole GPU suite, the bench line,
# then part 2 of the measurement set (Monte-Carlo records, config 4 in full, one solve under rocprofv3)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
echo "[$(date +%T)] tests"
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/pytest_final.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_final.log | cut -c1-300
echo "[$(date +%T)] bench"
timeout -k 10 500 python bench.py > gpurun_out/bench.log 2>&1 || tail gpurun_out/bench.log
tail -1 gpurun_out/bench.log | cut -c1-300
bash $R/tools/gpu_round4d.sh
