#!/bin/bash
# round 4, first GPU call: the GPU suite with the advisor fixes, the regression probe, a one-solve kernel profile, ring depth at the
# config-4 shard, Monte-Carlo baselines of this build
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_r4a.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/pytest_r4a.log
timeout -k 10 300 python tools/regress_probe.py > gpurun_out/regress_probe.log 2>&1
echo "probe rc=$?"; tail -20 gpurun_out/regress_probe.log
bash tools/gpu_mc_prof1.sh 0 2>&1 | tail -32
cp gpurun_out/mc1_kernel_stats.csv gpurun_out/mc1_kernel_stats_s0.csv
BATCHES="128" bash tools/gpu_nst.sh > gpurun_out/nst128.log 2>&1; tail -8 gpurun_out/nst128.log
: > gpurun_out/mc_r4a.jsonl
EMI_MC_GATHER=0 timeout -k 10 200 etol_amd/lib/etol_mi355x_montecarlo 8 64 6 8 > /dev/null 2>&1
for cfg in "64 1023 20 8" "32 512 20 8" "64 256 10 8"; do
  set -- $cfg
  EMI_MC_GATHER=0 timeout -k 10 300 etol_amd/lib/etol_mi355x_montecarlo $1 $2 $3 $4 > gpurun_out/mc_r4a_$2.log 2>&1
  echo "mc $cfg rc=$?"; tail -1 gpurun_out/mc_r4a_$2.log | tee -a gpurun_out/mc_r4a.jsonl
done
