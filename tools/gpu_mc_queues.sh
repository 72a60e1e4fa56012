#!/bin/bash
# Does the HIP runtime's hardware-queue count limit concurrent solver threads?  16 scenarios x 1024 nodes, 8 threads
export EMI_MC_GATHER=0
mkdir -p gpurun_out
timeout -k 5 60 etol_amd/lib/etol_mi355x_montecarlo 4 64 6 4 > /dev/null 2>&1
for q in default 8 16 2; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  for thr in 8 16; do
    echo "[$(date +%T)] queues $q threads $thr"
    timeout -k 5 200 etol_amd/lib/etol_mi355x_montecarlo 16 1023 20 $thr > gpurun_out/mcq_${q}_$thr.log 2>&1
    tail -1 gpurun_out/mcq_${q}_$thr.log | cut -c90-220
  done
done
