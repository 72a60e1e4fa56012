#!/bin/bash
# Monte-Carlo example over mesh sizes (8 host threads): scenarios solved, solves per second, mean iterations
mkdir -p gpurun_out; : > gpurun_out/mc_sizes.jsonl
EMI_MC_GATHER=0 timeout -k 10 300 etol_amd/lib/etol_mi355x_montecarlo 8 64 6 8 > /dev/null 2>&1      # loads the libraries
for cfg in "64 64 6" "64 128 10" "64 256 10" "32 512 20"; do
  set -- $cfg
  EMI_MC_GATHER=0 timeout -k 10 600 etol_amd/lib/etol_mi355x_montecarlo $1 $2 $3 8 > gpurun_out/mc_sizes_$2.log 2>&1 || exit 1
  tail -1 gpurun_out/mc_sizes_$2.log | tee -a gpurun_out/mc_sizes.jsonl
done
