#!/bin/bash
# Alg::scaling "none" against "automatic" on the Monte-Carlo sets (same scenarios, 8 host threads)
mkdir -p gpurun_out
export EMI_MC_GATHER=0
BIN=etol_amd/lib/etol_mi355x_montecarlo
say() { echo "[$(date +%T)] $*"; }
timeout -k 5 60 $BIN 4 64 6 4 > /dev/null 2>&1
for cfg in "32 1023 20 8" "64 256 10 8" "64 128 10 8"; do
  set -- $cfg
  for sc in none automatic; do
    say "montecarlo $cfg scaling=$sc"
    EMI_MC_SCALING=$sc timeout -k 10 300 $BIN $1 $2 $3 $4 > gpurun_out/mcs_$1_$2_$sc.log 2>&1
    tail -1 gpurun_out/mcs_$1_$2_$sc.log | cut -c1-220
  done
done
say done
