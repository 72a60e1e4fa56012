#!/bin/bash
# Monte-Carlo example at 1024 nodes: NLP tolerance 1e-7 (the example's) against 1e-6 (what ePSOPT sets, ePSOPT.cpp:67)
mkdir -p gpurun_out; : > gpurun_out/mc_tol.jsonl
EMI_MC_GATHER=0 timeout -k 10 300 etol_amd/lib/etol_mi355x_montecarlo 4 1023 20 4 > /dev/null 2>&1   # loads the libraries
for tol in 1e-7 1e-6; do
  for cfg in "8 4" "16 8"; do
    set -- $cfg
    EMI_MC_TOL=$tol EMI_MC_GATHER=0 timeout -k 10 300 etol_amd/lib/etol_mi355x_montecarlo $1 1023 20 $2 > gpurun_out/mc_tol_${tol}_$1.log 2>&1 || exit 1
    tail -1 gpurun_out/mc_tol_${tol}_$1.log | sed "s/^{/{\"nlp_tolerance\": $tol, /" | tee -a gpurun_out/mc_tol.jsonl
  done
done
