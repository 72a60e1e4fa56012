#!/bin/bash
# the four straggler scenarios of the 513-node Monte-Carlo set (10, 17, 27, 29) and two ordinary ones (0, 5) under other
# warm-start barrier parameters (Alg::warm_mu_init; default 1e-5)
mkdir -p gpurun_out; : > gpurun_out/mc_stragglers.txt
EMI_MC_GATHER=0 timeout -k 10 200 etol_amd/lib/etol_mi355x_montecarlo 2 64 6 1 > /dev/null 2>&1
for mu in 1e-5 1e-4 1e-3; do
  for s in 10 17 27 29 0 5; do
    EMI_MC_WARM_MU=$mu EMI_MC_ONLY=$s EMI_MC_GATHER=0 timeout -k 10 200 etol_amd/lib/etol_mi355x_montecarlo 32 512 20 1 2>&1 | grep "^scenario" | sed "s/^/warm_mu $mu  /" | tee -a gpurun_out/mc_stragglers.txt
  done
done
