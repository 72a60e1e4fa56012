#!/bin/bash
# Monte-Carlo A/B of the factorisation ladder: single scenarios with the time breakdown, then the 8-scenario set on 4 threads
export EMI_MC_GATHER=0
mkdir -p gpurun_out
timeout -k 5 60 etol_amd/lib/etol_mi355x_montecarlo 1 64 6 1 > /dev/null 2>&1
for prim in 1; do
  for s in 0 3 5; do
    EMI_MC_KKT_PRIMAL=$prim EMI_MC_ONLY=$s EMI_MC_PRINT_LEVEL=5 EMI_MC_KKT_DEBUG=1 timeout -k 5 200 etol_amd/lib/etol_mi355x_montecarlo 8 1023 20 1 > gpurun_out/mc_p${prim}_$s.log 2>&1
    echo "primal=$prim scenario $s: LU fallbacks $(grep -c 'gave up' gpurun_out/mc_p${prim}_$s.log), retries $(grep -c retrying gpurun_out/mc_p${prim}_$s.log)"
    grep "^time:" gpurun_out/mc_p${prim}_$s.log | tail -2 | cut -c1-150
    grep "^scenario" gpurun_out/mc_p${prim}_$s.log | cut -c1-120
  done
  EMI_MC_KKT_PRIMAL=$prim timeout -k 5 300 etol_amd/lib/etol_mi355x_montecarlo 8 1023 20 4 > gpurun_out/mc_p${prim}_set.log 2>&1
  tail -1 gpurun_out/mc_p${prim}_set.log
done
