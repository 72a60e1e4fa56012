#!/bin/bash
# A/B of the Monte-Carlo example at 1024 nodes (8 scenarios, 20 keep-outs): factorisation switch EMI_MC_KKT_STICKY 0 / 1
# on 1 and 4 host threads, then 16 scenarios on 8 threads.  To compare with an older tree instead: git worktree add _old <rev>,
# make -C _old all, and run _old/etol_amd/lib/etol_mi355x_montecarlo the same way (that is how the round-1 numbers in
# profiles/r02_montecarlo.jsonl were re-measured on the same box).
mkdir -p gpurun_out
: > gpurun_out/mc_ab.jsonl
for t in 4 1; do
  for st in 0 1; do
    EMI_MC_KKT_STICKY=$st EMI_MC_GATHER=0 timeout -k 10 300 etol_amd/lib/etol_mi355x_montecarlo 8 1023 20 $t > gpurun_out/mc_ab_sticky${st}_t$t.log 2>&1 || exit 1
    tail -1 gpurun_out/mc_ab_sticky${st}_t$t.log | sed "s/^{/{\"sticky_reg\": $st, /" | tee -a gpurun_out/mc_ab.jsonl
  done
done
EMI_MC_GATHER=0 timeout -k 10 300 etol_amd/lib/etol_mi355x_montecarlo 16 1023 20 8 > gpurun_out/mc_ab_16_t8.log 2>&1 || exit 1
tail -1 gpurun_out/mc_ab_16_t8.log | sed "s/^{/{\"sticky_reg\": 1, /" | tee -a gpurun_out/mc_ab.jsonl
