#!/bin/bash
# round 4, call 15: where the iterations and the solver's seconds of a Monte-Carlo set go, per mesh size and phase
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
MC=$GRAFT_REPO_ROOT/etol_amd/lib/etol_mi355x_montecarlo
EMI_MC_GATHER=0 timeout -k 10 200 $MC 8 64 6 8 > /dev/null 2>&1
: > $OUT/mc_r4m.jsonl
EMI_MC_GATHER=0 timeout -k 10 300 $MC 64 1023 20 8 > $OUT/mc_r4m_t8.log 2>&1; echo "rc=$?"
tail -1 $OUT/mc_r4m_t8.log | tee -a $OUT/mc_r4m.jsonl
EMI_MC_GATHER=0 timeout -k 10 300 $MC 8 1023 20 1 > $OUT/mc_r4m_t1.log 2>&1; echo "rc=$?"
tail -1 $OUT/mc_r4m_t1.log | tee -a $OUT/mc_r4m.jsonl
