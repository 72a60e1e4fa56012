#!/bin/bash
# config 4 in full with the step refined to 1e-8 instead of the default 1e-10 (EMI_MC_KKT_REFINE_EXP=8): the throughput variant of notes section 21
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
MC=$GRAFT_REPO_ROOT/etol_amd/lib/etol_mi355x_montecarlo
export EMI_MC_GATHER=0 EMI_MC_KKT_REFINE_EXP=8
timeout -k 5 60 $MC 4 64 6 4 > /dev/null 2>&1
timeout -k 10 700 $MC 1024 1023 20 8 > $OUT/mc_config4_e8.log 2>&1 &
pid=$!
while kill -0 $pid 2>/dev/null; do sleep 30; echo "  ... running $(date +%T)"; done
wait $pid; echo "rc=$?"
grep "^scenario" $OUT/mc_config4_e8.log | sed 's/  */ /g' | awk '{print $10}' | sort -n | awk '{a[NR]=$1} END {printf "{\"budget\": 1000, \"nlp_tolerance\": 1e-6, \"kkt_refine_exp\": 8, \"max_iterations\": %d, \"median_iterations\": %d, \"p90_iterations\": %d, ", a[NR], a[int(NR/2)], a[int(NR*0.9)]}' > $OUT/.pre
tail -1 $OUT/mc_config4_e8.log | sed "s/^{/$(cat $OUT/.pre)/" > $OUT/mc_config4_e8.jsonl
sed 's/"by_mesh": {.*}}, //' $OUT/mc_config4_e8.jsonl | cut -c1-400
grep "rc [^0]" $OUT/mc_config4_e8.log | cut -c1-220
