#!/bin/bash
# config 4 in full with the planned route as the first start of a climb, clearance weight 0.01
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
MC=$GRAFT_REPO_ROOT/etol_amd/lib/etol_mi355x_montecarlo
export EMI_MC_GATHER=0 EMI_MC_PLAN=2 EMI_MC_PLAN_CLEARANCE=0.01
timeout -k 5 60 $MC 4 64 6 4 > /dev/null 2>&1
timeout -k 10 700 $MC 1024 1023 20 8 > $OUT/mc_config4_pf.log 2>&1 &
pid=$!
while kill -0 $pid 2>/dev/null; do sleep 30; echo "  ... running $(date +%T)"; done
wait $pid; echo "rc=$?"
grep "^scenario" $OUT/mc_config4_pf.log | sed 's/  */ /g' | awk '{print $10}' | sort -n | awk '{a[NR]=$1} END {printf "{\"budget\": 1000, \"nlp_tolerance\": 1e-6, \"plan_first_start\": 1, \"plan_clearance\": 0.01, \"max_iterations\": %d, \"median_iterations\": %d, \"p90_iterations\": %d, ", a[NR], a[int(NR/2)], a[int(NR*0.9)]}' > $OUT/.pre
tail -1 $OUT/mc_config4_pf.log | sed "s/^{/$(cat $OUT/.pre)/" > $OUT/mc_config4_pf.jsonl
sed 's/"by_mesh": {.*}}, //' $OUT/mc_config4_pf.jsonl | cut -c1-420
grep "rc [^0]" $OUT/mc_config4_pf.log | cut -c1-160
grep "^scenario" $OUT/mc_config4_pf.log | awk '$6==0 {s+=$12; n++} END {print "mean cost of the solved", s/n, n}'
