#!/bin/bash
# the planned route as the second start of a ladder climb: the scenarios that stayed unsolved (938; 960 at tolerance 1e-7), then the 64- and 256-scenario sets
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
MC=$GRAFT_REPO_ROOT/etol_amd/lib/etol_mi355x_montecarlo
export EMI_MC_GATHER=0 EMI_MC_RUNS=1
timeout -k 5 60 $MC 4 64 6 4 > /dev/null 2>&1
for s in 938 960 17 10; do
  EMI_MC_ONLY=$s timeout -k 10 300 $MC 1024 1023 20 1 > $OUT/mc_plan_$s.log 2>&1
  echo "scenario $s rc=$?"; grep "^scenario" $OUT/mc_plan_$s.log | cut -c1-400
done
timeout -k 10 300 $MC 64 1023 20 8 > $OUT/mc_plan_64.log 2>&1; echo "rc=$?"; tail -1 $OUT/mc_plan_64.log | sed 's/"by_mesh": {.*}}, //' | cut -c1-300
timeout -k 10 400 $MC 256 1023 20 8 > $OUT/mc_plan_256.log 2>&1 &
pid=$!
while kill -0 $pid 2>/dev/null; do sleep 30; echo "  ... running $(date +%T)"; done
wait $pid; echo "rc=$?"; tail -1 $OUT/mc_plan_256.log | sed 's/"by_mesh": {.*}}, //' | cut -c1-300
