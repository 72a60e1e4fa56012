#!/bin/bash
# the clearance-weighted planned route: as the last cold-start attempt (default) and as the second start of a ladder climb (EMI_MC_PLAN=1)
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
MC=$GRAFT_REPO_ROOT/etol_amd/lib/etol_mi355x_montecarlo
export EMI_MC_GATHER=0 EMI_MC_RUNS=1
timeout -k 5 60 $MC 4 64 6 4 > /dev/null 2>&1
for plan in 1; do
for s in 938 960 17; do
  EMI_MC_PLAN=$plan EMI_MC_ONLY=$s timeout -k 10 300 $MC 1024 1023 20 1 > $OUT/mc_plan${plan}_$s.log 2>&1
  echo "plan $plan scenario $s rc=$?"; grep "^scenario" $OUT/mc_plan${plan}_$s.log | cut -c1-400
done
done
for plan in 0 1; do
EMI_MC_PLAN=$plan timeout -k 10 300 $MC 64 1023 20 8 > $OUT/mc_plan${plan}_64.log 2>&1; echo "plan $plan rc=$?"; tail -1 $OUT/mc_plan${plan}_64.log | sed 's/"by_mesh": {.*}}, //' | cut -c1-300
EMI_MC_PLAN=$plan timeout -k 10 400 $MC 256 1023 20 8 > $OUT/mc_plan${plan}_256.log 2>&1 &
pid=$!
while kill -0 $pid 2>/dev/null; do sleep 30; echo "  ... running $(date +%T)"; done
wait $pid; echo "plan $plan rc=$?"; tail -1 $OUT/mc_plan${plan}_256.log | sed 's/"by_mesh": {.*}}, //' | cut -c1-300
done
