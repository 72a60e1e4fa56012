#!/bin/bash
# Alg::target_patience (the warm start on the requested mesh while the ladder has unused starts): the scenarios it is for, then the 64- and 256-scenario sets
mkdir -p gpurun_out
MC=$GRAFT_REPO_ROOT/etol_amd/lib/etol_mi355x_montecarlo
export EMI_MC_GATHER=0 EMI_MC_RUNS=1
timeout -k 5 60 $MC 4 64 6 4 > /dev/null 2>&1
for s in 558 938 960; do
  EMI_MC_ONLY=$s timeout -k 10 300 $MC 1024 1023 20 1 > gpurun_out/mc_tp_$s.log 2>&1
  echo "scenario $s rc=$?"; grep "^scenario" gpurun_out/mc_tp_$s.log | cut -c1-500
done
for tp in 200 0; do
EMI_MC_TARGET_PATIENCE=$tp timeout -k 10 300 $MC 64 1023 20 8 > gpurun_out/mc_tp${tp}_64.log 2>&1; echo "tp $tp rc=$?"; tail -1 gpurun_out/mc_tp${tp}_64.log | sed 's/"by_mesh": {.*}}, //' | cut -c1-300
EMI_MC_TARGET_PATIENCE=$tp timeout -k 10 400 $MC 256 1023 20 8 > gpurun_out/mc_tp${tp}_256.log 2>&1 &
pid=$!
while kill -0 $pid 2>/dev/null; do sleep 30; echo "  ... running $(date +%T)"; done
wait $pid; echo "tp $tp rc=$?"; tail -1 gpurun_out/mc_tp${tp}_256.log | sed 's/"by_mesh": {.*}}, //' | cut -c1-300
done
