#!/bin/bash
# Round-4 measurement set, part 2: HBM traffic from PMC counters (separate passes per counter and batch size) for config 3 at 128 /
# 1024 / 4096 / 16384 instances and for config 5, then the FULL 1024-scenario config 4 on this one GPU
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
say() { echo "[$(date +%T)] $*"; }
say "pmc traffic c3"; bash tools/pmc_traffic.sh "1024 128 4096 16384" > gpurun_out/pmc_traffic.log 2>&1; grep "fetch" gpurun_out/pmc_traffic.log
say "pmc traffic c5"; PMC_M=4096 PMC_OUT=pmc_traffic_c5.json bash tools/pmc_traffic.sh "256" --config c5 > gpurun_out/pmc_traffic_c5.log 2>&1; grep "fetch" gpurun_out/pmc_traffic_c5.log
export EMI_MC_GATHER=0
timeout -k 5 60 etol_amd/lib/etol_mi355x_montecarlo 4 64 6 4 > /dev/null 2>&1
say "config 4: 1024 scenarios x 1024 nodes"
timeout -k 10 700 etol_amd/lib/etol_mi355x_montecarlo 1024 1023 20 8 > gpurun_out/mc_config4.log 2>&1 &
pid=$!
while kill -0 $pid 2>/dev/null; do sleep 30; echo "  ... running $(date +%T)"; done
wait $pid; echo "rc=$?"
grep "^scenario" gpurun_out/mc_config4.log | sed 's/  */ /g' | awk '{print $10}' | sort -n | awk '{a[NR]=$1} END {printf "{\"budget\": 1000, \"max_iterations\": %d, \"median_iterations\": %d, \"p90_iterations\": %d, ", a[NR], a[int(NR/2)], a[int(NR*0.9)]}' > gpurun_out/.pre
tail -1 gpurun_out/mc_config4.log | sed "s/^{/$(cat gpurun_out/.pre)/" | tee gpurun_out/mc_config4.jsonl
grep "rc [^0]" gpurun_out/mc_config4.log | cut -c1-220
say done
