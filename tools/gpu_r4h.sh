#!/bin/bash
# round 4, eighth GPU call: GPU suite with the new ladder defaults, single-solve times, the Monte-Carlo sets, the FULL 1024-scenario
# config 4 on one GPU, one-solve kernel profile
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/pytest_r4h.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/pytest_r4h.log
timeout -k 10 200 python tools/solve_times.py > gpurun_out/solve_times.log 2>&1; tail -8 gpurun_out/solve_times.log
: > gpurun_out/mc_r4h.jsonl
EMI_MC_GATHER=0 timeout -k 10 200 etol_amd/lib/etol_mi355x_montecarlo 8 64 6 8 > /dev/null 2>&1
run() {   # scenarios nsteps discs threads budget
  EMI_MC_BUDGET=$5 EMI_MC_GATHER=0 timeout -k 10 700 etol_amd/lib/etol_mi355x_montecarlo $1 $2 $3 $4 > gpurun_out/mc_r4h_$1x$2_t$4_b$5.log 2>&1 &
  pid=$!
  while kill -0 $pid 2>/dev/null; do sleep 30; echo "  ... $1 x $2 running $(date +%T)"; done
  wait $pid
  echo "mc $* rc=$?"
  grep "^scenario" gpurun_out/mc_r4h_$1x$2_t$4_b$5.log | sed 's/  */ /g' | awk '{print $10}' | sort -n | awk -v b=$5 '{a[NR]=$1} END {printf "{\"budget\": %d, \"max_iterations\": %d, \"median_iterations\": %d, \"p90_iterations\": %d, ", b, a[NR], a[int(NR/2)], a[int(NR*0.9)]}' > gpurun_out/.pre
  tail -1 gpurun_out/mc_r4h_$1x$2_t$4_b$5.log | sed "s/^{/$(cat gpurun_out/.pre)/" | tee -a gpurun_out/mc_r4h.jsonl
}
run 64 1023 20 8 0
run 64 1023 20 8 300
run 32 512 20 8 0
run 64 256 10 8 0
run 64 128 10 8 0
run 64 64 6 8 0
run 1024 1023 20 8 0
S=3
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/mc1_prof
cd /tmp && export TMPDIR=/tmp EMI_MC_GATHER=0 EMI_MC_ONLY=$S
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mc1_prof -- \
   $GRAFT_REPO_ROOT/etol_amd/lib/etol_mi355x_montecarlo 8 1023 20 1 > $OUT/mc1_prof.log 2>&1 &
pid=$!
while kill -0 $pid 2>/dev/null; do sleep 20; echo "profiling ... $(date +%T)"; done
wait $pid
echo "profile rc=$?"; tail -2 $OUT/mc1_prof.log | cut -c1-200
f=$(ls $OUT/mc1_prof/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then cp $f $OUT/mc1_kernel_stats_final.csv; fi
