#!/bin/bash
# round 4, second GPU call: the full GPU suite, where one 1024-node solve spends its time (solver timers), Monte-Carlo sets with
# scaling none (the new default) against automatic, then a one-solve kernel profile
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/pytest_r4b.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/pytest_r4b.log
for s in 0 3; do
  EMI_MC_GATHER=0 EMI_MC_ONLY=$s EMI_MC_PRINT_LEVEL=5 timeout -k 10 120 etol_amd/lib/etol_mi355x_montecarlo 8 1023 20 1 > gpurun_out/one_solve_$s.log 2>&1
  grep -E "^time:|mesh sequencing|^scenario" gpurun_out/one_solve_$s.log | cut -c1-330
done
: > gpurun_out/mc_r4b.jsonl
EMI_MC_GATHER=0 timeout -k 10 200 etol_amd/lib/etol_mi355x_montecarlo 8 64 6 8 > /dev/null 2>&1
for sc in none automatic; do
for cfg in "64 1023 20 8" "32 512 20 8" "64 256 10 8" "64 128 10 8"; do
  set -- $cfg
  EMI_MC_SCALING=$sc EMI_MC_GATHER=0 timeout -k 10 300 etol_amd/lib/etol_mi355x_montecarlo $1 $2 $3 $4 > gpurun_out/mc_r4b_${sc}_$2.log 2>&1
  echo "mc $sc $cfg rc=$?"; tail -1 gpurun_out/mc_r4b_${sc}_$2.log | sed "s/^{/{\"scaling\": \"$sc\", /" | tee -a gpurun_out/mc_r4b.jsonl
done
done
S=3
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/mc1_prof
cd /tmp && export TMPDIR=/tmp EMI_MC_GATHER=0 EMI_MC_ONLY=$S
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mc1_prof -- \
   $GRAFT_REPO_ROOT/etol_amd/lib/etol_mi355x_montecarlo 8 1023 20 1 > $OUT/mc1_prof.log 2>&1 &
pid=$!
while kill -0 $pid 2>/dev/null; do sleep 20; echo "profiling ... $(date +%T)"; done
wait $pid
echo "profile rc=$?"; tail -2 $OUT/mc1_prof.log | cut -c1-200
f=$(ls $OUT/mc1_prof/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then cp $f $OUT/mc1_kernel_stats.csv; head -30 $f | cut -c1-220; fi
