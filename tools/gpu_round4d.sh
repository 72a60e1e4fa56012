#!/bin/bash
# Round-4 measurement set of the FINAL build, part 2: Monte-Carlo record runs at the reference's NLP tolerance (1e-6, the example's
# default now), the full 1024-scenario config 4 on this one GPU, one solve under rocprofv3
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
MC=$R/etol_amd/lib/etol_mi355x_montecarlo
say() { echo "[$(date +%T)] $*"; }
export EMI_MC_GATHER=0
timeout -k 5 60 $MC 4 64 6 4 > /dev/null 2>&1
: > $OUT/mc_final.jsonl
run() {   # scenarios nsteps discs threads
  timeout -k 10 300 $MC $1 $2 $3 $4 > $OUT/mc_final_$1x$2.log 2>&1
  echo "mc $* rc=$?"; tail -1 $OUT/mc_final_$1x$2.log >> $OUT/mc_final.jsonl; tail -1 $OUT/mc_final.jsonl | sed 's/"by_mesh": {.*}}, //' | cut -c1-300
}
run 64 64 6 8
run 64 128 10 8
run 64 256 10 8
run 32 512 20 8
run 64 1023 20 8
say "config 4: 1024 scenarios x 1024 nodes"
timeout -k 10 700 $MC 1024 1023 20 8 > $OUT/mc_config4.log 2>&1 &
pid=$!
while kill -0 $pid 2>/dev/null; do sleep 30; echo "  ... running $(date +%T)"; done
wait $pid; echo "rc=$?"
grep "^scenario" $OUT/mc_config4.log | sed 's/  */ /g' | awk '{print $10}' | sort -n | awk '{a[NR]=$1} END {printf "{\"budget\": 1000, \"nlp_tolerance\": 1e-6, \"max_iterations\": %d, \"median_iterations\": %d, \"p90_iterations\": %d, ", a[NR], a[int(NR/2)], a[int(NR*0.9)]}' > $OUT/.pre
tail -1 $OUT/mc_config4.log | sed "s/^{/$(cat $OUT/.pre)/" > $OUT/mc_config4.jsonl
sed 's/"by_mesh": {.*}}, //' $OUT/mc_config4.jsonl | cut -c1-400
grep "rc [^0]" $OUT/mc_config4.log | cut -c1-220
say "one solve under rocprofv3"
rm -rf $OUT/mc1_prof
cd /tmp && export TMPDIR=/tmp EMI_MC_ONLY=3
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mc1_prof -- $MC 8 1023 20 1 > $OUT/mc1_prof.log 2>&1 &
pid=$!
while kill -0 $pid 2>/dev/null; do sleep 20; echo "profiling ... $(date +%T)"; done
wait $pid
echo "profile rc=$?"; tail -1 $OUT/mc1_prof.log | sed 's/"by_mesh": {.*}}, //' | cut -c1-300
f=$(ls $OUT/mc1_prof/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $OUT/mc1_kernel_stats_final.csv
rm -rf $OUT/mc1_prof
say done
