#!/bin/bash
# kernel statistics of the Monte-Carlo example (8 scenarios x 1024 nodes x 20 keep-outs): where the device time of the
# solves goes and how busy the GPU is (sum of kernel durations against wall time)
# usage: bash tools/gpu_mc_prof.sh [threads] [scenarios]
T=${1:-4}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT && rm -rf $OUT/mc_prof_t$T
cd /tmp && export TMPDIR=/tmp EMI_MC_GATHER=0
N=${2:-8}
timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mc_prof_t$T -- \
   $GRAFT_REPO_ROOT/etol_amd/lib/etol_mi355x_montecarlo $N 1023 20 $T > $OUT/mc_prof_t$T.log 2>&1 &
pid=$!
while kill -0 $pid 2>/dev/null; do sleep 30; echo "profiling ... $(date +%T)"; done      # (the profiler writes only when it ends)
wait $pid
echo "rc=$?"; tail -1 $OUT/mc_prof_t$T.log
f=$(ls $OUT/mc_prof_t$T/*/*kernel_stats.csv | head -1)
cp $f $OUT/mc_prof_t${T}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"sum of kernel durations {tot / 1e9:.2f} s over {sum(int(r['Calls']) for r in rows)} launches")
for r in rows[:22]:
    print(f"{float(r['TotalDurationNs']) / 1e9:7.3f} s {float(r['Percentage']):5.1f} %  {int(r['Calls']):7d} x {float(r['AverageNs']) / 1e3:9.1f} us  {r['Name'][:110]}")
PY
rm -rf $OUT/mc_prof_t$T
