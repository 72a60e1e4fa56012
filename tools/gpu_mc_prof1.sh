#!/bin/bash
# kernel statistics of ONE 1024-node Monte-Carlo solve (scenario $1, default 3) under rocprofv3 --kernel-trace --stats
S=${1:-3}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT && rm -rf $OUT/mc1_prof
cd /tmp && export TMPDIR=/tmp EMI_MC_GATHER=0 EMI_MC_ONLY=$S
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mc1_prof -- \
   $GRAFT_REPO_ROOT/etol_amd/lib/etol_mi355x_montecarlo 8 1023 20 1 > $OUT/mc1_prof.log 2>&1 &
pid=$!
while kill -0 $pid 2>/dev/null; do sleep 20; echo "profiling ... $(date +%T)" >> $OUT/mc1_progress.log; done
wait $pid
echo "rc=$?"; tail -1 $OUT/mc1_prof.log | cut -c1-200
f=$(ls $OUT/mc1_prof/*/*kernel_stats.csv | head -1)
cp $f $OUT/mc1_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"sum of kernel durations {tot / 1e9:.3f} s over {sum(int(r['Calls']) for r in rows)} launches")
for r in rows[:26]:
    print(f"{float(r['TotalDurationNs']) / 1e6:8.1f} ms {float(r['Percentage']):5.1f} %  {int(r['Calls']):6d} x {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:100]}")
PY
