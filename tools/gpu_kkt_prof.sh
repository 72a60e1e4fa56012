#!/bin/bash
# kernel statistics of 10 factorisations + 30 solves of the 1024-node KKT system
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT && rm -rf $OUT/kkt_prof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kkt_prof -- python3 $GRAFT_REPO_ROOT/tools/scratch/kkt_prof.py > $OUT/kkt_prof.log 2>&1
echo "rc=$?"
f=$(ls $OUT/kkt_prof/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"sum of kernel durations {tot / 1e6:.1f} ms over {sum(int(r['Calls']) for r in rows)} launches")
for r in rows[:24]:
    print(f"{float(r['TotalDurationNs']) / 1e6:8.2f} ms {float(r['Percentage']):5.1f} %  {int(r['Calls']):6d} x {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:90]}")
PY
