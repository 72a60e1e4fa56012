#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/trace.log 2>&1
echo "rc=$?"
python - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/trace/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if "emi" in r["Kernel_Name"]]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-9:]:
    print(r["Kernel_Name"][:40].ljust(40), "q", r["Queue_Id"], "start", (int(r["Start_Timestamp"]) - t0) / 1e3, "end", (int(r["End_Timestamp"]) - t0) / 1e3, "dur", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
PY
