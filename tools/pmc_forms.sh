#!/bin/bash
# FETCH_SIZE (and WRITE_SIZE) per launch of the pass kernel for several dispatch forms of tools/mid_sweep.py, one rocprofv3
# counter pass per form and counter.   usage: tools/pmc_forms.sh "<forms>" [batch] [counters]
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
FORMS=$1; B=${2:-1024}; CTRS=${3:-FETCH_SIZE}
cd /tmp && export TMPDIR=/tmp
for f in $FORMS; do
  for ctr in $CTRS; do
    d=$OUT/pmcf_${f}_$ctr
    rm -rf $d
    timeout -k 10 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -- python $GRAFT_REPO_ROOT/tools/mid_sweep.py --batches $B --forms $f --rounds 1 --ms 3 --out /tmp/pmcf.jsonl > $d.log 2>&1 || { echo "$f $ctr failed"; tail -3 $d.log; }
  done
done
PMC_FORMS="$FORMS" PMC_CTRS="$CTRS" PMC_B=$B python - <<'PY'
import csv, glob, json, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
res = []
for f in os.environ["PMC_FORMS"].split():
    row = {"form": f, "B": int(os.environ["PMC_B"])}
    for ctr in os.environ["PMC_CTRS"].split():
        acc = collections.defaultdict(lambda: [0.0, 0])
        rows = []
        for p in glob.glob(f"{out}/pmcf_{f}_{ctr}/*/*counter_collection.csv"):
            for r in csv.DictReader(open(p)):
                if r.get("Counter_Name") == ctr and ("emi_pass" in r["Kernel_Name"] or "emi_symdefect" in r["Kernel_Name"] or "emi_nodes" in r["Kernel_Name"]):
                    rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"].split("<")[0].split("(")[0][-40:], float(r["Counter_Value"])))
        rows.sort()
        for _, k, v in rows[-8:]:       # the last dispatches: the form under test (the run starts with passes of the default form)
            acc[k][0] += v; acc[k][1] += 1
        for k, v in acc.items():
            mb = v[0] / v[1] * 1024 / 1e6 * (2 if ctr == "FETCH_SIZE" else 1)
            row[f"{ctr}_MB:{k}"] = round(mb, 1)
    res.append(row)
    print(row)
json.dump(res, open(out + "/pmc_forms.json", "w"), indent=1)
PY
