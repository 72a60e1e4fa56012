#!/bin/bash
# HBM traffic of the bench kernels from PMC counters: separate passes for FETCH_SIZE and
# WRITE_SIZE (they do not fit one pass: MI355X_MICROARCH.md, rocprofv3 PMC slots), counters only
# with --kernel-trace.  Summary -> gpurun_out/pmc_traffic.json (copy to profiles/).
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_$ctr
  timeout -k 10 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_$ctr -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $OUT/pmc_$ctr.log 2>&1
  echo "$ctr rc=$?"
done
python - <<'PY'
import csv, glob, json, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"{out}/pmc_{ctr}/*/*counter_collection.csv")
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != ctr:
                continue
            name = r["Kernel_Name"]
            key = next((k for k in ("emi_nodes_kernel", "emi_pass_f64_kernel", "emi_symdefect_ring2_f64_kernel",
                                    "emi_symdefect_combine_kernel", "emi_symdefect_ring_f64_kernel", "emi_symdefect_f64_kernel",
                                    "emi_defect_f64_kernel", "emi_defect_f32_mfma_kernel", "emi_cost_finish_kernel") if k in name), None)
            if key:
                acc[key][0] += float(r["Counter_Value"])
                acc[key][1] += 1
    res[ctr] = {k: v[0] / v[1] for k, v in acc.items() if v[1]}
    print(ctr, {k: round(v, 1) for k, v in res[ctr].items()})
# FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B on rocprofv3 (guide: hbm_bytes = (F + W) * 1024);
# gfx950 correction: FETCH_SIZE reads exactly half of a wide coalesced stream -> doubled.
summary = {"unit_note": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024, per launch, averaged over dispatches",
           "raw": res, "bytes_per_launch": {}}
for k in set(res["FETCH_SIZE"]) | set(res["WRITE_SIZE"]):
    summary["bytes_per_launch"][k] = (2.0 * res["FETCH_SIZE"].get(k, 0.0) + res["WRITE_SIZE"].get(k, 0.0)) * 1024.0
json.dump(summary, open(out + "/pmc_traffic.json", "w"), indent=1)
print(json.dumps(summary["bytes_per_launch"]))
PY
