#!/bin/bash
# SQ counters of the fp32 MFMA defect kernel (config 5)
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
SET_A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU"
SET_B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM"
SET_C="GRBM_GUI_ACTIVE"
for set in A B C; do
  eval ctrs=\$SET_$set
  d=$OUT/pmc_c5_$set
  rm -rf $d
  timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $d -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 6 --warmup 2 > $d.log 2>&1
  echo "set $set rc=$?"
done
python - <<'PY'
import csv, glob, json, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(f"{out}/pmc_c5_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        key = next((k for k in ("emi_defect_f32_mfma_kernel", "emi_nodes_kernel", "emi_pass_f32_kernel") if k in r["Kernel_Name"]), None)
        if key:
            a = acc[key][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
res = {k: {c: v[0] / v[1] for c, v in d.items()} for k, d in acc.items()}
for k, c in res.items():
    if c.get("SQ_WAVE_CYCLES"):
        c["derived"] = {"wait_any_frac": c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"], "wait_inst_frac": c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"],
                        "active_frac": c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]}
        if c.get("GRBM_GUI_ACTIVE"):
            c["derived"]["mfma_busy_frac"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)
json.dump(res, open(out + "/pmc_c5.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
