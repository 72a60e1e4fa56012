#!/bin/bash
# Where the two kernels of the pass spend their cycles: SQ busy / wait / MFMA-busy counters per kernel, with the
# default dispatch (EMI_OVERLAP_MODE=0: at 1024 instances the one-launch pass kernel, the product path), the two
# kernels back to back on one stream (=1) and concurrent on two streams (=2).
# Two --pmc passes of <= 8 SQ counters each (MI355X_MICROARCH.md, rocprofv3 PMC slots); counters only with
# --kernel-trace.  Summary -> gpurun_out/pmc_busy.json (copy to profiles/).   usage: tools/pmc_busy.sh [sym_ct]
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
CT=${1:-0}
cd /tmp && export TMPDIR=/tmp
SET_A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES"
SET_B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES"
SET_C="GRBM_GUI_ACTIVE GRBM_COUNT"
for mode in 0 1 2; do
  for set in A B C; do
    eval ctrs=\$SET_$set
    d=$OUT/pmc_busy_m${mode}_$set
    rm -rf $d
    EMI_SYM_CT=$CT EMI_OVERLAP_MODE=$mode timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $d -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary > $d.log 2>&1
    echo "mode $mode set $set rc=$?"
  done
done
python - <<'PY'
import csv, glob, json, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
res = {}
for mode in (0, 1, 2):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(f"{out}/pmc_busy_m{mode}_*/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            key = next((k for k in ("emi_nodes_kernel", "emi_pass_f64_kernel", "emi_symdefect_ring2_f64_kernel",
                                    "emi_symdefect_ring_f64_kernel", "emi_cost_finish_kernel") if k in name), None)
            if key:
                a = acc[key][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
    res[{0: "default_dispatch", 1: "back_to_back", 2: "concurrent"}[mode]] = {k: {c: v[0] / v[1] for c, v in d.items()} for k, d in acc.items()}
for mode, ks in res.items():
    for k, c in ks.items():
        if "SQ_BUSY_CYCLES" in c and c.get("SQ_WAVE_CYCLES"):
            # SQ_WAVE_CYCLES / WAIT_* / ACTIVE_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs... per XCD-SE
            c["derived"] = {"wait_any_frac_of_wave_cycles": c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"],
                            "wait_inst_frac": c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"],
                            "active_inst_frac": c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]}
json.dump(res, open(out + "/pmc_busy.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
PY
