#!/bin/bash
# HBM traffic of the bench kernels from PMC counters: separate passes for FETCH_SIZE and
# WRITE_SIZE (they do not fit one pass: MI355X_MICROARCH.md, rocprofv3 PMC slots), counters only
# with --kernel-trace.  One pair of passes per batch size; summary -> gpurun_out/pmc_traffic.json
# (entries keyed by kernel, B, M; copy to profiles/rNN_pmc_traffic.json, where bench.py looks a
# figure up for exactly its own (kernel, B, M)).     usage: tools/pmc_traffic.sh "1024 128" [bench.py args]
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
BATCHES=${1:-1024}
shift
cd /tmp && export TMPDIR=/tmp
for b in $BATCHES; do
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_${ctr}_$b
  timeout -k 10 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_${ctr}_$b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --scenarios $b "$@" > $OUT/pmc_${ctr}_$b.log 2>&1
  echo "$ctr B=$b rc=$?"
done
done
PMC_BATCHES="$BATCHES" PMC_ARGS="$*" python - <<'PY'
import csv, glob, json, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
KEYS = ("emi_nodes_kernel", "emi_pass_f64_kernel", "emi_symdefect_ring2_f64_kernel", "emi_symdefect_combine_kernel",
        "emi_symdefect_ring_f64_kernel", "emi_symdefect_f64_kernel", "emi_defect_f64_kernel", "emi_defect_f32_mfma_kernel",
        "emi_defect_f32_ring_kernel", "emi_pass_f32_kernel", "emi_cost_finish_kernel")
MESH = int(os.environ.get("PMC_M", "1024"))
entries = []
for b in os.environ["PMC_BATCHES"].split():
    res = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for f in glob.glob(f"{out}/pmc_{ctr}_{b}/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") != ctr:
                    continue
                key = next((k for k in KEYS if k in r["Kernel_Name"]), None)
                if key:
                    acc[key][0] += float(r["Counter_Value"])
                    acc[key][1] += 1
        res[ctr] = {k: v[0] / v[1] for k, v in acc.items() if v[1]}
    # FETCH_SIZE / WRITE_SIZE are in units of 1024 B on rocprofv3; gfx950 correction: FETCH_SIZE reads exactly half of a
    # wide coalesced stream -> doubled (MI355X_MICROARCH.md, HBM)
    for k in sorted(set(res["FETCH_SIZE"]) | set(res["WRITE_SIZE"])):
        f, w = res["FETCH_SIZE"].get(k, 0.0), res["WRITE_SIZE"].get(k, 0.0)
        entries.append({"kernel": k, "B": int(b), "M": MESH, "fetch_bytes": 2.0 * f * 1024.0, "write_bytes": w * 1024.0,
                        "bytes_per_launch": (2.0 * f + w) * 1024.0, "raw": {"FETCH_SIZE": f, "WRITE_SIZE": w},
                        "source": f"tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) --kernel-trace -- "
                                  f"python bench.py --steps 5 --warmup 2 --scenarios {b} {os.environ.get('PMC_ARGS', '')}".strip()})
        print(b, k, "fetch %.1f MB  write %.1f MB  total %.1f MB" % (2 * f * 1024 / 1e6, w * 1024 / 1e6, (2 * f + w) * 1024 / 1e6))
summary = {"unit_note": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024, per launch, averaged over the dispatches of a run",
           "entries": entries}
json.dump(summary, open(out + "/" + os.environ.get("PMC_OUT", "pmc_traffic.json"), "w"), indent=1)
PY
