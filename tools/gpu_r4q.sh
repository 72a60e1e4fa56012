#!/bin/bash
# round 4, call 21: per-scenario NLP solves of 256 scenarios (which rungs fail, after how many iterations), and a finer default sweep
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
MC=$R/etol_amd/lib/etol_mi355x_montecarlo
export EMI_MC_GATHER=0
timeout -k 10 200 $MC 8 64 6 8 > /dev/null 2>&1
EMI_MC_RUNS=1 timeout -k 10 400 $MC 256 1023 20 8 > $OUT/mc_r4q_runs.log 2>&1 &
pid=$!
while kill -0 $pid 2>/dev/null; do sleep 30; echo "  ... running $(date +%T)"; done
wait $pid; echo "rc=$?"
tail -1 $OUT/mc_r4q_runs.log | sed 's/"by_mesh": {.*}}, //' | cut -c1-300
rm -f $OUT/default_sweep_fine.jsonl
timeout -k 10 400 python tools/mid_sweep.py --batches 8,24,32,48,64,80,96,112,128,144,160,224,240,1280,1536,1792,2048,2304,2560,3072 --forms default --rounds 3 --ms 40 --out $OUT/default_sweep_fine.jsonl > $OUT/default_sweep_fine.log 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/default_sweep_fine.jsonl'):
    d = json.loads(l); print(d['B'], round(d['ms_per_pass'], 4), '%.3g' % d['node_evals_per_s'], d['kernel'][20:])
PY
