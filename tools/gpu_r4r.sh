#!/bin/bash
# round 4, call 22: holes of the fine sweep -- 80 .. 112 instances (K slices / states per workgroup), 2304 .. 4096 (tile order, slicing); Monte-Carlo after the
# ladder no longer repeats a bend
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
rm -f $OUT/mid_sweep_r4r_small.jsonl $OUT/mid_sweep_r4r_large.jsonl
timeout -k 10 300 python tools/mid_sweep.py --batches 72,80,96,112,120 --forms default,sw1_ks1,sw1_ks2,sw2_ks1,sw2_bk16,sw1_bk16 --rounds 3 --ms 40 --out $OUT/mid_sweep_r4r_small.jsonl > $OUT/mid_sweep_r4r_small.log 2>&1
echo "small rc=$?"
timeout -k 10 400 python tools/mid_sweep.py --batches 2304,2560,3072,4096 --forms default,slice1024,slice1280,slice1536,ct1,cp2_ct1,cp2 --rounds 3 --ms 60 --out $OUT/mid_sweep_r4r_large.jsonl > $OUT/mid_sweep_r4r_large.log 2>&1
echo "large rc=$?"
python3 - <<'PY'
import json, collections
for f in ('small', 'large'):
    t = collections.defaultdict(dict)
    for l in open(f'gpurun_out/mid_sweep_r4r_{f}.jsonl'):
        d = json.loads(l); t[d['B']][d['form']] = round(d['ms_per_pass'], 4)
    for b in sorted(t): print(b, t[b])
PY
MC=$R/etol_amd/lib/etol_mi355x_montecarlo
export EMI_MC_GATHER=0
timeout -k 10 200 $MC 8 64 6 8 > /dev/null 2>&1
EMI_MC_RUNS=1 timeout -k 10 400 $MC 256 1023 20 8 > $OUT/mc_r4r_runs.log 2>&1 &
pid=$!
while kill -0 $pid 2>/dev/null; do sleep 30; echo "  ... running $(date +%T)"; done
wait $pid; echo "rc=$?"
tail -1 $OUT/mc_r4r_runs.log | sed 's/"by_mesh": {.*}}, //' | cut -c1-300
