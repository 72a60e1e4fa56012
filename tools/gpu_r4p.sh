#!/bin/bash
# round 4, call 19: (1) the 288 - 416 instance band (MFMA workgroups first: do they still fit beside the node role?), (2) one stream per context
# (second stream created on demand) and one hardware queue per Monte-Carlo thread
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "two_stream or overlap or f32 or rotated or variants" > $OUT/pytest_r4p.log 2>&1
echo "pytest rc=$?"; tail -2 $OUT/pytest_r4p.log | cut -c1-200
rm -f $OUT/mid_sweep_r4p.jsonl
timeout -k 10 400 python tools/mid_sweep.py --batches 272,288,320,352,384,416 --forms default,order0,order110,order125,order150,order200,bk16,order0_bk16,order125_bk16,order150_bk16,order1_bk16 --rounds 3 --ms 40 \
   --out $OUT/mid_sweep_r4p.jsonl > $OUT/mid_sweep_r4p.log 2>&1
echo "sweep rc=$?"; python3 - <<'PY'
import json, collections
t = collections.defaultdict(dict)
for l in open('gpurun_out/mid_sweep_r4p.jsonl'):
    d = json.loads(l); t[d['B']][d['form']] = round(d['ms_per_pass'], 4)
for b in sorted(t): print(b, t[b])
PY
MC=$R/etol_amd/lib/etol_mi355x_montecarlo
export EMI_MC_GATHER=0
timeout -k 10 200 $MC 8 64 6 8 > /dev/null 2>&1
: > $OUT/mc_r4p.jsonl
run() {   # scenarios nsteps discs threads queues(0 = runtime default)
  if [ "$5" != "0" ]; then export GPU_MAX_HW_QUEUES=$5; else unset GPU_MAX_HW_QUEUES; fi
  timeout -k 10 200 $MC $1 $2 $3 $4 > $OUT/mc_r4p_$2_t$4_q$5.log 2>&1
  echo "mc $* rc=$?"; tail -1 $OUT/mc_r4p_$2_t$4_q$5.log | sed "s/^{/{\"hw_queues\": $5, /" >> $OUT/mc_r4p.jsonl
  tail -1 $OUT/mc_r4p.jsonl | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d[k] for k in ('hw_queues', 'threads', 'solved', 'wall_s', 'solves_per_s')}, {m: round(1e3 * v['seconds'] / v['iterations'], 1) for m, v in d['by_mesh'].items()})"
  unset GPU_MAX_HW_QUEUES
}
run 64 1023 20 8 0
run 64 1023 20 8 8
run 64 1023 20 4 0
run 64 1023 20 6 6
run 64 1023 20 12 12
run 64 1023 20 8 2
