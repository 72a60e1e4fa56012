#!/bin/bash
# round 4, call 24: forms around the default at 128 / 144 instances (stores, tile order, block order)
mkdir -p gpurun_out
rm -f gpurun_out/mid_sweep_r4t.jsonl
timeout -k 10 400 python tools/mid_sweep.py --batches 128,144 --forms default,nt,sc1,ntsc1,cp1,cp2,cp4,cpm1,nt_order150,order150_only,order0_only,g2c2,g1c2 --rounds 5 --ms 40 --out gpurun_out/mid_sweep_r4t.jsonl > gpurun_out/mid_sweep_r4t.log 2>&1
echo "rc=$?"; tail -3 gpurun_out/mid_sweep_r4t.log | cut -c1-200
python3 - <<'PY'
import json, collections
t = collections.defaultdict(dict)
for l in open('gpurun_out/mid_sweep_r4t.jsonl'):
    d = json.loads(l); t[d['B']][d['form']] = round(d['ms_per_pass'], 4)
for b in sorted(t): print(b, t[b])
PY
