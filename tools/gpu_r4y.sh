#!/bin/bash
# round 4, call 30: 8- against 16-deep K tiles from 1280 to 2048 instances with the final register allocation
mkdir -p gpurun_out
rm -f gpurun_out/mid_sweep_r4y.jsonl
timeout -k 10 500 python tools/mid_sweep.py --batches 1280,1536,1792,2048 --forms default,bk16,o110,default,bk16 --rounds 5 --ms 60 --out gpurun_out/mid_sweep_r4y.jsonl > gpurun_out/mid_sweep_r4y.log 2>&1
echo "rc=$?"
python3 - <<'PY'
import json, collections
t = collections.defaultdict(lambda: collections.defaultdict(list))
for l in open('gpurun_out/mid_sweep_r4y.jsonl'):
    d = json.loads(l); t[d['B']][d['form']].append(round(d['ms_per_pass'], 4))
for b in sorted(t): print(b, dict(t[b]))
PY
