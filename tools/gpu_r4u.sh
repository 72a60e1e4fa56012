#!/bin/bash
# round 4, call 25: write-through (sc1) against plain result stores below the non-temporal range
mkdir -p gpurun_out
rm -f gpurun_out/mid_sweep_r4u.jsonl
timeout -k 10 500 python tools/mid_sweep.py --batches 1,16,32,64,80,96,112,128,144,160,192,224,240 --forms default,sc1 --rounds 5 --ms 40 --out gpurun_out/mid_sweep_r4u.jsonl > gpurun_out/mid_sweep_r4u.log 2>&1
echo "rc=$?"
python3 - <<'PY'
import json, collections
t = collections.defaultdict(dict)
for l in open('gpurun_out/mid_sweep_r4u.jsonl'):
    d = json.loads(l); t[d['B']][d['form']] = round(d['ms_per_pass'], 4)
for b in sorted(t): print(b, t[b], round(t[b]['sc1'] / t[b]['default'], 3))
PY
