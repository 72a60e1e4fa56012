#!/bin/bash
# round 4, call 29: forms around the default at 512 / 1024 / 2048 instances with the final register allocation (block order, column partitions, stores)
mkdir -p gpurun_out
rm -f gpurun_out/mid_sweep_r4x.jsonl
timeout -k 10 500 python tools/mid_sweep.py --batches 512,1024,2048 --forms default,o100,o105,o110,o115,o120,o130,cp1,cp2,cp4,g2c2,ntsc1,bk16 --rounds 5 --ms 60 --out gpurun_out/mid_sweep_r4x.jsonl > gpurun_out/mid_sweep_r4x.log 2>&1
echo "rc=$?"
python3 - <<'PY'
import json, collections
t = collections.defaultdict(dict)
for l in open('gpurun_out/mid_sweep_r4x.jsonl'):
    d = json.loads(l); t[d['B']][d['form']] = round(d['ms_per_pass'], 4)
for b in sorted(t): print(b, t[b])
PY
