#!/bin/bash
# final default sweep of the round (after the last policy changes): the whole batch range through the default dispatch, and the bench lines of the sizes whose form changed
mkdir -p gpurun_out
rm -f gpurun_out/default_sweep.jsonl
timeout -k 10 700 python tools/mid_sweep.py --batches 1,4,16,32,64,80,96,112,128,144,160,192,224,256,288,320,352,384,416,448,512,576,640,704,768,896,1024,1280,1536,1792,2048,2304,2560,3072,4096,16384 --forms default --rounds 3 --ms 40 --out gpurun_out/default_sweep.jsonl > gpurun_out/default_sweep.log 2>&1
echo "rc=$?"
python3 - <<'PY'
import json
for l in open('gpurun_out/default_sweep.jsonl'):
    d = json.loads(l); print(d['B'], round(d['ms_per_pass'], 4), '%.3g' % d['node_evals_per_s'], d['kernel'][20:])
PY
for b in 256 512 2048; do timeout -k 10 300 python bench.py --batch $b --no-cpu-baseline > gpurun_out/bench_b$b.log 2>&1; tail -1 gpurun_out/bench_b$b.log | cut -c1-200; done
