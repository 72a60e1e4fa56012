#!/bin/bash
# round 4, call 31: 16-deep K tiles up to 2048 instances in the policy (parity + sweep); wide tiles with 16-deep K tiles again under the final register allocation
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "default_dispatch or deep_k" > gpurun_out/pytest_r4z.log 2>&1
echo "pytest rc=$?"; tail -2 gpurun_out/pytest_r4z.log | cut -c1-200
rm -f gpurun_out/mid_sweep_r4z.jsonl
timeout -k 10 600 python tools/mid_sweep.py --batches 1536,2048,2304,4096,16384 --forms default,ct2_bk16,ct2_bk8 --rounds 3 --ms 60 --out gpurun_out/mid_sweep_r4z.jsonl > gpurun_out/mid_sweep_r4z.log 2>&1
echo "rc=$?"
python3 - <<'PY'
import json, collections
t = collections.defaultdict(dict)
for l in open('gpurun_out/mid_sweep_r4z.jsonl'):
    d = json.loads(l); t[d['B']][d['form']] = (round(d['ms_per_pass'], 4), d['kernel'][44:])
for b in sorted(t): print(b, t[b])
PY
