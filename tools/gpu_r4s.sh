#!/bin/bash
# round 4, call 23: the small-batch policy (no K slices above 64 instances) -- parity of the default dispatch, then the default sweep over the small sizes
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "default_dispatch" > gpurun_out/pytest_r4s.log 2>&1
echo "pytest rc=$?"; tail -2 gpurun_out/pytest_r4s.log | cut -c1-200
rm -f gpurun_out/default_sweep_small.jsonl
timeout -k 10 300 python tools/mid_sweep.py --batches 48,64,72,80,96,112,120,128,144 --forms default --rounds 3 --ms 40 --out gpurun_out/default_sweep_small.jsonl > gpurun_out/default_sweep_small.log 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/default_sweep_small.jsonl'):
    d = json.loads(l); print(d['B'], round(d['ms_per_pass'], 4), '%.3g' % d['node_evals_per_s'], d['kernel'][20:])
PY
