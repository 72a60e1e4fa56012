#!/bin/bash
# last call of the round: parity of the default dispatch after the store-mode detail, the small sizes, PMC traffic of the headline with the final build, bench line
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "default_dispatch" > gpurun_out/pytest_last.log 2>&1
echo "pytest rc=$?"; tail -2 gpurun_out/pytest_last.log | cut -c1-200
rm -f gpurun_out/default_sweep_last.jsonl
timeout -k 10 300 python tools/mid_sweep.py --batches 16,32,48,64,80 --forms default,sc1,plain --rounds 5 --ms 40 --out gpurun_out/default_sweep_last.jsonl > gpurun_out/default_sweep_last.log 2>&1
python3 - <<'PY'
import json, collections
t = collections.defaultdict(dict)
for l in open('gpurun_out/default_sweep_last.jsonl'):
    d = json.loads(l); t[d['B']][d['form']] = round(d['ms_per_pass'], 4)
for b in sorted(t): print(b, t[b])
PY
PMC_OUT=pmc_traffic_1024.json bash tools/pmc_traffic.sh "1024" > gpurun_out/pmc_traffic_1024.log 2>&1; grep "fetch" gpurun_out/pmc_traffic_1024.log
timeout -k 10 500 python bench.py > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-260
for b in 1 64; do timeout -k 10 300 python bench.py --batch $b --no-cpu-baseline > gpurun_out/bench_b$b.log 2>&1; tail -1 gpurun_out/bench_b$b.log | cut -c1-200; done
