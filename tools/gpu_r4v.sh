#!/bin/bash
# round 4, call 26: write-through stores below the non-temporal range -- parity, the default sweep, the 128-instance shard again (bench line, rocprofv3, PMC)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_delays.py -m gpu -q -x > gpurun_out/pytest_r4v.log 2>&1
echo "pytest rc=$?"; tail -2 gpurun_out/pytest_r4v.log | cut -c1-200
rm -f gpurun_out/default_sweep_sc1.jsonl
timeout -k 10 400 python tools/mid_sweep.py --batches 1,4,16,32,48,64,80,96,112,128,144,160,192,224,256 --forms default,plain --rounds 5 --ms 40 --out gpurun_out/default_sweep_sc1.jsonl > gpurun_out/default_sweep_sc1.log 2>&1
python3 - <<'PY'
import json, collections
t = collections.defaultdict(dict)
for l in open('gpurun_out/default_sweep_sc1.jsonl'):
    d = json.loads(l); t[d['B']][d['form']] = round(d['ms_per_pass'], 4)
for b in sorted(t): print(b, t[b])
PY
timeout -k 10 300 python bench.py --batch 128 --no-cpu-baseline > gpurun_out/bench_b128.log 2>&1; tail -1 gpurun_out/bench_b128.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_b128
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b128 -- python $R/bench.py --steps 200 --warmup 20 --batch 128 --no-cpu-baseline > $R/gpurun_out/prof_b128.log 2>&1
echo "rc=$?"; head -3 $R/gpurun_out/prof_b128/*/*kernel_stats.csv | cut -c1-200
cd $R
PMC_OUT=pmc_traffic_128.json bash tools/pmc_traffic.sh "128" > gpurun_out/pmc_traffic_128.log 2>&1; grep "fetch" gpurun_out/pmc_traffic_128.log
