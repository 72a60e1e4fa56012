#!/bin/bash
# final build, after the band policy and the on-demand second stream: parity of the default dispatch, the band again, then part 2 of the set
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/pytest_r4e.log 2>&1
echo "pytest rc=$?"; tail -2 gpurun_out/pytest_r4e.log | cut -c1-200
rm -f gpurun_out/default_sweep_band.jsonl
timeout -k 10 300 python tools/mid_sweep.py --batches 256,272,288,320,352,384,416,448 --forms default --rounds 3 --ms 40 --out gpurun_out/default_sweep_band.jsonl > gpurun_out/default_sweep_band.log 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/default_sweep_band.jsonl'):
    d = json.loads(l); print(d['B'], round(d['ms_per_pass'], 4), '%.3g' % d['node_evals_per_s'], d['kernel'][-40:])
PY
bash $R/tools/gpu_round4d.sh
