#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1
rc=$?
tail -5 gpurun_out/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
run() { echo "== $*"; env "$@" timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']; print('ms/step', round(d['ms_per_step'],4), r['kernel'], 'avg_ms', round(r['avg_ms'],4), 'frac', round(r['frac'],3), r.get('other_kernel_ms'))" || exit 1; }
run EMI_SYM_CT=1 EMI_OVERLAP_MODE=2
run EMI_SYM_CT=1 EMI_OVERLAP_MODE=1
run EMI_SYM_CT=2 EMI_OVERLAP_MODE=2
run EMI_SYM_CT=2 EMI_OVERLAP_MODE=1
run EMI_OVERLAP=0
