#!/bin/bash
run() { echo "== $*"; env "$@" timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']; print('ms/step', round(d['ms_per_step'],4), 'pass', round(r.get('pass_ms',0),4), r['kernels_ms'])" || exit 1; }
for ab in 0 1 2 3 4 6 7; do run EMI_SYM_CT=3 EMI_OVERLAP_MODE=1 EMI_SYM_ABLATE=$ab; done
