#!/bin/bash
# config-5 bench with the main build and every build under _alt/ on this box
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
for d in . $(ls -d _alt/*/ 2>/dev/null); do
  (cd $R/$d && timeout -k 10 200 python bench.py --config c5 --steps 50 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('$d', 'pass %.4f ms' % j['ms_per_step'], 'kernel %.4f ms' % r['avg_ms'], 'frac %.3f' % r['frac'])") || exit 1
done
