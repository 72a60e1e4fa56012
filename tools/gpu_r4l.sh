#!/bin/bash
# round 4, call 13: epilogue prefetch ahead of the K loop + two waves per SIMD stated -- parity, then A/B against builds without them on this box
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_traced.py -m gpu -q -x > gpurun_out/pytest_r4l.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/pytest_r4l.log | cut -c1-300
for d in . _alt/nopre _alt/noprefree . _alt/nopre; do
  n=$(basename $d); [ "$n" = "." ] && n=main
  echo "== build $n $(date +%T)"
  (cd $R/$d && timeout -k 10 400 python tools/mid_sweep.py --batches 64,128,192,256,512,1024,2048,4096 --forms default --rounds 5 --ms 40 \
      --out $R/gpurun_out/ab_r4l_$n.jsonl > $R/gpurun_out/ab_r4l_$n.log 2>&1) || { tail -5 $R/gpurun_out/ab_r4l_$n.log; exit 1; }
done
python3 - <<'PY'
import json, collections
t = collections.defaultdict(dict)
for n in ("main", "nopre", "noprefree"):
    rows = collections.defaultdict(list)
    for l in open(f"gpurun_out/ab_r4l_{n}.jsonl"):
        d = json.loads(l); rows[d["B"]].append(d["ms_per_pass"])
    for b, v in rows.items(): t[b][n] = v
for b in sorted(t):
    print(b, {n: [round(x, 4) for x in v] for n, v in t[b].items()})
PY
