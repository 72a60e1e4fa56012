#!/bin/bash
# first GPU pass: parity tests, short bench, rocprof kernel stats
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1
rc=$?
tail -25 gpurun_out/pytest.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
timeout -k 10 300 python bench.py --steps 50 --warmup 5 --cpu-budget 5 > gpurun_out/bench.log 2>&1
brc=$?
tail -5 gpurun_out/bench.log
if [ $brc -ne 0 ]; then echo "bench rc=$brc: stopping"; exit $brc; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1
echo "rocprof rc=$?"
find $GRAFT_REPO_ROOT/gpurun_out/prof -name "*stats*" | head
