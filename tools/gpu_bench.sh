#!/bin/bash
# parity tests, then bench with the fused kernel and with the general path, then rocprof stats
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1
rc=$?
tail -15 gpurun_out/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --cpu-budget 4 > gpurun_out/bench.log 2>&1 || exit 1
tail -2 gpurun_out/bench.log
EMI_FUSED=0 timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench_general.log 2>&1 || exit 1
tail -1 gpurun_out/bench_general.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1
echo "rocprof rc=$?"
cat $GRAFT_REPO_ROOT/gpurun_out/prof/*/*kernel_stats.csv
