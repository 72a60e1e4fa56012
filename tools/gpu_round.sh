#!/bin/bash
# full measurement set of a round: tests, smoke, bench, rocprof kernel stats, PMC traffic
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1 || { tail -20 gpurun_out/pytest.log; exit 1; }
tail -2 gpurun_out/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || { tail gpurun_out/smoke.log; exit 1; }
tail -2 gpurun_out/smoke.log
timeout -k 10 400 python bench.py > gpurun_out/bench.log 2>&1 || { tail gpurun_out/bench.log; exit 1; }
tail -1 gpurun_out/bench.log
EMI_OVERLAP=0 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_sequential.log 2>&1 || exit 1
tail -1 gpurun_out/bench_sequential.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1
echo "rocprof rc=$?"
cat $GRAFT_REPO_ROOT/gpurun_out/prof/*/*kernel_stats.csv
cd $GRAFT_REPO_ROOT && bash tools/pmc_traffic.sh
