#!/bin/bash
# round 3, first call: the whole GPU test suite, the default bench line (with `checked` and the config-5 secondary block),
# then the build-time A/B of the ticket order and the m0 wait state (tools/ab_build.sh relaxed / nonop)
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1
rc=$?
tail -15 gpurun_out/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 500 python bench.py > gpurun_out/bench.log 2>&1 || { tail gpurun_out/bench.log; exit 1; }
tail -1 gpurun_out/bench.log | cut -c1-1500
bash tools/gpu_ab.sh "1024 128" "default" 2>&1 | tail -30
