#!/bin/bash
mkdir -p gpurun_out
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1
rc=$?
tail -40 gpurun_out/pytest.log
exit $rc
