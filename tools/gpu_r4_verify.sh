#!/bin/bash
# the planned route first as the default: the solve tests
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_solve.py tests/test_gpu_traced.py tests/test_gpu_delays.py -m gpu -q > gpurun_out/pytest_solve.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/pytest_solve.log | cut -c1-300
