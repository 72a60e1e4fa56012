#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 200 python -m pytest tests/test_gpu_solve.py -m gpu -q > gpurun_out/pytest_solve.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/pytest_solve.log | cut -c1-300
