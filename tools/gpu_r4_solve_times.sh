#!/bin/bash
# single-solve times of the final tree
mkdir -p gpurun_out
timeout -k 10 400 python tools/solve_times.py > gpurun_out/solve_times.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/solve_times.log | cut -c1-220
timeout -k 10 300 python tools/regress_probe.py > gpurun_out/regress_probe.log 2>&1; echo "rc=$?"; grep "scaling none" gpurun_out/regress_probe.log | cut -c1-200
