#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_solve.py -m gpu -q -x > gpurun_out/pytest_solve.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/pytest_solve.log | cut -c1-200
MC=$GRAFT_REPO_ROOT/etol_amd/lib/etol_mi355x_montecarlo
export EMI_MC_GATHER=0 EMI_MC_RUNS=1
for plan in 1 0; do
  EMI_MC_PLAN=$plan EMI_MC_ONLY=558 timeout -k 10 300 $MC 1024 1023 20 1 > gpurun_out/mc_558_plan$plan.log 2>&1
  echo "plan $plan rc=$?"; grep "^scenario" gpurun_out/mc_558_plan$plan.log | cut -c1-500
done
