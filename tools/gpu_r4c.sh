#!/bin/bash
# round 4, third GPU call: the batched Newton step -- its tests, then Monte-Carlo sets with shared launches (EMI_MC_BATCH groups)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kkt.py tests/test_gpu_solve.py tests/test_gpu_delays.py -m gpu -q -x > gpurun_out/pytest_r4c.log 2>&1
echo "pytest rc=$?"; tail -25 gpurun_out/pytest_r4c.log
: > gpurun_out/mc_r4c.jsonl
EMI_MC_GATHER=0 timeout -k 10 200 etol_amd/lib/etol_mi355x_montecarlo 8 64 6 8 > /dev/null 2>&1
run() {   # scenarios nsteps discs threads groups
  EMI_MC_BATCH=$5 EMI_MC_GATHER=0 timeout -k 10 300 etol_amd/lib/etol_mi355x_montecarlo $1 $2 $3 $4 > gpurun_out/mc_r4c_$2_t$4_g$5.log 2>&1
  echo "mc $* rc=$?"; grep -E "^batcher" gpurun_out/mc_r4c_$2_t$4_g$5.log | head -4; tail -1 gpurun_out/mc_r4c_$2_t$4_g$5.log | tee -a gpurun_out/mc_r4c.jsonl
}
run 64 128 10 8 0
run 64 128 10 16 1
run 64 128 10 32 2
run 64 256 10 16 1
run 64 256 10 32 2
run 32 512 20 16 1
run 32 512 20 32 2
run 64 1023 20 16 1
run 64 1023 20 32 2
run 64 1023 20 32 4
run 64 1023 20 64 2
