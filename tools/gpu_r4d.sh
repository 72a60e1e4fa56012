#!/bin/bash
# round 4, fourth GPU call: K tiles of 16 (parity, then timing against the 8-deep form over the batch range), refined solves on the
# device, Monte-Carlo sets with the batcher in step (flush 20 ms)
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kkt.py -m gpu -q -x -k "deep_k or refined or batched or default_dispatch" > gpurun_out/pytest_r4d.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/pytest_r4d.log
rm -f gpurun_out/mid_sweep_r4d.jsonl
timeout -k 10 300 python tools/mid_sweep.py --batches 64,128,192,256 --rounds 5 --out gpurun_out/mid_sweep_r4d.jsonl \
  --forms default,sw1_bk16,sw2_bk16,sw1_bk16_first,sw2_bk16_first,sw1_bk8_node_off,sw1_bk16_node_off,sw2_bk8_node_off,sw2_bk16_node_off 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/mid_sweep.py --batches 512,1024,2048,4096 --rounds 5 --out gpurun_out/mid_sweep_r4d.jsonl \
  --forms default,bk16,sw2_bk16_inter,sw2_bk8_node_off,sw2_bk16_node_off 2>&1 | grep -v amdgpu.ids
: > gpurun_out/mc_r4d.jsonl
EMI_MC_GATHER=0 timeout -k 10 200 etol_amd/lib/etol_mi355x_montecarlo 8 64 6 8 > /dev/null 2>&1
run() {   # scenarios nsteps discs threads groups
  EMI_MC_BATCH=$5 EMI_MC_GATHER=0 timeout -k 10 200 etol_amd/lib/etol_mi355x_montecarlo $1 $2 $3 $4 > gpurun_out/mc_r4d_$2_t$4_g$5.log 2>&1
  echo "mc $* rc=$?"; grep -E "^batcher" gpurun_out/mc_r4d_$2_t$4_g$5.log | head -4 | cut -c1-200; tail -1 gpurun_out/mc_r4d_$2_t$4_g$5.log | tee -a gpurun_out/mc_r4d.jsonl
}
run 64 1023 20 8 0
run 64 1023 20 32 2
run 64 1023 20 48 3
run 64 1023 20 64 2
run 64 256 10 8 0
run 64 256 10 32 2
run 64 128 10 32 2
run 32 512 20 32 2
