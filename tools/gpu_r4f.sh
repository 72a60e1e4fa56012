#!/bin/bash
# round 4, sixth GPU call: batched factorisation with per-scenario rocBLAS calls on large meshes against the *_batched forms, the
# batcher without polling, rung patience against stragglers
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kkt.py -m gpu -q -x -k "deep_k or refined or batched or default_dispatch" > gpurun_out/pytest_r4f.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/pytest_r4f.log
rm -f gpurun_out/kkt_times.jsonl
timeout -k 10 200 python tools/kkt_times.py --nodes 1024 --batch 1,8,32 --lowrank 770 2>&1 | grep -v amdgpu.ids | cut -c1-400
timeout -k 10 200 python tools/kkt_times.py --nodes 1024 --batch 1,8 --lowrank 770 --opt kkt_batch_syrk_rows=100000 2>&1 | grep batched | cut -c1-400
timeout -k 10 200 python tools/kkt_times.py --nodes 1024 --batch 1,8 --lowrank 770 --opt kkt_batch_gemm_rows=100000 2>&1 | grep batched | cut -c1-400
timeout -k 10 200 python tools/kkt_times.py --nodes 1024 --batch 1,8 --lowrank 770 --opt kkt_batch_trtri_rows=100000 2>&1 | grep batched | cut -c1-400
timeout -k 10 100 python tools/kkt_times.py --nodes 513 --batch 1,8,32 --lowrank 300 2>&1 | grep -v amdgpu.ids | cut -c1-400
timeout -k 10 100 python tools/kkt_times.py --nodes 513 --batch 8 --lowrank 300 --opt kkt_batch_syrk_rows=100,kkt_batch_gemm_rows=100,kkt_batch_trtri_rows=100 2>&1 | grep batched | cut -c1-400
: > gpurun_out/mc_r4f.jsonl
EMI_MC_GATHER=0 timeout -k 10 200 etol_amd/lib/etol_mi355x_montecarlo 8 64 6 8 > /dev/null 2>&1
run() {   # scenarios nsteps discs threads groups patience
  EMI_MC_RUNG_PATIENCE=$6 EMI_MC_BATCH=$5 EMI_MC_GATHER=0 timeout -k 10 200 etol_amd/lib/etol_mi355x_montecarlo $1 $2 $3 $4 > gpurun_out/mc_r4f_$2_t$4_g$5_p$6.log 2>&1
  echo "mc $* rc=$?"; grep -E "^batcher" gpurun_out/mc_r4f_$2_t$4_g$5_p$6.log | head -4 | cut -c1-200; tail -1 gpurun_out/mc_r4f_$2_t$4_g$5_p$6.log | tee -a gpurun_out/mc_r4f.jsonl
}
run 64 256 10 8 0 0
run 64 256 10 16 1 0
run 64 256 10 32 1 0
run 64 256 10 64 1 0
run 64 1023 20 8 0 0
run 64 1023 20 8 0 100
run 64 1023 20 16 0 100
run 64 1023 20 16 1 100
run 64 1023 20 32 1 100
run 64 1023 20 32 2 100
