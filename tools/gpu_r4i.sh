#!/bin/bash
# round 4, ninth GPU call: two column sub-tiles per MFMA workgroup -- parity, then timing against the one-sub-tile forms
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "deep_k or default_dispatch or partitioned" > gpurun_out/pytest_r4i.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/pytest_r4i.log
rm -f gpurun_out/mid_sweep_r4i.jsonl
timeout -k 10 300 python tools/mid_sweep.py --batches 64,128,192,256 --rounds 5 --out gpurun_out/mid_sweep_r4i.jsonl \
  --forms default,sw1_hs2,sw1_hs2_bk16,sw2_hs2,sw2_hs2_bk16,sw1_hs2_node_off,sw2_hs2_node_off,sw2_hs2_bk16_node_off,abl1_mfma_off 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/mid_sweep.py --batches 256,512,1024,2048 --rounds 5 --out gpurun_out/mid_sweep_r4i.jsonl \
  --forms default,ct2,ct2_bk16,ct2_bk8,sw2_bk8_node_off,sw2_ct2_node_off,sw2_ct2_bk16_node_off 2>&1 | grep -v amdgpu.ids
timeout -k 10 400 python tools/mid_sweep.py --batches 4096,16384 --rounds 3 --ms 120 --out gpurun_out/mid_sweep_r4i.jsonl \
  --forms default,ct2,ct2_bk16,ct2_g2c1,ct2_g4c1,g4c2,sw2_bk16_g2c2 2>&1 | grep -v amdgpu.ids
timeout -k 10 200 python tools/mid_sweep.py --batches 1024 --rounds 3 --rotate 8 --out gpurun_out/mid_sweep_r4i.jsonl \
  --forms default,ct2,ct2_bk16,one_sw2_g2c2,ct2_g2c2 2>&1 | grep -v amdgpu.ids
