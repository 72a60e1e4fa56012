#!/bin/bash
# ring depth of the one-launch pass (3 or 4 stages = 2 or 3 K tiles in flight), wall time per pass
mkdir -p gpurun_out; rm -f gpurun_out/pv_nst.jsonl
for b in ${BATCHES:-1024 768 128}; do
  timeout -k 10 300 python tools/pass_variants.py --batch $b --rounds 5 --steps 200 --no-profile --out gpurun_out/pv_nst.jsonl \
     --only default,one_launch_sw2,one_launch_sw2_nst4,one_launch_sw1_ks1,one_launch_sw1_nst4 2>&1 | grep -v amdgpu.ids || exit 1
done
