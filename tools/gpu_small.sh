#!/bin/bash
# small batches (shards of config 4): wall time per pass of the forms that compete there
mkdir -p gpurun_out; rm -f gpurun_out/pv_small.jsonl
for b in ${BATCHES:-128 256}; do
  timeout -k 10 300 python tools/pass_variants.py --batch $b --rounds 5 --steps 300 --no-profile --out gpurun_out/pv_small.jsonl \
     --only default,one_launch_sw1_ks1,one_launch_sw2,one_launch_sw1_ks2,one_launch_sw2_ks2,one_launch_sw2_ks4,one_launch_sw3_ks2,one_launch_sw6_ks4,sw2_ks2_conc_ticket,sw6_ks4_conc_kernel,ring2_sw2_conc_nt 2>&1 | grep -v amdgpu.ids || exit 1
done
