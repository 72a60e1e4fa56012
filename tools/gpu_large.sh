#!/bin/bash
# large batches: two streams against the one-launch pass, wall time per pass (one process per batch size, interleaved rounds)
mkdir -p gpurun_out; rm -f gpurun_out/pv_large.jsonl
for b in ${BATCHES:-640 768 896 1024 1536 2048}; do
  timeout -k 10 300 python tools/pass_variants.py --batch $b --rounds 5 --steps 150 --no-profile --out gpurun_out/pv_large.jsonl \
     --only default,one_launch_sw2,one_launch_sw3,one_launch_sw6,ring2_sw2_conc_nt,ring2_sw6_conc_nt,ring_bk16_conc_nt,one_launch_sw2_plain 2>&1 | grep -v amdgpu.ids || exit 1
done
