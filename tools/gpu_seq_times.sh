#!/bin/bash
# stand-alone kernel times (HIP events, kernels back to back on one stream) of the MFMA defect kernel forms
mkdir -p gpurun_out; rm -f gpurun_out/pv_seq.jsonl
for b in 1024 512 128; do
  timeout -k 10 300 python tools/pass_variants.py --batch $b --rounds 3 --steps 200 --out gpurun_out/pv_seq.jsonl \
     --only ring_bk16_sequential,ring2_sw6_sequential,ring2_sw2_sequential,ring2_sw1_sequential,ring2_sw3_sequential 2>&1 | grep -v amdgpu.ids || exit 1
done
