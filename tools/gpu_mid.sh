#!/bin/bash
# middle batch sizes: which form of the pass wins where (wall time per pass)
mkdir -p gpurun_out; rm -f gpurun_out/pv_mid.jsonl
for b in ${BATCHES:-192 256 320 384 448 512 640}; do
  timeout -k 10 300 python tools/pass_variants.py --batch $b --rounds 5 --steps 200 --no-profile --out gpurun_out/pv_mid.jsonl \
     --only default,one_launch_sw1_ks1,one_launch_sw2,one_launch_sw3,ring_bk16_conc_nt,ring2_sw6_conc_nt 2>&1 | grep -v amdgpu.ids || exit 1
done
