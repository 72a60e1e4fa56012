#!/bin/bash
# tile order of the MFMA role: column partitions per XCD group (sym_cpart), wall time per pass
mkdir -p gpurun_out; rm -f gpurun_out/pv_cpart.jsonl
for b in ${BATCHES:-1024 768 512 128}; do
  timeout -k 10 300 python tools/pass_variants.py --batch $b --rounds 5 --steps 200 --no-profile --out gpurun_out/pv_cpart.jsonl \
     --only default,one_launch_sw2_plain_order,one_launch_sw2_cpart2,one_launch_sw2_cpart4,one_launch_sw2_cpart8,one_launch_sw3_cpart4,one_launch_sw1_plain_order,one_launch_sw1_cpart4,one_launch_sw1_cpart8,sw2_conc_plain_order,sw2_conc_cpart4,sw6_conc_cpart4,ring_bk16_conc_nt 2>&1 | grep -v amdgpu.ids || exit 1
done
