#!/bin/bash
# after a change to the MFMA defect kernels: parity of every kernel variant, then the timings that show what it did
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "variant or large_batches or two_state or fast or symmetric or mfma" > gpurun_out/ring_check_pytest.log 2>&1 || { tail -30 gpurun_out/ring_check_pytest.log; exit 1; }
tail -2 gpurun_out/ring_check_pytest.log
timeout -k 10 300 python tools/small_batch_anatomy.py --out gpurun_out/small_batch_anatomy.json > gpurun_out/small_batch_anatomy.log 2>&1 || exit 1
grep -E "ablate 0" gpurun_out/small_batch_anatomy.log
rm -f gpurun_out/pv_ring.jsonl
for b in 128 256 512 1024; do
  timeout -k 10 300 python tools/pass_variants.py --batch $b --rounds 3 --steps 200 --no-profile --out gpurun_out/pv_ring.jsonl \
     --only default,one_launch_sw1,one_launch_sw2,ring2_sw2_conc_nt,ring2_sw6_conc_nt,ring_bk16_conc_nt,ring2_sw1_conc_nt,ring2_auto_sequential 2>&1 | grep -v amdgpu.ids || exit 1
done
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_ring.log 2>&1 || exit 1
tail -1 gpurun_out/bench_ring.log | cut -c1-600
