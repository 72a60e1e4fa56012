#!/bin/bash
# measurement set of a round (tests are run separately by tools/gpu_tests.sh): bench lines, rocprof kernel stats,
# PMC traffic and busy counters, variant A/B, batch sweep, probes.  Everything lands in gpurun_out/.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py > gpurun_out/bench.log 2>&1 || { tail gpurun_out/bench.log; exit 1; }
tail -1 gpurun_out/bench.log | cut -c1-400
timeout -k 10 300 python bench.py --config c5 --steps 50 --warmup 5 > gpurun_out/bench_c5.log 2>&1; tail -1 gpurun_out/bench_c5.log | cut -c1-300
for b in 128 256 512; do timeout -k 10 200 python bench.py --batch $b --no-cpu-baseline > gpurun_out/bench_b$b.log 2>&1; tail -1 gpurun_out/bench_b$b.log | cut -c1-200; done
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof $R/gpurun_out/prof_c5
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > $R/gpurun_out/prof.log 2>&1
echo "rocprof c3 rc=$?"; cat $R/gpurun_out/prof/*/*kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c5 -- python $R/bench.py --config c5 --steps 30 --warmup 5 > $R/gpurun_out/prof_c5.log 2>&1
echo "rocprof c5 rc=$?"; cat $R/gpurun_out/prof_c5/*/*kernel_stats.csv
cd $R && bash tools/pmc_traffic.sh "1024 128" > gpurun_out/pmc_traffic.log 2>&1; tail -2 gpurun_out/pmc_traffic.log
bash tools/pmc_busy.sh > gpurun_out/pmc_busy.log 2>&1; tail -3 gpurun_out/pmc_busy.log | cut -c1-200
rm -f gpurun_out/pass_variants.jsonl gpurun_out/pass_variants_noprof.jsonl
for b in 1024 512 256 128; do
  timeout -k 10 300 python tools/pass_variants.py --batch $b --rounds 3 --steps 200 --only default,ring_bk16_concurrent,ring_bk16_conc_nt,ring2_sw6_conc_nt,ring2_sw2_conc_nt,ring2_sw1_conc_nt,ring_bk16_sequential,ring2_auto_sequential,one_launch_sw2,one_launch_sw1,general_sequential 2>&1 | tail -12
  timeout -k 10 300 python tools/pass_variants.py --batch $b --rounds 3 --steps 300 --no-profile --out gpurun_out/pass_variants_noprof.jsonl --only default,ring_bk16_concurrent,ring_bk16_conc_nt,ring2_sw6_conc_nt,ring2_sw2_conc_nt,ring2_sw1_conc_nt,ring_bk16_sequential,ring2_auto_sequential,one_launch_sw2,one_launch_sw1,general_sequential 2>&1 | tail -12
done
(timeout -k 5 120 etol_amd/lib/clock_probe 1; timeout -k 5 120 etol_amd/lib/clock_probe 2; timeout -k 5 120 etol_amd/lib/store_probe; timeout -k 5 200 etol_amd/lib/mfma_mix_probe) > gpurun_out/probes.jsonl 2>&1
timeout -k 10 600 python tools/batch_sweep.py > gpurun_out/batch_sweep.log 2>&1; tail -3 gpurun_out/batch_sweep.log
