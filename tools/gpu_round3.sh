#!/bin/bash
# Round-3 measurement set (everything lands in gpurun_out/, one progress line per step).  Tests run separately (tools/gpu_tests.sh).
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
say() { echo "[$(date +%T)] $*"; }
say bench; timeout -k 10 500 python bench.py > gpurun_out/bench.log 2>&1 || { tail gpurun_out/bench.log; exit 1; }
tail -1 gpurun_out/bench.log | cut -c1-300
for b in 128 256 512 2048 16384; do say "bench --batch $b"; timeout -k 10 300 python bench.py --batch $b --no-cpu-baseline > gpurun_out/bench_b$b.log 2>&1; tail -1 gpurun_out/bench_b$b.log | cut -c1-160; done
say "bench c5"; timeout -k 10 300 python bench.py --config c5 --steps 50 --warmup 5 > gpurun_out/bench_c5.log 2>&1; tail -1 gpurun_out/bench_c5.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof $R/gpurun_out/prof_c5 $R/gpurun_out/prof_b128
say "rocprof c3"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > $R/gpurun_out/prof.log 2>&1
echo "rc=$?"; head -4 $R/gpurun_out/prof/*/*kernel_stats.csv | cut -c1-200
say "rocprof c3 B=128"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b128 -- python $R/bench.py --steps 200 --warmup 20 --batch 128 --no-cpu-baseline > $R/gpurun_out/prof_b128.log 2>&1
echo "rc=$?"; head -3 $R/gpurun_out/prof_b128/*/*kernel_stats.csv | cut -c1-200
say "rocprof c5"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c5 -- python $R/bench.py --config c5 --steps 30 --warmup 5 > $R/gpurun_out/prof_c5.log 2>&1
echo "rc=$?"; head -4 $R/gpurun_out/prof_c5/*/*kernel_stats.csv | cut -c1-200
cd $R
say "pmc traffic"; bash tools/pmc_traffic.sh "1024 128 16384" > gpurun_out/pmc_traffic.log 2>&1; grep "fetch" gpurun_out/pmc_traffic.log
say "pmc busy"; bash tools/pmc_busy.sh > gpurun_out/pmc_busy.log 2>&1; grep "rc=" gpurun_out/pmc_busy.log | tr '\n' ' '; echo
say "batch sweep"; timeout -k 10 600 python tools/batch_sweep.py > gpurun_out/batch_sweep.log 2>&1; tail -3 gpurun_out/batch_sweep.log | cut -c1-200
say "mid sweep"; rm -f gpurun_out/mid_sweep.jsonl; timeout -k 10 500 python tools/mid_sweep.py --batches 1,4,16,64,128,192,256,320,384,448,512,576,640,704,768,896,1024,2048,4096,16384 --forms default --rounds 3 --ms 40 > gpurun_out/mid_sweep.log 2>&1; tail -2 gpurun_out/mid_sweep.log | cut -c1-100
say "solve times"; timeout -k 10 300 python tools/solve_times.py > gpurun_out/solve_times.log 2>&1; tail -8 gpurun_out/solve_times.log | cut -c1-160
export EMI_MC_GATHER=0
: > gpurun_out/montecarlo.jsonl
timeout -k 5 60 etol_amd/lib/etol_mi355x_montecarlo 4 64 6 4 > /dev/null 2>&1
for cfg in "8 1023 20 4" "64 1023 20 8" "32 512 20 8" "64 256 10 8" "64 128 10 8"; do
  set -- $cfg
  say "montecarlo $cfg"; timeout -k 10 400 etol_amd/lib/etol_mi355x_montecarlo $1 $2 $3 $4 > gpurun_out/mc_$1_$2.log 2>&1
  tail -1 gpurun_out/mc_$1_$2.log | tee -a gpurun_out/montecarlo.jsonl | cut -c1-220
done
say done
