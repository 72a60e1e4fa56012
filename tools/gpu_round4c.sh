#!/bin/bash
# Round-4 measurement set of the FINAL build, part 1 (everything lands in gpurun_out/, one progress line per step): GPU suite, bench lines, rocprofv3
# kernel statistics (config 3 headline, its 128-instance shard, config 2, config 5), the default dispatch over the batch range
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
say() { echo "[$(date +%T)] $*"; }
say tests; timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/pytest_round4.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_round4.log
say bench; timeout -k 10 500 python bench.py > gpurun_out/bench.log 2>&1 || { tail gpurun_out/bench.log; }
tail -1 gpurun_out/bench.log | cut -c1-300
for b in 128 256 512 2048 4096 16384; do say "bench --batch $b"; timeout -k 10 300 python bench.py --batch $b --no-cpu-baseline > gpurun_out/bench_b$b.log 2>&1; tail -1 gpurun_out/bench_b$b.log | cut -c1-160; done
say "bench c2"; timeout -k 10 300 python bench.py --nodes 256 --obstacles 0 --batch 1024 --no-cpu-baseline > gpurun_out/bench_c2.log 2>&1; tail -1 gpurun_out/bench_c2.log | cut -c1-200
say "bench c5"; timeout -k 10 300 python bench.py --config c5 --steps 50 --warmup 5 > gpurun_out/bench_c5.log 2>&1; tail -1 gpurun_out/bench_c5.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof $R/gpurun_out/prof_c5 $R/gpurun_out/prof_b128 $R/gpurun_out/prof_c2
say "rocprof c3"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > $R/gpurun_out/prof.log 2>&1
echo "rc=$?"; head -4 $R/gpurun_out/prof/*/*kernel_stats.csv | cut -c1-200
say "rocprof c3 B=128"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b128 -- python $R/bench.py --steps 200 --warmup 20 --batch 128 --no-cpu-baseline > $R/gpurun_out/prof_b128.log 2>&1
echo "rc=$?"; head -3 $R/gpurun_out/prof_b128/*/*kernel_stats.csv | cut -c1-200
say "rocprof c2"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c2 -- python $R/bench.py --steps 100 --warmup 10 --nodes 256 --obstacles 0 --batch 1024 --no-cpu-baseline > $R/gpurun_out/prof_c2.log 2>&1
echo "rc=$?"; head -3 $R/gpurun_out/prof_c2/*/*kernel_stats.csv | cut -c1-200
say "rocprof c5"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c5 -- python $R/bench.py --config c5 --steps 30 --warmup 5 > $R/gpurun_out/prof_c5.log 2>&1
echo "rc=$?"; head -4 $R/gpurun_out/prof_c5/*/*kernel_stats.csv | cut -c1-200
cd $R
say "default sweep"; rm -f gpurun_out/default_sweep.jsonl; timeout -k 10 500 python tools/mid_sweep.py --batches 1,4,16,64,128,192,256,320,384,448,512,576,640,704,768,896,1024,2048,4096,16384 --forms default --rounds 3 --ms 40 --out gpurun_out/default_sweep.jsonl > gpurun_out/default_sweep.log 2>&1; grep -c . gpurun_out/default_sweep.jsonl
say done
