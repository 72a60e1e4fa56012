#!/bin/bash
# Is the Monte-Carlo solve rate bound by the launch path of ONE process?  The same 32 scenarios (1024 nodes) as one process with 8 / 16 host threads
# and as 4 processes (ranks of a world of 4 on the SAME GPU, no gather) with 2 / 4 threads each.  Diagnostics: the product is one process per GPU.
mkdir -p gpurun_out
export EMI_MC_GATHER=0
BIN=etol_amd/lib/etol_mi355x_montecarlo
say() { echo "[$(date +%T)] $*"; }
timeout -k 5 60 $BIN 4 64 6 4 > /dev/null 2>&1
for th in 8 16; do
  say "one process, $th threads"; t0=$(date +%s.%N)
  timeout -k 10 300 $BIN 32 1023 20 $th > gpurun_out/mcp_1x$th.log 2>&1
  t1=$(date +%s.%N); echo "1 x $th: $(python3 -c "print('%.2f s, %.3f solves/s' % ($t1-$t0, 32/($t1-$t0)))")"; tail -1 gpurun_out/mcp_1x$th.log | cut -c1-200
done
for th in 2 4; do
  say "four processes, $th threads each"; t0=$(date +%s.%N)
  pids=""
  for r in 0 1 2 3; do
    RANK=$r WORLD_SIZE=4 LOCAL_RANK=0 timeout -k 10 300 $BIN 32 1023 20 $th > gpurun_out/mcp_4x${th}_r$r.log 2>&1 &
    pids="$pids $!"
  done
  for p in $pids; do wait $p; done
  t1=$(date +%s.%N); echo "4 x $th: $(python3 -c "print('%.2f s, %.3f solves/s' % ($t1-$t0, 32/($t1-$t0)))")"
  for r in 0 1 2 3; do tail -1 gpurun_out/mcp_4x${th}_r$r.log | cut -c1-160; done
done
say done
