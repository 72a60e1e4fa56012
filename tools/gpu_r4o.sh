#!/bin/bash
# round 4, call 17: Monte-Carlo set at the reference's own NLP tolerance (ePSOPT.cpp:67: 1e-6; the example ran at 1e-7), thread counts
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
MC=$GRAFT_REPO_ROOT/etol_amd/lib/etol_mi355x_montecarlo
EMI_MC_GATHER=0 timeout -k 10 200 $MC 8 64 6 8 > /dev/null 2>&1
: > $OUT/mc_r4o.jsonl
run() {   # scenarios nsteps discs threads tol
  EMI_MC_TOL=$5 EMI_MC_GATHER=0 timeout -k 10 300 $MC $1 $2 $3 $4 > $OUT/mc_r4o_$2_t$4_tol$5.log 2>&1
  echo "mc $* rc=$?"; tail -1 $OUT/mc_r4o_$2_t$4_tol$5.log | sed "s/^{/{\"nlp_tolerance\": $5, /" >> $OUT/mc_r4o.jsonl
  tail -1 $OUT/mc_r4o.jsonl | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d[k] for k in ('nlp_tolerance', 'threads', 'scenarios', 'solved', 'wall_s', 'solves_per_s', 'mean_iterations')}, {m: int(v['iterations']) for m, v in d['by_mesh'].items()})"
}
run 64 1023 20 8 1e-7
run 64 1023 20 8 1e-6
run 64 1023 20 12 1e-6
run 64 1023 20 6 1e-6
run 32 512 20 8 1e-6
