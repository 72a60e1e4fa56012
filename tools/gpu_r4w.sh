#!/bin/bash
# round 4, call 27: how far the device refinement of a Newton step has to go (kkt_refine_exp: stop below 10^-e of the right-hand side)
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
MC=$GRAFT_REPO_ROOT/etol_amd/lib/etol_mi355x_montecarlo
export EMI_MC_GATHER=0
timeout -k 10 200 $MC 8 64 6 8 > /dev/null 2>&1
: > $OUT/mc_r4w.jsonl
run() {   # scenarios nsteps discs threads exp
  EMI_MC_KKT_REFINE_EXP=$5 timeout -k 10 300 $MC $1 $2 $3 $4 > $OUT/mc_r4w_$2_e$5.log 2>&1
  echo "mc $* rc=$?"; tail -1 $OUT/mc_r4w_$2_e$5.log | sed "s/^{/{\"kkt_refine_exp\": $5, /" >> $OUT/mc_r4w.jsonl
  tail -1 $OUT/mc_r4w.jsonl | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d[k] for k in ('kkt_refine_exp', 'solved', 'wall_s', 'solves_per_s', 'mean_iterations')}, {m: (int(v['iterations']), round(v['solves'] / v['iterations'], 2)) for m, v in d['by_mesh'].items()})"
}
run 64 1023 20 8 14
run 64 1023 20 8 12
run 64 1023 20 8 10
run 64 1023 20 8 8
run 32 512 20 8 14
run 32 512 20 8 10
