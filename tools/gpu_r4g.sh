#!/bin/bash
# round 4, seventh GPU call: ladder knobs (rung patience / tolerance, thread count) on the 64 x 1024-node set
mkdir -p gpurun_out
: > gpurun_out/mc_r4g.jsonl
EMI_MC_GATHER=0 timeout -k 10 200 etol_amd/lib/etol_mi355x_montecarlo 8 64 6 8 > /dev/null 2>&1
run() {   # scenarios nsteps discs threads patience rungtol budget
  EMI_MC_BUDGET=$7 EMI_MC_RUNG_TOL=$6 EMI_MC_RUNG_PATIENCE=$5 EMI_MC_GATHER=0 timeout -k 10 200 etol_amd/lib/etol_mi355x_montecarlo $1 $2 $3 $4 > gpurun_out/mc_r4g_$2_t$4_p$5_r$6_b$7.log 2>&1
  echo "mc $* rc=$?"; tail -1 gpurun_out/mc_r4g_$2_t$4_p$5_r$6_b$7.log | sed "s/^{/{\"rung_patience\": $5, \"rung_tolerance\": $6, \"budget\": $7, /" | tee -a gpurun_out/mc_r4g.jsonl
}
run 64 1023 20 8 100 1e-6 0
run 64 1023 20 8 60 1e-6 0
run 64 1023 20 8 80 1e-6 0
run 64 1023 20 8 150 1e-6 0
run 64 1023 20 8 100 1e-4 0
run 64 1023 20 8 100 1e-3 0
run 64 1023 20 12 100 1e-4 0
run 64 1023 20 8 100 1e-4 400
run 32 512 20 8 100 1e-4 0
run 64 256 10 8 100 1e-4 0
run 64 256 10 16 100 1e-4 0
