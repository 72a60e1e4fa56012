#!/bin/bash
# Monte-Carlo example, few scenarios on one thread, iteration log + factorisation diagnostics: which iterations leave
# the Schur path and where the time of a solve goes once the libraries are warm (the first scenario loads them)
# usage: bash tools/gpu_mc_debug.sh [scenarios] [steps] [keep-outs]
N=${1:-2}; M=${2:-1023}; K=${3:-20}
mkdir -p gpurun_out
EMI_MC_KKT_DEBUG=${EMI_MC_KKT_DEBUG:-1} EMI_MC_PRINT_LEVEL=5 EMI_MC_GATHER=0 \
  timeout -k 10 400 etol_amd/lib/etol_mi355x_montecarlo $N $M $K 1 > gpurun_out/mc_debug_${N}_${M}.log 2>&1
echo "rc=$?"
grep -E "mesh sequencing|^time:|-> LU|^scenario" gpurun_out/mc_debug_${N}_${M}.log | cut -c1-200 | tail -60
