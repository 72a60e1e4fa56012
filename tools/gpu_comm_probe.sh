#!/bin/bash
# two Monte-Carlo ranks at once on ONE GPU: does RCCL form a 2-rank communicator on a shared device?
# (diagnostic; a refusal is expected -- "Duplicate GPU detected" -- and is not an error of the build)
mkdir -p gpurun_out
export MASTER_PORT=29611 EMI_COMM_FILE=/tmp/emi_comm_probe.id
rm -f $EMI_COMM_FILE
for r in 0 1; do
  RANK=$r WORLD_SIZE=2 LOCAL_RANK=0 timeout -k 5 120 etol_amd/lib/etol_mi355x_montecarlo 4 40 2 1 > gpurun_out/comm_probe_$r.log 2>&1 &
done
wait
tail -3 gpurun_out/comm_probe_0.log gpurun_out/comm_probe_1.log
