#!/bin/bash
# round 4, call 11: new Newton-step test, ladder ratio, bench line with this round's PMC traffic
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kkt.py -m gpu -q -x -s -k "primal_regularisation" > gpurun_out/pytest_r4j.log 2>&1
echo "pytest rc=$?"; grep -E "state curvature|shift dw|passed|failed|skipped|Error|assert" gpurun_out/pytest_r4j.log | cut -c1-200
: > gpurun_out/mc_r4j.jsonl
EMI_MC_GATHER=0 timeout -k 10 200 etol_amd/lib/etol_mi355x_montecarlo 8 64 6 8 > /dev/null 2>&1
run() {   # scenarios nsteps discs threads ratio
  EMI_MC_LADDER_RATIO=$5 EMI_MC_GATHER=0 timeout -k 10 200 etol_amd/lib/etol_mi355x_montecarlo $1 $2 $3 $4 > gpurun_out/mc_r4j_$2_t$4_l$5.log 2>&1
  echo "mc $* rc=$?"; tail -1 gpurun_out/mc_r4j_$2_t$4_l$5.log | sed "s/^{/{\"ladder_ratio\": $5, /" | tee -a gpurun_out/mc_r4j.jsonl
}
run 64 1023 20 8 2
run 64 1023 20 8 4
run 64 1023 20 8 3
run 32 512 20 8 2
run 32 512 20 8 4
timeout -k 10 300 python bench.py > gpurun_out/bench_final.log 2>&1; tail -1 gpurun_out/bench_final.log | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
r = j['roofline']; c5 = j['secondary']['c5']['roofline']
print('value', j['value'], 'ms', j['ms_per_step'], 'frac', r['frac'], 'traffic', r['traffic'], r.get('traffic_over_algorithmic'), r['traffic_source'])
print('c5', j['secondary']['c5']['value'], 'traffic', c5['traffic'], c5['traffic_source'], 'checked', j['checked']['ok'])"
