#!/bin/bash
# round 4, call 12: where a Monte-Carlo run's time goes -- hardware queues, GPU busy time from a kernel trace
mkdir -p gpurun_out
# the two-halves form of the MFMA role with both halves of a node-role workgroup at work: parity first, then the small batches
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "deep_k_tiles" > gpurun_out/pytest_r4k.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/pytest_r4k.log | cut -c1-200
timeout -k 10 300 python tools/mid_sweep.py --batches 64,128,192,256,384 --forms default,sw2_hs2,sw2_hs2_bk16,sw1_hs2,sw1_hs2_bk16 --rounds 5 \
   --out gpurun_out/mid_sweep_r4k.jsonl > gpurun_out/mid_sweep_r4k.log 2>&1
echo "sweep rc=$?"; python3 - <<'PY'
import json
for l in open('gpurun_out/mid_sweep_r4k.jsonl'):
    d = json.loads(l); print(d['B'], d['form'], round(d['ms_per_pass'], 4), '%.3e' % d['node_evals_per_s'])
PY
OUT=$GRAFT_REPO_ROOT/gpurun_out
MC=$GRAFT_REPO_ROOT/etol_amd/lib/etol_mi355x_montecarlo
: > $OUT/mc_r4k.jsonl
EMI_MC_GATHER=0 timeout -k 10 200 $MC 8 64 6 8 > /dev/null 2>&1
run() {   # scenarios nsteps discs threads queues(0 = runtime default)
  if [ "$5" != "0" ]; then export GPU_MAX_HW_QUEUES=$5; else unset GPU_MAX_HW_QUEUES; fi
  EMI_MC_GATHER=0 timeout -k 10 200 $MC $1 $2 $3 $4 > $OUT/mc_r4k_$2_t$4_q$5.log 2>&1
  echo "mc $* rc=$?"; tail -1 $OUT/mc_r4k_$2_t$4_q$5.log | sed "s/^{/{\"hw_queues\": $5, /" | tee -a $OUT/mc_r4k.jsonl
  unset GPU_MAX_HW_QUEUES
}
run 64 1023 20 8 0
run 64 1023 20 8 8
run 64 1023 20 8 16
run 64 1023 20 16 16
run 64 1023 20 4 0
# kernel trace of a 16-scenario run with 8 threads: union of the kernel intervals against the span
rm -rf $OUT/mc8_trace
cd /tmp && export TMPDIR=/tmp EMI_MC_GATHER=0
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $OUT/mc8_trace -- $MC 16 1023 20 8 > $OUT/mc8_trace.log 2>&1 &
pid=$!
while kill -0 $pid 2>/dev/null; do sleep 20; echo "tracing ... $(date +%T)"; done
wait $pid
echo "trace rc=$?"; tail -1 $OUT/mc8_trace.log | cut -c1-300
f=$(ls $OUT/mc8_trace/*/*kernel_trace.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then
  ls -la $f
  python3 $GRAFT_REPO_ROOT/tools/trace_busy.py $f --top 25 --out $OUT/mc8_trace_busy.json
fi
rm -rf $OUT/mc8_trace
# the same for ONE thread (what a single solve looks like from the GPU)
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/mc1_trace -- $MC 4 1023 20 1 > $OUT/mc1_trace.log 2>&1 &
pid=$!
while kill -0 $pid 2>/dev/null; do sleep 20; echo "tracing ... $(date +%T)"; done
wait $pid
echo "trace rc=$?"; tail -1 $OUT/mc1_trace.log | cut -c1-300
f=$(ls $OUT/mc1_trace/*/*kernel_trace.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then
  python3 $GRAFT_REPO_ROOT/tools/trace_busy.py $f --top 12 --out $OUT/mc1_trace_busy.json
fi
rm -rf $OUT/mc1_trace
