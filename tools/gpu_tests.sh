#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1
rc=$?
tail -30 gpurun_out/pytest.log
[ $rc -ne 0 ] && exit $rc
# 2-rank rehearsal of bench.py on the one GPU: control plane over gloo, both ranks on device 0 (strong-scaled batch of
# 512 scenarios = 256 per rank, then the weak-scaled repeat)
EMI_BENCH_BACKEND=gloo EMI_BENCH_SHARE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
   --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 3 --scenarios 512 > gpurun_out/bench2.log 2>&1
echo "2-rank rehearsal rc=$?"; tail -3 gpurun_out/bench2.log | cut -c1-900
