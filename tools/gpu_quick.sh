#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1
rc=$?
tail -5 gpurun_out/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
bash tools/gpu_ablate.sh
