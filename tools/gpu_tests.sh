#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1
rc=$?
tail -30 gpurun_out/pytest.log
exit $rc
