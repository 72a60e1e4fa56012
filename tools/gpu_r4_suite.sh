#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q > gpurun_out/pytest_final.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_final.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
