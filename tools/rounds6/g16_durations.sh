#!/bin/bash
# round 6, call 16: where the -m gpu suite spends its time (the driver's step has a 1200 s limit)
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 --durations=25 > gpurun_out/r06_pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -A85 "slowest" gpurun_out/r06_pytest_gpu.log | cut -c1-200; tail -3 gpurun_out/r06_pytest_gpu.log
