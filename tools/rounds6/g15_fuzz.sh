#!/bin/bash
# round 6, call 15: seeded random walk over the planner's routes through the modules
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_fuzz_routes_gpu.py -m gpu -q --timeout 900 > gpurun_out/r06w_fuzz.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/r06w_fuzz.log | cut -c1-400
