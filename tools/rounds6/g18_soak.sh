#!/bin/bash
# round 6, call 18: soak run of the route fuzzer (400 single layers, 100 sibling groups)
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
QLLM_FUZZ_SINGLE=400 QLLM_FUZZ_GROUP=100 timeout 1500 python -m pytest tests/test_fuzz_routes_gpu.py -m gpu -q --timeout 900 > gpurun_out/r06y_soak.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r06y_soak.log | cut -c1-600
