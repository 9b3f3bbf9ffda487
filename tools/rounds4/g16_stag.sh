#!/bin/bash
# round 4: gemm3 with the activation piece of half the matrix waves requested between the MFMA halves of a sub-step (QLLM_GEMM3_STAG)
tag=${1:-r04p}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
export LD_LIBRARY_PATH=$R/tools/lab:$LD_LIBRARY_PATH
timeout 300 tools/lab/g4lab check > gpurun_out/${tag}_check.log 2>&1; echo "check rc=$?"; grep -c bit-exact gpurun_out/${tag}_check.log; grep MISMATCH gpurun_out/${tag}_check.log | head
timeout 300 tools/lab/g4lab time 2048 native > gpurun_out/${tag}_time_native.log 2>&1; cat gpurun_out/${tag}_time_native.log
