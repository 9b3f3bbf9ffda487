#!/bin/bash
# round 4, GPU call 9: gemm3 with 8 dequant waves (two packed words per thread)
tag=${1:-r04i}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
export LD_LIBRARY_PATH=$R/tools/lab:$LD_LIBRARY_PATH
timeout 400 tools/lab/g4lab check > gpurun_out/${tag}_check.log 2>&1; echo "check rc=$?"; tail -2 gpurun_out/${tag}_check.log; grep -c bit-exact gpurun_out/${tag}_check.log; grep MISMATCH gpurun_out/${tag}_check.log | head -20
timeout 600 tools/lab/g4lab time 2048 native > gpurun_out/${tag}_time_native.log 2>&1; cat gpurun_out/${tag}_time_native.log
timeout 600 tools/lab/g4lab time 2048 gptq > gpurun_out/${tag}_time_gptq.log 2>&1; cat gpurun_out/${tag}_time_gptq.log
