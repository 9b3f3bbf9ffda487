#!/bin/bash
# round 4, GPU call 8: gemm4 variants where the matrix waves request the activation tiles (paced one piece per sub-step)
tag=${1:-r04h}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
export LD_LIBRARY_PATH=$R/tools/lab:$LD_LIBRARY_PATH
timeout 400 tools/lab/g4lab check > gpurun_out/${tag}_check.log 2>&1; echo "check rc=$?"; tail -2 gpurun_out/${tag}_check.log; grep -c bit-exact gpurun_out/${tag}_check.log; grep MISMATCH gpurun_out/${tag}_check.log | head -20
timeout 600 tools/lab/g4lab time 2048 native > gpurun_out/${tag}_time_native.log 2>&1; cat gpurun_out/${tag}_time_native.log
for v in 5 6; do
  timeout 120 tools/lab/g4lab timeline $v 2048 4096 4096 > gpurun_out/${tag}_timeline_v$v.log 2>&1; tail -15 gpurun_out/${tag}_timeline_v$v.log | grep -v "entry after\|exit after\|block lifetime \[us\]"
done
