#!/bin/bash
# round 4, GPU call 3: gemm4 with separate loader / dequant / matrix waves: check, timing, timeline
tag=${1:-r04c}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
export LD_LIBRARY_PATH=$R/tools/lab:$LD_LIBRARY_PATH
timeout 400 tools/lab/g4lab check > gpurun_out/${tag}_check.log 2>&1; echo "check rc=$?"; tail -2 gpurun_out/${tag}_check.log; grep -c bit-exact gpurun_out/${tag}_check.log; grep MISMATCH gpurun_out/${tag}_check.log | head -20
timeout 300 tools/lab/g4lab time 2048 native > gpurun_out/${tag}_time_native.log 2>&1; cat gpurun_out/${tag}_time_native.log
for v in 2 3; do
  timeout 120 tools/lab/g4lab timeline $v 2048 4096 4096 > gpurun_out/${tag}_timeline_v$v.log 2>&1; tail -16 gpurun_out/${tag}_timeline_v$v.log
done
