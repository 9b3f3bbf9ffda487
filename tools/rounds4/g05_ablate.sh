#!/bin/bash
# round 4, GPU call 5: ablations of gemm4's dequant waves (timing only): what makes a dequant iteration 1600 cycles long?
tag=${1:-r04e}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
export LD_LIBRARY_PATH=$R/tools/lab:$LD_LIBRARY_PATH
timeout 600 tools/lab/g4lab time 2048 native > gpurun_out/${tag}_time_native.log 2>&1; head -16 gpurun_out/${tag}_time_native.log
for v in 4 5 10; do
  timeout 120 tools/lab/g4lab timeline $v 2048 4096 4096 > gpurun_out/${tag}_timeline_v$v.log 2>&1; tail -15 gpurun_out/${tag}_timeline_v$v.log | grep -v "entry after\|exit after\|block lifetime \[us\]"
done
