#!/bin/bash
# round 4, GPU call 2: in-kernel timeline of gemm4 (where do the 70 us go?)
tag=${1:-r04b}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
export LD_LIBRARY_PATH=$R/tools/lab:$LD_LIBRARY_PATH
for v in 2 4; do
  timeout 120 tools/lab/g4lab timeline $v 2048 4096 4096 > gpurun_out/${tag}_timeline_v$v.log 2>&1; tail -15 gpurun_out/${tag}_timeline_v$v.log
done
timeout 120 tools/lab/g4lab timeline 2 2048 11008 4096 > gpurun_out/${tag}_timeline_v2_k11008.log 2>&1; tail -15 gpurun_out/${tag}_timeline_v2_k11008.log
