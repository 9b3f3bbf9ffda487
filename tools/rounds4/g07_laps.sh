#!/bin/bash
# round 4, GPU call 7: where a dequant iteration's cycles go (lap counters)
tag=${1:-r04g}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
export LD_LIBRARY_PATH=$R/tools/lab:$LD_LIBRARY_PATH
for v in 1 2 3 4 9; do
  timeout 120 tools/lab/g4lab timeline $v 2048 4096 4096 > gpurun_out/${tag}_timeline_v$v.log 2>&1; tail -15 gpurun_out/${tag}_timeline_v$v.log | grep -v "entry after\|exit after\|block lifetime \[us\]"
done
