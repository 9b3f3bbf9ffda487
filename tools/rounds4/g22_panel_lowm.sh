#!/bin/bash
# round 4: the panel kernel below 33 rows (lab build, QLLM_PANEL_MIN_M): one / two row tiles against the strip kernels
tag=${1:-r04aa}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
export QLLM_MI355X_LIB=$R/tools/lab/libqllm_lab.so
for mm in 33 2; do
  for g in 128 64; do
    for m in 8 16 32; do
      QLLM_PANEL_MIN_M=$mm timeout 200 python tools/midm_bench.py $g $m 2>&1 | grep "M=" | cut -c1-100
    done
  done
done > gpurun_out/${tag}_lowm.log 2>&1
cat gpurun_out/${tag}_lowm.log
