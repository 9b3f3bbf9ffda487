#!/bin/bash
# round 4: where the panel kernel's time goes: timing-only ablations through the lab build (QLLM_PANEL_ABL bits: 1 no activation
# pieces, 2 no word loads, 4 no compute, 8 no split-K sum)
tag=${1:-r04y}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
export QLLM_MI355X_LIB=$R/tools/lab/libqllm_lab.so
for a in 0 1 2 4 8 3 7 15; do
  echo "== QLLM_PANEL_ABL=$a"
  QLLM_PANEL_ABL=$a timeout 200 python tools/midm_bench.py 128 64 2>&1 | grep "M= 64" | cut -c1-40
done > gpurun_out/${tag}_abl.log 2>&1
cat gpurun_out/${tag}_abl.log
