#!/bin/bash
# round 5, call 15: forced strips-per-block widths at batch 16 (lab library: QLLM_DMA_CPL) -- q/k/v as 258 blocks of three instead of 192 of four
tag=${1:-r05p}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for b in 4 3; do for c in 0 2 3 4; do
  echo "== bits $b QLLM_DMA_CPL=$c"; QLLM_DMA_CPL=$c timeout 300 tools/lab/gbench_lab --cfg3 --bits $b --m 16 2>&1 | tee gpurun_out/${tag}_w${b}_cpl$c.log | grep -v layer
done; done
