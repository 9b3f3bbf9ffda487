#!/bin/bash
# round 5, call 17: a sibling group as ONE layer of the summed width (q/k/v: 768 strips = 256 blocks of three) -- what contiguous native copies would buy
tag=${1:-r05r}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for b in 4 3; do GBENCH_FUSED=1 timeout 300 tools/lab/gbench --cfg3 --bits $b --m 8 16 2>&1 | tee gpurun_out/${tag}_fused_w$b.log; done
GBENCH_FUSED=1 timeout 300 tools/lab/gbench --cfg3 --gptq --group 128 --bits 4 --m 4 16 2>&1 | tee gpurun_out/${tag}_fused_g128.log
timeout 300 tools/lab/gbench --cfg3 --gptq --group 128 --bits 4 --m 4 16 2>&1 | tee gpurun_out/${tag}_g128.log
