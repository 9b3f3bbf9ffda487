#!/bin/bash
# round 5, call 13: where K >= 2 N (down_proj) at 9..32 rows -- panel kernel vs the strips (lab library: QLLM_PANEL=0), by batch size,
# group size and bit width; per launch kind, 20 rotating layers
tag=${1:-r05n}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for cfg in "--cfg3 --bits 4" "--cfg3 --bits 3" "--cfg3 --gptq --group 128 --bits 4"; do
  n=$(echo $cfg | tr -d ' -')
  for pan in 1 0; do
    echo "== $cfg QLLM_PANEL=$pan"; QLLM_PANEL=$pan timeout 300 tools/lab/gbench_lab $cfg --m 9 12 16 17 24 32 2>&1 | tee gpurun_out/${tag}_${n}_panel$pan.log | grep -v "q/k/v\|gate/up"
  done
done
