#!/bin/bash
# round 5, call 14: the two round-5 panel -> strips rules on 32-wide groups (lab library: the old routes through QLLM_PANEL_MIN_M=9 QLLM_PANEL_SMALL=1)
tag=${1:-r05o}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== g32, round-5 rules"; timeout 300 tools/lab/gbench_lab --cfg3 --gptq --group 32 --bits 4 --m 9 16 17 32 2>&1 | tee gpurun_out/${tag}_g32_new.log | grep "o_proj\|down"
echo "== g32, round-4 rules"; QLLM_PANEL_MIN_M=9 QLLM_PANEL_SMALL=1 timeout 300 tools/lab/gbench_lab --cfg3 --gptq --group 32 --bits 4 --m 9 16 17 32 2>&1 | tee gpurun_out/${tag}_g32_old.log | grep "o_proj\|down"
