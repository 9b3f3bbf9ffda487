#!/bin/bash
# round 5: gemm5 v4 (packed words by one LDS-DMA piece per wave and k-tile): check + timing + ablations
tag=${1:-r05l}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 tools/lab/g4lab check > gpurun_out/${tag}_g5_check.log 2>&1; echo "check rc=$?"; grep -c "bit-exact" gpurun_out/${tag}_g5_check.log; grep -v "bit-exact" gpurun_out/${tag}_g5_check.log | tail -8
timeout 900 tools/lab/g4lab time 2048 native > gpurun_out/${tag}_g5_abl.log 2>&1; echo "time rc=$?"; cat gpurun_out/${tag}_g5_abl.log
timeout 300 tools/lab/g4lab time 2048 gptq 2>&1 | head -8 | tee gpurun_out/${tag}_g5_gptq.log
