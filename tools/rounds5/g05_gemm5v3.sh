#!/bin/bash
# round 5, call 5: gemm5 v3 (requests three k-tiles ahead, four register sets): check + timing + ablations
tag=${1:-r05e}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 tools/lab/g4lab check > gpurun_out/${tag}_g5_check.log 2>&1; echo "check rc=$?"; grep -c "bit-exact" gpurun_out/${tag}_g5_check.log; grep -v "bit-exact" gpurun_out/${tag}_g5_check.log | tail -12
timeout 900 tools/lab/g4lab time 2048 native > gpurun_out/${tag}_g5_abl.log 2>&1; echo "time rc=$?"; cat gpurun_out/${tag}_g5_abl.log
