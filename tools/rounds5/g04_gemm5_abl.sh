#!/bin/bash
# round 5, call 4: timing-only ablations of gemm5's one-wave-per-SIMD form
tag=${1:-r05d}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 tools/lab/g4lab time 2048 native > gpurun_out/${tag}_g5_abl.log 2>&1; echo "time rc=$?"; cat gpurun_out/${tag}_g5_abl.log
