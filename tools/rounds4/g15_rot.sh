#!/bin/bash
# round 4, GPU call 15: gemm3 with per-column-tile k rotation (L2 same-line contention)
tag=${1:-r04o}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
export LD_LIBRARY_PATH=$R/tools/lab:$LD_LIBRARY_PATH
timeout 600 tools/lab/g4lab time 2048 native > gpurun_out/${tag}_time_native.log 2>&1; cat gpurun_out/${tag}_time_native.log
timeout 600 tools/lab/g4lab time 8192 native > gpurun_out/${tag}_time_native_m8192.log 2>&1; cat gpurun_out/${tag}_time_native_m8192.log
