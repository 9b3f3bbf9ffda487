#!/bin/bash
# round 5, call 3: gemm5 v2 (static ring slots in the ds_read immediates, 5-op extraction, no clamps) vs gemm3
tag=${1:-r05c}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 tools/lab/g4lab check > gpurun_out/${tag}_g5_check.log 2>&1; echo "check rc=$?"; grep -c "bit-exact" gpurun_out/${tag}_g5_check.log; grep -v "bit-exact" gpurun_out/${tag}_g5_check.log | tail -12
timeout 600 tools/lab/g4lab time 2048 native > gpurun_out/${tag}_g5_time.log 2>&1; echo "time rc=$?"; cat gpurun_out/${tag}_g5_time.log
timeout 300 tools/lab/g4lab time 8192 native > gpurun_out/${tag}_g5_time_8192.log 2>&1; echo "time rc=$?"; cat gpurun_out/${tag}_g5_time_8192.log
