#!/bin/bash
# round 4, GPU call 11: what the act-order prefill path adds: gather kernel micro-bench + per-kernel stats of the two stacks
tag=${1:-r04k}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/gather_bench.py > gpurun_out/${tag}_gather.log 2>&1; cat gpurun_out/${tag}_gather.log | grep -v amdgpu.ids
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ao -o ao -- python $R/tools/actorder_leg.py 6 > $R/gpurun_out/${tag}_leg.log 2>&1
grep -v amdgpu.ids $R/gpurun_out/${tag}_leg.log | tail -3
f=$(find /tmp/prof_ao -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-200; cp $f $R/gpurun_out/${tag}_kernel_stats.csv
