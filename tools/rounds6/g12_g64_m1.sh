#!/bin/bash
# round 6, call 12: where 64-wide groups stand at batch 1 (the batch-1 kernel takes 128-wide groups only): HQQ g64 / GPTQ g128, M = 1, grouped launches
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 300 python tools/kbench.py --m 1 --layouts HQQ --g 64; timeout 300 python tools/kbench.py --m 1 --layouts GPTQ --g 128; timeout 300 python tools/hqq_m1.py) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06s_g64_m1.log
