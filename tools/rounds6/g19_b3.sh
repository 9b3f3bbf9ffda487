#!/bin/bash
# round 6, call 19: 3-bit layers on the batch-1 kernel: parity of every form, then the batch-1 stacks (AWQ g128, HQQ g64 4 / 3 bits)
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_strip1_gpu.py -m gpu -q -x --timeout 600 -k "3_bit" > gpurun_out/r06z_pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r06z_pytest.log | cut -c1-300
timeout 300 python tools/hqq_m1.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06z_hqq_m1.log | cut -c1-200
