#!/bin/bash
# round 6, call 13: 64-wide groups on the batch-1 kernel: parity of every form, then the batch-1 stacks (AWQ g128, HQQ g64 4 / 3 bits, HQQ g128)
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_strip1_gpu.py tests/test_numerics_contract_gpu.py tests/test_gpu_parity.py -m gpu -q -x --timeout 900 > gpurun_out/r06t_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r06t_pytest.log
timeout 300 python tools/hqq_m1.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06t_hqq_m1.log
