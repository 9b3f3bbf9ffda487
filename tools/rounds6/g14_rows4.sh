#!/bin/bash
# round 6, call 14: batches 2..4 on the batch-1 kernel's four-row forms: parity, then the stack at M = 1, 2, 3, 4, 8 against strip_dma
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_strip1_gpu.py tests/test_decode_step_gpu.py tests/test_numerics_contract_gpu.py tests/test_native_layout_gpu.py -m gpu -q -x --timeout 900  > gpurun_out/r06v_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r06v_pytest.log
timeout 600 python tools/small_batch.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06v_small_batch.log
