#!/bin/bash
# round 6, call 11: the batch-1 kernel with rounds of 40 .. 64 k-steps (K up to 32768): parity of every form, the 70B shapes, and the two
# families whose down_proj sat on the general strip kernel (Qwen2-7B K = 18944, Llama-2-70B K = 28672)
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_strip1_gpu.py tests/test_tp_shapes_gpu.py tests/test_numerics_contract_gpu.py -m gpu -q -x --timeout 900 > gpurun_out/r06r_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r06r_pytest.log
timeout 600 python tools/shape_table.py --families qwen2-7b llama2-70b --m 1 2 > gpurun_out/r06r_shape_long_k.md 2>&1; grep "down_proj" gpurun_out/r06r_shape_long_k.md
