#!/bin/bash
# round 6, call 6: the planner refactor (one decision for forward + describe), run-time knobs, gemm3's tail split: GPU parity of everything
# that goes through the planner + the new tests; the shape table again (13B / Llama-3 prefill rows should move)
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_tail_split_gpu.py tests/test_mixed_bits_gpu.py tests/test_numerics_contract_gpu.py tests/test_gpu_parity.py tests/test_native_layout_gpu.py tests/test_decode_step_gpu.py -m gpu -q -x --timeout 900 > gpurun_out/r06f_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r06f_pytest.log
timeout 900 python tools/shape_table.py --families llama2-7b llama2-13b llama3-8b/mistral-7b qwen2-7b --m 2048 > gpurun_out/r06f_shape_table_prefill.md 2> gpurun_out/r06f_shape_table.err; echo "shapes rc=$?"; cat gpurun_out/r06f_shape_table_prefill.md; tail -3 gpurun_out/r06f_shape_table.err
