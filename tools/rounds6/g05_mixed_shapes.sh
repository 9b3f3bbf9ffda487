#!/bin/bash
# round 6, call 5: the mixed 3/4-bit model tests (verdict Missing #2), the mixed stacks' rates, the shape table of other model families (Missing #5)
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mixed_bits_gpu.py tests/test_decode_step_gpu.py -m gpu -q -x --timeout 600 > gpurun_out/r06e_pytest_mixed.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r06e_pytest_mixed.log
timeout 600 python tools/hqq_leg.py 20 32 4,3,by_layer,by_module > gpurun_out/r06e_hqq_mixed.log 2>&1; echo "hqq rc=$?"; cat gpurun_out/r06e_hqq_mixed.log
timeout 1500 python tools/shape_table.py > gpurun_out/r06e_shape_table.md 2> gpurun_out/r06e_shape_table.err; echo "shapes rc=$?"; cat gpurun_out/r06e_shape_table.md; tail -5 gpurun_out/r06e_shape_table.err
