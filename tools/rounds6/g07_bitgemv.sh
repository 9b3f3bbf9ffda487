#!/bin/bash
# round 6, call 7: the bit-stream matvec (2 / 5 / 6 / 7 / 8 bits): parity, the goldens through the modules, timing against dequant + GEMM
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_bitgemv_gpu.py -m gpu -q -x --timeout 900 > gpurun_out/r06l_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r06l_pytest.log
timeout 600 python tools/bitgemv_bench.py > gpurun_out/r06l_bitgemv_bench.md 2>&1; echo "bench rc=$?"; grep "| 8 | " gpurun_out/r06l_bitgemv_bench.md
