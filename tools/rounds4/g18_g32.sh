#!/bin/bash
# round 4: 32-wide groups on the strip path (native layout): parity, then decode timing against the split-K kernel
tag=${1:-r04q}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_native_layout_gpu.py tests/test_gpu_parity.py -m gpu -q -x -k "native_decode or grouped_launch or act_order_decode or release or modules_decode or multi_strip" --timeout 600 > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/${tag}_pytest.log
timeout 300 python tools/g32_bench.py > gpurun_out/${tag}_g32_bench.log 2>&1; cat gpurun_out/${tag}_g32_bench.log
