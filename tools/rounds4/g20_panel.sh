#!/bin/bash
# round 4: the mid-batch panel kernel (33 <= M <= 128, native layout): parity, then timing against gemm2 on the reference layout in place
tag=${1:-r04t}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_native_layout_gpu.py -m gpu -q -x -k "panel or native_decode" --timeout 600 > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/${tag}_pytest.log
timeout 300 python tools/midm_bench.py 128 > gpurun_out/${tag}_midm.log 2>&1; cat gpurun_out/${tag}_midm.log
