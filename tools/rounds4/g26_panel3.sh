#!/bin/bash
# round 4: the 3-bit form of the panel kernel: parity (single layers, sibling groups, the existing 3-bit tests that now reach it)
tag=${1:-r04ag}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_native_layout_gpu.py tests/test_decode_step_gpu.py tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "panel or (native_decode and 3) or three_bit or sibling_groups_take" > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/${tag}_pytest.log | cut -c1-300
