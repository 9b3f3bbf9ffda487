#!/bin/bash
# round 4: bf16 activations on the panel kernel: the tile converted once per block in LDS (release build) against once per fragment
# (the lab library built before the change), + parity
tag=${1:-r04af}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_native_layout_gpu.py tests/test_decode_step_gpu.py -m gpu -q -x --timeout 600 -k "panel or sibling_groups_take" 2>&1 | tail -3
for m in 32 64 128; do
  echo "== per fragment (old)"; QLLM_MI355X_LIB=$R/tools/lab/libqllm_lab.so timeout 200 python tools/midm_bench.py 128 $m bf16 2>&1 | grep "M=" | cut -c1-44
  echo "== per tile (new)"; timeout 200 python tools/midm_bench.py 128 $m bf16 2>&1 | grep "M=" | cut -c1-44
done 2>&1 | tee gpurun_out/${tag}_bf16.log
