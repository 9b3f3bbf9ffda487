#!/bin/bash
# round 5, call 7: configs[3] per-launch timing: product vs timing-only ablations of strip_dma (no scale/zero loads, no packed-word loads, neither);
# parity of the changed batch-2..32 paths
tag=${1:-r05g}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for b in 4 3; do
  for v in "" _abl1 _abl2 _abl3; do
    echo "== bits $b variant ${v:-product}"; timeout 300 tools/lab/gbench$v --cfg3 --bits $b --m 16 2>&1 | tee gpurun_out/${tag}_cfg3_w${b}${v}.log
  done
done
timeout 900 python -m pytest tests/test_native_layout_gpu.py -m gpu -q -x --timeout 600 -k "multi_strip or decode_kernels or grouped" > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/${tag}_pytest.log
