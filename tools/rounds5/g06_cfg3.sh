#!/bin/bash
# round 5, call 6: configs[3] (HQQ g64, batch 16): g64 group step from minus-sum-x-bias accumulators, six 3-bit strips per block; parity + per-launch timing
tag=${1:-r05f}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for b in 4 3; do timeout 300 tools/lab/gbench --cfg3 --bits $b --m 16 > gpurun_out/${tag}_cfg3_w$b.log 2>&1; echo "gbench w$b rc=$?"; cat gpurun_out/${tag}_cfg3_w$b.log; done
timeout 300 tools/lab/gbench --cfg3 --bits 4 --m 2 4 8 > gpurun_out/${tag}_cfg3_w4_lowm.log 2>&1; tail -20 gpurun_out/${tag}_cfg3_w4_lowm.log
timeout 900 python -m pytest tests/test_native_layout_gpu.py -m gpu -q -x --timeout 600 -k "multi_strip or decode_kernels or grouped" > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/${tag}_pytest.log
timeout 300 python tools/hqq_leg.py 10 > gpurun_out/${tag}_hqq_leg.log 2>&1; cat gpurun_out/${tag}_hqq_leg.log | tail -4
