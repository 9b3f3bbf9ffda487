#!/bin/bash
# round 5: native layout version 2 -- parity of every reader, then timing (configs[3] per launch, decode step)
tag=${1:-r05k}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_native_layout_gpu.py tests/test_strip1_gpu.py tests/test_decode_step_gpu.py -m gpu -q --timeout 600 > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/${tag}_pytest.log
for b in 4 3; do timeout 300 tools/lab/gbench --cfg3 --bits $b --m 16 2>&1 | tee gpurun_out/${tag}_cfg3_w$b.log; done
timeout 300 tools/lab/cbench > gpurun_out/${tag}_cbench.log 2>&1; tail -3 gpurun_out/${tag}_cbench.log
timeout 300 tools/lab/dbisect --no-tp > gpurun_out/${tag}_dbisect.log 2>&1; grep -A 13 "us per launch" gpurun_out/${tag}_dbisect.log | grep -E "variant|C ABI|LVL0|LVL4: "; grep "C ABI" gpurun_out/${tag}_dbisect.log | tail -2
