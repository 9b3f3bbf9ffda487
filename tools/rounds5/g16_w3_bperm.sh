#!/bin/bash
# round 5, call 16: 3-bit strips at batch 2..32 with ONE word load per fragment (the pair's lower word through ds_bpermute); the new panel / strips
# routes; parity of the native-layout kernels, then configs[3] per launch
tag=${1:-r05q}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_native_layout_gpu.py tests/test_numerics_contract_gpu.py -m gpu -q --timeout 600 > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/${tag}_pytest.log
for b in 3 4; do timeout 300 tools/lab/gbench --cfg3 --bits $b --m 16 2>&1 | tee gpurun_out/${tag}_cfg3_w$b.log; done
timeout 300 tools/lab/gbench --cfg3 --bits 3 --m 2 4 8 32 2>&1 | tee gpurun_out/${tag}_cfg3_w3_m.log | grep layer
