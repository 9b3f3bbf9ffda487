#!/bin/bash
# round 5, call 18: 3-bit panels with one word load per fragment (ds_bpermute for the pair's lower word): parity, then 17..64 rows per launch
tag=${1:-r05s}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_native_layout_gpu.py -m gpu -q --timeout 600 -k "panel" > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/${tag}_pytest.log
timeout 300 tools/lab/gbench --cfg3 --bits 3 --m 32 48 2>&1 | tee gpurun_out/${tag}_cfg3_w3_m.log
