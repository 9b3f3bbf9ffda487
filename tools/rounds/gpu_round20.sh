#!/bin/bash
tag=${1:-r02x}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 420 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 150 -k "three_bit_prefill" > gpurun_out/${tag}_pytest_3bit.log 2>&1; tail -6 gpurun_out/${tag}_pytest_3bit.log
