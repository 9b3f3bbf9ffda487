#!/bin/bash
tag=${1:-r02p}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "gather or act_order or three_bit or odd_bits or wave_specialised" > gpurun_out/${tag}_pytest_sel.log 2>&1; tail -5 gpurun_out/${tag}_pytest_sel.log
timeout 200 python tools/gather_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${tag}_gather.log
