#!/bin/bash
tag=${1:-r02i}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/engine_timeline.py ${LAYERS:-4} 1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${tag}_engine_timeline.log
timeout 600 python -m pytest tests/test_chain_gpu.py -m gpu -x -q --timeout 200 > gpurun_out/${tag}_pytest_chain.log 2>&1; tail -5 gpurun_out/${tag}_pytest_chain.log
