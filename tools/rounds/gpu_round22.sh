#!/bin/bash
tag=${1:-r02z}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 560 python -m pytest tests/test_gpu_parity.py tests/test_tp_shapes_gpu.py tests/test_eval_gpu.py -m gpu -q --timeout 150 -k "three_bit_prefill or prefill or wave_specialised or workspace or c_abi or determinis or llama70b or eval or properties" > gpurun_out/${tag}_pytest_split.log 2>&1; tail -6 gpurun_out/${tag}_pytest_split.log
