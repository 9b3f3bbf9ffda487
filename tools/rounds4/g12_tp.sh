#!/bin/bash
# round 4, GPU call 12: tensor-parallel tests (two ranks on one GPU): act-order shards, one-shot all-reduce, tp_bench
tag=${1:-r04l}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_tp_collective_gpu.py tests/test_tp_shapes_gpu.py -m gpu -x -q --timeout 800 -s > gpurun_out/${tag}_pytest_tp.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/${tag}_pytest_tp.log | tail -25
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "three_bit_act_order or act_order or gather" > gpurun_out/${tag}_pytest_ao.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/${tag}_pytest_ao.log
