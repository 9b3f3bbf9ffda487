#!/bin/bash
tag=${1:-r02f}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 600 > gpurun_out/${tag}_pytest.log 2>&1; tail -6 gpurun_out/${tag}_pytest.log
for cfg in "QLLM_GEMM3_PRIO=1" "QLLM_GEMM3_PRIO=0"; do env $cfg timeout 200 python tools/kbench.py --m 2048 --iters 100 --layouts GPTQ GEMM 2>&1 | grep -v amdgpu.ids | sed "s/^/[$cfg] /"; done > gpurun_out/${tag}_prefill.log; cat gpurun_out/${tag}_prefill.log
timeout 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -c 5000 gpurun_out/${tag}_bench.json; tail -3 gpurun_out/${tag}_bench.err
