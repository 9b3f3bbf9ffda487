#!/bin/bash
tag=${1:-r02c}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/chain_timeline.py 4 1 1 > gpurun_out/${tag}_timeline_graph.txt 2>&1; cat gpurun_out/${tag}_timeline_graph.txt
timeout 300 python tools/chain_timeline.py 4 0 1 > gpurun_out/${tag}_timeline_eager.txt 2>&1; cat gpurun_out/${tag}_timeline_eager.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 -k "wave_specialised or prefill_kernel_vs_oracle" > gpurun_out/${tag}_pytest_g3.log 2>&1; tail -4 gpurun_out/${tag}_pytest_g3.log
for g3 in 1 0; do QLLM_GEMM3=$g3 timeout 200 python tools/kbench.py --m 2048 8192 --iters 100 --layouts GPTQ GEMM 2>&1 | grep -v amdgpu.ids | sed "s/^/GEMM3=$g3 /"; done > gpurun_out/${tag}_prefill.log; cat gpurun_out/${tag}_prefill.log
