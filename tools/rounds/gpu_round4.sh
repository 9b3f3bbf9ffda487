#!/bin/bash
tag=${1:-r02d}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_chain_gpu.py -m gpu -x -q --timeout 300 > gpurun_out/${tag}_pytest_chain.log 2>&1; tail -4 gpurun_out/${tag}_pytest_chain.log
for lw in 1 0; do
  QLLM_CHAIN_LW=$lw timeout 300 python tools/chain_timeline.py 4 1 1 2>&1 | grep -v amdgpu.ids | sed "s/^/LW=$lw /" | head -14
  QLLM_CHAIN_LW=$lw timeout 300 python bench.py --no-extra --no-pmc --steps 30 2>&1 | grep -v amdgpu.ids | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('LW=$lw bench', d['value'], 'tok/s', d['ms_per_step'], 'ms', d['roofline']['frac'])"
done > gpurun_out/${tag}_chain.log 2>&1; cat gpurun_out/${tag}_chain.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 -k "wave_specialised or prefill_kernel_vs_oracle" > gpurun_out/${tag}_pytest_g3.log 2>&1; tail -4 gpurun_out/${tag}_pytest_g3.log
for g3 in 1 0; do QLLM_GEMM3=$g3 timeout 200 python tools/kbench.py --m 2048 8192 --iters 100 --layouts GPTQ GEMM 2>&1 | grep -v amdgpu.ids | sed "s/^/GEMM3=$g3 /"; done > gpurun_out/${tag}_prefill.log; cat gpurun_out/${tag}_prefill.log
