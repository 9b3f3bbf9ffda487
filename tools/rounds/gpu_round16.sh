#!/bin/bash
tag=${1:-r02t}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "wave_specialised or prefill_kernel_vs_oracle or act_order" > gpurun_out/${tag}_pytest_g3.log 2>&1; tail -3 gpurun_out/${tag}_pytest_g3.log
for v in "QLLM_GEMM3_MW=8" "QLLM_GEMM3_MW=4"; do env $v timeout 200 python tools/kbench.py --m 2048 --iters 100 --layouts GPTQ GEMM 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"; done > gpurun_out/${tag}_prefill.log; cat gpurun_out/${tag}_prefill.log
bash tools/pmc_pass.sh ${tag}_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -- python tools/one_shape.py | head -12
