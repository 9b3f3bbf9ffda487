#!/bin/bash
tag=${1:-r02e}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for mw in 4 8; do QLLM_GEMM3_MW=$mw timeout 200 python tools/kbench.py --m 2048 --iters 100 --layouts GPTQ GEMM 2>&1 | grep -v amdgpu.ids | sed "s/^/MW=$mw /"; done > gpurun_out/${tag}_prefill_mw.log; cat gpurun_out/${tag}_prefill_mw.log
for mw in 4 8; do
QLLM_GEMM3_MW=$mw bash tools/pmc_pass.sh ${tag}_g3mw${mw}_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -- python tools/one_shape.py | grep -A12 gemm3
QLLM_GEMM3_MW=$mw bash tools/pmc_pass.sh ${tag}_g3mw${mw}_sq2 SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_MISC -- python tools/one_shape.py | grep -A12 gemm3
done
timeout 200 python tools/narrow_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_narrow.log; cat gpurun_out/${tag}_narrow.log
for cfg in "" "QLLM_GEMM2_MIN_M=65" "QLLM_STRIP_MAX_M=64" "QLLM_STRIP_MAX_M=64 QLLM_GEMM2_MIN_M=65"; do env $cfg timeout 300 python tools/kbench.py --m 33 48 64 96 128 160 --iters 100 --layouts GPTQ 2>&1 | grep -v amdgpu.ids | sed "s/^/[$cfg] /"; done > gpurun_out/${tag}_midm.log; cat gpurun_out/${tag}_midm.log
