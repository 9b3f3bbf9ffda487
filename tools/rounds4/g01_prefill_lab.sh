#!/bin/bash
# round 4, GPU call 1: gemm4 (matrix waves + all-DMA producer waves) vs gemm3 -- bit-exactness, interleaved timing, SQ counters
tag=${1:-r04a}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
export LD_LIBRARY_PATH=$R/tools/lab:$LD_LIBRARY_PATH
timeout 400 tools/lab/g4lab check > gpurun_out/${tag}_check.log 2>&1; echo "check rc=$?"; tail -4 gpurun_out/${tag}_check.log; grep -c bit-exact gpurun_out/${tag}_check.log; grep MISMATCH gpurun_out/${tag}_check.log | head -20
timeout 300 tools/lab/g4lab time 2048 native > gpurun_out/${tag}_time_native.log 2>&1; cat gpurun_out/${tag}_time_native.log
timeout 300 tools/lab/g4lab time 2048 awq > gpurun_out/${tag}_time_awq.log 2>&1; cat gpurun_out/${tag}_time_awq.log
for v in 0 2 4; do
  bash tools/pmc_pass.sh ${tag}_sq_v$v SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -- tools/lab/g4lab prof $v | grep -A10 "gemm[34]_kernel"
done
