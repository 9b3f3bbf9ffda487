#!/bin/bash
# round 4, GPU call 14: one-strip batch 2..32 blocks with the packed words requested up front (WT): A/B per launch + parity
tag=${1:-r04n}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
export LD_LIBRARY_PATH=$R/tools/lab:$LD_LIBRARY_PATH
for wt in 0 1; do
  echo "== QLLM_DMA_WT=$wt"
  QLLM_DMA_WT=$wt timeout 200 tools/lab/gbench_lab --cfg3 --bits 4 --m 2 8 16 2>&1 | grep -v amdgpu | tee gpurun_out/${tag}_cfg3_w4_wt$wt.log
  QLLM_DMA_WT=$wt timeout 200 tools/lab/gbench_lab --cfg3 --group 128 --gptq --bits 4 --m 16 2>&1 | grep -v amdgpu | tee gpurun_out/${tag}_g128_w4_wt$wt.log
done
timeout 900 python -m pytest tests/test_native_layout_gpu.py tests/test_gpu_parity.py -m gpu -x -q -k "native or decode_kernel or three_bit_decode" > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/${tag}_pytest.log
