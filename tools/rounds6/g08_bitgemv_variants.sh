#!/bin/bash
# round 6, call 8: bitgemv block shapes A/B -- 16 / 32 / 64 columns per block, K split until 1 or 2 blocks per CU (lab variants of the library)
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in release bg16f2 bg32f1 bg32f2 bg64f2; do
  echo "=== $v" >> gpurun_out/r06j_bitgemv_variants.log
  if [ $v = release ]; then timeout 600 python tools/bitgemv_bench.py --quick >> gpurun_out/r06j_bitgemv_variants.log 2>&1
  else QLLM_MI355X_LIB=$R/tools/lab/libqllm_$v.so timeout 600 python tools/bitgemv_bench.py --quick >> gpurun_out/r06j_bitgemv_variants.log 2>&1; fi
done
grep -v "amdgpu.ids" gpurun_out/r06j_bitgemv_variants.log
