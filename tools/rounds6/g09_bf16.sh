#!/bin/bash
# round 6, call 9: native bf16 prefill (gemm3 BF): parity (the prefill tests incl. the bf16 block, tail split, numerics contract), then
# the prefill legs: fp16 / act-order / bf16 native / bf16 through the conversion pre-pass, three rounds
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_tail_split_gpu.py tests/test_decode_step_gpu.py tests/test_mixed_bits_gpu.py tests/test_eval_gpu.py -m gpu -q -x --timeout 900 > gpurun_out/r06o_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r06o_pytest.log
for i in 1 2 3; do timeout 300 python tools/prefill_legs.py 20 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r06o_prefill_legs.log
