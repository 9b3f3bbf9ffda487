#!/bin/bash
# round 6, call 10: grouped prefill launches A/B in one process pool (same box, alternating): QLLM_FUSE_PREFILL=1 / 0
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_tail_split_gpu.py tests/test_decode_step_gpu.py tests/test_gpu_parity.py tests/test_eval_gpu.py -m gpu -q -x --timeout 900 > gpurun_out/r06q_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r06q_pytest.log
for i in 1 2 3; do for f in 1 0; do echo "--- QLLM_FUSE_PREFILL=$f"; QLLM_FUSE_PREFILL=$f timeout 300 python tools/prefill_legs.py 20 2>&1 | grep -v amdgpu.ids; done; done | tee gpurun_out/r06q_group_ab.log
