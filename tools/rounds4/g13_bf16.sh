#!/bin/bash
# round 4, GPU call 13: shared bf16 -> fp16 conversion: parity + the three prefill legs
tag=${1:-r04m}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wave_specialised or bf16" > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/${tag}_pytest.log
timeout 300 python tools/prefill_legs.py 10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${tag}_legs.log
