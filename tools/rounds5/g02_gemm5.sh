#!/bin/bash
# round 5, call 2: gemm5 (register-B-fragment prefill kernel) vs gemm3: bit-exactness + timing; o_proj on the 4-wave form
tag=${1:-r05b}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 tools/lab/g4lab check > gpurun_out/${tag}_g5_check.log 2>&1; echo "check rc=$?"; grep -c "bit-exact" gpurun_out/${tag}_g5_check.log; grep -v "bit-exact" gpurun_out/${tag}_g5_check.log | tail -12
timeout 600 tools/lab/g4lab time 2048 native > gpurun_out/${tag}_g5_time.log 2>&1; echo "time rc=$?"; cat gpurun_out/${tag}_g5_time.log
timeout 300 tools/lab/g4lab time 2048 gptq > gpurun_out/${tag}_g5_time_gptq.log 2>&1; echo "time rc=$?"; cat gpurun_out/${tag}_g5_time_gptq.log
timeout 300 tools/lab/g4lab time 8192 native > gpurun_out/${tag}_g5_time_8192.log 2>&1; echo "time rc=$?"; cat gpurun_out/${tag}_g5_time_8192.log
timeout 300 tools/lab/dbisect --no-tp > gpurun_out/${tag}_dbisect.log 2>&1; echo "dbisect rc=$?"; grep -A 14 "us per launch" gpurun_out/${tag}_dbisect.log; grep -A 12 "per decoder layer" gpurun_out/${tag}_dbisect.log | grep "C ABI"
timeout 600 python -m pytest tests/test_strip1_gpu.py -m gpu -q -x --timeout 600 > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/${tag}_pytest.log
