#!/bin/bash
# round 5, call 1: the decode bisect (tools/lab/dbisect), in-kernel timelines of both batch-1 paths, parity of the new kernel, headline
tag=${1:-r05a}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 tools/lab/dbisect > gpurun_out/${tag}_dbisect.log 2>&1; echo "dbisect rc=$?"; cat gpurun_out/${tag}_dbisect.log
timeout 300 tools/lab/cbench --timeline > gpurun_out/${tag}_cbench_new.log 2>&1; echo "cbench new rc=$?"; tail -16 gpurun_out/${tag}_cbench_new.log
QLLM_STRIP1=0 timeout 300 tools/lab/cbench_lab --timeline > gpurun_out/${tag}_cbench_old.log 2>&1; echo "cbench old rc=$?"; tail -16 gpurun_out/${tag}_cbench_old.log
timeout 900 python -m pytest tests/test_strip1_gpu.py tests/test_decode_step_gpu.py -m gpu -q -x --timeout 600 > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/${tag}_pytest.log
timeout 400 python bench.py --no-extra --no-pmc > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/${tag}_bench.json; tail -3 gpurun_out/${tag}_bench.err
