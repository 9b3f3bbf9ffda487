#!/bin/bash
# round 6, call 1: where the round starts on this pool -- smoke, the multi-rank tests (incl. the new `bench.py --gpus 2` end-to-end
# test: two ranks on the one GPU), the default bench line
tag=${1:-r06a}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
git rev-parse HEAD > /dev/null 2>&1 || true
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${tag}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/${tag}_smoke.log
timeout 1200 python -m pytest tests/test_tp_collective_gpu.py -m gpu -q -x --timeout 900 > gpurun_out/${tag}_pytest_tp.log 2>&1; echo "pytest tp rc=$?"; tail -15 gpurun_out/${tag}_pytest_tp.log
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"; tail -c 2500 gpurun_out/${tag}_bench.json; tail -3 gpurun_out/${tag}_bench.err
