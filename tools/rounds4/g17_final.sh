#!/bin/bash
# round 4, final evidence run: the full -m gpu suite, the default bench line, then the rocprofv3 passes (tools/profile_round3.sh r04)
tag=${1:-r04}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/${tag}_pytest_gpu.log
timeout 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"; tail -c 2500 gpurun_out/${tag}_bench.json; tail -3 gpurun_out/${tag}_bench.err
timeout 1200 bash tools/profile_round3.sh $tag 2>&1 | tail -30
