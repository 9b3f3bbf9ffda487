#!/bin/bash
# round 5: the full -m gpu suite + the default bench line (mid-round safety run)
tag=${1:-r05i}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/${tag}_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"; tail -c 6000 gpurun_out/${tag}_bench.json; tail -3 gpurun_out/${tag}_bench.err
