#!/bin/bash
# round 4, GPU call 10: the full -m gpu suite + a quick headline bench after the round's refactors
tag=${1:-r04j}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 600 > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/${tag}_pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"; tail -c 3000 gpurun_out/${tag}_bench.json; tail -5 gpurun_out/${tag}_bench.err
