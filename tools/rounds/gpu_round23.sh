#!/bin/bash
# bench + profiler passes of tools/profile_round2.sh without the test suite (second box for the same commit)
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -c 600 gpurun_out/${tag}_bench.json
PROFILE_ONLY=1 bash tools/profile_round2.sh $tag
