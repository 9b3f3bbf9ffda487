#!/bin/bash
# round 5, call 23: bench.py's leg timers got warm-up calls after the evidence run on a1d16f1 (library and tests unchanged since):
# the default bench line and the rocprofv3 passes again, same file names
tag=${1:-r05}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "commit: $(cat .git_rev 2>/dev/null) (bench + profiles; pytest log: a1d16f1)" > gpurun_out/${tag}_commit.txt
timeout 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"; tail -c 300 gpurun_out/${tag}_bench.json
bash tools/profile_round3.sh $tag 2>&1 | tail -6
