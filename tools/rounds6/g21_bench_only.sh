#!/bin/bash
# round 6, call 21: bench.py got the batch 2 / 4 legs after the evidence run (library and tests unchanged since): the default bench line and
# the rocprofv3 passes again under the same file names, + the small-batch and batch-1 HQQ stacks
tag=${1:-r06}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "commit: $(cat .git_rev 2>/dev/null) (bench + profiles; pytest log: $(cat gpurun_out/${tag}_commit.txt 2>/dev/null | head -1))" > gpurun_out/${tag}_commit_bench.txt
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"; tail -c 400 gpurun_out/${tag}_bench.json
bash tools/profile_round3.sh $tag 2>&1 | tail -6
timeout 300 python tools/small_batch.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_small_batch.log; cat gpurun_out/${tag}_small_batch.log
timeout 300 python tools/hqq_m1.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_hqq_m1.log; cat gpurun_out/${tag}_hqq_m1.log
