#!/bin/bash
# round 6, last call: the full -m gpu suite on the final library, then the bench line + rocprofv3 passes + small-batch / batch-1 stacks again
tag=${1:-r06}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "commit: $(cat .git_rev 2>/dev/null)" > gpurun_out/${tag}_commit.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${tag}_smoke.log 2>&1; echo "smoke rc=$?"
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --durations=15 > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${tag}_pytest_gpu.log
bash tools/rounds6/g21_bench_only.sh $tag 2>&1 | tail -22
