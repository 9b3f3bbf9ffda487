#!/bin/bash
# round 5, final evidence run (one gpurun call): smoke, the full -m gpu suite, the default bench line, then tools/profile_round3.sh
# (kernel trace + PMC passes of the headline step, configs[3], prefill).  tools/summarize_prof3.py r05 condenses it into profiles/r05_*.
tag=${1:-r05}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
git_rev=$(cat .git_rev 2>/dev/null)
echo "commit: $git_rev" > gpurun_out/${tag}_commit.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${tag}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/${tag}_smoke.log
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/${tag}_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/${tag}_bench.json; tail -3 gpurun_out/${tag}_bench.err
bash tools/profile_round3.sh $tag 2>&1 | tail -25
