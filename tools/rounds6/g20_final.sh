#!/bin/bash
# round 6, evidence run (one gpurun call): smoke, the full -m gpu suite, the default bench line, tools/profile_round3.sh (kernel trace + PMC
# passes of the headline step, configs[3], prefill incl. the per-launch table of the module step), the shape table, the two-rank TP bench.
# tools/summarize_prof3.py r06 condenses it into profiles/r06_*.
tag=${1:-r06}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "commit: $(cat .git_rev 2>/dev/null)" > gpurun_out/${tag}_commit.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${tag}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/${tag}_smoke.log
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/${tag}_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/${tag}_bench.json; tail -3 gpurun_out/${tag}_bench.err
bash tools/profile_round3.sh $tag 2>&1 | tail -25
timeout 1500 python tools/shape_table.py > gpurun_out/${tag}_shape_table.md 2> gpurun_out/${tag}_shape_table.err; echo "shapes rc=$?"
timeout 300 python tools/hqq_leg.py 20 32 4,3,by_layer,by_module > gpurun_out/${tag}_hqq_mixed.log 2>&1; grep "^hqq" gpurun_out/${tag}_hqq_mixed.log
