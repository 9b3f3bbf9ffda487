#!/bin/bash
# One gpurun call: GPU parity suite, the default bench line, narrow-shape A/B.  Output under gpurun_out/<tag>_*.
tag=${1:-r02a}
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 > gpurun_out/${tag}_pytest.log 2>&1; tail -5 gpurun_out/${tag}_pytest.log
fi
timeout 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -c 6000 gpurun_out/${tag}_bench.json; tail -5 gpurun_out/${tag}_bench.err
if [ -z "$SKIP_NARROW" ]; then
for m in 192 48 16; do QLLM_STRIP_MIN=$m timeout 200 python tools/narrow_ab.py; done > gpurun_out/${tag}_narrow.log 2>&1; cat gpurun_out/${tag}_narrow.log
fi
