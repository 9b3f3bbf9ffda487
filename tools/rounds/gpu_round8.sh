#!/bin/bash
tag=${1:-r02h}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/engine_debug.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${tag}_engine_debug.log
for mode in engine streams; do
  timeout 300 python bench.py --no-extra --no-pmc --steps 30 --chain-mode $mode 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode bench', d['value'], 'tok/s', d['ms_per_step'], 'ms', d['roofline']['frac'])"
done 2>&1 | tee gpurun_out/${tag}_chain.log
timeout 1500 python -m pytest tests -m gpu -q --timeout 300 > gpurun_out/${tag}_pytest_all.log 2>&1; tail -25 gpurun_out/${tag}_pytest_all.log
