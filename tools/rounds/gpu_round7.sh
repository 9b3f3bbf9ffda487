#!/bin/bash
tag=${1:-r02g}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_chain_gpu.py -m gpu -x -q --timeout 200 > gpurun_out/${tag}_pytest_chain.log 2>&1; tail -8 gpurun_out/${tag}_pytest_chain.log
for mode in engine streams; do
  timeout 300 python bench.py --no-extra --no-pmc --steps 30 --chain-mode $mode 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode bench', d['value'], 'tok/s', d['ms_per_step'], 'ms', d['roofline']['frac'])"
done 2>&1 | tee gpurun_out/${tag}_chain.log
