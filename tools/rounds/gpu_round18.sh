#!/bin/bash
tag=${1:-r02v}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "QLLM_STRIP_CPL=0" "QLLM_STRIP_CPL=2" "QLLM_STRIP_CPL=1"; do
  for rep in 1 2; do
  env $v timeout 300 python bench.py --no-extra --no-pmc --steps 40 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], 'tok/s', d['ms_per_step'], 'ms', d['roofline']['frac'])"
  done
  env $v timeout 200 python tools/kbench.py --grouped 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
done 2>&1 | tee gpurun_out/${tag}_cpl.log
