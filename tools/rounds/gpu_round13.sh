#!/bin/bash
tag=${1:-r02q}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for lib in default hdr; do
  for chain in 0 1; do
    if [ $lib = hdr ]; then export QLLM_MI355X_LIB=$R/tools/lab/libqllm_hdr.so; else unset QLLM_MI355X_LIB; fi
    for rep in 1 2; do
    timeout 300 python bench.py --no-extra --no-pmc --steps 40 --chain $chain 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$lib chain=$chain', d['value'], 'tok/s', d['ms_per_step'], 'ms', d['roofline']['frac'])"
    done
  done
done 2>&1 | tee gpurun_out/${tag}_hdr.log
