#!/bin/bash
# round 4: does the placement of kernel arguments (HIP_FORCE_DEV_KERNARG) move the decode step?  + the g32 repack test
tag=${1:-r04s}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in default 1 0 1 default 0; do
  if [ $v = default ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
  timeout 300 python bench.py --no-extra --no-pmc --steps 50 --warmup 5 --min-timed-s 0.5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HIP_FORCE_DEV_KERNARG=$v', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
unset HIP_FORCE_DEV_KERNARG
timeout 600 python -m pytest tests/test_native_layout_gpu.py -m gpu -q -x -k "other_group_sizes" --timeout 600 2>&1 | tail -3
