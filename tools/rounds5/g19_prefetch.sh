#!/bin/bash
# round 5, call 19: a side stream reading the NEXT launch's packed words while the current launch runs (cbench --prefetch): does HBM
# streaming through the launch boundaries + Infinity Cache hits shorten the decode step?
tag=${1:-r05t}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { echo "== $*"; timeout 120 tools/lab/cbench "$@" 2>&1 | grep -v "^parity" | tail -1; }
run
run --prefetch 1 --pf-blocks 256 --pf-threads 512
run --prefetch 1 --pf-blocks 128 --pf-threads 256
run --prefetch 1 --pf-blocks 1024 --pf-threads 256
run --prefetch 2 --pf-blocks 256 --pf-threads 512
run --prefetch 2 --pf-blocks 128 --pf-threads 256
run --prefetch 3 --pf-blocks 256 --pf-threads 256
