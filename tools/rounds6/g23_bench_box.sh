#!/bin/bash
# round 6: the default bench line on another box of the pool (spread of the legs), nothing else
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python bench.py > gpurun_out/r06_bench_box2.json 2> gpurun_out/r06_bench_box2.err; echo "bench rc=$?"; tail -c 300 gpurun_out/r06_bench_box2.json
