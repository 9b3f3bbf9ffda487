#!/bin/bash
# round 6, call 3: in-kernel timeline of the chained step (two graphs side by side) beside the plain 5-launch step
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 120 tools/lab/chainlab --timeline --spin 4000 "$@" > gpurun_out/r06c_chain_timeline.log 2>&1; echo "rc=$?"; cat gpurun_out/r06c_chain_timeline.log
