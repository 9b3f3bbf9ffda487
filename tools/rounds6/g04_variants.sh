#!/bin/bash
# round 6, call 4: verdict item 1 (a) / (c) as A/B variants of the batch-1 kernel inside the product's launch structure (dependent chain,
# 32 layers, hipGraph), interleaved rounds
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 tools/lab/chainlab --variants --replays 20 > gpurun_out/r06d_variants.log 2>&1; echo "rc=$?"; cat gpurun_out/r06d_variants.log
