#!/bin/bash
# round 6, call 2: the chained-links prototype (tools/lab/chainlab.hip) against the product's launch structure, same buffers, same minute;
# the single two-stream graph under the runtime's default and with DEBUG_HIP_FORCE_GRAPH_QUEUES=2 / 4
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for q in default 2 4; do
  echo "=== DEBUG_HIP_FORCE_GRAPH_QUEUES=$q" >> gpurun_out/r06b_chainlab.log
  if [ $q = default ]; then timeout 300 tools/lab/chainlab --spin 1000 --replays 10 "$@" >> gpurun_out/r06b_chainlab.log 2>&1
  else DEBUG_HIP_FORCE_GRAPH_QUEUES=$q timeout 300 tools/lab/chainlab --spin 1000 --replays 10 "$@" >> gpurun_out/r06b_chainlab.log 2>&1; fi
  echo "chainlab rc=$?" >> gpurun_out/r06b_chainlab.log
done
cat gpurun_out/r06b_chainlab.log
