#!/bin/bash
# round 5, call 21: batch-1 blocks of 15 x 24 (K = 11008) and 7 x 16 (K = 3584) waves x k-steps -- no dead wave: parity, the decode step,
# the TP shard leg; tp_bench on two ranks sharing the GPU (reports the faster of fused / unfused)
tag=${1:-r05u}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_strip1_gpu.py tests/test_tp_shapes_gpu.py tests/test_tp_collective_gpu.py tests/test_decode_step_gpu.py -m gpu -q --timeout 600 > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/${tag}_pytest.log
cat gpurun_out/tp_bench_two_ranks_one_gpu.log | grep "fused"
for odd in 1 0; do echo "== QLLM_S1_ODD=$odd"; QLLM_S1_ODD=$odd timeout 200 tools/lab/cbench_lab 2>&1 | tail -1; done
for odd in 1 0; do echo "== QLLM_S1_ODD=$odd"; QLLM_S1_ODD=$odd timeout 300 tools/lab/dbisect 2>&1 | grep -A 4 "TP = 8 shard" | tail -3; done
timeout 300 python -c "
import torch, json
from tools import tp_bench
r = tp_bench.shard_shapes_leg(torch.device('cuda:0'))
print(json.dumps(r['tp_shard_decode_m1']))
" 2>&1 | tail -2
