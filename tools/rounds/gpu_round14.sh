#!/bin/bash
tag=${1:-r02r}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${tag}_hqq.log
import torch, bench
from qllm_amd import ops
dev = torch.device("cuda:0")
print(bench.hqq_leg(dev))
from qllm_amd.modeling.q_layers import QuantLinearHQQ
hs = bench.Stack(QuantLinearHQQ, 1, dev, seed=1, bits=4, group=64)
b = hs.blocks[0]
for m in (1, 16):
    print(m, "qkv:", b.q_proj._siblings.describe(m), "| gate/up:", b.gate_proj._siblings.describe(m))
hs3 = bench.Stack(QuantLinearHQQ, 1, dev, seed=1, bits=3, group=64)
b = hs3.blocks[0]
for m in (1, 16):
    print("3-bit", m, "qkv:", b.q_proj._siblings.describe(m), "| gate/up:", b.gate_proj._siblings.describe(m))
PY
