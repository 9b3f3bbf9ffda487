#!/bin/bash
tag=${1:-r02aa}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 120 -k "three_bit_prefill or wave_specialised or prefill_kernel_vs_oracle" > gpurun_out/${tag}_pytest_odd.log 2>&1; tail -6 gpurun_out/${tag}_pytest_odd.log
timeout 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${tag}_3bit_midm.log
import torch
from qllm_amd import ops
from qllm_amd.modeling.q_layers import QuantLinearHQQ
dev = torch.device("cuda:0")
K, N = 11008, 4096
ls = []
for i in range(6):
    l = QuantLinearHQQ(3, 64, K, N, False, dtype=torch.float16)
    l.qweight = torch.randint(-2**31, 2**31 - 1, l.qweight.shape, dtype=torch.int32)
    l.qzeros = (torch.rand(l.qzeros.shape) * 7).half()
    l.scales = (torch.rand(l.scales.shape) * 0.01 + 0.002).half()
    ls.append(l.to(dev))
for M in (128, 300, 512, 1024, 2048):
    x = torch.randn(M, K, device=dev, dtype=torch.float16)
    def fused():
        for l in ls: l(x)
    fused(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fused()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"HQQ w3 g64 {K}x{N} M={M}: plan={ops.plan_describe([ls[0].decode_descriptor()], M)[-24:]}  fused {e0.elapsed_time(e1) / 10 / len(ls) * 1e3:.1f} us")
PY
