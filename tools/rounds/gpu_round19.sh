#!/bin/bash
tag=${1:-r02w}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "three_bit or odd_bits or wave_specialised" > gpurun_out/${tag}_pytest_3bit.log 2>&1; tail -8 gpurun_out/${tag}_pytest_3bit.log
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${tag}_3bit_prefill.log
import torch, sys
sys.path.insert(0, "tests")
from qllm_amd import ops
from qllm_amd.modeling.q_layers import QuantLinearHQQ
dev = torch.device("cuda:0")
for (K, N) in ((4096, 4096), (4096, 11008), (11008, 4096)):
    ls = []
    for i in range(6):
        l = QuantLinearHQQ(3, 64, K, N, False, dtype=torch.float16)
        l.qweight = torch.randint(-2**31, 2**31 - 1, l.qweight.shape, dtype=torch.int32)
        l.qzeros = (torch.rand(l.qzeros.shape) * 7).half()
        l.scales = (torch.rand(l.scales.shape) * 0.01 + 0.002).half()
        ls.append(l.to(dev))
    x = torch.randn(2048, K, device=dev, dtype=torch.float16)
    def fused():
        for l in ls: l(x)
    def twostep():
        for l in ls:
            w = ops.dequant(l.decode_descriptor(), dev, torch.float16)
            torch.matmul(x, w)
    for name, fn in (("fused gemm3 3-bit", fused), ("dequant kernel + dense GEMM", twostep)):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): g.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 / len(ls) * 1e3
        print(f"HQQ w3 g64 M=2048 {K}x{N}: {name}: {us:.1f} us  {2.0*2048*K*N/us/1e6:.0f} TFLOP/s")
PY
