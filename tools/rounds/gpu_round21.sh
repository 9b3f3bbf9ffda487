#!/bin/bash
tag=${1:-r02y}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 150 -k "three_bit_prefill or prefill_kernel_vs_oracle or wave_specialised or workspace or c_abi" > gpurun_out/${tag}_pytest_split.log 2>&1; tail -6 gpurun_out/${tag}_pytest_split.log
for v in "QLLM_GEMM3_MIN_M=100000" "QLLM_GEMM3_MIN_M=65"; do env $v timeout 200 python tools/kbench.py --m 128 256 512 1024 --iters 60 --layouts GPTQ 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"; done > gpurun_out/${tag}_midm.log; cat gpurun_out/${tag}_midm.log
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${tag}_3bit_midm.log
import torch, sys
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
    for M in (128, 300, 512, 1024):
        x = torch.randn(M, K, device=dev, dtype=torch.float16)
        def fused():
            for l in ls: l(x)
        def twostep():
            for l in ls:
                w = ops.dequant(l.decode_descriptor(), dev, torch.float16)
                torch.matmul(x, w)
        out = []
        for name, fn in (("fused", fused), ("two-step", twostep)):
            fn(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g): fn()
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): g.replay()
            e1.record(); torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / 10 / len(ls) * 1e3)
        print(f"HQQ w3 g64 {K}x{N} M={M}: plan={ops.plan_describe([ls[0].decode_descriptor()], M)[-24:]}  fused {out[0]:.1f} us, dequant + dense GEMM {out[1]:.1f} us")
PY
