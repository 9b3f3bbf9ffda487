#!/bin/bash
tag=${1:-r02s}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
export QLLM_MI355X_LIB=$R/qllm_amd/libqllm_mi355x.so
timeout 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${tag}_l1_decode.log
import importlib.util, os, sys, torch
sys.path.insert(0, "tests")
from gpu_util import synth
spec = importlib.util.spec_from_file_location("stub", "integration/awq_inference_engine.py"); eng = importlib.util.module_from_spec(spec); spec.loader.exec_module(eng)
dev = "cuda:0"
for (K, N) in ((4096, 4096), (4096, 11008), (11008, 4096)):
    ws = []
    for i in range(12):
        d = synth("GEMM", 4, 128, K, N, seed=i)
        ws.append(tuple(torch.from_numpy(d[k]).to(dev) for k in ("qweight", "scales", "qzeros")))
    x = torch.randn(1, K, device=dev, dtype=torch.float16)
    res = {}
    for sh in ("1", "0"):
        os.environ["QLLM_AWQ_DECODE_SHADOW"] = sh
        for w in ws: eng.gemm_forward_cuda(x, *w, 8)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for w in ws: eng.gemm_forward_cuda(x, *w, 8)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): g.replay()
        e1.record(); torch.cuda.synchronize()
        res[sh] = e0.elapsed_time(e1) / 20 / len(ws) * 1e3
    print(f"Level-1 gemm_forward_cuda M=1 {K}x{N}: row-stream copy {res['1']:.2f} us, in place {res['0']:.2f} us")
PY
