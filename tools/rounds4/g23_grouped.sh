#!/bin/bash
# round 4: sibling groups around the panel kernel: the regression test, grouped strips vs layer-by-layer panel launches at 17..32
# rows, the HQQ batch-16 leg with down_proj on the panel kernel
tag=${1:-r04ad}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_decode_step_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -3
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from gpu_util import synth, to_layer
from qllm_amd import ops
DEV = "cuda:0"
def timed(fn, reps=30):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(4): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * 4)
for name, widths in (("q/k/v", (4096, 4096, 4096)), ("gate/up", (11008, 11008))):
    layers = [to_layer(synth("GPTQ", 4, 128, 4096, n, seed=n + i), DEV) for i, n in enumerate(widths)]
    ws = [l.native_descriptor(0) for l in layers]
    for m in (17, 24, 32):
        x = torch.from_numpy(np.random.default_rng(m).standard_normal((m, 4096)).astype(np.float16)).to(DEV)
        tg = timed(lambda: ops.linear_forward_grouped(ws, x))
        ts = timed(lambda: [ops.linear_forward(w, x) for w in ws])
        print(f"{name:8s} M={m:2d}  grouped {tg:6.2f} us [{ops.plan_describe(ws, m)[:50]}]   one by one {ts:6.2f} us [{ops.plan_describe([ws[0]], m)[:40]}]", flush=True)
PY
timeout 200 python tools/hqq_leg.py 10 2>&1 | grep -v amdgpu.ids
