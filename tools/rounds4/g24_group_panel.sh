#!/bin/bash
# round 4: grouped launches of the panel kernel (q/k/v, gate/up from 17 rows): parity through the modules, timing against the layers one by one
tag=${1:-r04ae}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_decode_step_gpu.py tests/test_native_layout_gpu.py tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "sibling or group or panel or mid_batch or refused" > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/${tag}_pytest.log
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${tag}_grouped.log
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
    for m in (16, 17, 32, 64, 128):
        x = torch.from_numpy(np.random.default_rng(m).standard_normal((m, 4096)).astype(np.float16)).to(DEV)
        tg = timed(lambda: ops.linear_forward_grouped(ws, x))
        ts = timed(lambda: [ops.linear_forward(w, x) for w in ws])
        print(f"{name:8s} M={m:3d}  grouped {tg:6.2f} us [{ops.plan_describe(ws, m)[:58]}]   one by one {ts:6.2f} us [{ops.plan_describe([ws[0]], m)[:30]}]", flush=True)
PY
