#!/bin/bash
tag=${1:-r02u}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "act_order or gather" > gpurun_out/${tag}_pytest_ao.log 2>&1; tail -5 gpurun_out/${tag}_pytest_ao.log
timeout 900 python bench.py --no-pmc 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/${tag}_bench.json
python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["unit"], d["roofline"]["frac"])
for k,v in d.get("extra",{}).items():
    if k.startswith(("prefill","hqq","decode_stack")): print(k, v)
PY
