#!/bin/bash
tag=${1:-r02o}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 -k "gather or act_order" > gpurun_out/${tag}_pytest_gather.log 2>&1; tail -5 gpurun_out/${tag}_pytest_gather.log
timeout 200 python tools/gather_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${tag}_gather.log
for v in "QLLM_GEMM3_MW=4" "QLLM_GEMM3_MW=8" "QLLM_GEMM2_RASTER=0"; do env $v timeout 200 python tools/kbench.py --m 2048 --iters 100 --layouts GPTQ GEMM 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"; done > gpurun_out/${tag}_prefill.log; cat gpurun_out/${tag}_prefill.log
timeout 600 python bench.py --no-pmc 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/${tag}_bench.json; python - <<'PY'
import json
d=json.loads(open("gpurun_out/'${tag}'_bench.json".replace("'","")).read().strip().splitlines()[-1])
print(d["value"], d["unit"], d["roofline"]["frac"])
for k,v in d.get("extra",{}).items(): print(k, v)
PY
