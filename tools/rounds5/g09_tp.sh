#!/bin/bash
# round 5: fused row-parallel GEMV + one-shot all-reduce (two processes on one GPU), the tests whose plan strings changed, bench line
tag=${1:-r05j}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_tp_collective_gpu.py tests/test_tp_shapes_gpu.py -m gpu -q -x --timeout 800 -s > gpurun_out/${tag}_pytest_tp.log 2>&1; echo "pytest tp rc=$?"; grep -E "tp_bench\]|passed|failed|Error|error" gpurun_out/${tag}_pytest_tp.log | tail -20
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "act_order_decode_sizes or wave_specialised" > gpurun_out/${tag}_pytest2.log 2>&1; echo "pytest2 rc=$?"; tail -4 gpurun_out/${tag}_pytest2.log
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/%s_bench.json" % "r05j").read().strip().splitlines()[-1])
print(json.dumps({k:d[k] for k in ("value","ms_per_step","sustained","roofline","roofline_prefill")}, indent=1))
print({k:v for k,v in d["extra"].items() if k.startswith(("hqq","tp_shard","prefill"))})
print(d["cpu_baseline"])
PY
tail -3 gpurun_out/${tag}_bench.err
