#!/bin/bash
# GPU call: (1) chain timeline under rocprofv3 --kernel-trace (chain on / off), (2) gemm3 parity + prefill perf, (3) strip cpl sweep on TP shards
tag=${1:-r02b}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 -k "wave_specialised or prefill_kernel_vs_oracle or load_and_eval" > gpurun_out/${tag}_pytest_g3.log 2>&1; tail -4 gpurun_out/${tag}_pytest_g3.log
for g3 in 1 0; do QLLM_GEMM3=$g3 timeout 200 python tools/kbench.py --m 2048 8192 --iters 100 --layouts GPTQ GEMM 2>&1 | sed "s/^/GEMM3=$g3 /"; done > gpurun_out/${tag}_prefill.log; cat gpurun_out/${tag}_prefill.log
cd /tmp
for ch in 1 0; do
  rm -rf /tmp/tr_$ch
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$ch -o t -- python $R/bench.py --steps 3 --warmup 1 --no-extra --no-pmc --chain $ch > $R/gpurun_out/${tag}_trace_bench_$ch.json 2> /tmp/tr_$ch.err
  python $R/tools/trace_overlap.py /tmp/tr_$ch 256 > $R/gpurun_out/${tag}_overlap_chain$ch.txt 2>&1; head -60 $R/gpurun_out/${tag}_overlap_chain$ch.txt; tail -8 $R/gpurun_out/${tag}_overlap_chain$ch.txt
done
cd $R
for c in 0 2 4; do QLLM_STRIP_MIN=8 QLLM_STRIP_CPL=$c timeout 200 python tools/narrow_ab.py 2>&1 | sed "s/^/CPL=$c /"; done > gpurun_out/${tag}_cpl.log; cat gpurun_out/${tag}_cpl.log
