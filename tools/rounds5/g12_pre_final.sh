#!/bin/bash
# round 5, call 12: the tests added since the last full run; kernel trace of the headline step (durations and gaps per launch);
# configs[3] with down_proj on the strips instead of the panel kernel (lab library: QLLM_PANEL=0)
tag=${1:-r05m}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_numerics_contract_gpu.py tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "contract or bf16 or act_order" > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/${tag}_pytest.log
for b in 4 3; do
  echo "== bits $b, product plans"; timeout 200 tools/lab/gbench_lab --cfg3 --bits $b --m 16 2>&1 | tee gpurun_out/${tag}_cfg3_w$b.log
  echo "== bits $b, QLLM_PANEL=0"; QLLM_PANEL=0 timeout 200 tools/lab/gbench_lab --cfg3 --bits $b --m 16 2>&1 | tee gpurun_out/${tag}_cfg3_w${b}_nopanel.log
done
P=$R/gpurun_out/prof_$tag; rm -rf $P; mkdir -p $P
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -o t -- python $R/bench.py --steps 20 --warmup 3 --no-extra --min-timed-s 0 > $P/bench_under_rocprof.json 2> $P/rocprof_trace.err)
for f in $(find $P -name "*kernel_trace.csv"); do (head -1 $f; grep "qllm::" $f) > $f.tmp && mv $f.tmp $f; done
find $P -name "*agent_info*" -delete
python tools/trace_overlap.py $P/trace 256 > gpurun_out/${tag}_overlap.log 2>&1; tail -12 gpurun_out/${tag}_overlap.log
tail -c 400 $P/bench_under_rocprof.json
timeout 200 tools/lab/cbench 2>&1 | tail -3
