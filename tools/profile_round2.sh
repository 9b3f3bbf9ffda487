#!/bin/bash
# End-of-round evidence run on the GPU box (one gpurun call):  bash tools/profile_round2.sh <tag>
#   1. full -m gpu parity suite                                  -> gpurun_out/<tag>_pytest.log
#   2. default bench.py line (incl. its own PMC passes)           -> gpurun_out/<tag>_bench.json
#   3. rocprofv3 --kernel-trace --stats of `bench.py --no-extra` (the headline form: plain grouped graph, one stream), then
#      FETCH_SIZE / WRITE_SIZE in their own passes                  -> gpurun_out/prof_<tag>/{trace,pmc_fetch,pmc_write}
#      (QLLM_CHAIN_SERIAL=1 only matters for `--chain 1` runs: under the tracer an overlapped chain shows 60 us kernels that wait)
#   4. prefill: kbench under --kernel-trace --stats, two SQ counter passes on the 4096x4096 M=2048 GEMM, hipBLASLt context line
# tools/summarize_prof2.py condenses (3)/(4) into profiles/.
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
P=$R/gpurun_out/prof_$tag
rm -rf $P; mkdir -p $P
cd $R
export TMPDIR=/tmp
if [ -z "$PROFILE_ONLY" ]; then
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > gpurun_out/${tag}_pytest.log 2>&1; tail -3 gpurun_out/${tag}_pytest.log
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -c 1500 gpurun_out/${tag}_bench.json
fi
cd /tmp
QLLM_CHAIN_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -o t -- python $R/bench.py --steps 20 --warmup 3 --no-extra > $P/bench_under_rocprof.json 2> $P/rocprof_trace.err
QLLM_CHAIN_SERIAL=1 timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch -o f -- python $R/bench.py --steps 5 --warmup 2 --no-extra > /dev/null 2> $P/rocprof_fetch.err
QLLM_CHAIN_SERIAL=1 timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/pmc_write -o w -- python $R/bench.py --steps 5 --warmup 2 --no-extra > /dev/null 2> $P/rocprof_write.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $P/prefill -o p -- python $R/tools/kbench.py --m 2048 --iters 40 --layouts GPTQ GEMM > $P/prefill_kbench.log 2> $P/rocprof_prefill.err
cd $R
bash tools/pmc_pass.sh ${tag}_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -- python tools/one_shape.py > /dev/null
bash tools/pmc_pass.sh ${tag}_sq2 SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F16 -- python tools/one_shape.py > /dev/null
timeout 100 python tools/one_shape.py --ref --iters 40 > gpurun_out/${tag}_hipblaslt_ref.log 2>&1; tail -2 gpurun_out/${tag}_hipblaslt_ref.log
# keep what the summaries need (gpurun_out is capped at 64 MiB): only this library's kernels in the per-dispatch CSVs
for f in $(find $P -name "*kernel_trace.csv" -o -name "*counter_collection.csv"); do
  (head -1 $f; grep "qllm::" $f) > $f.tmp && mv $f.tmp $f
done
find $P -name "*agent_info*" -delete
du -sh $R/gpurun_out
grep -h "GPTQ\|GEMM" $P/prefill_kbench.log
