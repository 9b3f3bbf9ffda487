#!/bin/bash
# Round-3 / round-4 evidence run on the GPU box (one gpurun call, profiling only):  bash tools/profile_round3.sh [tag]
#   1. rocprofv3 --kernel-trace --stats of `bench.py --no-extra` (the headline decode step), then FETCH_SIZE / WRITE_SIZE in
#      their own passes                                             -> gpurun_out/prof_<tag>/{trace,pmc_fetch,pmc_write}
#   2. BASELINE configs[3] (tools/hqq_leg.py): kernel trace + FETCH_SIZE / WRITE_SIZE passes   -> .../hqq_{trace,fetch,write}
#   3. prefill: kbench under --kernel-trace --stats, two SQ counter passes on the 4096x4096 M=2048 GEMM, hipBLASLt context line
# tools/summarize_prof3.py condenses everything into profiles/.
tag=${1:-r03}
R=$GRAFT_REPO_ROOT
P=$R/gpurun_out/prof_$tag
rm -rf $P; mkdir -p $P
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -o t -- python $R/bench.py --steps 20 --warmup 3 --no-extra --min-timed-s 0 > $P/bench_under_rocprof.json 2> $P/rocprof_trace.err
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch -o f -- python $R/bench.py --steps 5 --warmup 2 --no-extra --min-timed-s 0 > /dev/null 2> $P/rocprof_fetch.err
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/pmc_write -o w -- python $R/bench.py --steps 5 --warmup 2 --no-extra --min-timed-s 0 > /dev/null 2> $P/rocprof_write.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $P/hqq_trace -o h -- python $R/tools/hqq_leg.py 10 > $P/hqq_leg.log 2> $P/rocprof_hqq.err
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/hqq_fetch -o f -- python $R/tools/hqq_leg.py 3 8 > /dev/null 2> $P/rocprof_hqq_fetch.err
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/hqq_write -o w -- python $R/tools/hqq_leg.py 3 8 > /dev/null 2> $P/rocprof_hqq_write.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $P/prefill -o p -- python $R/tools/kbench.py --m 2048 --iters 40 --layouts GPTQ GEMM > $P/prefill_kbench.log 2> $P/rocprof_prefill.err
# (round 6) the prefill step as the modules run it: 4 launches per decoder layer (q/k/v and gate/up grouped), AWQ fp16 leg only
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $P/prefill_step -o s -- python $R/tools/prefill_legs.py 10 awq > $P/prefill_step.log 2> $P/rocprof_prefill_step.err
cd $R
bash tools/pmc_pass.sh ${tag}_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -- python tools/one_shape.py > /dev/null
timeout 100 python tools/one_shape.py --ref --iters 40 > gpurun_out/${tag}_hipblaslt_ref.log 2>&1; tail -2 gpurun_out/${tag}_hipblaslt_ref.log
# keep what the summaries need (gpurun_out is capped at 64 MiB): only this library's kernels in the per-dispatch CSVs
for f in $(find $P -name "*kernel_trace.csv" -o -name "*counter_collection.csv"); do
  (head -1 $f; grep "qllm::" $f) > $f.tmp && mv $f.tmp $f
done
find $P -name "*agent_info*" -delete
du -sh $R/gpurun_out
cat $P/hqq_leg.log; grep -h "GPTQ\|GEMM" $P/prefill_kbench.log; tail -c 600 $P/bench_under_rocprof.json
