#!/bin/bash
# usage: tools/prof_encode_pmc.sh <tag>   (GPU box) -- SQ counters of the encoder's bf16 GEMM kernels
# (one --pmc pass, kernel-trace only) + the per-kernel stats of the same command
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --workload encode --steps 3 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o r -- $B > $out/bench_stats.json 2> $out/bench_stats.err
echo "stats rc=$?"
f=$(find $out/stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -12 "$f" | cut -c1-160
find $out/stats -name "*kernel_trace.csv" -size +8M -delete
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES \
  --kernel-trace --kernel-include-regex "gemm_bf16_(ring|slab)" --output-format csv -d $out/pmc -o r -- $B > $out/bench_pmc.json 2> $out/bench_pmc.err
echo "pmc rc=$?"
f=$(find $out/pmc -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python $GRAFT_REPO_ROOT/tools/pmc_summarize.py "$f" gemm_bf16_
find $out/pmc -name "*kernel_trace.csv" -size +8M -delete
