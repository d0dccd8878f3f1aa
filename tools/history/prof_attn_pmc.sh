#!/bin/bash
# HISTORY (rounds 1-3): kept for the record; NOT maintained -- knobs it sets may no longer exist (silent no-ops), paths may have moved.
# usage: tools/prof_attn_pmc.sh <tag>   (GPU box) -- SQ counters of the encoder's attention kernel (one --pmc pass)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --workload encode --steps 3 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU \
  --kernel-trace --kernel-include-regex "attn_kernel" --output-format csv -d $out/pmc -o r -- $B "$@" > $out/bench_pmc.json 2> $out/bench_pmc.err
echo "pmc rc=$?"
f=$(find $out/pmc -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python $GRAFT_REPO_ROOT/tools/pmc_summarize.py "$f" attn_kernel
find $out/pmc -name "*kernel_trace.csv" -size +8M -delete
