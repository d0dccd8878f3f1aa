#!/bin/bash
# HISTORY (rounds 1-3): kept for the record; NOT maintained -- knobs it sets may no longer exist (silent no-ops), paths may have moved.
# usage: tools/prof_r03.sh   (GPU box, from the repo root) -- round-3 rocprofv3 evidence in one go:
#   encode (cfg3 shape): --kernel-trace --stats; one SQ-counter pass; FETCH_SIZE and WRITE_SIZE passes (GEMM kernels only)
#   cfg4 (207 M): --kernel-trace --stats of the default line (with the whole-index refine point); FETCH / WRITE passes of the scan kernel
# Counters always in their own passes with --kernel-trace only (never with other trace domains).  Summaries -> gpurun_out/r03_prof/
out=$GRAFT_REPO_ROOT/gpurun_out/r03_prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
E="python $GRAFT_REPO_ROOT/bench.py --workload encode --steps 4 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_enc_stats -o r -- $E > $out/encode_under_stats.json 2> $out/encode_stats.err
cp $(find /tmp/p_enc_stats -name "*kernel_stats.csv" | head -1) $out/encode_kernel_stats.csv
timeout 600 $E > $out/encode_plain.json 2> /dev/null
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES \
  --kernel-trace --kernel-include-regex "gemm_bf16_(ring|slab)" --output-format csv -d /tmp/p_enc_sq -o r -- $E > /dev/null 2> $out/encode_sq.err
python $GRAFT_REPO_ROOT/tools/pmc_summarize.py $(find /tmp/p_enc_sq -name "*counter_collection.csv" | head -1) gemm_bf16_ > $out/encode_gemm_sq.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "gemm_bf16_(ring|slab)" --output-format csv -d /tmp/p_enc_$c -o r -- $E > $out/encode_under_$c.json 2> $out/encode_$c.err
  python $GRAFT_REPO_ROOT/tools/pmc_summarize.py $(find /tmp/p_enc_$c -name "*counter_collection.csv" | head -1) gemm_bf16_ > $out/encode_gemm_$c.txt
done
B="python $GRAFT_REPO_ROOT/bench.py --no-encode --no-cpu-baseline --streams 1"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_cfg4_stats -o r -- $B > $out/cfg4_under_stats.json 2> $out/cfg4_stats.err
cp $(find /tmp/p_cfg4_stats -name "*kernel_stats.csv" | head -1) $out/cfg4_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "scan_kernel|rerank_sq8|select_pairs" --output-format csv -d /tmp/p_cfg4_$c -o r -- $B --no-recall --steps 10 > $out/cfg4_under_$c.json 2> $out/cfg4_$c.err
  python $GRAFT_REPO_ROOT/tools/pmc_summarize.py $(find /tmp/p_cfg4_$c -name "*counter_collection.csv" | head -1) "" > $out/cfg4_$c.txt
done
cd $GRAFT_REPO_ROOT
head -12 $out/encode_kernel_stats.csv | cut -c1-180
cat $out/encode_gemm_FETCH_SIZE.txt $out/encode_gemm_WRITE_SIZE.txt
cat $out/cfg4_FETCH_SIZE.txt $out/cfg4_WRITE_SIZE.txt
python -c "
import json
for f in ('encode_plain','encode_under_stats'):
    d=json.load(open('$out/'+f+'.json')); print(f, d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['gemm_ms_per_step'])"
