#!/bin/bash
# usage: tools/prof_r04.sh   (GPU box, from the repo root) -- round-4 rocprofv3 evidence:
#   cfg4 (207 M, incl. the recall-0.95 point): --kernel-trace --stats; FETCH_SIZE / WRITE_SIZE passes of scan / re-rank / selection
#   the query-time encoder (one 31-token query): --kernel-trace --stats; FETCH_SIZE pass of its six kernels
#   encode (cfg3): --kernel-trace --stats; FETCH_SIZE / WRITE_SIZE passes of its GEMM kernels (tools/pmc_encode_json.py makes the json bench.py reads)
# Counters always in their own passes with --kernel-trace only.  Summaries -> gpurun_out/r04_prof/
out=$GRAFT_REPO_ROOT/gpurun_out/r04_prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-encode --no-cpu-baseline --streams 1"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_cfg4_stats -o r -- $B > $out/cfg4_under_stats.json 2> $out/cfg4_stats.err
cp $(find /tmp/p_cfg4_stats -name "*kernel_stats.csv" | head -1) $out/cfg4_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "scan_kernel|rerank_sq8|select_pairs" --output-format csv -d /tmp/p_cfg4_$c -o r -- $B --no-recall --steps 10 > $out/cfg4_under_$c.json 2> $out/cfg4_$c.err
  python $GRAFT_REPO_ROOT/tools/pmc_summarize.py $(find /tmp/p_cfg4_$c -name "*counter_collection.csv" | head -1) "" > $out/cfg4_$c.txt
done
Q="python $GRAFT_REPO_ROOT/tools/encode_b1.py 31 40 1"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_b1_stats -o r -- $Q > $out/b1_under_stats.txt 2> $out/b1_stats.err
cp $(find /tmp/p_b1_stats -name "*kernel_stats.csv" | head -1) $out/b1_kernel_stats.csv
timeout 300 $Q > $out/b1_plain.txt 2>/dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "few_" --output-format csv -d /tmp/p_b1_fetch -o r -- $Q > /dev/null 2> $out/b1_fetch.err
python $GRAFT_REPO_ROOT/tools/pmc_summarize.py $(find /tmp/p_b1_fetch -name "*counter_collection.csv" | head -1) few_ > $out/b1_FETCH_SIZE.txt
E="python $GRAFT_REPO_ROOT/bench.py --workload encode --steps 4 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_enc_stats -o r -- $E > $out/encode_under_stats.json 2> $out/encode_stats.err
cp $(find /tmp/p_enc_stats -name "*kernel_stats.csv" | head -1) $out/encode_kernel_stats.csv
timeout 600 $E > $out/encode_plain.json 2> $out/encode_plain.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "gemm_bf16_(ring|slab)" --output-format csv -d /tmp/p_enc_$c -o r -- $E > /dev/null 2> $out/encode_$c.err
  python $GRAFT_REPO_ROOT/tools/pmc_summarize.py $(find /tmp/p_enc_$c -name "*counter_collection.csv" | head -1) gemm_bf16_ > $out/encode_gemm_$c.txt
done
cd $GRAFT_REPO_ROOT
head -12 $out/cfg4_kernel_stats.csv | cut -c1-170
cat $out/cfg4_FETCH_SIZE.txt $out/cfg4_WRITE_SIZE.txt $out/b1_FETCH_SIZE.txt
head -9 $out/b1_kernel_stats.csv | cut -c1-150; cat $out/b1_plain.txt $out/b1_under_stats.txt | tail -2
head -10 $out/encode_kernel_stats.csv | cut -c1-150
