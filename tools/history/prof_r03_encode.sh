#!/bin/bash
# HISTORY (rounds 1-3): kept for the record; NOT maintained -- knobs it sets may no longer exist (silent no-ops), paths may have moved.
# the encode half of tools/prof_r03.sh (GPU box): kernel stats + FETCH / WRITE passes -> gpurun_out/r03_prof/
out=$GRAFT_REPO_ROOT/gpurun_out/r03_prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
E="python $GRAFT_REPO_ROOT/bench.py --workload encode --steps 4 --warmup 1 --no-cpu-baseline"
rm -rf /tmp/p_enc_stats; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_enc_stats -o r -- $E > $out/encode_under_stats.json 2> $out/encode_stats.err
cp $(find /tmp/p_enc_stats -name "*kernel_stats.csv" | head -1) $out/encode_kernel_stats.csv
timeout 600 $E > $out/encode_plain.json 2> /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_enc_$c; timeout 600 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "gemm_bf16_(ring|slab)" --output-format csv -d /tmp/p_enc_$c -o r -- $E > $out/encode_under_$c.json 2> $out/encode_$c.err
  python $GRAFT_REPO_ROOT/tools/pmc_summarize.py $(find /tmp/p_enc_$c -name "*counter_collection.csv" | head -1) gemm_bf16_ > $out/encode_gemm_$c.txt
done
cd $GRAFT_REPO_ROOT; head -9 $out/encode_kernel_stats.csv | cut -c1-170
