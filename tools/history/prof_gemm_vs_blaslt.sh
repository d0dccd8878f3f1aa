#!/bin/bash
# HISTORY (rounds 1-3): kept for the record; NOT maintained -- knobs it sets may no longer exist (silent no-ops), paths may have moved.
# L2-miss traffic and kernel names of the slab GEMM against hipBLASLt on the encoder's shapes (GPU box) -> gpurun_out/r03_blaslt/
out=$GRAFT_REPO_ROOT/gpurun_out/r03_blaslt
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
E="python $GRAFT_REPO_ROOT/tools/gemm_bench.py"
export M=29312
rm -rf /tmp/p_g_stats; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_g_stats -o r -- $E > $out/bench_under_stats.txt 2> $out/stats.err
cp $(find /tmp/p_g_stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_g_$c; timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/p_g_$c -o r -- $E > $out/bench_under_$c.txt 2> $out/$c.err
  cp $(find /tmp/p_g_$c -name "*counter_collection.csv" | head -1) $out/counters_$c.csv
done
cd $GRAFT_REPO_ROOT; cut -c1-200 $out/kernel_stats.csv | head -14; ls -la $out
