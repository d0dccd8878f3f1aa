#!/bin/bash
# cfg2 (1 M x 1024, IVF4096,PQ64, batch 64, nprobe 16): FETCH_SIZE / WRITE_SIZE of the scan kernel, each in its own pass, and the
# kernel stats + bench line of the same command (GPU box).  -> gpurun_out/r04_cfg2/
out=$GRAFT_REPO_ROOT/gpurun_out/r04_cfg2
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --workload cfg2 --streams 1 --steps 100 --no-cpu-baseline"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c2s -o r -- $B > $out/cfg2_under_stats.json 2> $out/stats.err
cp $(find /tmp/p_c2s -name "*kernel_stats.csv" | head -1) $out/cfg2_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "scan_kernel" --output-format csv -d /tmp/p_c2_$c -o r -- $B > /dev/null 2> $out/$c.err
  python $GRAFT_REPO_ROOT/tools/pmc_summarize.py $(find /tmp/p_c2_$c -name "*counter_collection.csv" | head -1) scan_kernel > $out/cfg2_$c.txt
done
timeout 300 python $GRAFT_REPO_ROOT/bench.py --workload cfg2 --no-cpu-baseline > $out/cfg2_plain.json 2> $out/plain.err
cat $out/cfg2_FETCH_SIZE.txt $out/cfg2_WRITE_SIZE.txt; head -4 $out/cfg2_kernel_stats.csv | cut -c1-160
