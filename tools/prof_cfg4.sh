#!/bin/bash
# usage: tools/prof_cfg4.sh <tag> [bench args...]   (GPU box, from the repo root)
# rocprofv3 evidence for the default bench line (cfg4, 207 M): one --kernel-trace --stats run and
# two separate --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) restricted to the
# scan kernel.  Summaries land in gpurun_out/<tag>/; copy them to profiles/.
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-encode --no-cpu-baseline --no-refine-point --no-recall --streams 1 $*"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o r -- $B > $out/bench_stats.json 2> $out/bench_stats.err
echo "stats rc=$?"
f=$(find $out/stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -14 "$f" | cut -c1-200
find $out/stats -name "*kernel_trace.csv" -size +8M -delete
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "scan_kernel" --output-format csv -d $out/pmc_$c -o r -- $B --steps 10 > $out/bench_pmc_$c.json 2> $out/bench_pmc_$c.err
  echo "pmc $c rc=$?"
  f=$(find $out/pmc_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $GRAFT_REPO_ROOT/tools/pmc_summarize.py "$f" scan_kernel
  find $out/pmc_$c -name "*kernel_trace.csv" -size +8M -delete
done
