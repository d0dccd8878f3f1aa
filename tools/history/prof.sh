#!/bin/bash
# HISTORY (rounds 1-3): kept for the record; NOT maintained -- knobs it sets may no longer exist (silent no-ops), paths may have moved.
# usage: tools_prof.sh <tag> <bench args...>   (run on the GPU box from the repo root)
# rocprofv3 kernel trace + stats of bench.py, bounded by timeout; CSV summaries land in gpurun_out/<tag>/
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o r -- python $GRAFT_REPO_ROOT/bench.py "$@" > $out/bench.log 2>&1
echo "rocprofv3 rc=$?"
cd $out
f=$(find . -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -14 "$f" | cut -c1-200
# the raw per-dispatch trace is large: keep only the stats summaries
find . -name "*kernel_trace.csv" -size +8M -delete
