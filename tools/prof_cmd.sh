#!/bin/bash
# usage: tools/prof_cmd.sh <tag> <command...>  -- rocprofv3 kernel stats of an arbitrary command (GPU box)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o r -- "$@" > $out/cmd.log 2>&1
echo "rocprofv3 rc=$?"
cd $out
f=$(find . -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -16 "$f" | cut -c1-180
find . -name "*kernel_trace.csv" -size +8M -delete
