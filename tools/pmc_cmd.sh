#!/bin/bash
# usage: tools/pmc_cmd.sh <tag> "<counters>" <command...>   (GPU box) -- PMC pass (own run, kernel-trace only)
tag=$1; shift; ctrs=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o r -- "$@" > $out/cmd.log 2>&1
echo "rocprofv3 rc=$?"
ls $out
