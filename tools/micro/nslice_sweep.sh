#!/bin/bash
# the all-scores scan of the recall-0.95 point (nprobe 8: ~395 groups per query) by the number of slices a query's groups are cut into
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r04
cd /tmp && export TMPDIR=/tmp
for ns in "$@"; do
  MI_NSLICE=$ns timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ns$ns -o r -- python $GRAFT_REPO_ROOT/bench.py --no-encode --no-cfg5 --no-cpu-baseline --streams 1 --steps 10 > /dev/null 2>&1
  python - <<PY
import csv, glob
f = glob.glob("/tmp/p_ns$ns/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "scan_kernel<64, 8, true, false>" in r["Name"] or "select_pairs_kernel<32, true>" in r["Name"]:
        print("nslice $ns", r["Name"][:44], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
done
