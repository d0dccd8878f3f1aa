"""HISTORY (rounds 1-3): kept for the record of how a number in DESIGN_HISTORY.md / profiles/ was produced.  NOT maintained: knobs it
sets may no longer exist (MI_RERANK, MI_REFINE_DEBUG, MI_SCAN_DEBUG ... are silent no-ops now) and paths may have moved."""
"""Per-phase kernel durations from a rocprofv3 kernel_trace.csv.
usage: python tools/trace_groups.py <trace.csv> <kernel-substring> <launches-per-group>
Prints the median duration of every consecutive group of launches of that kernel."""
import csv
import statistics
import sys

path, sub, per = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        if sub in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows.sort()
d = [x[1] for x in rows]
for i in range(0, len(d), per):
    g = d[i:i + per]
    print(f"group {i // per}: n={len(g)} median {statistics.median(g) / 1e3:.2f} us  min {min(g) / 1e3:.2f} us")
