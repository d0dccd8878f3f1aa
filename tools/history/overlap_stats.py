"""HISTORY (rounds 1-3): kept for the record of how a number in DESIGN_HISTORY.md / profiles/ was produced.  NOT maintained: knobs it
sets may no longer exist (MI_RERANK, MI_REFINE_DEBUG, MI_SCAN_DEBUG ... are silent no-ops now) and paths may have moved."""
"""Concurrency statistics from a rocprofv3 kernel_trace.csv of the multi-stream bench:
per kernel the duration under overlap, plus how long 1, 2, 3... kernels ran at once.
usage: python tools/overlap_stats.py <kernel_trace.csv> [last_n_dispatches]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 1800
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        if "mi::" in n:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0][:60]))
rows.sort()
# drop the trailing back-to-back scan replays of mi_index_profile_scan
while rows and "scan_kernel" in rows[-1][2] and len(rows) > 2 and "scan_kernel" in rows[-2][2]:
    rows.pop()
rows = rows[-last:]
dur = defaultdict(list)
for s, e, n in rows:
    dur[n].append(e - s)
for n, v in dur.items():
    v.sort()
    print(f"{n:62s} n={len(v):5d} median {v[len(v)//2]/1e3:7.2f} us  p10 {v[len(v)//10]/1e3:7.2f}  p90 {v[len(v)*9//10]/1e3:7.2f}")
ev = []
for s, e, _ in rows:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
busy = defaultdict(int)
cur, t_prev = 0, ev[0][0]
for t, d in ev:
    busy[cur] += t - t_prev
    cur += d
    t_prev = t
tot = sum(busy.values())
print("span %.1f us for %d dispatches (%.2f us per 3 dispatches)" % (tot / 1e3, len(rows), tot / 1e3 / (len(rows) / 3)))
for k in sorted(busy):
    print(f"  {k} kernels in flight: {100.0 * busy[k] / tot:5.1f} %")
