"""HISTORY (rounds 1-3): kept for the record of how a number in DESIGN_HISTORY.md / profiles/ was produced.  NOT maintained: knobs it
sets may no longer exist (MI_RERANK, MI_REFINE_DEBUG, MI_SCAN_DEBUG ... are silent no-ops now) and paths may have moved."""
"""Durations of the coarse-selection kernel per search step from a rocprofv3 kernel trace
(steps are recognised by the small-batch score GEMM in front of it), grouped by nprobe run.
usage: python tools/select_from_trace.py <kernel_trace.csv> <steps_per_run>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
per = int(sys.argv[2]) if len(sys.argv) > 2 else 53
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
steps = []
for i, (a, b) in enumerate(zip(rows, rows[1:])):
    if "ip_gemm_kernel<1, 1," in a["Kernel_Name"] and "select" in b["Kernel_Name"]:
        scan = next((dur(c) for c in rows[i + 2:i + 5] if "scan_kernel" in c["Kernel_Name"]), 0.0)
        steps.append((b["Kernel_Name"].split("(")[0].replace("mi::", ""), dur(a), dur(b), scan))
for g in range(0, len(steps), per):
    s = steps[g:g + per]
    print(f"run {g // per}: {s[0][0]:20s} gemm {sum(x[1] for x in s) / len(s):8.1f}  select {sum(x[2] for x in s) / len(s):8.1f}  "
          f"scan {sum(x[3] for x in s) / len(s):8.1f} us  ({len(s)} steps)")
