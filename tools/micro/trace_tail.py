"""Per-kernel durations of the LAST launches in a rocprofv3 kernel trace (csv): the timed calls at the end of a run whose
setup (training, fills) launched the same kernels thousands of times.  usage: trace_tail.py <kernel_trace.csv> [n=40] [substr ...]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
subs = sys.argv[3:]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
if subs:
    rows = [r for r in rows if any(s in r["Kernel_Name"] for s in subs)]
for r in rows[-n:]:
    print(f'{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:10.2f} us  lds {r["LDS_Block_Size"]:>6} vgpr {r["VGPR_Count"]:>3} scr {r["Scratch_Size"]:>4}  {r["Kernel_Name"][:90]}')
