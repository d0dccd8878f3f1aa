cd /tmp && export TMPDIR=/tmp
BENCH_NO_EXHAUSTIVE=1 timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o r -- python $GRAFT_REPO_ROOT/bench.py --no-encode --no-cpu-baseline --no-refine-point --no-recall --steps 6 --warmup 2 > /tmp/tr.log 2>&1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the host_io loop: the last launches before the profile replays; print the last 120 launches with gaps
tail=rows[-140:]
t0=int(tail[0]["Start_Timestamp"])
for r in tail:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print(f'{(s-t0)/1e3:10.1f} +{(e-s)/1e3:8.1f} us  {r["Kernel_Name"][:70]}')
P
