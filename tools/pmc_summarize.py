"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, mean of each counter per dispatch."""
import csv, sys, collections
f = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r.get("Kernel_Name", "")
    if pat and pat not in k: continue
    acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} n={len(v):5d} mean={sum(v)/len(v):16.1f}")
