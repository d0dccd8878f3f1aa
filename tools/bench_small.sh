#!/bin/bash
# a short end-to-end run of the default bench line on a small corpus (code-path check of every sub-object): GPU box
mkdir -p gpurun_out/r04
timeout 1300 python bench.py --corpus 8388608 --nlist 8192 --steps 10 --encode-steps 4 "$@" > gpurun_out/r04/bench_small.json 2> gpurun_out/r04/bench_small.log
echo rc=$?
tail -5 gpurun_out/r04/bench_small.log
python - <<PY
import json
o=json.load(open("gpurun_out/r04/bench_small.json"))
a=o["at_recall_095"]
print({k:a[k] for k in ("index","refine_store","nprobe","k_factor_rf","recall_at_10","recall_at_10_selection_batch","qps","ms_per_step")})
print(a["roofline"]); print(a["parity_vs_oracle"])
for c in o["cfg5"]["curve"]: print(c)
print(o["cfg5"]["cpu_baseline"]); print(o["cfg5"]["parity_vs_oracle"])
print(o["value"], o["roofline"]["frac"], o["encode"]["abstracts_per_s"])
PY
