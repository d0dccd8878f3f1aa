#!/bin/bash
# HISTORY (rounds 1-3): kept for the record; NOT maintained -- knobs it sets may no longer exist (silent no-ops), paths may have moved.
# round-3 closing run (GPU box): the whole GPU suite, then the default bench line
mkdir -p gpurun_out
timeout 2000 python -m pytest tests -m gpu -q -rf 2>&1 | grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl" | tail -8 | tee gpurun_out/r03_gputest_final.log
python bench.py > gpurun_out/r03_bench_final.json 2> gpurun_out/r03_bench_final.err; echo "bench rc=$?"; tail -3 gpurun_out/r03_bench_final.err
python -c "
import json; d=json.load(open('gpurun_out/r03_bench_final.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac']); a=d['at_recall_095']; print(a['qps'], a['ms_per_step'], a['recall_at_10']); e=d['encode']; print(e['abstracts_per_s'], e['ms_per_step'], e['roofline']['achieved'], e['roofline']['kernel_trace_cross_check'], e['parity_vs_oracle']['bulk_path']['min_cosine'])"
