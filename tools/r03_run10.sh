mkdir -p gpurun_out
timeout 2000 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee gpurun_out/r03_gputest_full.log
python bench.py > gpurun_out/r03_bench_v5.json 2> gpurun_out/r03_bench_v5.err; echo "bench rc=$?"; tail -4 gpurun_out/r03_bench_v5.err
python -c "
import json; d=json.load(open('gpurun_out/r03_bench_v5.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac']); print(json.dumps(d['at_recall_095'])[:400]); print(json.dumps(d['encode'])[:1500])"
