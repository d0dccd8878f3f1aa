F='^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
python -m pytest tests/test_ivfpq_gpu.py -m gpu -x -q 2>&1 | grep -v "$F" | tail -6 > gpurun_out/s3_gputests5.txt
for nt in 512; do echo "== NT $nt"; MI_IVFPQ_LIB=libmi_ivfpq_ts.so MI_SELP_NT=$nt python tools/micro/selp_stamps.py 2>&1 | grep -v "$F"; done > gpurun_out/s3_selp_stamps4.txt
python bench.py --no-encode --no-cpu-baseline > gpurun_out/s3_bench_selp3.json 2> gpurun_out/s3_bench_selp3.err
cat gpurun_out/s3_gputests5.txt gpurun_out/s3_selp_stamps4.txt; python - <<'P'
import json
d=json.load(open('gpurun_out/s3_bench_selp3.json'))
print(d['value'], d['ms_per_step']); r=d['at_recall_095']; print(r['ms_per_step'], r.get('recall_at_10'), r['roofline']['step_split_ms'], r.get('parity_vs_oracle'))
P
