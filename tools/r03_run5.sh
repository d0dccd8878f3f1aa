mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ivfpq_gpu.py tests/test_pipeline_encoder_gpu.py tests/test_threads_gpu.py tests/test_pipeline_search_e2e_gpu.py -m gpu -q 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof5 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof5 -o r03 -- python $GRAFT_REPO_ROOT/bench.py --no-encode --no-cpu-baseline --streams 1 > $GRAFT_REPO_ROOT/gpurun_out/r03_bench_v3_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r03_bench_v3.err
cd $GRAFT_REPO_ROOT
tail -4 gpurun_out/r03_bench_v3.err | cut -c1-200
cp $(find /tmp/prof5 -name "*kernel_stats.csv" | head -1) gpurun_out/r03_cfg4_kernel_stats_v1.csv; head -24 gpurun_out/r03_cfg4_kernel_stats_v1.csv | cut -c1-220
python -c "
import json; d=json.load(open('gpurun_out/r03_bench_v3_under_rocprof.json')); print(d['value'], d['ms_per_step'], json.dumps(d['at_recall_095'])[:600])"
echo "== encode default (async path)"; python bench.py --workload encode --no-cpu-baseline --steps 16 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['roofline'])"
echo "== gemm zero vs random"; for z in 0 1; do ZERO=$z M=29312 python tools/gemm_bench.py 2>&1 | grep -v "^RCCL\|version\|Hostname\|Librccl\|amdgpu.ids"; done
