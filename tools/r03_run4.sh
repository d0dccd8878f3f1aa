mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ivfpq_gpu.py tests/test_pipeline_encoder_gpu.py tests/test_threads_gpu.py tests/test_pipeline_search_e2e_gpu.py -m gpu -q -x 2>&1 | tail -15
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof4 && rocprofv3 --kernel-trace --stats -d /tmp/prof4 -o r03 -- python $GRAFT_REPO_ROOT/bench.py --no-encode --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r03_bench_v2_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r03_bench_v2.err
cd $GRAFT_REPO_ROOT
tail -20 gpurun_out/r03_bench_v2.err
find /tmp/prof4 -name "*kernel_stats*" | head; cp $(find /tmp/prof4 -name "*kernel_stats.csv" | head -1) gpurun_out/r03_cfg4_kernel_stats_v1.csv; head -30 gpurun_out/r03_cfg4_kernel_stats_v1.csv | cut -c1-200
python -c "
import json; d=json.load(open('gpurun_out/r03_bench_v2_under_rocprof.json')); print(d['value'], d['ms_per_step'], json.dumps(d['at_recall_095'])[:1500])"
