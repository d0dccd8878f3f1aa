mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ivfpq_gpu.py tests/test_threads_gpu.py -m gpu -q -x 2>&1 | tail -6
cd /tmp && export TMPDIR=/tmp
for nq in 16 256; do
rm -rf /tmp/pm$nq; ENC_NQ=$nq rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm$nq -o r -- python $GRAFT_REPO_ROOT/tools/encode_mid_prof.py 2>/dev/null | tail -1
cp $(find /tmp/pm$nq -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r03_encode_nq${nq}_kernel_stats.csv
head -16 $GRAFT_REPO_ROOT/gpurun_out/r03_encode_nq${nq}_kernel_stats.csv | cut -c1-150
done
cd $GRAFT_REPO_ROOT
python bench.py --no-encode --no-cpu-baseline > gpurun_out/r03_bench_v4.json 2> gpurun_out/r03_bench_v4.err; tail -3 gpurun_out/r03_bench_v4.err
python -c "
import json; d=json.load(open('gpurun_out/r03_bench_v4.json')); print(d['value'], d['ms_per_step'], json.dumps(d['at_recall_095'])[:700])"
