mkdir -p gpurun_out
python bench.py > gpurun_out/r03_bench_v1.json 2> gpurun_out/r03_bench_v1.err; echo "bench rc=$?"; tail -25 gpurun_out/r03_bench_v1.err; cat gpurun_out/r03_bench_v1.json | head -c 6000
for o in 0 1; do echo "== MI_TILE_ORDER=$o"; MI_TILE_ORDER=$o M=29312 python tools/gemm_bench.py 2>&1 | grep -v "^RCCL\|version\|Hostname\|Librccl"; done > gpurun_out/r03_gemm_order.txt 2>&1; cat gpurun_out/r03_gemm_order.txt
for o in 0 1; do echo "== encode MI_TILE_ORDER=$o"; MI_TILE_ORDER=$o python bench.py --workload encode --no-cpu-baseline --steps 16 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"; done > gpurun_out/r03_encode_order.txt 2>&1; cat gpurun_out/r03_encode_order.txt
