mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_pipeline_encoder_gpu.py tests/test_pipeline_search_e2e_gpu.py tests/test_threads_gpu.py -m gpu -q 2>&1 | tail -8
echo "== mid gemm bench"; python tools/gemm_mid_bench.py 2>&1 | grep -v "^RCCL\|version\|Hostname\|Librccl\|amdgpu.ids" | tee gpurun_out/r03_gemm_mid_bench_v1.txt
echo "== encode (replay timing)"; python bench.py --workload encode --no-cpu-baseline --steps 16 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['roofline'])"
echo "== e2e curve"
for mid in old slab; do echo "MI_MID=$mid"; MI_MID=$mid E2E_N=2000000 E2E_NLIST=4096 python tools/e2e_curve.py 2>&1 | grep -v "^RCCL\|version\|Hostname\|Librccl\|amdgpu.ids" | tail -6; done
