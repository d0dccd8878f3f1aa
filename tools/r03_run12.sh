timeout 1500 python -m pytest tests/test_bench_multirank_gpu.py tests/test_pipeline_search_e2e_gpu.py -m gpu -q -x -rf 2>&1 | grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl" | tail -30
