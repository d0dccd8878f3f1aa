timeout 900 python -m pytest tests/test_ivfpq_gpu.py -m gpu -q -k "large_k or candidates or sq8 or merge or refine" 2>&1 | tail -2
for cfg in "MI_GEMM_RING=1" "MI_GEMM_TILE=big32"; do for z in 0 1; do echo "== $cfg ZERO=$z"; env $cfg ZERO=$z M=29312 python tools/gemm_bench.py 2>&1 | grep -v "^RCCL\|version\|Hostname\|Librccl\|amdgpu.ids\|hipBLASLt"; done; done
