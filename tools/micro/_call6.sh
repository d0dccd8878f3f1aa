F='^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
for mb in 0 12 0 6 12 20 30; do echo "PF_MB=$mb $(MI_FEW_PF_MB=$mb python tools/encode_b1.py 31 200 1 2>&1 | grep 'ms per encode')"; done > gpurun_out/s3_few_pf.txt
for wv in 512 2048; do echo "PF_MB=12 WAVES=$wv $(MI_FEW_PF_MB=12 MI_FEW_PF_WAVES=$wv python tools/encode_b1.py 31 200 1 2>&1 | grep 'ms per encode')"; done >> gpurun_out/s3_few_pf.txt
cat gpurun_out/s3_few_pf.txt
python -m pytest tests/test_encoder_gpu.py -m gpu -x -q 2>&1 | grep -v "$F" | tail -4
