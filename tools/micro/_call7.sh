F='^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
MI_FEW_PF_MB=12 MI_FEW_SYNC=1 timeout 120 python tools/encode_b1.py 31 1 1 2>&1 | grep -v "$F" | grep -v "no error" | head -10 | cut -c1-300
for mb in 0 12 0 6 12 20 30; do echo "PF_MB=$mb $(MI_FEW_PF_MB=$mb timeout 120 python tools/encode_b1.py 31 200 1 2>&1 | grep 'ms per encode')"; done > gpurun_out/s3_few_pf.txt
cat gpurun_out/s3_few_pf.txt
