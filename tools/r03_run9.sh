mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pipeline_encoder_gpu.py -m gpu -q -x -k "tile_configs or query_batches" 2>&1 | tail -4
python tools/gemm_mid_bench.py 2>&1 | grep -v "^RCCL\|version\|Hostname\|Librccl\|amdgpu.ids" | grep -v gate_up | tee gpurun_out/r03_gemm_mid_bench_v2.txt
for v in 0 1; do echo "MI_MID64=$v"; if [ $v = 1 ]; then export MI_MID64=1; fi; for nq in 16 64; do ENC_NQ=$nq python tools/encode_mid_prof.py 2>/dev/null | tail -1; done; done
