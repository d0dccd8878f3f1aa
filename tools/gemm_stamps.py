"""In-kernel phase stamps (s_memtime ticks) of the slab GEMM on one shape, one tile per workgroup vs persistent:
MI_GEMM_TS=1 makes mi_enc_gemm_bf16 print mean / max of: first slabs landed, K loop done, epilogue issued, stores acknowledged,
and the gap between consecutive units of a CU.  usage: python tools/gemm_stamps.py [M N K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import abstracts_search_amd.sentence_transformers as st
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (27958, 17920, 1536)
A = torch.randn((M, K), device="cuda").bfloat16(); W = (torch.randn((N, K), device="cuda") / K ** 0.5).bfloat16()
for persist in (0, 1):
    os.environ["MI_GEMM_PERSIST"] = str(persist)
    os.environ.pop("MI_GEMM_TS", None)
    st.reload_env()
    for _ in range(3):
        st.gemm_bf16(A, W)
    torch.cuda.synchronize()
    os.environ["MI_GEMM_TS"] = "1"
    st.reload_env()
    print(f"--- MI_GEMM_PERSIST={persist}", file=sys.stderr, flush=True)
    st.gemm_bf16(A, W)
    torch.cuda.synchronize()
