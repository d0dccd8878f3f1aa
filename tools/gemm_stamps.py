"""In-kernel phase stamps of the 256x256 ring GEMM (MI_GEMM_TS=1): where a tile's time goes.
usage: python tools/gemm_stamps.py   (GPU box; prints to stderr)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MI_GEMM_TS"] = "1"
import torch
import abstracts_search_amd.sentence_transformers as st
M = 29696
for N, K in ((17920, 64), (17920, 512), (17920, 1536), (2048, 1536)):
    A = torch.randn((M, K), device="cuda").bfloat16(); W = (torch.randn((N, K), device="cuda") / K ** 0.5).bfloat16()
    for _ in range(2): st.gemm_bf16(A, W)
    torch.cuda.synchronize()
