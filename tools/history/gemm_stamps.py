"""HISTORY (rounds 1-3): kept for the record of how a number in DESIGN_HISTORY.md / profiles/ was produced.  NOT maintained: knobs it
sets may no longer exist (MI_RERANK, MI_REFINE_DEBUG, MI_SCAN_DEBUG ... are silent no-ops now) and paths may have moved."""
"""In-kernel phase stamps of the 256x256 GEMM kernels (MI_GEMM_TS=1): where a tile's time goes -- pipeline fill, K loop,
epilogue, store drain, and (slab kernel) the gap between consecutive workgroups of a CU.
usage: [ZERO=1] [M=32768] python tools/gemm_stamps.py   (GPU box; prints to stderr)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MI_GEMM_TS"] = "1"
import torch
import abstracts_search_amd.sentence_transformers as st
M = int(os.environ.get("M", 32768))
for N, K in ((1536, 1536), (1536, 8960), (4096, 4096), (17920, 1536)):
    A = torch.randn((M, K), device="cuda").bfloat16(); W = (torch.randn((N, K), device="cuda") / K ** 0.5).bfloat16()
    if os.environ.get("ZERO") == "1":
        A.zero_(); W.zero_()
    for _ in range(2): st.gemm_bf16(A, W)
    torch.cuda.synchronize()
