"""HISTORY (rounds 1-3): kept for the record of how a number in DESIGN_HISTORY.md / profiles/ was produced.  NOT maintained: knobs it
sets may no longer exist (MI_RERANK, MI_REFINE_DEBUG, MI_SCAN_DEBUG ... are silent no-ops now) and paths may have moved."""
"""bf16 GEMM time at the encoder's shapes for few-hundred- to few-thousand-token batches (cfg5's batch 16 / 256),
per tile configuration (MI_GEMM_TILE): which existing kernel is the best starting point for the mid-batch path."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    import abstracts_search_amd.sentence_transformers as st
    for M in (576, 1152):
        for name, N, K in (("qkv", 2048, 1536), ("o", 1536, 1536), ("gate_up", 17920, 1536), ("down", 1536, 8960)):
            A = torch.randn((M, K), device="cuda").bfloat16(); W = (torch.randn((N, K), device="cuda") / K ** 0.5).bfloat16()
            for _ in range(3): C = st.gemm_bf16(A, W)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record()
            for _ in range(reps): C = st.gemm_bf16(A, W)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            print(f"{os.environ.get('MI_GEMM_TILE','default'):8s} M={M:5d} {name:8s} {us:8.1f} us {2*M*N*K/us/1e6:7.1f} TF  (weights once at 5 TB/s: {N*K*2/5e6:5.1f} us)", flush=True)
    sys.exit(0)
for tile in ("", "mid64", "small"):
    env = dict(os.environ)
    if tile: env["MI_GEMM_TILE"] = tile
    else: env.pop("MI_GEMM_TILE", None)
    out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout
    print(out, end="", flush=True)
