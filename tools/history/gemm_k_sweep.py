"""HISTORY (rounds 1-3): kept for the record of how a number in DESIGN_HISTORY.md / profiles/ was produced.  NOT maintained: knobs it
sets may no longer exist (MI_RERANK, MI_REFINE_DEBUG, MI_SCAN_DEBUG ... are silent no-ops now) and paths may have moved."""
"""bf16 ring GEMM time vs K at fixed M, N (big 256x256 tiles): slope = per-K-step cost, intercept =
per-tile fixed cost (launch, pipeline fill, epilogue).  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import abstracts_search_amd.sentence_transformers as st
M, N = int(os.environ.get("M", 29696)), int(os.environ.get("N", 2048))
tiles = ((M + 255) // 256) * ((N + 255) // 256)
pts = []
for K in [int(k) for k in os.environ.get("KS", "512,1024,1536,3072,6144,8960").split(",")]:
    A = torch.randn((M, K), device="cuda").bfloat16(); W = (torch.randn((N, K), device="cuda") / K ** 0.5).bfloat16()
    if os.environ.get("ZERO") == "1":           # no data toggling: the schedule alone, at the unthrottled clock
        A.zero_(); W.zero_()
    if os.environ.get("BLASLT") == "1":         # the vendor library on the same shapes (comparison point only)
        C_ = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
        st.gemm_bf16 = lambda a, w: torch.matmul(a, w.T, out=C_)
    for _ in range(3): st.gemm_bf16(A, W)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): st.gemm_bf16(A, W)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    per_tile = us / (tiles / 256)
    pts.append((K // 32, per_tile))
    print(f"K {K:5d}: {us:8.1f} us  {2*M*N*K/us/1e6:7.1f} TF   {per_tile:7.2f} us per tile-round ({tiles} tiles = {tiles/256:.2f} rounds)")
(x0, y0), (x1, y1) = pts[1], pts[-2]
c = (y1 - y0) / (x1 - x0)
print(f"slope {c:.3f} us per K step of 32, intercept {y0 - c * x0:.1f} us per tile")
