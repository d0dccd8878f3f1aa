"""HISTORY (rounds 1-3): kept for the record of how a number in DESIGN_HISTORY.md / profiles/ was produced.  NOT maintained: knobs it
sets may no longer exist (MI_RERANK, MI_REFINE_DEBUG, MI_SCAN_DEBUG ... are silent no-ops now) and paths may have moved."""
"""Kernel-time sweep of the exact-f32 score GEMM (run under tools/prof_cmd.sh).

IndexFlatIP.search = ip_gemm_kernel + select_kernel; sweeping d at fixed
(nq, nb) separates the per-K-chunk cost from the fixed launch cost.
usage: python tools/ipgemm_sweep.py [nq] [nb]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import abstracts_search_amd.faiss as faiss
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
g = torch.Generator(device="cuda").manual_seed(1)
for d in (128, 256, 512, 1024, 2048, 4096):
    x = torch.randn(nb, d, device="cuda", generator=g)
    q = torch.randn(nq, d, device="cuda", generator=g)
    ix = faiss.IndexFlatIP(d)
    ix.add(x)
    for _ in range(5):
        ix.search(q, 16)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(50):
        ix.search(q, 16)
    ev1.record()
    torch.cuda.synchronize()
    print(f"d={d} nq={nq} nb={nb}: {ev0.elapsed_time(ev1) / 50 * 1e3:.1f} us per search (gemm+select)")
