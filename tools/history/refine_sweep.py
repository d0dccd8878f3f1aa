"""HISTORY (rounds 1-3): kept for the record of how a number in DESIGN_HISTORY.md / profiles/ was produced.  NOT maintained: knobs it
sets may no longer exist (MI_RERANK, MI_REFINE_DEBUG, MI_SCAN_DEBUG ... are silent no-ops now) and paths may have moved."""
"""Recall@10 and step time of IVF4096,PQ64 + RFlat (IndexRefineFlat) on the bench corpus:
k_factor x nprobe sweep against exact search.  GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import abstracts_search_amd.faiss as faiss
import abstracts_search_amd.synth as synth

n, nlist, batch, k = 1_000_000, 4096, 64, 10
x = synth.corpus_cuda(n, 1024)
base = faiss.IndexIVFPQ(1024, nlist, 64, 8, faiss.METRIC_INNER_PRODUCT)
base.cp.niter = 10
base.train(x)
idx = faiss.IndexRefineFlat(base)
idx.add(x)
q = synth.queries_cuda(x, batch * 16, seed=4321).view(16, batch, 1024)
exact = [idx.refine_index.search(q[b], k)[1].cpu().numpy() for b in range(16)]
D = torch.empty((batch, k), device="cuda"); I = torch.empty((batch, k), dtype=torch.int64, device="cuda")
for nprobe in (8, 16, 32, 64):
    idx.nprobe = nprobe
    for kf in (1, 3, 6, 10, 20):
        kb = k * kf
        cD = torch.empty((batch, kb), device="cuda"); cI = torch.empty((batch, kb), dtype=torch.int64, device="cuda")
        hits = 0
        for b in range(16):
            if kf == 1:
                base.search_into(q[b], k, D, I)
            else:
                idx.search_into(q[b], k, D, I, cD, cI)
            got = I.cpu().numpy()
            hits += sum(len(set(a.tolist()) & set(e.tolist())) for a, e in zip(got, exact[b]))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(200):
            if kf == 1:
                base.search_into(q[r % 16], k, D, I)
            else:
                idx.search_into(q[r % 16], k, D, I, cD, cI)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 200
        print(f"nprobe {nprobe:3d} k_factor {kf:3d} (k_base {kb:3d}): recall@10 {hits / (16 * batch * k):.4f}  "
              f"step {dt * 1e6:7.1f} us  {batch / dt:10.0f} QPS (1 stream)", flush=True)
