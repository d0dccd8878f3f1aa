"""faiss.search(numpy) vs search on resident tensors: what the host round trip adds to a 1024-query step (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import abstracts_search_amd.faiss as faiss
import abstracts_search_amd.synth as synth
N, d, nlist = 8 << 20, 1024, 65536
idx = faiss.IndexIVFPQ(d, nlist, 64, 8, faiss.METRIC_INNER_PRODUCT)
idx.cp.niter = idx.pq.cp.niter = 2
x = synth.corpus_cuda(4 << 20, d)
idx.train(x)
for c0 in range(0, N, 1 << 20):
    idx.add(synth.corpus_cuda(1 << 20, d, row0=c0))
idx.nprobe = 64
q = synth.queries_cuda(x, 1024)
qh = q.cpu().numpy()
D = torch.empty((1024, 10), device="cuda"); I = torch.empty((1024, 10), dtype=torch.int64, device="cuda")
for name, fn in (("resident tensors (search_into)", lambda: idx.search_into(q, 10, D, I)), ("host numpy (faiss.search(numpy))", lambda: idx.search(qh, 10))):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t) / 50 * 1e3:.3f} ms per 1024-query step", flush=True)
