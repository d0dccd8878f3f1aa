"""IndexFlatIP.rerank (gather GEMM + candidate merge) on random candidates: 1024 queries x
640 rows of an n-row store.  usage: python tools/gather_bench.py [n_rows]   (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import abstracts_search_amd.faiss as faiss
import abstracts_search_amd.synth as synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
flat = faiss.IndexFlatIP(1024)
flat.reserve(n)
for c0 in range(0, n, 1048576):
    flat.add(synth.corpus_cuda(min(1048576, n - c0), 1024, row0=c0))
q = synth.queries_cuda(synth.corpus_cuda(65536, 1024), 1024)
g = torch.Generator(device="cuda").manual_seed(1)
cand = torch.randint(0, n, (1024, 640), generator=g, device="cuda")
D = torch.empty((1024, 10), device="cuda"); I = torch.empty((1024, 10), dtype=torch.int64, device="cuda")
for _ in range(3): flat.rerank(q, cand, 10, D, I)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): flat.rerank(q, cand, 10, D, I)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"rerank 1024 x 640 of {n} rows: {dt * 1e3:.3f} ms  ({1024 * 640 * 4096 / dt / 1e12:.2f} TB/s)")
