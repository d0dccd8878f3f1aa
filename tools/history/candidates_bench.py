"""HISTORY (rounds 1-3): kept for the record of how a number in DESIGN_HISTORY.md / profiles/ was produced.  NOT maintained: knobs it
sets may no longer exist (MI_RERANK, MI_REFINE_DEBUG, MI_SCAN_DEBUG ... are silent no-ops now) and paths may have moved."""
"""Time of the first stage of a refine search (mi_index_search_candidates) at a whole-index-like shape: ~25 k pairs per
query, the best 5120 kept (CAND_KC); run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import abstracts_search_amd.faiss as faiss
import abstracts_search_amd.synth as synth
N, NLIST, NPROBE, KC = 8 * 1048576, 4096, 12, int(os.environ.get("CAND_KC", 5120))
idx = faiss.IndexIVFPQ(1024, NLIST, 64, 8, faiss.METRIC_INNER_PRODUCT)
idx.cp.niter = 4
x = synth.corpus_cuda(1048576, 1024)
idx.train(x)
for c0 in range(0, N, 1048576):
    idx.add(synth.corpus_cuda(1048576, 1024, row0=c0))
q = synth.queries_cuda(x, 1024, seed=4321)
I = torch.empty((1024, KC), dtype=torch.int64, device="cuda")
D = torch.empty((1024, KC), dtype=torch.float32, device="cuda")
for _ in range(3): idx.search_candidates_into(q, KC, I, NPROBE)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): idx.search_candidates_into(q, KC, I, NPROBE)
torch.cuda.synchronize(); t1 = time.perf_counter()
for _ in range(3): idx.search_into(q, KC, D, I, NPROBE)
torch.cuda.synchronize(); t2 = time.perf_counter()
for _ in range(20): idx.search_into(q, KC, D, I, NPROBE)
torch.cuda.synchronize(); t3 = time.perf_counter()
print(f"kc {KC}: candidates (set) {(t1 - t0) / 20 * 1e3:.3f} ms, sorted search {(t3 - t2) / 20 * 1e3:.3f} ms per 1024 queries", flush=True)
