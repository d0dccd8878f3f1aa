"""HISTORY (rounds 1-3): kept for the record of how a number in DESIGN_HISTORY.md / profiles/ was produced.  NOT maintained: knobs it
sets may no longer exist (MI_RERANK, MI_REFINE_DEBUG, MI_SCAN_DEBUG ... are silent no-ops now) and paths may have moved."""
"""In-kernel phase timing of the scan kernel at cfg2 (GPU box): MI_SCAN_TS=1 makes
mi_index_profile_scan replay the last scan once with s_memtime stamps and print the
per-phase statistics (stderr)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MI_SCAN_TS"] = "1"
import torch
import abstracts_search_amd.faiss as faiss
import abstracts_search_amd.synth as synth

n, nlist, batch, nprobe = int(os.environ.get("N", 1000000)), int(os.environ.get("NLIST", 4096)), int(os.environ.get("BATCH", 64)), int(os.environ.get("NPROBE", 16))
x = synth.corpus_cuda(n, 1024)
idx = faiss.IndexIVFPQ(1024, nlist, 64, 8, faiss.METRIC_INNER_PRODUCT)
idx.cp.niter = 6
idx.train(x); idx.add(x); idx.nprobe = nprobe
q = synth.queries_cuda(x, batch)
D = torch.empty((batch, 10), device="cuda"); I = torch.empty((batch, 10), dtype=torch.int64, device="cuda")
for _ in range(5):
    idx.search_into(q, 10, D, I)
torch.cuda.synchronize()
p = idx.profile_scan(50)
print("scan avg us", p["scan_ms_avg"] * 1e3)
