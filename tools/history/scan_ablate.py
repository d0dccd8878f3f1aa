"""HISTORY (rounds 1-3): kept for the record of how a number in DESIGN_HISTORY.md / profiles/ was produced.  NOT maintained: knobs it
sets may no longer exist (MI_RERANK, MI_REFINE_DEBUG, MI_SCAN_DEBUG ... are silent no-ops now) and paths may have moved."""
"""Ablation / tuning harness for the scan kernel (GPU box): builds cfg2 once and
times scan_kernel (HIP events inside the library) under MI_NSLICE / MI_SCAN_DEBUG."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import abstracts_search_amd.faiss as faiss
import abstracts_search_amd.synth as synth

n, nlist, batch, nprobe = int(os.environ.get("N", 1000000)), int(os.environ.get("NLIST", 4096)), int(os.environ.get("BATCH", 64)), int(os.environ.get("NPROBE", 16))
x = synth.corpus_cuda(n, 1024)
idx = faiss.IndexIVFPQ(1024, nlist, 64, 8, faiss.METRIC_INNER_PRODUCT)
idx.cp.niter = 6
idx.train(x); idx.add(x); idx.nprobe = nprobe
q = synth.queries_cuda(x, batch * 8).view(8, batch, 1024)
D = torch.empty((batch, 10), device="cuda"); I = torch.empty((batch, 10), dtype=torch.int64, device="cuda")

def run(env):
    for k_, v in env.items(): os.environ[k_] = str(v)
    for b in range(8): idx.search_into(q[b], 10, D, I)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range(64): idx.search_into(q[b % 8], 10, D, I)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 64
    p = idx.profile_scan(20)
    for k_ in env: os.environ.pop(k_)
    gbs = p["scan_bytes"] / (p["scan_ms_avg"] * 1e-3) / 1e9
    print(f"{env!s:45s} scan {p['scan_ms_avg']*1e3:8.1f} us  {gbs:8.1f} GB/s   step {dt*1e6:8.1f} us", flush=True)

for ns in (2, 4, 8):
    run({"MI_NSLICE": ns})
for dbg in (1, 2, 3, 8, 9, 11):
    run({"MI_NSLICE": 4, "MI_SCAN_DEBUG": dbg})
run({"MI_NSLICE": 4, "MI_NO_FUSED_MERGE": 1})
