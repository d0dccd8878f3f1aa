"""HISTORY (rounds 1-3): kept for the record of how a number in DESIGN_HISTORY.md / profiles/ was produced.  NOT maintained: knobs it
sets may no longer exist (MI_RERANK, MI_REFINE_DEBUG, MI_SCAN_DEBUG ... are silent no-ops now) and paths may have moved."""
"""Scan-kernel throughput vs (batch, nprobe) on the cfg2 index (GPU box): where
the kernel leaves the launch-latency regime and what it reaches when
bandwidth/LDS-bound."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import abstracts_search_amd.faiss as faiss
import abstracts_search_amd.synth as synth
n, nlist = int(os.environ.get("N", 1000000)), int(os.environ.get("NLIST", 4096))
x = synth.corpus_cuda(n, 1024)
idx = faiss.IndexIVFPQ(1024, nlist, 64, 8, faiss.METRIC_INNER_PRODUCT)
idx.cp.niter = 6
idx.train(x); idx.add(x)
qall = synth.queries_cuda(x, 4096)
for batch, nprobe in ((64, 16), (64, 64), (64, 256), (256, 64), (1024, 16), (1024, 64), (1024, 256), (4096, 64)):
    q = qall[:batch].contiguous()
    D = torch.empty((batch, 10), device="cuda"); I = torch.empty((batch, 10), dtype=torch.int64, device="cuda")
    idx.nprobe = nprobe
    for _ in range(3): idx.search_into(q, 10, D, I)
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps): idx.search_into(q, 10, D, I)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    p = idx.profile_scan(20)
    gbs = p["scan_bytes"] / (p["scan_ms_avg"] * 1e-3) / 1e9
    print(f"batch {batch:5d} nprobe {nprobe:4d}: scan {p['scan_ms_avg']*1e3:9.1f} us {p['scan_bytes']/1e6:9.1f} MB "
          f"{gbs:8.1f} GB/s ({gbs/80:.1f}% of 8 TB/s)  step {dt*1e6:9.1f} us  {batch/dt:10.0f} QPS", flush=True)
