"""HISTORY (rounds 1-3): kept for the record of how a number in DESIGN_HISTORY.md / profiles/ was produced.  NOT maintained: knobs it
sets may no longer exist (MI_RERANK, MI_REFINE_DEBUG, MI_SCAN_DEBUG ... are silent no-ops now) and paths may have moved."""
"""Search steps at a large nprobe on the cfg2 index (per-kernel profile target: run under
tools/prof_cmd.sh).  usage: python tools/nprobe_prof.py [nprobe[,nprobe...]] [k] [batch] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import abstracts_search_amd.faiss as faiss
import abstracts_search_amd.synth as synth
nprobes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1024").split(",")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
n, nlist = int(os.environ.get("N", 1000000)), int(os.environ.get("NLIST", 4096))
x = synth.corpus_cuda(n, 1024)
idx = faiss.IndexIVFPQ(1024, nlist, 64, 8, faiss.METRIC_INNER_PRODUCT)
idx.cp.niter = 6
idx.train(x); idx.add(x)
q = synth.queries_cuda(x, batch)
D = torch.empty((batch, k), device="cuda"); I = torch.empty((batch, k), dtype=torch.int64, device="cuda")
for nprobe in nprobes:
    idx.nprobe = nprobe
    for _ in range(3): idx.search_into(q, k, D, I)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): idx.search_into(q, k, D, I)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"batch {batch} nprobe {nprobe} k {k}: step {dt * 1e6:.1f} us  {batch / dt:.0f} QPS", flush=True)
