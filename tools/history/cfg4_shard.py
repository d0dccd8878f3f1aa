"""HISTORY (rounds 1-3): kept for the record of how a number in DESIGN_HISTORY.md / profiles/ was produced.  NOT maintained: knobs it
sets may no longer exist (MI_RERANK, MI_REFINE_DEBUG, MI_SCAN_DEBUG ... are silent no-ops now) and paths may have moved."""
"""One shard of BASELINE.json configs[3] on one MI355X: 207M x 1024 IVF65536,PQ64
sharded by vector over 8 GPUs = 25.9M vectors per GPU, every GPU searching all
queries (batch 1024) over its shard.  Builds the shard from the synthetic
generator in chunks (never holding more than one chunk of raw vectors), then
times search and the scan kernel.  GPU box; ~2-3 minutes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import abstracts_search_amd.faiss as faiss
import abstracts_search_amd.synth as synth

N_SHARD = int(os.environ.get("NSHARD_VEC", 207_000_000 // 8)); NLIST = int(os.environ.get("NLIST", 65536))
CH = 65536 * 16
t0 = time.time()
idx = faiss.IndexIVFPQ(1024, NLIST, 64, 8, faiss.METRIC_INNER_PRODUCT)
xs = synth.corpus_cuda(4 * 1024 * 1024, 1024)             # training sample (64 points per centroid)
idx.cp.niter = int(os.environ.get("NITER", 4))
idx.train(xs)
print(f"train {time.time()-t0:.1f}s", flush=True)
t1 = time.time()
done = 0
while done < N_SHARD:
    m = min(CH, N_SHARD - done)
    x = synth.corpus_cuda(m, 1024, row0=done)
    # shard r of 8 holds global rows 8*i + r: ids carry the global numbering
    idx.add_with_ids(x, torch.arange(done, done + m, device="cuda") * 8)
    done += m
print(f"add {N_SHARD} vectors {time.time()-t1:.1f}s ({N_SHARD/(time.time()-t1)/1e6:.2f} M vec/s)", flush=True)
q = synth.queries_cuda(xs, 4096)
for batch, nprobe in ((1, 64), (16, 64), (256, 64), (1024, 16), (1024, 64), (1024, 256)):
    qq = q[:batch].contiguous()
    D = torch.empty((batch, 10), device="cuda"); I = torch.empty((batch, 10), dtype=torch.int64, device="cuda")
    idx.nprobe = nprobe
    t2 = time.time()
    for _ in range(2): idx.search_into(qq, 10, D, I)
    torch.cuda.synchronize()
    if batch == 1: print(f"first search (device image build) {time.time()-t2:.1f}s", flush=True)
    reps = 10
    ta = time.perf_counter()
    for _ in range(reps): idx.search_into(qq, 10, D, I)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - ta) / reps
    p = idx.profile_scan(5)
    gbs = p["scan_bytes"] / (p["scan_ms_avg"] * 1e-3) / 1e9
    print(f"batch {batch:5d} nprobe {nprobe:4d}: step {dt*1e3:8.3f} ms {batch/dt:10.0f} QPS/shard-GPU | scan {p['scan_ms_avg']*1e3:9.1f} us "
          f"{p['scan_bytes']/1e6:9.1f} MB {gbs:7.1f} GB/s ({gbs/80:.1f}% of 8 TB/s)", flush=True)
print("sorted:", bool((D[:, :-1] >= D[:, 1:]).all()), "ids%8==0:", bool(((I % 8 == 0) | (I < 0)).all()), "mem GB", torch.cuda.max_memory_allocated() / 1e9)
