"""One-query search over an IVF65536,PQ64 index (the coarse stage reads all 65536 centroids whatever the corpus): wall time
per call with a sync, back-to-back rate, to set beside the kernel times of a rocprofv3 --kernel-trace --stats run of this
script.  usage: python tools/micro/search_b1_overhead.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import abstracts_search_amd.faiss as faiss
dev = torch.device("cuda", 0)
d, nlist, n = 1024, 65536, 1 << 22
g = torch.Generator(device=dev).manual_seed(1)
cent = torch.randn((nlist, d), generator=g, device=dev)
idx = faiss.index_factory(d, f"IVF{nlist},PQ64", faiss.METRIC_INNER_PRODUCT)
x0 = cent[torch.randint(0, nlist, (1 << 18,), generator=g, device=dev)] + 0.3 * torch.randn((1 << 18, d), generator=g, device=dev)
idx.train(x0)
for c in range(n >> 18):
    idx.add(cent[torch.randint(0, nlist, (1 << 18,), generator=g, device=dev)] + 0.3 * torch.randn((1 << 18, d), generator=g, device=dev))
idx.nprobe = 64
q = x0[:1].contiguous()
for _ in range(20): idx.search(q, 10)
torch.cuda.synchronize()
reps = 300
t0 = time.perf_counter()
for _ in range(reps):
    idx.search(q, 10); torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(reps): idx.search(q, 10)
torch.cuda.synchronize()
t2 = time.perf_counter()
t3 = time.perf_counter()
for _ in range(reps): idx.search(q, 10)
t4 = time.perf_counter()
torch.cuda.synchronize()
print(f"one query, nlist {nlist}: {(t1 - t0) / reps * 1e6:.1f} us per call with a sync; {(t2 - t1) / reps * 1e6:.1f} us back to back; host issue alone {(t4 - t3) / reps * 1e6:.1f} us")
