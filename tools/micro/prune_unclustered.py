"""The exact list pruning on data it cannot prune: i.i.d. Gaussian unit vectors (no clusters: a query's neighbours are anywhere, the
coarse scores of its probed lists are all alike).  32 M x 1024, IVF8192,PQ64 (~61 code groups per list, cfg4's are 49), batch 1024,
nprobe 64: the step with MI_SCAN_PRUNE=0, then the default -- whose first calls take the one-workgroup-per-query shape and whose
later calls, the index having seen that nothing prunes, go back to the balanced slices (csrc/ivfpq.hip: prunes_well)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
faiss = importlib.import_module("abstracts-search_amd.faiss")
d, N, nlist, nq, nprobe, k = 1024, int(os.environ.get("PU_N", 32 << 20)), 8192, 1024, 64, 10
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(7)
def rows(n):
    x = torch.randn(n, d, device=dev, generator=g)
    return x / x.norm(dim=1, keepdim=True)
index = faiss.index_factory(d, f"IVF{nlist},PQ64", faiss.METRIC_INNER_PRODUCT)
index.train(rows(64 * nlist))
for i in range(0, N, 1 << 20):
    index.add(rows(min(1 << 20, N - i)))
index.nprobe = nprobe
q = rows(nq)
D = torch.empty(nq, k, device=dev); I = torch.empty(nq, k, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
def timed(label, reps=20):
    for _ in range(3):
        index.search_into(q, k, D, I, None, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        index.search_into(q, k, D, I, None, st)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"{label}: {ms:.3f} ms per step; checksum {int(I.sum().item())}")
    return ms
os.environ["MI_SCAN_PRUNE"] = "0"; faiss.reload_env()
timed("exhaustive (MI_SCAN_PRUNE=0)")
del os.environ["MI_SCAN_PRUNE"]; faiss.reload_env()
index.prune_stats(reset=True)
timed("default, first calls (one workgroup per query until the counters say otherwise)", reps=4)
print("   ", index.prune_stats(reset=True))
timed("default, later calls")
print("   ", index.prune_stats())
os.environ["MI_SCAN_PRUNE_MODE"] = "1"; os.environ["MI_SCAN_PRUNE_MIN_GROUPS"] = "0"; faiss.reload_env()
timed("two launches forced (MI_SCAN_PRUNE_MODE=1)")
