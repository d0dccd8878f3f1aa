"""How batches through IndexRefine overlap by the way they are issued (small index, batch 1024): whole searches round-robin on
1 / 2 / 4 streams, and a stage pipeline (candidates of batch b+1 on stream A beside the re-rank of batch b on stream B) linked by
torch events or by raw hipEvents without the system fence.  usage: python tools/micro/refine_overlap.py [mode]"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import abstracts_search_amd.faiss as faiss
only = sys.argv[1] if len(sys.argv) > 1 else None
dev = torch.device("cuda", 0)
d, nlist, n, batch, k, kf = 1024, 2048, 4 << 20, 1024, 10, 256
g = torch.Generator(device=dev).manual_seed(1)
cent = torch.randn((nlist, d), generator=g, device=dev)
idx = faiss.index_factory(d, f"IVF{nlist},PQ64,Refine(SQ8)", faiss.METRIC_INNER_PRODUCT)
x0 = cent[torch.randint(0, nlist, (1 << 18,), generator=g, device=dev)] + 0.3 * torch.randn((1 << 18, d), generator=g, device=dev)
idx.train(x0)
for c in range(n >> 18):
    x = cent[torch.randint(0, nlist, (1 << 18,), generator=g, device=dev)] + 0.3 * torch.randn((1 << 18, d), generator=g, device=dev)
    idx.add(x)
idx.nprobe = 16
idx.k_factor = kf
base, flat = idx.base_index, idx.refine_index
qs = [x0[i * batch:(i + 1) * batch].contiguous() + 0.05 for i in range(8)]
kb = k * kf
def bufs(m): return [(torch.empty((batch, k), dtype=torch.float32, device=dev), torch.empty((batch, k), dtype=torch.int64, device=dev),
                      torch.empty((batch, kb), dtype=torch.int64, device=dev)) for _ in range(m)]
def timeit(fn, n=40):
    for b in range(8): fn(b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for b in range(n): fn(b)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def rr(ns):
    ss = [torch.cuda.Stream(device=dev) for _ in range(ns)]; B = bufs(ns)
    return lambda b: idx.search_into(qs[b % 8], k, B[b % ns][0], B[b % ns][1], None, B[b % ns][2], int(ss[b % ns].cuda_stream))
def pipe_torch(depth=3):
    A, Bs = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev); B = bufs(depth)
    ea = [torch.cuda.Event() for _ in range(depth)]; eb = [torch.cuda.Event() for _ in range(depth)]
    def step(b):
        j = b % depth
        A.wait_event(eb[j]); base.search_candidates_into(qs[b % 8], kb, B[j][2], None, int(A.cuda_stream)); ea[j].record(A)
        Bs.wait_event(ea[j]); flat.rerank(qs[b % 8], B[j][2], k, B[j][0], B[j][1], int(Bs.cuda_stream)); eb[j].record(Bs)
    return step
def pipe_nodep(depth=3):
    """the same two streams WITHOUT the events (wrong results possible: the re-rank may read a half-written candidate list) --
    only to see whether the dependency mechanism is what serialises"""
    A, Bs = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev); B = bufs(depth)
    def step(b):
        j = b % depth
        base.search_candidates_into(qs[b % 8], kb, B[j][2], None, int(A.cuda_stream))
        flat.rerank(qs[(b + 7) % 8], B[(j + depth - 1) % depth][2], k, B[j][0], B[j][1], int(Bs.cuda_stream))
    return step
hip = ctypes.CDLL("libamdhip64.so")
def pipe_raw(depth=3, flags=0x2 | 0x20000000):     # hipEventDisableTiming | hipEventDisableSystemFence
    A, Bs = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev); B = bufs(depth)
    def mk():
        e = ctypes.c_void_p(); assert hip.hipEventCreateWithFlags(ctypes.byref(e), ctypes.c_uint(flags)) == 0; return e
    ea = [mk() for _ in range(depth)]; eb = [mk() for _ in range(depth)]; used = [False] * depth
    sa, sb = ctypes.c_void_p(A.cuda_stream), ctypes.c_void_p(Bs.cuda_stream)
    def step(b):
        j = b % depth
        if used[j]: hip.hipStreamWaitEvent(sa, eb[j], 0)
        base.search_candidates_into(qs[b % 8], kb, B[j][2], None, int(A.cuda_stream)); hip.hipEventRecord(ea[j], sa)
        hip.hipStreamWaitEvent(sb, ea[j], 0)
        flat.rerank(qs[b % 8], B[j][2], k, B[j][0], B[j][1], int(Bs.cuda_stream)); hip.hipEventRecord(eb[j], sb); used[j] = True
    return step
modes = {"rr1": lambda: rr(1), "rr2": lambda: rr(2), "rr3": lambda: rr(3), "rr4": lambda: rr(4), "rr6": lambda: rr(6), "rr8": lambda: rr(8), "pipe_torch": pipe_torch, "pipe_nodep": pipe_nodep,
         "pipe_raw_nofence": pipe_raw, "pipe_raw_fence": lambda: pipe_raw(3, 0x2)}
s = torch.cuda.current_stream(dev)
c0 = timeit(lambda b: base.search_candidates_into(qs[b % 8], kb, bufs(1)[0][2], None, int(s.cuda_stream)), 10)
print(f"index {n} x {d}, IVF{nlist},PQ64,Refine(SQ8), batch {batch}, nprobe 16, {kb} candidates per query")
for name, mk in modes.items():
    if only and name != only: continue
    print(f"{name}: {timeit(mk()):.3f} ms per batch", flush=True)
