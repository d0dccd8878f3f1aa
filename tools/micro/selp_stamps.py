"""Phase stamps of select_pairs_kernel<32,true> at the recall-0.95 shape (profiling build: hipcc -DMI_SELP_TS).
1024 queries x 8 probes of ~3.2 k codes -> 5120 of ~25.9 k pairs per query.  usage: python tools/micro/selp_stamps.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import abstracts_search_amd.faiss as faiss
dev = torch.device("cuda", 0)
d, nlist, n, batch, kb = 1024, 2048, 6553600, 1024, 5120
g = torch.Generator(device=dev).manual_seed(1)
cent = torch.randn((nlist, d), generator=g, device=dev)
idx = faiss.index_factory(d, f"IVF{nlist},PQ64", faiss.METRIC_INNER_PRODUCT)
x0 = cent[torch.randint(0, nlist, (1 << 18,), generator=g, device=dev)] + 0.3 * torch.randn((1 << 18, d), generator=g, device=dev)
idx.train(x0)
for c in range(n >> 18):
    idx.add(cent[torch.randint(0, nlist, (1 << 18,), generator=g, device=dev)] + 0.3 * torch.randn((1 << 18, d), generator=g, device=dev))
idx.nprobe = int(os.environ.get('SELP_NPROBE', 8))
q = x0[:batch].contiguous() + 0.05
cI = torch.empty((batch, kb), dtype=torch.int64, device=dev)
for _ in range(3): idx.search_candidates_into(q, kb, cI)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(10): idx.search_candidates_into(q, kb, cI)
ev[1].record(); torch.cuda.synchronize()
print(f"nprobe {idx.nprobe}: {ev[0].elapsed_time(ev[1]) / 10:.3f} ms per candidate search of {batch} queries (coarse + LUT + all-scores scan + set selection)")
if 'ts' not in os.environ.get('MI_IVFPQ_LIB', ''): sys.exit(0)
lib = faiss._Lib.get()
out = (ctypes.c_ulonglong * (8 * 4096))()
assert lib.mi_debug_selp_stamps(out) == 0
a = np.frombuffer(out, dtype=np.uint64).reshape(4096, 8)[:batch].astype(np.float64)
names = ["pass 1 (group maxima)", "descent over the maxima", "compaction pass", "survivor keys/ids + second descent", "output"]
print("survivors per query: mean %.0f max %.0f" % (a[:, 6].mean(), a[:, 6].max()))
for i, nm in enumerate(names):
    dlt = a[:, i + 1] - a[:, i]
    print(f"  {nm:38s} mean {dlt.mean() / 1000:8.2f} us   max {dlt.max() / 1000:8.2f} us")
print(f"  whole workgroup                        mean {(a[:, 5] - a[:, 0]).mean() / 1000:8.2f} us; first start -> last end {(a[:, 5].max() - a[:, 0].min()) / 1000:.1f} us   (one s_memtime tick ~ 1 ns on this part)")
