"""Does the PQ scan lose time when some CUs are taken away from it (hipExtStreamCreateWithCUMask)?  If not, the coarse stage of the
next batch could run on those CUs beside it.  8 M x 1024, IVF8192,PQ64, batch 1024, nprobe 64 (HBM-bound scan like cfg4's).
usage: python tools/micro/cu_mask_scan.py"""
import os, sys, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import abstracts_search_amd.faiss as faiss
dev = torch.device("cuda", 0)
d, nlist, n, batch, k = 1024, 8192, 8 << 20, 1024, 10
g = torch.Generator(device=dev).manual_seed(1)
cent = torch.randn((nlist, d), generator=g, device=dev)
idx = faiss.index_factory(d, f"IVF{nlist},PQ64", faiss.METRIC_INNER_PRODUCT)
x0 = cent[torch.randint(0, nlist, (1 << 18,), generator=g, device=dev)] + 0.3 * torch.randn((1 << 18, d), generator=g, device=dev)
idx.train(x0)
for c in range(n >> 18):
    idx.add(cent[torch.randint(0, nlist, (1 << 18,), generator=g, device=dev)] + 0.3 * torch.randn((1 << 18, d), generator=g, device=dev))
idx.nprobe = 64
q = x0[:batch].contiguous() + 0.05
D = torch.empty((batch, k), dtype=torch.float32, device=dev); I = torch.empty((batch, k), dtype=torch.int64, device=dev)
hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xffffffff for i in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(8), words)
    assert rc == 0, rc
    return s.value
ALL = (1 << 256) - 1
cases = {"all 256 CUs (plain stream)": None, "mask: all 256": ALL, "mask: bits 0..239": (1 << 240) - 1, "mask: bits 0..223": (1 << 224) - 1,
         "mask: bits 0..191": (1 << 192) - 1, "mask: 30 of every 32 bits": int("".join(["0011" + "1" * 28] * 8), 2),
         "mask: 28 of every 32 bits": int("".join(["0000" + "1" * 28] * 8), 2)}
for name, bits in cases.items():
    st = int(torch.cuda.current_stream(dev).cuda_stream) if bits is None else masked_stream(bits)
    for _ in range(3): idx.search_into(q, k, D, I, None, st)
    torch.cuda.synchronize()
    pr = idx.profile_scan(10, st)
    t0 = time.perf_counter()
    for _ in range(20): idx.search_into(q, k, D, I, None, st)
    hip.hipStreamSynchronize(ctypes.c_void_p(st)); torch.cuda.synchronize()
    whole = (time.perf_counter() - t0) / 20 * 1e3
    print(f"{name:32s} scan {pr['scan_ms_avg']:.4f} ms ({pr['scan_bytes'] / pr['scan_ms_avg'] / 1e6:.0f} GB/s)   whole search {whole:.3f} ms", flush=True)
