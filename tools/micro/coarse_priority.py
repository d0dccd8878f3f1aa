"""Experiment (GPU box): can the coarse stage of batch i + 1 run beside the scan of batch i when it is issued on a HIGH-PRIORITY
stream?  An IVF65536,PQ64 index of N random codes (add_codes: no encoding), batch 1024, nprobe 64:
  mode plain   : mi_index_search on 2 streams round-robin (what bench.py times)
  mode split   : coarse_slice on stream H -> event -> search_preassigned on stream S[i % 2], H of the given priority
MI_F16_GEMM_128=1 puts the approximate coarse scores on 128 x 128 tiles (half a CU: co-resident with one scan workgroup)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import abstracts_search_amd.faiss as faiss
N = int(os.environ.get("CP_N", 40_000_000)); nlist, M, d, nq, nprobe, k = 65536, 64, 1024, 1024, 64, 10
rng = np.random.default_rng(0)
cent = rng.standard_normal((nlist, d), dtype=np.float32); cent /= np.linalg.norm(cent, axis=1, keepdims=True)
cb = (0.05 * rng.standard_normal((M, 256, d // M))).astype(np.float32)
idx = faiss.IndexIVFPQ(d, nlist, M, 8, faiss.METRIC_INNER_PRODUCT)
idx.set_centroids(cent); idx.set_codebook(cb)
B = 4_000_000
for b0 in range(0, N, B):
    m = min(B, N - b0)
    idx.add_codes(rng.integers(0, nlist, m, dtype=np.int32), rng.integers(0, 256, (m, M), dtype=np.uint8), np.arange(b0, b0 + m, dtype=np.int64))
idx.nprobe = nprobe
g = torch.Generator(device="cuda").manual_seed(1)
qs = [torch.nn.functional.normalize(torch.randn((nq, d), generator=g, device="cuda"), dim=1) for _ in range(8)]
S = [torch.cuda.Stream() for _ in range(2)]
D = [torch.empty((nq, k), device="cuda") for _ in range(2)]; I = [torch.empty((nq, k), dtype=torch.int64, device="cuda") for _ in range(2)]

def plain(steps):
    for i in range(steps):
        idx.search_into(qs[i % 8], k, D[i % 2], I[i % 2], stream=int(S[i % 2].cuda_stream))

def split(H, steps):
    evs = []
    for i in range(steps):
        with torch.cuda.stream(H):
            cI, cD = idx.coarse_slice(qs[i % 8], nprobe, 0, nlist)
            ev = torch.cuda.Event(); ev.record(H)
        s = S[i % 2]
        s.wait_event(ev)
        with torch.cuda.stream(s):
            idx.search_preassigned(qs[i % 8], k, cI, cD)

def timeit(f, steps=40):
    f(6); torch.cuda.synchronize()
    t = []
    for _ in range(3):
        t0 = time.perf_counter(); f(steps); torch.cuda.synchronize(); t.append((time.perf_counter() - t0) / steps * 1e3)
    return sorted(t)[1]

print(f"N {N} gemm128={os.environ.get('MI_F16_GEMM_128')}: plain 2 streams {timeit(plain):.3f} ms/step", flush=True)
for pr, name in ((0, "normal"), (-1, "high")):
    H = torch.cuda.Stream(priority=pr)
    print(f"   split, coarse on a {name}-priority stream: {timeit(lambda n: split(H, n)):.3f} ms/step", flush=True)
with torch.cuda.stream(S[0]):
    t0 = time.perf_counter()
    for i in range(20): idx.coarse_slice(qs[i % 8], nprobe, 0, nlist)
    torch.cuda.synchronize(); print(f"   coarse alone {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms", flush=True)
