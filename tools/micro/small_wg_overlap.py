"""Experiment (GPU box): do ONE-WAVE workgroups of a second kernel run beside the PQ scan?  (The coarse stage's 256 x 256 and
128 x 128 GEMM workgroups do not: profiles/r05_coarse_overlap_priority.txt.)  IVF65536,PQ64 over N random codes, batch 1024,
nprobe 64: the scan alone (search_preassigned on stream A), an MFMA-only spin kernel of one-wave workgroups alone (stream B, sized
to ~0.15 ms: the coarse GEMM's time), and both at once."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import abstracts_search_amd.faiss as faiss
spin = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libspin.so"))
N = int(os.environ.get("CP_N", 40_000_000)); nlist, M, d, nq, nprobe, k = 65536, 64, 1024, 1024, 64, 10
rng = np.random.default_rng(0)
cent = rng.standard_normal((nlist, d), dtype=np.float32); cent /= np.linalg.norm(cent, axis=1, keepdims=True)
idx = faiss.IndexIVFPQ(d, nlist, M, 8, faiss.METRIC_INNER_PRODUCT)
idx.set_centroids(cent); idx.set_codebook((0.05 * rng.standard_normal((M, 256, d // M))).astype(np.float32))
B = 4_000_000
for b0 in range(0, N, B):
    m = min(B, N - b0)
    idx.add_codes(rng.integers(0, nlist, m, dtype=np.int32), rng.integers(0, 256, (m, M), dtype=np.uint8), np.arange(b0, b0 + m, dtype=np.int64))
idx.nprobe = nprobe
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.nn.functional.normalize(torch.randn((nq, d), generator=g, device="cuda"), dim=1)
cI, cD = idx.coarse_slice(q, nprobe, 0, nlist)
SA, SB = torch.cuda.Stream(), torch.cuda.Stream(priority=int(os.environ.get("SPIN_PRIO", "0")))
sink = torch.zeros(1 << 20, device="cuda")

def scan(n):
    with torch.cuda.stream(SA):
        for _ in range(n): idx.search_preassigned(q, k, cI, cD)

def spin_(n, nwg, iters, nv):
    for _ in range(n): spin.spin_launch(nwg, iters, nv, ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(SB.cuda_stream))

def t(f, n=20):
    f(3); torch.cuda.synchronize(); t0 = time.perf_counter(); f(n); torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

print(f"scan alone (tables + LUT + scan, N {N}): {t(scan):.3f} ms")
for nv in (4, 24):
    for nwg, iters in ((16384, 512), (65536, 128)):
        a = t(lambda n: spin_(n, nwg, iters, nv))
        both = t(lambda n: (scan(n), spin_(n, nwg, iters, nv)))
        print(f"  spin nv {nv:2d} ({'~40' if nv == 4 else '~128'} VGPRs) {nwg} x 1 wave x {iters} MFMA rounds: alone {a:.3f} ms; beside the scan {both:.3f} ms per (scan + spin)")
