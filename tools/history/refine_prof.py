"""HISTORY (rounds 1-3): kept for the record of how a number in DESIGN_HISTORY.md / profiles/ was produced.  NOT maintained: knobs it
sets may no longer exist (MI_RERANK, MI_REFINE_DEBUG, MI_SCAN_DEBUG ... are silent no-ops now) and paths may have moved."""
"""The recall >= 0.95 operating point on one cfg4 sub-shard (25.9 M vectors, IVF65536,PQ64 + refine
store), timed per configuration; run under tools/prof_cmd.sh for the per-kernel split.
usage: python tools/refine_prof.py [f16|f32] [nprobe] [k_factor] [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import abstracts_search_amd.faiss as faiss
import abstracts_search_amd.synth as synth

store = sys.argv[1] if len(sys.argv) > 1 else "f16"
nprobe = int(sys.argv[2]) if len(sys.argv) > 2 else 8
kf = int(sys.argv[3]) if len(sys.argv) > 3 else 80
N = int(sys.argv[4]) if len(sys.argv) > 4 else 25_875_000
CH = 1 << 20
base = faiss.IndexIVFPQ(1024, 65536, 64, 8, faiss.METRIC_INNER_PRODUCT)
base.cp.niter = 4
base.train(synth.corpus_cuda(4 * CH, 1024))
flat = faiss.IndexScalarQuantizer(1024) if store == "f16" else faiss.IndexFlatIP(1024)
flat.reserve(N)
idx = faiss.IndexRefine(base, flat)
t0 = time.time()
for c0 in range(0, N, CH):
    idx.add(synth.corpus_cuda(min(CH, N - c0), 1024, row0=c0))
torch.cuda.synchronize()
print(f"add {time.time()-t0:.0f}s", flush=True)
xq = synth.corpus_cuda(CH, 1024, row0=(N // 2) // CH * CH)
q = synth.queries_cuda(xq, 8 * 1024).view(8, 1024, 1024)
base.nprobe, idx.k_factor = nprobe, kf
k = 10
D = torch.empty((1024, k), device="cuda"); I = torch.empty((1024, k), dtype=torch.int64, device="cuda")
cD = torch.empty((1024, k * kf), device="cuda"); cI = torch.empty((1024, k * kf), dtype=torch.int64, device="cuda")
for b in range(5):
    idx.search_into(q[b % 8], k, D, I, cD, cI)
torch.cuda.synchronize()
t = time.perf_counter()
steps = 30
for b in range(steps):
    idx.search_into(q[b % 8], k, D, I, cD, cI)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / steps
print(f"store {store} nprobe {nprobe} k_factor {kf}: {dt*1e3:.3f} ms per 1024-query step, {1024/dt:.0f} QPS")
