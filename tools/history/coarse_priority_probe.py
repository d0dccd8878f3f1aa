"""HISTORY (rounds 1-3): kept for the record of how a number in DESIGN_HISTORY.md / profiles/ was produced.  NOT maintained: knobs it
sets may no longer exist (MI_RERANK, MI_REFINE_DEBUG, MI_SCAN_DEBUG ... are silent no-ops now) and paths may have moved."""
"""Does a high-priority stream for the coarse stage hide it behind the previous batch's scan?  207 M index (or PROBE_N),
batch 1024, nprobe 64: (a) whole searches round-robin on 2 streams (bench.py's loop); (b) coarse quantiser of batch i+1
on a high-priority stream, LUT + scan (search_preassigned) on a normal one, chained by events."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import abstracts_search_amd.faiss as faiss
import abstracts_search_amd.synth as synth
N, NLIST, B, NPROBE, K = int(os.environ.get("PROBE_N", 207_000_000)), 65536, 1024, 64, 10
CH = 1 << 20
idx = faiss.IndexIVFPQ(1024, NLIST, 64, 8, faiss.METRIC_INNER_PRODUCT)
idx.cp.niter = 4
idx.train(synth.corpus_cuda(4 * CH, 1024))
idx.reserve(N + 1)
for c0 in range(0, N, CH):
    idx.add(synth.corpus_cuda(min(CH, N - c0), 1024, row0=c0))
idx.nprobe = NPROBE
xq = synth.corpus_cuda(CH, 1024, row0=(N // 2) // CH * CH)
qs = synth.queries_cuda(xq, 8 * B, seed=4321).view(8, B, 1024)
dev = torch.device("cuda")
D = [torch.empty((B, K), dtype=torch.float32, device=dev) for _ in range(2)]
I = [torch.empty((B, K), dtype=torch.int64, device=dev) for _ in range(2)]
idx.search(qs[0], K); torch.cuda.synchronize()

def timeit(step, n=60):
    for i in range(10): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): step(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

s2 = [torch.cuda.Stream() for _ in range(2)]
def step_a(i):
    idx.search_into(qs[i % 8], K, D[i % 2], I[i % 2], None, s2[i % 2].cuda_stream)
print(f"(a) 2 streams, whole searches: {timeit(step_a):.4f} ms per step", flush=True)

hi = torch.cuda.Stream(priority=-1)
lo = [torch.cuda.Stream(priority=0) for _ in range(2)]
ev = [torch.cuda.Event() for _ in range(4)]
def step_b(i):
    with torch.cuda.stream(hi):
        cI, cD = idx.coarse_slice(qs[i % 8], NPROBE, 0, NLIST)
        e = ev[i % 4]; e.record(hi)
    st = lo[i % 2]
    st.wait_event(e)
    with torch.cuda.stream(st):
        cI.record_stream(st); cD.record_stream(st)
        idx.search_preassigned(qs[i % 8], K, cI, cD)
print(f"(b) coarse stage on a high-priority stream + search_preassigned: {timeit(step_b):.4f} ms per step", flush=True)
hi0 = torch.cuda.Stream(priority=0)
def step_c(i):
    with torch.cuda.stream(hi0):
        cI, cD = idx.coarse_slice(qs[i % 8], NPROBE, 0, NLIST)
        e = ev[i % 4]; e.record(hi0)
    st = lo[i % 2]
    st.wait_event(e)
    with torch.cuda.stream(st):
        cI.record_stream(st); cD.record_stream(st)
        idx.search_preassigned(qs[i % 8], K, cI, cD)
print(f"(c) the same split, coarse stream at normal priority: {timeit(step_c):.4f} ms per step", flush=True)
