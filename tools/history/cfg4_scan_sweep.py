"""HISTORY (rounds 1-3): kept for the record of how a number in DESIGN_HISTORY.md / profiles/ was produced.  NOT maintained: knobs it
sets may no longer exist (MI_RERANK, MI_REFINE_DEBUG, MI_SCAN_DEBUG ... are silent no-ops now) and paths may have moved."""
"""Builds the full cfg4 index (207 M x 1024, IVF65536,PQ64) once on one MI355X and times the
batch-1024 search + the scan kernel under launch variants (MI_NSLICE, MI_SCAN_NW, nprobe).
GPU box; ~2.5 min.  usage: python tools/cfg4_scan_sweep.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import abstracts_search_amd.faiss as faiss
import abstracts_search_amd.synth as synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 207_000_000
CH = 1 << 20
idx = faiss.IndexIVFPQ(1024, 65536, 64, 8, faiss.METRIC_INNER_PRODUCT)
idx.cp.niter = 4
t0 = time.time()
idx.train(synth.corpus_cuda(4 * CH, 1024))
idx.reserve(N)
for c0 in range(0, N, CH):
    idx.add(synth.corpus_cuda(min(CH, N - c0), 1024, row0=c0))
torch.cuda.synchronize()
print(f"build {time.time()-t0:.0f}s", flush=True)
xq = synth.corpus_cuda(CH, 1024, row0=(N // 2) // CH * CH)
q = synth.queries_cuda(xq, 8 * 1024).view(8, 1024, 1024)
del xq
D = [torch.empty((1024, 10), device="cuda") for _ in range(2)]
I = [torch.empty((1024, 10), dtype=torch.int64, device="cuda") for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
sp = [int(s.cuda_stream) for s in streams]

def run(tag, nprobe=64, S=2, steps=30):
    idx.nprobe = nprobe
    for b in range(4):
        idx.search_into(q[b % 8], 10, D[b % S], I[b % S], None, sp[b % S])
    torch.cuda.synchronize()
    t = time.perf_counter()
    for b in range(steps):
        idx.search_into(q[b % 8], 10, D[b % S], I[b % S], None, sp[b % S])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    p = idx.profile_scan(5, sp[0])
    gbs = p["scan_bytes"] / (p["scan_ms_avg"] * 1e-3) / 1e9
    print(f"{tag:28s} nprobe {nprobe:4d} streams {S}: step {dt*1e3:7.3f} ms {1024/dt:9.0f} QPS | scan {p['scan_ms_avg']:7.3f} ms "
          f"{gbs:7.1f} GB/s ({gbs/80:.1f}%)", flush=True)

for npb in (8, 16, 32, 64, 128):
    run("default (sliced)", nprobe=npb)
    os.environ["MI_NSLICE"] = "1"
    run("MI_NSLICE=1", nprobe=npb)
    os.environ.pop("MI_NSLICE")
run("default (sliced)", S=1)
for ns in (2, 4, 8):
    os.environ["MI_NSLICE"] = str(ns)
    run(f"MI_NSLICE={ns}", nprobe=16)
os.environ.pop("MI_NSLICE")
