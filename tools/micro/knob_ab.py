"""A / B of one MI_* knob of libmi_ivfpq.so on the search step, alternating in ONE process on one box (boxes of the pool differ by
2-3 %: a change of a per cent only shows this way).  IVF65536,PQ64 over AB_N (default 40 M) synthetic vectors, batch 1024,
nprobe AB_NPROBE (64), k 10; the results of both settings compared bit for bit first.
    python tools/micro/knob_ab.py MI_NO_FUSED_MERGE"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import abstracts_search_amd.faiss as faiss
import abstracts_search_amd.synth as synth
KNOB = sys.argv[1]
N, NLIST, B, NPROBE, K = int(os.environ.get("AB_N", 40_000_000)), 65536, 1024, int(os.environ.get("AB_NPROBE", 64)), 10
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
S = [torch.cuda.Stream() for _ in range(2)]


def env(v):
    if v is None:
        os.environ.pop(KNOB, None)
    else:
        os.environ[KNOB] = v
    faiss.reload_env()


def step(ns):
    def f(n):
        for i in range(n):
            idx.search_into(qs[i % 8], K, D[i % 2], I[i % 2], None, S[i % ns].cuda_stream)
    return f


def timeit(f, steps=60):
    f(8); torch.cuda.synchronize()
    t = []
    for _ in range(3):
        t0 = time.perf_counter(); f(steps); torch.cuda.synchronize(); t.append((time.perf_counter() - t0) / steps * 1e3)
    return sorted(t)[1]


res = {}
for v in (None, "1"):
    env(v)
    idx.search_into(qs[3], K, D[0], I[0], None, S[0].cuda_stream); torch.cuda.synchronize()
    res[v] = (D[0].clone(), I[0].clone())
same = torch.equal(res[None][1], res["1"][1]) and torch.equal(res[None][0].view(torch.int32), res["1"][0].view(torch.int32))
print(f"{KNOB} unset == set (ids and score bits): {same}", flush=True)
assert same
for rep in range(3):
    for v in (None, "1"):
        env(v)
        print(f"N {N} nprobe {NPROBE} {KNOB}={'1' if v else '-'}: 1 stream {timeit(step(1)):.4f} ms/step, 2 streams {timeit(step(2)):.4f} ms/step", flush=True)
