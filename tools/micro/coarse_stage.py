"""The coarse stage of the headline step alone: 1024 queries x 65 536 trained centroids x 1024 dims, nprobe 64 and 8
(COARSE_SHAPES="rows:nprobe,..." for others: "4096:1" is the assignment step of add()).
COARSE_ONCE=1: one pass (for rocprofv3 --kernel-trace; tools/micro/trace_tail.py prints the last launches).
Prints ms per call, candidates per row and the in-kernel phase stamps of the second stage (MI_REFINE_STATS,
MI_REFINE_TS) and checks that the two-stage quantiser returns the lists and score bits of the one-stage exact one.
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import abstracts_search_amd.faiss as faiss
import abstracts_search_amd.synth as synth

NLIST, B, D = 65536, 1024, 1024
CH = 1 << 20
idx = faiss.IndexIVFPQ(D, NLIST, 64, 8, faiss.METRIC_INNER_PRODUCT)
idx.cp.niter = 4
idx.train(synth.corpus_cuda(4 * CH, D))
xq = synth.corpus_cuda(CH, D, row0=64 * CH)
SHAPES = [tuple(int(v) for v in t.split(":")) for t in os.environ.get("COARSE_SHAPES", "1024:64,1024:8").split(",")]
QALL = synth.queries_cuda(xq, 8 * max(b for b, _ in SHAPES), seed=4321)
once = os.environ.get("COARSE_ONCE") == "1"


def env(**kv):
    for k, v in kv.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    faiss.reload_env()


def timeit(nprobe, n=200):
    for i in range(10):
        idx.coarse_slice(qs[i % 8], nprobe, 0, NLIST)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        idx.coarse_slice(qs[i % 8], nprobe, 0, NLIST)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for B, nprobe in SHAPES:
    qs = QALL[:8 * B].view(8, B, D)
    env(MI_TWO_STAGE="0")
    I0, D0 = idx.coarse_slice(qs[0], nprobe, 0, NLIST)
    env(MI_TWO_STAGE=None, MI_REFINE_STATS="1", MI_REFINE_TS="1")
    I1, D1 = idx.coarse_slice(qs[0], nprobe, 0, NLIST)
    torch.cuda.synchronize()
    same = bool(torch.equal(I0, I1)) and bool(torch.equal(D0.view(torch.int32), D1.view(torch.int32)))
    print(f"{B} rows, nprobe {nprobe}: identical to the exact quantiser: {same}", flush=True)
    assert same
    env(MI_REFINE_STATS=None, MI_REFINE_TS=None)
    if once:
        continue
    for rep in range(3):
        print(f"{B} rows, nprobe {nprobe}: {timeit(nprobe):.4f} ms per coarse call", flush=True)
    env(MI_TWO_STAGE="0")
    print(f"{B} rows, nprobe {nprobe} exact one-stage quantiser: {timeit(nprobe, 20):.4f} ms per coarse call", flush=True)
    env(MI_TWO_STAGE=None)
