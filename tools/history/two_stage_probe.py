"""HISTORY (rounds 1-3): kept for the record of how a number in DESIGN_HISTORY.md / profiles/ was produced.  NOT maintained: knobs it
sets may no longer exist (MI_RERANK, MI_REFINE_DEBUG, MI_SCAN_DEBUG ... are silent no-ops now) and paths may have moved."""
import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import abstracts_search_amd.faiss as faiss
import abstracts_search_amd.synth as synth
n, nlist = 1_000_000, 4096
x = synth.corpus_cuda(n, 1024)
idx = faiss.IndexIVFPQ(1024, nlist, 64, 8, faiss.METRIC_INNER_PRODUCT)
idx.cp.niter = 4
idx.train(x); idx.add(x)
q = synth.queries_cuda(x, 1024)
D = torch.empty((1024, 10), device="cuda"); I = torch.empty((1024, 10), dtype=torch.int64, device="cuda")
for nprobe in (16, 64, 256):
    idx.nprobe = nprobe
    res = {}
    for mode in ("0", "1"):
        os.environ["MI_TWO_STAGE"] = mode
        os.environ["MI_REFINE_STATS"] = "1"
        idx.search_into(q, 10, D, I)
        os.environ.pop("MI_REFINE_STATS")
        for _ in range(3): idx.search_into(q, 10, D, I)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): idx.search_into(q, 10, D, I)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        res[mode] = (D.clone(), I.clone(), dt)
    same = bool((res["0"][1] == res["1"][1]).all() and (res["0"][0].view(torch.int32) == res["1"][0].view(torch.int32)).all())
    print(f"cfg2 batch 1024 nprobe {nprobe}: one-stage {res['0'][2]*1e6:.0f} us, two-stage {res['1'][2]*1e6:.0f} us, identical {same}", flush=True)
