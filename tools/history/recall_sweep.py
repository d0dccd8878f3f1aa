"""HISTORY (rounds 1-3): kept for the record of how a number in DESIGN_HISTORY.md / profiles/ was produced.  NOT maintained: knobs it
sets may no longer exist (MI_RERANK, MI_REFINE_DEBUG, MI_SCAN_DEBUG ... are silent no-ops now) and paths may have moved."""
"""recall@10 vs nprobe on the cfg2 synthetic corpus (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import abstracts_search_amd.faiss as faiss
import abstracts_search_amd.synth as synth
n, nlist = int(os.environ.get("N", 1000000)), int(os.environ.get("NLIST", 4096))
cos = float(os.environ.get("COS", 0.7))
x = synth.corpus_cuda(n, 1024, cos=cos)
idx = faiss.IndexIVFPQ(1024, nlist, 64, 8, faiss.METRIC_INNER_PRODUCT)
idx.cp.niter = int(os.environ.get("NITER", 10))
t0 = time.time(); idx.train(x); t1 = time.time(); idx.add(x); t2 = time.time()
print(f"train {t1-t0:.1f}s add {t2-t1:.1f}s", flush=True)
q = synth.queries_cuda(x, 512, cos=cos)
flat = faiss.IndexFlatIP(1024); flat.add(x)
De, Ie = flat.search(q, 10); Ie = Ie.cpu().numpy()
for nprobe in (1, 4, 16, 32, 64, 128, 256, 1024, nlist):
    idx.nprobe = nprobe
    torch.cuda.synchronize(); t0 = time.perf_counter()
    D, I = idx.search(q, 10); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    I = I.cpu().numpy()
    r = np.mean([len(set(a) & set(b)) / 10 for a, b in zip(I, Ie)])
    r1 = np.mean([a[0] == b[0] for a, b in zip(I, Ie)])
    print(f"nprobe {nprobe:5d} recall@10 {r:.4f} top1 {r1:.4f}  ({dt*1e3:.2f} ms for 512 queries)", flush=True)
