"""HISTORY (rounds 1-3): kept for the record of how a number in DESIGN_HISTORY.md / profiles/ was produced.  NOT maintained: knobs it
sets may no longer exist (MI_RERANK, MI_REFINE_DEBUG, MI_SCAN_DEBUG ... are silent no-ops now) and paths may have moved."""
"""Empirical |approximate - exact| of the two-stage coarse quantiser's first stage against the
margin its second stage assumes (eps_rel |q| max|c|).  MI_REFINE_DEBUG=1 makes the second stage
report the approximate scores of the selected centroids instead of their exact ones.  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import abstracts_search_amd.faiss as faiss

rng = np.random.default_rng(0)
for d, nlist, nq, kind in ((1024, 8192, 2048, "unit gaussian"), (1024, 8192, 2048, "clustered, cos 0.7"), (128, 8192, 2048, "unit gaussian")):
    cent = rng.standard_normal((nlist, d)).astype(np.float32)
    if kind.startswith("clustered"):
        base = rng.standard_normal((256, d)).astype(np.float32)
        cent = base[rng.integers(0, 256, nlist)] + 0.65 * cent
    cent /= np.linalg.norm(cent, axis=1, keepdims=True)
    q = cent[rng.integers(0, nlist, nq)] + 0.3 * rng.standard_normal((nq, d)).astype(np.float32) / np.sqrt(d)
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    cb = rng.standard_normal((d // 16, 256, 16)).astype(np.float32)
    idx = faiss.IndexIVFPQ(d, nlist, d // 16, 8, faiss.METRIC_INNER_PRODUCT)
    idx.set_centroids(cent); idx.set_codebook(cb)
    os.environ["MI_TWO_STAGE"] = "0"
    cI0, cD0, _ = idx.coarse_and_lut(q, 64, want_lut=False)
    os.environ["MI_TWO_STAGE"] = "1"; os.environ["MI_REFINE_DEBUG"] = "1"
    cI1, cD1, _ = idx.coarse_and_lut(q, 64, want_lut=False)
    os.environ.pop("MI_REFINE_DEBUG")
    exact = {}
    err = 0.0
    for i in range(nq):
        m = dict(zip(cI0[i].tolist(), cD0[i].tolist()))
        for c, s in zip(cI1[i].tolist(), cD1[i].tolist()):
            if c in m:
                err = max(err, abs(s - m[c]))
    eps_rel = (2.0 ** -10 + d * (2.0 ** -22 + 2.0 ** -24) + 2.0 ** -26 * np.sqrt(d)) * 1.01
    print(f"d {d} nlist {nlist} {kind}: max |approx - exact| over {nq} x 64 top scores = {err:.3e} (|q| = |c| = 1); "
          f"assumed bound eps = {eps_rel:.3e}: safety factor {eps_rel / err:.0f}x", flush=True)
