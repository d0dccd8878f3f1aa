"""HISTORY (rounds 1-3): kept for the record of how a number in DESIGN_HISTORY.md / profiles/ was produced.  NOT maintained: knobs it
sets may no longer exist (MI_RERANK, MI_REFINE_DEBUG, MI_SCAN_DEBUG ... are silent no-ops now) and paths may have moved."""
"""select_refine_kernel at the cfg4 coarse size (1024 queries x 65536 centroids): time by nprobe, with and without
the exact chains (MI_REFINE_DEBUG=1), through mi_index_coarse (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import abstracts_search_amd.faiss as faiss
import abstracts_search_amd.synth as synth
d, nlist = 1024, 65536
idx = faiss.IndexIVFPQ(d, nlist, 64, 8, faiss.METRIC_INNER_PRODUCT)
cent = synth.corpus_cuda(nlist, d, row0=1 << 20)
idx.set_centroids(cent)
idx.set_codebook(torch.zeros((64, 256, 16), device="cuda"))
q = synth.queries_cuda(synth.corpus_cuda(4096, d), 1024)
for nprobe in (8, 64):
    for dbg in ("0", "1"):
        os.environ["MI_REFINE_DEBUG"] = dbg
        for _ in range(3): idx.coarse_slice(q, nprobe, 0, nlist)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): idx.coarse_slice(q, nprobe, 0, nlist)
        torch.cuda.synchronize()
        print(f"nprobe {nprobe} debug {dbg}: {(time.perf_counter()-t)/20*1e6:.0f} us per coarse call (to_f16 + f16 GEMM + select_refine)", flush=True)
