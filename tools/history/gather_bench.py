"""HISTORY (rounds 1-3): kept for the record of how a number in DESIGN_HISTORY.md / profiles/ was produced.  NOT maintained: knobs it
sets may no longer exist (MI_RERANK, MI_REFINE_DEBUG, MI_SCAN_DEBUG ... are silent no-ops now) and paths may have moved."""
"""IndexFlatIP.rerank (gather GEMM + candidate merge) on random candidates: 1024 queries x kc
rows of an n-row store, f32 or half store -- is the gather bound by bytes or by rows (TLB reach /
scattered-row rate)?  Candidates: uniformly random over the store, or clustered into `nreg`
contiguous regions per query (what a list-ordered store would give).
usage: python tools/gather_bench.py [n_rows ...]   (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import abstracts_search_amd.faiss as faiss
import abstracts_search_amd.synth as synth

sizes = [int(a) for a in sys.argv[1:]] or [262_144, 2_097_152, 25_875_000]
q = synth.queries_cuda(synth.corpus_cuda(65536, 1024), 1024)
g = torch.Generator(device="cuda").manual_seed(1)
D = torch.empty((1024, 10), device="cuda"); I = torch.empty((1024, 10), dtype=torch.int64, device="cuda")
for n in sizes:
    for store in ("f32", "f16"):
        flat = faiss.IndexScalarQuantizer(1024) if store == "f16" else faiss.IndexFlatIP(1024)
        flat.reserve(n)
        blk = synth.corpus_cuda(1 << 20, 1024)
        for c0 in range(0, n, 1 << 20):
            flat.add(blk[:min(1 << 20, n - c0)])          # contents do not matter for the timing
        for kc, mode, env in [(640, "random", "gemm"), (800, "random", "gemm")] + [(kc, m, e) for e in ("2", "3") for kc in (640, 800)
                                                                                   for m in ("random", "8 regions")]:
            os.environ["MI_RERANK"] = env
            if True:
                if mode == "random":
                    cand = torch.randint(0, n, (1024, kc), generator=g, device="cuda")
                else:   # 8 contiguous regions of 400 rows per query, kc/8 random rows inside each
                    base = torch.randint(0, n - 400, (1024, 8, 1), generator=g, device="cuda")
                    cand = (base + torch.randint(0, 400, (1024, 8, kc // 8), generator=g, device="cuda")).view(1024, kc)
                for _ in range(3): flat.rerank(q, cand, 10, D, I)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(10): flat.rerank(q, cand, 10, D, I)
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
                b = 1024 * kc * 1024 * (2 if store == "f16" else 4)
                print(f"n {n:9d} ({n * 1024 * (2 if store == 'f16' else 4) / 1e9:6.1f} GB) {store} kc {kc} {mode:10s} MI_RERANK={env:4s}: {dt * 1e3:7.3f} ms  "
                      f"{b / dt / 1e12:5.2f} TB/s  {1024 * kc / dt / 1e9:5.2f} G rows/s", flush=True)
        del flat
        torch.cuda.empty_cache()
