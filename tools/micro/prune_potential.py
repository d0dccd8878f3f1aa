"""How much of the PQ scan's table gathers an EXACT early abandon could skip, on the bench's data.

A code's score is the f32 chain s_0 = coarse term, s_{m+1} = s_m + LUT[m][code[m]].  With R_m = sum_{j >= m} max_c LUT[j][c],
s_64 <= s_m + R_m (+ rounding slack), so once every lane of a 64-code group has s_m + R_m < T (T = a score already known to be
beaten by k others) the rest of the group's gathers cannot change the result.  This script measures, for a corpus with the
per-cluster density and list length of BASELINE.json configs[3] (12.6 k rows per cluster, ~3 159 codes per list) at nprobe 64:
the fraction of (group, sub-quantiser) gathers left when the test runs every STEP sub-quantisers, for T = the final k-th best
of the whole query (optimistic) and T = the k-th best of the codes an 8-list slice has seen so far (what a workgroup knows).
"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

faiss = importlib.import_module("abstracts-search_amd.faiss")
synth = importlib.import_module("abstracts-search_amd.synth")

N = int(os.environ.get("PP_N", 8_400_000))
NC = int(os.environ.get("PP_CLUSTERS", 664))            # 12.6 k rows per cluster, as 207 M / 16 384
NLIST = int(os.environ.get("PP_NLIST", 2656))           # ~3 160 codes per list
NPROBE, K, NQ, STEP, SLICE = 64, 10, int(os.environ.get("PP_NQ", 24)), int(os.environ.get("PP_STEP", 8)), 8
d = 1024
x = synth.corpus_cuda(N, d, ncentres=NC)
q = synth.queries_cuda(x, NQ)
index = faiss.index_factory(d, f"IVF{NLIST},PQ64", faiss.METRIC_INNER_PRODUCT)
perm = torch.randperm(N, device=x.device)[: min(N, 256 * NLIST)]
index.train(x[perm])
for i in range(0, N, 1 << 20):
    index.add(x[i:i + (1 << 20)])
index.nprobe = NPROBE
D, I = index.search(q, K)
cI, cD, lut = index.coarse_and_lut(q.cpu().numpy(), NPROBE)
D = D.cpu().numpy() if hasattr(D, "cpu") else D
tot = np.zeros(2)
left = np.zeros(2)
hist = np.zeros((2, 64 // STEP + 1))
for qi in range(NQ):
    L = lut[qi]                                           # [64][256]
    mx = L.max(axis=1)
    R = np.concatenate([np.cumsum(mx[::-1], dtype=np.float64)[::-1], [0.0]])      # R[m] = sum_{j >= m} max_j
    T_final = D[qi, K - 1]
    for s0 in range(0, NPROBE, SLICE):
        seen = np.full(K, -np.inf, np.float32)            # the slice's running k best
        for p in range(s0, min(s0 + SLICE, NPROBE)):
            ln = int(cI[qi, p])
            if ln < 0:
                continue
            codes, _ = index.get_list(ln)
            if codes.shape[0] == 0:
                continue
            terms = L[np.arange(64)[None, :], codes]      # [n][64]
            part = np.cumsum(np.concatenate([np.full((codes.shape[0], 1), cD[qi, p], np.float32), terms], axis=1), axis=1, dtype=np.float32)
            n = codes.shape[0]
            ng = (n + 63) // 64
            padded = np.full((ng * 64, 65), -np.inf, np.float32)
            padded[:n] = part
            ub = padded[:, :64:STEP] + R[None, :64:STEP].astype(np.float32)          # bound at the check points m = 0, STEP, ...
            gub = ub.reshape(ng, 64, -1).max(axis=1)                                   # per group: the best any lane can still reach
            for g in range(ng):
                T_run = seen[K - 1]
                for which, T in enumerate((T_final, T_run)):
                    dead = np.nonzero(gub[g] < T)[0]
                    stop = int(dead[0]) * STEP if dead.size else 64
                    tot[which] += 64
                    left[which] += stop
                    hist[which, stop // STEP] += 1
                sc = padded[g * 64:(g + 1) * 64, 64]
                seen = np.sort(np.concatenate([seen, sc]))[::-1][:K]
for which, name in enumerate(("T = the query's final k-th best", "T = the slice's running k-th best")):
    print(f"{name}: {left[which] / tot[which]:.3f} of the gathers left (test every {STEP} sub-quantisers)")
    print("   groups by the sub-quantiser they stop at:", {int(i * STEP): int(c) for i, c in enumerate(hist[which]) if c})
print("list length mean", float(index.list_sizes().mean()), "rows per cluster", N / NC)
