"""Generates tests/golden/ivfpq_tiny.npz -- inputs and expected outputs of the
IVF-PQ path on a tiny index.

PARITY UNPINNED: neither faiss nor any other implementation of the reference's
index arithmetic is importable in the build container, so the
expected outputs come from oracle/ivfpq_oracle.c and are cross-checked here
against the independent float64 numpy brute force (decode every code, dot,
argsort).  Run from the repository root:  python tests/golden/make_ivfpq_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ivfpq_oracle as O  # noqa: E402

O.build()
rng = np.random.default_rng(20260928)
d, M, nlist, n, nq, k = 64, 8, 16, 512, 16, 10
dsub = d // M
cent = rng.standard_normal((nlist, d)).astype(np.float32)
cent /= np.linalg.norm(cent, axis=1, keepdims=True)
x = cent[rng.integers(0, nlist, n)] + 0.08 * rng.standard_normal((n, d)).astype(np.float32)
x /= np.linalg.norm(x, axis=1, keepdims=True)
x[500:512] = x[0:12]          # exact duplicates -> exact score ties, ordered by id
x = x.astype(np.float32)
cb = (0.08 * rng.standard_normal((M, 256, dsub))).astype(np.float32)
q = x[rng.integers(0, n, nq)] + 0.02 * rng.standard_normal((nq, d)).astype(np.float32)
q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
ids = (np.arange(n, dtype=np.int64) * 7 + 1000)          # add_with_ids-style ids

list_no, codes = O.encode(x, cent, cb, True)
off, lcodes, lids = O.build_lists(list_no, codes, ids, nlist)
out = dict(centroids=cent, codebook=cb, x=x, q=q, ids=ids, list_no=list_no, codes=codes,
           lut_q0=O.lut(q[0], cb), k=np.int64(k))
for nprobe in (1, 4, 16):
    D, I, cI, cD = O.search(q, cent, cb, off, lcodes, lids, nprobe, k, True, return_coarse=True)
    D2, I2 = O.brute_force_search(q, cent, cb, off, lcodes, lids, nprobe, k, True)
    fin = I >= 0
    assert np.array_equal(I < 0, I2 < 0)
    # the float64 brute force may order near-ties differently; sets must agree
    # wherever the k-th and (k+1)-th scores are separated by more than rounding
    assert np.allclose(D[fin], D2[fin], rtol=0, atol=2e-5), np.abs(D[fin] - D2[fin]).max()
    agree = np.mean([len(set(a[a >= 0]) ^ set(b[b >= 0])) == 0 for a, b in zip(I, I2)])
    assert agree >= 0.9, agree
    out[f"D_np{nprobe}"], out[f"I_np{nprobe}"] = D, I
    out[f"cI_np{nprobe}"], out[f"cD_np{nprobe}"] = cI, cD
# flat IP (config #1 plumbing) on the same vectors
out["flat_D"], out["flat_I"] = O.flat_ip(q, x, k)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ivfpq_tiny.npz"), **out)
print("wrote ivfpq_tiny.npz", {k_: v.shape for k_, v in out.items() if hasattr(v, "shape")})
