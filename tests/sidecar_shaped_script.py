"""What a `sidecar-search index train|fill|tune` + query-time caller does with faiss, written
against the real faiss names only (reference Makefile:39,25,32; README.md:28).  Run unchanged
under the sitecustomize alias by tests/test_dropin_alias.py; prints one line of JSON."""
import json
import sys

import numpy as np

import faiss                                                   # resolved by integration/sitecustomize.py

d, nlist, M = 64, 32, 8
rng = np.random.default_rng(5)
centres = rng.standard_normal((nlist, d)).astype(np.float32)
x = centres[rng.integers(0, nlist, 20000)] + 0.3 * rng.standard_normal((20000, d)).astype(np.float32)
faiss.normalize_L2(x)
q = x[:50] + 0.02 * rng.standard_normal((50, d)).astype(np.float32)
faiss.normalize_L2(q)
out = sys.argv[1]

index = faiss.index_factory(d, f"IVF{nlist},PQ{M}", faiss.METRIC_INNER_PRODUCT)        # train
assert not index.is_trained
index.train(x)
faiss.write_index(index, out + "/empty.faiss")
index = faiss.read_index(out + "/empty.faiss")                                          # fill
assert index.is_trained and index.ntotal == 0
for lo in range(0, len(x), 4096):
    index.add(x[lo:lo + 4096])
faiss.write_index(index, out + "/index.faiss")
index = faiss.read_index(out + "/index.faiss")                                          # tune / query
ivf = faiss.extract_index_ivf(index)
ivf.nprobe = 8
D, I = index.search(q, 10)
# the real-faiss constructor signature, with a quantizer object
quantizer = faiss.IndexFlatIP(d)
index2 = faiss.IndexIVFPQ(quantizer, d, nlist, M, 8, faiss.METRIC_INNER_PRODUCT)
index2.train(x)
index2.add(x)
index2.nprobe = 8
D2, I2 = index2.search(q, 10)
flat = faiss.IndexFlatIP(d)
flat.add(x)
_, Igt = flat.search(q, 10)
recall = np.mean([len(set(a) & set(b)) / 10 for a, b in zip(I.tolist(), Igt.tolist())])
print(json.dumps({"ntotal": int(index.ntotal), "shape": list(I.shape), "sorted": bool((np.diff(D, axis=1) <= 0).all()),
                  "self_hit": float(np.mean(I[:, 0] == np.arange(50))), "recall_at_10": float(recall),
                  "ctor_signature_ok": bool(I2.shape == (50, 10) and index2.quantizer.ntotal == nlist),
                  "module": faiss.__name__}))
