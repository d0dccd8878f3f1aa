"""First contact with the real faiss as ONE command (nothing around it: no bench, no fixtures to prepare):

    python tools/crosscheck_faiss.py [--n 200000] [--nlist 256] [--M 16] [--d 128] [--nprobe 8] [--k 10] [--self]

faiss is on no box of this pool (parity unpinned); the day a box has it, this prints the answers the
parity claim is missing, both directions of the file format included:

  A  index built HERE (this package: train + add on the GPU) -> write_index -> REAL faiss.read_index -> search at the same
     nprobe: per-rank exact-match rate of the ids, every mismatch classified -- `tie` (the two libraries return the same
     score there to 1 ulp: heap order against this package's (score desc, id asc) total order), `set` (same ids in the top
     k, another order: f32 summation order) or `real` (a different id set: a format or semantics disagreement) -- and the
     largest score difference in ulps;
  B  index trained + filled by REAL faiss (index_factory "IVF{nlist},PQ{M}") -> faiss.write_index -> read_index HERE ->
     the same comparison;  with --hnsw also "IVF{nlist}_HNSW32,PQ{M}" (read here as its flat storage, searched exactly:
     every mismatch there is first checked against the probe sets).

Exit code: 0 compared and no `real` mismatch, 1 `real` mismatches (the report says where), 3 faiss absent.
--self swaps this package in for the real one (exercises the tool itself: tests/test_ivfpq_gpu.py)."""
import argparse
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def ulps(a, b):
    ia, ib = a.astype(np.float32).view(np.int32).astype(np.int64), b.astype(np.float32).view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, np.int64(-(1 << 31)) - ia, ia)
    ib = np.where(ib < 0, np.int64(-(1 << 31)) - ib, ib)
    return np.abs(ia - ib)


def compare(tag, D1, I1, D2, I2):
    """(D1, I1) = this package, (D2, I2) = the other library; -> report dict"""
    nq, k = I1.shape
    same = I1 == I2
    rep = {"case": tag, "queries": int(nq), "k": int(k), "id_match_rate": float(same.mean()),
           "per_rank_match": [round(float(same[:, r].mean()), 6) for r in range(k)], "tie": 0, "set": 0, "real": 0, "examples": []}
    fin = np.isfinite(D1) & np.isfinite(D2)
    rep["max_score_ulps"] = int(ulps(D1[fin], D2[fin]).max()) if fin.any() else 0
    for q in np.flatnonzero(~same.all(1)):
        s1, s2 = set(I1[q].tolist()), set(I2[q].tolist())
        for r in np.flatnonzero(~same[q]):
            if np.isfinite(D1[q, r]) and np.isfinite(D2[q, r]) and ulps(D1[q, r:r + 1], D2[q, r:r + 1])[0] <= 1:
                kind = "tie"
            elif s1 == s2:
                kind = "set"
            else:
                kind = "real"
            rep[kind] += 1
            if kind == "real" and len(rep["examples"]) < 5:
                rep["examples"].append({"query": int(q), "rank": int(r), "here": [int(I1[q, r]), float(D1[q, r])],
                                        "there": [int(I2[q, r]), float(D2[q, r])]})
    return rep


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--n", type=int, default=200000)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--nlist", type=int, default=256)
    ap.add_argument("--M", type=int, default=16)
    ap.add_argument("--nprobe", type=int, default=8)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--nq", type=int, default=500)
    ap.add_argument("--hnsw", action="store_true")
    ap.add_argument("--self", dest="self_", action="store_true")
    args = ap.parse_args()
    import abstracts_search_amd.faiss as mine
    if args.self_:
        real, version = mine, "self"
    else:
        try:
            import faiss as real
            if getattr(real, "__file__", "").startswith(os.path.dirname(os.path.abspath(mine.__file__))):
                raise ImportError("`faiss` resolves to this package's drop-in alias, not the real library")
            version = getattr(real, "__version__", "?")
        except Exception as e:
            print(json.dumps({"faiss": "absent", "why": f"{type(e).__name__}: {e}"}))
            return 3
    rng = np.random.default_rng(1234)
    cen = rng.standard_normal((max(16, args.nlist // 4), args.d)).astype(np.float32)
    x = cen[rng.integers(0, cen.shape[0], args.n)] + 0.35 * rng.standard_normal((args.n, args.d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    q = x[rng.integers(0, args.n, args.nq)] + 0.05 * rng.standard_normal((args.nq, args.d)).astype(np.float32)
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    out = {"faiss": version, "shape": vars(args), "cases": []}
    tmp = tempfile.mkdtemp(prefix="crosscheck_")
    # ---- A: built here, read by the other library
    a = mine.index_factory(args.d, f"IVF{args.nlist},PQ{args.M}", mine.METRIC_INNER_PRODUCT)
    a.cp.niter = a.pq.cp.niter = 6
    a.train(x)
    a.add(x)
    a.nprobe = args.nprobe
    Da, Ia = a.search(q, args.k)
    fa = os.path.join(tmp, "built_here.faiss")
    mine.write_index(a, fa)
    try:
        theirs = real.read_index(fa)
        theirs.nprobe = args.nprobe
        Dt, It = theirs.search(q, args.k)
        out["cases"].append(compare("A: built here -> write_index -> faiss.read_index -> search", Da, Ia, np.asarray(Dt), np.asarray(It)))
    except Exception as e:
        out["cases"].append({"case": "A", "error": f"the other library could not read this package's file: {type(e).__name__}: {e}", "real": 1})
    # ---- B: trained, filled and written by the other library, read here
    for spec in [f"IVF{args.nlist},PQ{args.M}"] + ([f"IVF{args.nlist}_HNSW32,PQ{args.M}"] if args.hnsw else []):
        try:
            b = real.index_factory(args.d, spec, real.METRIC_INNER_PRODUCT)
            b.train(x)
            b.add(x)
            b.nprobe = args.nprobe
            Db, Ib = b.search(q, args.k)
            fb = os.path.join(tmp, "built_there.faiss")
            real.write_index(b, fb)
            here = mine.read_index(fb)
            here.nprobe = args.nprobe
            Dh, Ih = here.search(q, args.k)
            out["cases"].append(compare(f"B: faiss {spec} -> faiss.write_index -> read_index here -> search", Dh, Ih, np.asarray(Db), np.asarray(Ib)))
        except Exception as e:
            out["cases"].append({"case": f"B: {spec}", "error": f"{type(e).__name__}: {e}", "real": 1})
    print(json.dumps(out, indent=1))
    return 1 if any(c.get("real", 0) for c in out["cases"]) else 0


if __name__ == "__main__":
    sys.exit(main())
