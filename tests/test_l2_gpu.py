"""METRIC_L2 (faiss's default metric): IndexFlatL2, IndexIVFPQ and IndexRefineFlat on the HIP
kernels against the oracle's L2 restatement (oracle/ivfpq_oracle.c, section METRIC_L2), bit for
bit: list numbers, codes, per-vector terms, coarse lists, squared distances, ids, padding."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
FMAX = np.float32(np.finfo(np.float32).max)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def faiss():
    import abstracts_search_amd.faiss as f
    return f


def problem(seed, d, M, nlist, n, nq):
    rng = np.random.default_rng(seed)
    cent = rng.standard_normal((nlist, d)).astype(np.float32)
    cb = (0.3 * rng.standard_normal((M, 256, d // M))).astype(np.float32)
    x = (cent[rng.integers(0, nlist, n)] + 0.35 * rng.standard_normal((n, d))).astype(np.float32)
    q = (x[rng.integers(0, n, nq)] + 0.1 * rng.standard_normal((nq, d))).astype(np.float32)
    return cent, cb, x, q


def test_flat_l2(faiss, oracle):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((5000, 96)).astype(np.float32)
    q = rng.standard_normal((37, 96)).astype(np.float32)
    idx = faiss.index_factory(96, "Flat")                      # faiss's default metric: L2
    assert idx.metric_type == faiss.METRIC_L2
    idx.add(x[:3000]); idx.add(x[3000:])
    for k in (1, 10, 100):
        D, I = idx.search(q, k)
        De, Ie = oracle.flat_l2(q, x, k)
        assert np.array_equal(I, Ie) and np.array_equal(bits(D), bits(De))
    assert np.array_equal(bits(idx.reconstruct_n(10, 5)), bits(x[10:15]))
    import torch
    Dt, It = idx.search(torch.from_numpy(q).cuda(), 10)
    De, Ie = oracle.flat_l2(q, x, 10)
    assert np.array_equal(It.cpu().numpy(), Ie) and np.array_equal(bits(Dt.cpu().numpy()), bits(De))
    e = faiss.IndexFlatL2(96)
    D, I = e.search(q, 3)
    assert (I == -1).all() and (D == FMAX).all()


@pytest.mark.parametrize("d,M,nlist,n,nq", [(64, 8, 16, 4000, 33), (128, 16, 64, 20000, 150), (1024, 64, 128, 12000, 64)])
@pytest.mark.parametrize("by_residual", [True, False])
def test_ivfpq_l2_matches_oracle(faiss, oracle, d, M, nlist, n, nq, by_residual):
    cent, cb, x, q = problem(d + M + nlist, d, M, nlist, n, nq)
    idx = faiss.IndexIVFPQ(d, nlist, M, 8, faiss.METRIC_L2, by_residual)
    idx.set_centroids(cent)
    idx.set_codebook(cb)
    idx.add(x[: n // 2]); idx.add(x[n // 2:])
    ln, codes, t = oracle.encode_l2(x, cent, cb, by_residual)
    ln_h, codes_h = idx.encode(x[:500])
    assert np.array_equal(ln_h, ln[:500]) and np.array_equal(codes_h, codes[:500])
    off, lc, li, lt = oracle.build_lists_l2(ln, codes, np.arange(n), t, nlist)
    for l in (0, nlist - 1):
        c, i = idx.get_list(l)
        assert np.array_equal(c, lc[off[l]:off[l + 1]]) and np.array_equal(i, li[off[l]:off[l + 1]])
    for nprobe, k in ((1, 10), (5, 10), (nlist, 1), (min(nlist, 20), 64), (8, 100)):
        idx.nprobe = nprobe
        D, I = idx.search(q, k)
        De, Ie, cIe, cDe = oracle.search_l2(q, cent, cb, off, lc, li, lt, nprobe, k, by_residual, return_coarse=True)
        cI, cD, _ = idx.coarse_and_lut(q, nprobe, want_lut=False)
        assert np.array_equal(cI, cIe) and np.array_equal(bits(cD), bits(cDe)), (nprobe, k)
        assert np.array_equal(I, Ie), (nprobe, k, np.argwhere(I != Ie)[:5])
        assert np.array_equal(bits(D), bits(De)), (nprobe, k)
        assert (np.diff(D, axis=1) >= 0).all()                 # ascending squared distances
    import torch
    Dt, It = idx.search(torch.from_numpy(q).cuda(), 100)
    assert np.array_equal(It.cpu().numpy(), Ie) and np.array_equal(bits(Dt.cpu().numpy()), bits(De))


def test_ivfpq_l2_factory_train_write_read(faiss, oracle, tmp_path):
    """the default factory call (no metric argument = METRIC_L2), trained on the GPU == the oracle's
    training; the index survives write_index / read_index (IwPQ over IxF2)"""
    from oracle import train_oracle as T
    import abstracts_search_amd.faiss_io as fio
    rng = np.random.default_rng(5)
    d, nlist, M, n = 64, 40, 8, 9000
    c = rng.standard_normal((30, d)).astype(np.float32)
    x = (c[rng.integers(0, 30, n)] + 0.4 * rng.standard_normal((n, d))).astype(np.float32)
    idx = faiss.index_factory(d, f"IVF{nlist},PQ{M}")
    assert idx.metric_type == faiss.METRIC_L2
    idx.cp.niter = idx.pq.cp.niter = 4
    idx.train(x)
    ce, cbe = T.train_ivfpq_l2(x, nlist, M, True, niter=4, max_points_per_centroid=idx.cp.max_points_per_centroid, seed=idx.cp.seed)
    assert np.array_equal(bits(idx.get_centroids()), bits(ce)) and np.array_equal(bits(idx.get_codebook()), bits(cbe))
    idx.add(x)
    idx.nprobe = 6
    q = x[:40] + 0.05 * rng.standard_normal((40, d)).astype(np.float32)
    D, I = idx.search(q, 10)
    ln, codes, t = oracle.encode_l2(x, ce, cbe)
    off, lc, li, lt = oracle.build_lists_l2(ln, codes, np.arange(n), t, nlist)
    De, Ie = oracle.search_l2(q, ce, cbe, off, lc, li, lt, 6, 10)
    assert np.array_equal(I, Ie) and np.array_equal(bits(D), bits(De))
    f = str(tmp_path / "l2.faiss")
    faiss.write_index(idx, f)
    z = fio.parse(f)
    assert z["metric"] == faiss.METRIC_L2 and z["ntotal"] == n
    back = faiss.read_index(f)
    assert back.metric_type == faiss.METRIC_L2
    D2, I2 = back.search(q, 10)
    assert np.array_equal(I2, I) and np.array_equal(bits(D2), bits(D))
    # recall sanity against exact L2
    flat = faiss.IndexFlatL2(d)
    flat.add(x)
    _, Ig = flat.search(q, 10)
    idx.nprobe = nlist
    _, Ia = idx.search(q, 10)
    assert np.mean([len(set(a) & set(b)) / 10 for a, b in zip(Ia.tolist(), Ig.tolist())]) > 0.3


def test_refine_flat_l2(faiss, oracle):
    d, M, nlist, n, nq, k = 64, 8, 32, 6000, 30, 10
    cent, cb, x, q = problem(91, d, M, nlist, n, nq)
    idx = faiss.index_factory(d, f"IVF{nlist},PQ{M},RFlat")
    idx.base_index.set_centroids(cent)
    idx.base_index.set_codebook(cb)
    idx.add(x)
    idx.nprobe, idx.k_factor = 6, 8
    D, I = idx.search(q, k)
    ln, codes, t = oracle.encode_l2(x, cent, cb)
    off, lc, li, lt = oracle.build_lists_l2(ln, codes, np.arange(n), t, nlist)
    _, cand = oracle.search_l2(q, cent, cb, off, lc, li, lt, 6, k * 8)
    # exact squared distances of the candidates, the oracle's flat arithmetic restricted to them
    for j in range(nq):
        ids = cand[j][cand[j] >= 0]
        De, Ie = oracle.flat_l2(q[j:j + 1], x[ids], k)
        assert np.array_equal(I[j], ids[Ie[0]]) or np.array_equal(bits(D[j]), bits(De[0]))
        assert np.array_equal(bits(D[j]), bits(De[0]))


def test_ivfpq_l2_two_stage_coarse(faiss, oracle, monkeypatch):
    """METRIC_L2 on a big coarse table (nlist >= 8192, d % 128 == 0: augmented width d + 128): the
    two-stage quantiser (f16 MFMA scores + exact re-scoring inside the proven margin) on the
    augmented vectors gives the oracle's lists, in search and in add"""
    monkeypatch.setenv("MI_TWO_STAGE", "1")
    d, M, nlist, n, nq, k = 128, 16, 8192, 30000, 96, 10
    rng = np.random.default_rng(17)
    cent = rng.standard_normal((nlist, d)).astype(np.float32)
    cent *= (0.5 + rng.random((nlist, 1))).astype(np.float32)              # norms differ: L2 order != inner-product order
    cb = (0.2 * rng.standard_normal((M, 256, d // M))).astype(np.float32)
    x = (cent[rng.integers(0, nlist, n)] + 0.2 * rng.standard_normal((n, d))).astype(np.float32)
    q = (x[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, d))).astype(np.float32)
    idx = faiss.IndexIVFPQ(d, nlist, M, 8, faiss.METRIC_L2)
    idx.set_centroids(cent)
    idx.set_codebook(cb)
    idx.add(x)
    ln, codes, t = oracle.encode_l2(x, cent, cb)
    ln_h, codes_h = idx.encode(x[:2000])
    assert np.array_equal(ln_h, ln[:2000]) and np.array_equal(codes_h, codes[:2000])
    off, lc, li, lt = oracle.build_lists_l2(ln, codes, np.arange(n), t, nlist)
    for nprobe in (1, 32):
        idx.nprobe = nprobe
        D, I = idx.search(q, k)
        De, Ie, cIe, cDe = oracle.search_l2(q, cent, cb, off, lc, li, lt, nprobe, k, return_coarse=True)
        cI, cD, _ = idx.coarse_and_lut(q, nprobe, want_lut=False)
        assert np.array_equal(cI, cIe) and np.array_equal(bits(cD), bits(cDe))
        assert np.array_equal(I, Ie) and np.array_equal(bits(D), bits(De))
    # inner-product order would pick other lists for some queries (the test data makes the metrics differ)
    ip = faiss.IndexIVFPQ(d, nlist, M, 8, faiss.METRIC_INNER_PRODUCT)
    ip.set_centroids(cent)
    ip.set_codebook(cb)
    cI_ip, _, _ = ip.coarse_and_lut(q, 1, want_lut=False)
    assert (cI_ip[:, 0] != cIe[:, 0]).any()
