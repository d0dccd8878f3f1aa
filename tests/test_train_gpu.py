"""IndexIVFPQ.train (reference Makefile:39): the HIP training
path against its CPU restatement (oracle/train_oracle.py) -- same seeded draws, every
arithmetic step in a fixed order -> centroids and PQ codebook equal bit for bit, run to run
and against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _data(seed, n, d, nclu):
    rng = np.random.default_rng(seed)
    c = rng.standard_normal((nclu, d)).astype(np.float32)
    x = c[rng.integers(0, nclu, n)] + 0.4 * rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float32)


@pytest.mark.parametrize("n,d,nlist,M,by_residual", [(6000, 64, 32, 8, True), (3000, 32, 50, 4, False), (20000, 128, 300, 16, True)])
def test_train_is_bit_equal_to_the_oracle(oracle, n, d, nlist, M, by_residual):
    import abstracts_search_amd.faiss as faiss
    from oracle import train_oracle as T
    x = _data(n + d, n, d, 40)
    idx = faiss.IndexIVFPQ(d, nlist, M, 8, faiss.METRIC_INNER_PRODUCT, by_residual)
    idx.cp.niter = idx.pq.cp.niter = 5
    idx.train(x)
    cent, cb = idx.get_centroids(), idx.get_codebook()
    ce, cbe = T.train_ivfpq(x, nlist, M, by_residual, niter=5, max_points_per_centroid=idx.cp.max_points_per_centroid,
                            seed=idx.cp.seed)
    assert np.array_equal(bits(cent), bits(ce))
    assert np.array_equal(bits(cb), bits(cbe))
    # run to run
    idx2 = faiss.IndexIVFPQ(d, nlist, M, 8, faiss.METRIC_INNER_PRODUCT, by_residual)
    idx2.cp.niter = idx2.pq.cp.niter = 5
    idx2.train(x)
    assert np.array_equal(bits(idx2.get_centroids()), bits(cent)) and np.array_equal(bits(idx2.get_codebook()), bits(cb))
    # a training run is k-means: the quantisation error of the trained tables beats the initial draw
    a = oracle.flat_ip(x, cent, 1)[1][:, 0]
    assert len(np.unique(a)) > 0.5 * min(nlist, 40)


@pytest.mark.parametrize("n,d,nlist,M", [(6000, 64, 32, 8), (20000, 128, 300, 16)])
def test_spherical_train_is_bit_equal_to_the_oracle(oracle, n, d, nlist, M):
    """ClusteringParameters.spherical (what faiss's index_factory sets for METRIC_INNER_PRODUCT [PRIOR]; reference Makefile:39
    trains through the factory): unit-norm coarse centroids, inner-product assignment -- product == oracle bit for bit, and
    index_factory turns it on while the constructor leaves it off."""
    import abstracts_search_amd.faiss as faiss
    from oracle import train_oracle as T
    x = _data(n + d + 1, n, d, 40)
    idx = faiss.index_factory(d, f"IVF{nlist},PQ{M}", faiss.METRIC_INNER_PRODUCT)
    assert idx.cp.spherical is True and idx.pq.cp.spherical is False
    assert faiss.IndexIVFPQ(d, nlist, M, 8, faiss.METRIC_INNER_PRODUCT).cp.spherical is False
    assert faiss.index_factory(d, f"IVF{nlist},PQ{M}").cp.spherical is False          # METRIC_L2: not spherical
    idx.cp.niter = idx.pq.cp.niter = 5
    idx.train(x)
    cent, cb = idx.get_centroids(), idx.get_codebook()
    ce, cbe = T.train_ivfpq(x, nlist, M, True, niter=5, max_points_per_centroid=idx.cp.max_points_per_centroid,
                            seed=idx.cp.seed, spherical=True)
    assert np.array_equal(bits(cent), bits(ce))
    assert np.array_equal(bits(cb), bits(cbe))
    assert np.allclose(np.linalg.norm(cent.astype(np.float64), axis=1), 1.0, atol=1e-6)
    plain = faiss.IndexIVFPQ(d, nlist, M, 8, faiss.METRIC_INNER_PRODUCT)
    plain.cp.niter = plain.pq.cp.niter = 5
    plain.train(x)
    assert not np.array_equal(bits(plain.get_centroids()), bits(cent))              # the flag changes the index
    idx.pq.cp.spherical = True
    with pytest.raises(NotImplementedError, match="pq.cp.spherical"):
        idx.train(x)


def test_cluster_means_kernel(oracle):
    """mi_cluster_means on skewed cluster sizes (one huge, many empty) == the plain loop"""
    import torch
    import abstracts_search_amd._train as tr
    rng = np.random.default_rng(9)
    n, d, k = 70000, 48, 500
    x = rng.standard_normal((n, d)).astype(np.float32)
    a = rng.integers(0, 40, n).astype(np.int32)
    a[rng.random(n) < 0.5] = 7                              # half the rows in one cluster
    a[:3] = [499, 499, 123]
    c0 = rng.standard_normal((k, d)).astype(np.float32)     # rows of empty clusters must survive
    ce = c0.copy()
    cnt_e = oracle.cluster_means(x, a, ce)
    xc, ac, cc = torch.from_numpy(x).cuda(), torch.from_numpy(a).cuda(), torch.from_numpy(c0.copy()).cuda()
    cnt = tr._cluster_means(xc, ac, cc, 0)
    assert np.array_equal(cnt.cpu().numpy(), cnt_e)
    assert np.array_equal(bits(cc.cpu().numpy()), bits(ce))
    assert np.array_equal(bits(tr._neg_half_sqnorm(xc, 0).cpu().numpy()), bits(oracle.neg_half_sqnorm(x)))
