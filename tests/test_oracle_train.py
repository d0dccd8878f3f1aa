"""CPU checks of the training restatement (oracle/train_oracle.py): it is k-means (the
objective falls), deterministic, and its building blocks agree with float64 numpy."""
import numpy as np


def test_kmeans_restatement_reduces_the_objective(oracle):
    from oracle import train_oracle as T
    rng = np.random.default_rng(0)
    c = rng.standard_normal((12, 16)).astype(np.float32)
    x = (c[rng.integers(0, 12, 3000)] + 0.2 * rng.standard_normal((3000, 16))).astype(np.float32)

    def objective(cent):
        d2 = ((x[:, None, :].astype(np.float64) - cent[None].astype(np.float64)) ** 2).sum(2)
        return d2.min(1).sum()

    c1, c8 = T.kmeans_l2(x, 12, 1, 5), T.kmeans_l2(x, 12, 8, 5)
    assert objective(c8) < objective(c1)
    assert np.array_equal(c8, T.kmeans_l2(x, 12, 8, 5))            # deterministic
    # the assignment is the L2 nearest centroid (up to float rounding of exact ties: none here)
    a = T.assign_l2(x, c8, 3)
    d2 = ((x[:, None, :].astype(np.float64) - c8[None].astype(np.float64)) ** 2).sum(2)
    assert (a == d2.argmin(1)).mean() > 0.999


def test_cluster_means_and_norms(oracle):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((500, 8)).astype(np.float32)
    a = rng.integers(0, 5, 500).astype(np.int32)
    a[a == 3] = 2                                                   # cluster 3 is empty
    c = np.full((5, 8), 7.0, np.float32)
    cnt = oracle.cluster_means(x, a, c)
    assert cnt.tolist() == np.bincount(a, minlength=5).tolist() and cnt[3] == 0
    assert (c[3] == 7.0).all()
    for j in (0, 1, 2, 4):
        assert np.allclose(c[j], x[a == j].astype(np.float64).mean(0), rtol=1e-5, atol=1e-6)
    assert np.allclose(oracle.neg_half_sqnorm(x), -0.5 * (x.astype(np.float64) ** 2).sum(1), rtol=1e-6)


def test_train_ivfpq_shapes(oracle):
    from oracle import train_oracle as T
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2000, 32)).astype(np.float32)
    cent, cb = T.train_ivfpq(x, 16, 4, True, niter=3)
    assert cent.shape == (16, 32) and cb.shape == (4, 256, 8)
    assert np.isfinite(cent).all() and np.isfinite(cb).all()


def test_spherical_kmeans_restatement(oracle):
    """ClusteringParameters.spherical: unit-norm centroids after every update, assignment by inner product; on unit-norm
    data it is still k-means (the cosine objective rises with the iterations) and deterministic."""
    from oracle import train_oracle as T
    rng = np.random.default_rng(3)
    c = rng.standard_normal((12, 16)).astype(np.float32)
    x = (c[rng.integers(0, 12, 3000)] + 0.3 * rng.standard_normal((3000, 16))).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)

    def objective(cent):
        return (x.astype(np.float64) @ cent.astype(np.float64).T).max(1).sum()

    c1, c8 = T.kmeans_l2(x, 12, 1, 5, spherical=True), T.kmeans_l2(x, 12, 8, 5, spherical=True)
    assert objective(c8) > objective(c1)
    assert np.allclose(np.linalg.norm(c8.astype(np.float64), axis=1), 1.0, atol=1e-6)
    assert np.array_equal(c8, T.kmeans_l2(x, 12, 8, 5, spherical=True))
    assert not np.array_equal(c8, T.kmeans_l2(x, 12, 8, 5))
    z = np.zeros((2, 4), np.float32)
    z[1] = [3, 0, 4, 0]
    T.renorm_rows(z)                                                 # a zero row stays zero (faiss: nr > 0 only)
    assert (z[0] == 0).all() and np.allclose(z[1], [0.6, 0, 0.8, 0])
