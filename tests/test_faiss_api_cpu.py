"""Host-side pieces of the faiss mirror that need no GPU: normalize_L2, the faiss-gpu cloning
entry points (identities here: the index is born on the GPU), per-call search parameter objects."""
import importlib

import numpy as np
import pytest
import torch

faiss = importlib.import_module("abstracts_search_amd.faiss")


def test_normalize_l2_in_place():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((7, 16)).astype(np.float32)
    x[3] = 0
    want = x / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-30)
    want[3] = 0
    y = x
    assert faiss.normalize_L2(x) is None and y is x
    assert np.allclose(x, want, atol=1e-6) and (x[3] == 0).all()
    t = torch.from_numpy(rng.standard_normal((4, 8)).astype(np.float32))
    t[1] = 0
    faiss.normalize_L2(t)
    assert torch.allclose(t.norm(dim=1), torch.tensor([1.0, 0.0, 1.0, 1.0]), atol=1e-6)
    for bad in (x.astype(np.float64), x[:, ::2], x[0]):
        with pytest.raises(TypeError):
            faiss.normalize_L2(bad)


def test_gpu_cloning_entry_points_are_identities():
    class Dummy:
        device = 0

    idx = Dummy()
    res = faiss.StandardGpuResources()
    res.noTempMemory()
    co = faiss.GpuMultipleClonerOptions()
    co.shard = True
    assert faiss.index_cpu_to_gpu(res, 0, idx, co) is idx
    assert faiss.index_cpu_to_all_gpus(idx, co) is idx
    assert faiss.index_gpu_to_cpu(idx) is idx
    with pytest.raises(NotImplementedError):
        faiss.index_cpu_to_gpu(res, 1, idx)


def test_search_parameter_objects():
    p = faiss.SearchParametersIVF(nprobe=32)
    assert p.nprobe == 32 and p.max_codes == 0 and p.sel is None
    r = faiss.IndexRefineSearchParameters(k_factor=4, base_index_params=p)
    assert r.k_factor == 4.0 and r.base_index_params is p


def test_clustering_parameters_mirror_faiss_defaults():
    """faiss: ClusteringParameters.niter = 25, max_points_per_centroid = 256, seed = 1234; Level1Quantizer (index.cp) sets
    niter = 10; the product quantizer trains with its own pq.cp (25).  Fields train() would silently ignore raise."""
    import abstracts_search_amd.faiss as faiss
    cp = faiss.ClusteringParameters()
    assert (cp.niter, cp.nredo, cp.max_points_per_centroid, cp.min_points_per_centroid, cp.seed) == (25, 1, 256, 39, 1234)
    assert not (cp.spherical or cp.int_centroids or cp.update_index or cp.frozen_centroids or cp.verbose)
    cp.check_supported("x")
    c2 = faiss.ClusteringParameters()
    c2.spherical = True                                           # implemented for the coarse quantiser (round 6): accepted
    c2.check_supported("train")
    for field in ("int_centroids", "update_index", "frozen_centroids"):
        c2 = faiss.ClusteringParameters()
        setattr(c2, field, True)
        with pytest.raises(NotImplementedError, match=field):
            c2.check_supported("train")
    c2 = faiss.ClusteringParameters()
    c2.nredo = 3
    with pytest.raises(NotImplementedError, match="nredo"):
        c2.check_supported("train")
    pq = faiss._PQ(64, 8, 8)
    assert pq.cp.niter == 25 and pq.cp is not faiss._PQ(64, 8, 8).cp
