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
