"""Synthetic corpora for BASELINE.json's configs.

Row i of the corpus is a pure function of (seed, i): x_i = normalise(c[z_i] +
sigma * g_i) with `ncentres` unit-norm centres, so any rank can regenerate any
row range without storing the corpus.  numpy (Philox) version for tests and
small sizes; a torch/CUDA version for bench-sized corpora (same distribution,
different stream).
"""
from __future__ import annotations

import functools

import numpy as np


@functools.lru_cache(maxsize=4)
def _centres(d: int, ncentres: int, seed: int = 99) -> np.ndarray:
    rng = np.random.Generator(np.random.Philox(seed))
    c = rng.standard_normal((ncentres, d), dtype=np.float32)
    c /= np.linalg.norm(c, axis=1, keepdims=True)
    return c


def sigma_for_cosine(d: int, cos: float = 0.7) -> float:
    """sigma such that cos(x_i, centre) ~= `cos`: |c|=1, |sigma g| ~= sigma sqrt(d)."""
    return float(np.sqrt((1.0 / (cos * cos) - 1.0) / d))


def corpus_rows(lo: int, hi: int, d: int = 1024, ncentres: int = 16384, seed: int = 1234,
                cos: float = 0.7) -> np.ndarray:
    """Rows [lo, hi) of the clustered unit-norm corpus (float32)."""
    c = _centres(d, ncentres)
    sig = sigma_for_cosine(d, cos)
    out = np.empty((hi - lo, d), np.float32)
    B = 8192
    for b0 in range(lo - lo % B, hi, B):  # blocks keyed by (seed, block) -> any range is reproducible
        rng = np.random.Generator(np.random.Philox(key=seed, counter=[0, 0, 0, b0 // B]))
        z = rng.integers(0, ncentres, B)
        g = rng.standard_normal((B, d), dtype=np.float32)
        x = c[z] + sig * g
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        s, e = max(lo, b0), min(hi, b0 + B)
        out[s - lo:e - lo] = x[s - b0:e - b0]
    return out


def queries_from(x: np.ndarray, nq: int, seed: int = 4321, cos: float = 0.7) -> np.ndarray:
    """q_j = normalise(x_pi(j) + 0.3 sigma g'_j)."""
    d = x.shape[1]
    rng = np.random.Generator(np.random.Philox(seed))
    pick = rng.integers(0, x.shape[0], nq)
    q = x[pick] + 0.3 * sigma_for_cosine(d, cos) * rng.standard_normal((nq, d), dtype=np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q.astype(np.float32)


@functools.lru_cache(maxsize=4)
def _centres_cuda(d: int, ncentres: int, device: int):
    import torch
    return torch.from_numpy(_centres(d, ncentres)).to(torch.device("cuda", device))


def corpus_cuda(n: int, d: int = 1024, ncentres: int = 16384, seed: int = 1234, cos: float = 0.7,
                device: int = 0, row0: int = 0):
    """Bench-sized corpus generated on the GPU with torch's Philox generator
    (rows row0 .. row0+n of a stream keyed by (seed, block))."""
    import torch
    dev = torch.device("cuda", device)
    c = _centres_cuda(d, ncentres, device)
    sig = sigma_for_cosine(d, cos)
    out = torch.empty((n, d), dtype=torch.float32, device=dev)
    B = 65536
    assert row0 % B == 0
    for b0 in range(0, n, B):
        g = torch.Generator(device=dev).manual_seed(seed * 1000003 + (row0 + b0) // B)
        m = min(B, n - b0)
        z = torch.randint(0, ncentres, (B,), generator=g, device=dev)[:m]
        x = c[z] + sig * torch.randn((B, d), generator=g, device=dev)[:m]
        out[b0:b0 + m] = x / x.norm(dim=1, keepdim=True)
    return out


def queries_cuda(x, nq: int, seed: int = 4321, cos: float = 0.7):
    import torch
    g = torch.Generator(device=x.device).manual_seed(seed)
    pick = torch.randint(0, x.shape[0], (nq,), generator=g, device=x.device)
    q = x[pick] + 0.3 * sigma_for_cosine(x.shape[1], cos) * torch.randn((nq, x.shape[1]), generator=g, device=x.device)
    return (q / q.norm(dim=1, keepdim=True)).contiguous()
