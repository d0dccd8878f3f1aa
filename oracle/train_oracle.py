"""CPU restatement of IndexIVFPQ.train as the product implements it
(abstracts-search_amd/_train.py; reference call site Makefile:39 `sidecar-search index train`,
README.md:60 `-c 65536`): Lloyd k-means for the coarse centroids, then one 256-codeword
k-means per sub-quantiser on the residuals.

TEST INFRASTRUCTURE ONLY (imported by tests/).  PARITY UNPINNED against faiss: faiss's
Clustering object is absent here, and it differs in details this file does not claim
(faiss splits the largest cluster to refill an empty one; here an empty cluster is re-seeded
from a random training point).  `spherical` restates faiss's ClusteringParameters.spherical (centroids
renormalised after every update, assignment by inner product) from its published source [PRIOR].  What IS pinned, bit for bit, is the product against this
file: same seeded draws, and every arithmetic step in a fixed order --

  assignment   arg max over j of  fmaf-chain(<x, c_j>) + 1 * (-0.5 * fmaf-chain(<c_j, c_j>))
               (= arg min |x - c_j|^2; ties: smallest j), evaluated as one ascending-k chain
               over the augmented vectors [x, 1, 0...] . [c_j, -|c_j|^2/2, 0...]
  update       mean of the members in ascending row order, sequential f32 adds, one division
  PQ encode    nearest codeword by the l2sqr chain (ties: smallest index)
"""
from __future__ import annotations

import numpy as np
import torch

from . import ivfpq_oracle as O


def sample(x: np.ndarray, nmax: int, seed: int) -> np.ndarray:
    n = x.shape[0]
    if n > nmax:
        sel = np.sort(np.random.default_rng(seed).choice(n, nmax, replace=False))
        x = x[sel]
    return np.ascontiguousarray(x, np.float32)


def assign_l2(x: np.ndarray, c: np.ndarray, pad: int) -> np.ndarray:
    n, k = x.shape[0], c.shape[0]
    xa = np.concatenate([x, np.ones((n, 1), np.float32), np.zeros((n, pad), np.float32)], 1)
    ca = np.concatenate([c, O.neg_half_sqnorm(c)[:, None], np.zeros((k, pad), np.float32)], 1)
    return O.flat_ip(xa, ca, 1)[1][:, 0].astype(np.int32)


def renorm_rows(c: np.ndarray) -> None:
    """faiss fvec_renorm_L2 (utils/distances.cpp [PRIOR]: `nr = fvec_norm_L2sqr(xi); if (nr > 0) xi *= 1.0 / sqrtf(nr)`), in place:
    |row|^2 by the ascending-k fmaf chain, the reciprocal root in double rounded to f32, one f32 multiply per element."""
    nr = (O.neg_half_sqnorm(c) * np.float32(-2.0)).astype(np.float32)
    inv = np.ones_like(nr)
    pos = nr > 0
    inv[pos] = (1.0 / np.sqrt(nr[pos]).astype(np.float64)).astype(np.float32)
    c *= inv[:, None]


def kmeans_l2(x: np.ndarray, k: int, niter: int, seed: int, spherical: bool = False) -> np.ndarray:
    """spherical = faiss ClusteringParameters.spherical [PRIOR: Clustering::post_process_centroids renormalises the centroids
    after every update; index_factory sets it for METRIC_INNER_PRODUCT]: unit-norm centroids, assignment by largest inner
    product (faiss assigns with the quantiser being trained, an IndexFlatIP)."""
    n, d = x.shape
    g = torch.Generator(device="cpu").manual_seed(seed)
    perm = torch.randperm(n, generator=g)[:k].numpy()
    c = x[perm].copy()
    if n <= k:
        c = np.concatenate([c, c[torch.randint(0, max(n, 1), (k - c.shape[0],), generator=g).numpy()]])
    c = np.ascontiguousarray(c)
    if spherical:
        renorm_rows(c)
    pad = 127 if d % 128 == 0 and k >= 8192 else 3
    for _ in range(niter):
        a = O.flat_ip(x, c, 1)[1][:, 0].astype(np.int32) if spherical else assign_l2(x, c, pad)
        cnt = O.cluster_means(x, a, c)
        ne = int((cnt == 0).sum())
        if ne:
            idx = torch.randint(0, n, (ne,), generator=g).numpy()
            c[cnt == 0] = x[idx]
        if spherical:
            renorm_rows(c)
    return np.ascontiguousarray(c)


def train_ivfpq(x: np.ndarray, nlist: int, M: int, by_residual: bool = True, niter: int = 25,
                max_points_per_centroid: int = 256, seed: int = 1234, spherical: bool = False):
    x = np.ascontiguousarray(x, np.float32)
    d = x.shape[1]
    dsub = d // M
    cent = kmeans_l2(sample(x, max_points_per_centroid * nlist, seed), nlist, niter, seed, spherical)
    xp = sample(x, max_points_per_centroid * 256, seed + 1)
    if by_residual:
        xp = np.ascontiguousarray(xp - cent[O.flat_ip(xp, cent, 1)[1][:, 0]])
    n = xp.shape[0]
    g = torch.Generator(device="cpu").manual_seed(seed + 2)
    init = torch.randperm(n, generator=g)[:256]
    if init.numel() < 256:
        init = torch.cat([init, torch.randint(0, n, (256 - init.numel(),), generator=g)])
    cb = np.ascontiguousarray(xp[init.numpy()].reshape(256, M, dsub).transpose(1, 0, 2))      # [M, 256, dsub]
    rows = xp.reshape(n * M, dsub)
    zero = np.zeros((1, d), np.float32)
    offs = (np.arange(M, dtype=np.int32) * 256)[None, :]
    for _ in range(niter):
        codes = O.encode(xp, zero, cb, by_residual=False)[1]                                # nearest codeword per sub-vector
        a = (codes.astype(np.int32) + offs).reshape(-1)
        flat = cb.reshape(M * 256, dsub)
        O.cluster_means(rows, a, flat)
    return cent, cb


def train_ivfpq_l2(x: np.ndarray, nlist: int, M: int, by_residual: bool = True, niter: int = 25,
                   max_points_per_centroid: int = 256, seed: int = 1234):
    """METRIC_L2: as train_ivfpq, with the residuals taken to the L2-nearest centroid."""
    x = np.ascontiguousarray(x, np.float32)
    d = x.shape[1]
    dsub = d // M
    cent = kmeans_l2(sample(x, max_points_per_centroid * nlist, seed), nlist, niter, seed)
    xp = sample(x, max_points_per_centroid * 256, seed + 1)
    if by_residual:
        xp = np.ascontiguousarray(xp - cent[assign_l2(xp, cent, 3)])
    n = xp.shape[0]
    g = torch.Generator(device="cpu").manual_seed(seed + 2)
    init = torch.randperm(n, generator=g)[:256]
    if init.numel() < 256:
        init = torch.cat([init, torch.randint(0, n, (256 - init.numel(),), generator=g)])
    cb = np.ascontiguousarray(xp[init.numpy()].reshape(256, M, dsub).transpose(1, 0, 2))
    rows = xp.reshape(n * M, dsub)
    zero = np.zeros((1, d), np.float32)
    offs = (np.arange(M, dtype=np.int32) * 256)[None, :]
    for _ in range(niter):
        codes = O.encode(xp, zero, cb, by_residual=False)[1]
        O.cluster_means(rows, (codes.astype(np.int32) + offs).reshape(-1), cb.reshape(M * 256, dsub))
    return cent, cb
