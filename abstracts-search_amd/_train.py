"""IndexIVFPQ.train (setup; reference Makefile:39 `index train`): k-means for
the nlist coarse centroids and for the M sub-quantiser codebooks.

Not on the timed path.  The assignment steps -- the only
heavy part -- run on the library's own kernels through the C ABI
(``mi_ip_assign``: exact-f32 MFMA GEMM + arg max; ``mi_pq_encode``: nearest
codeword per sub-vector).  The update step is the library's deterministic
``mi_cluster_means`` (a cluster's members summed in ascending row order), so
training is bit-reproducible and restated by oracle/train_oracle.py.
"""
from __future__ import annotations

import ctypes
from ctypes import c_void_p

import numpy as np


def _lib():
    from .faiss import _Lib
    return _Lib.get()


def _check(rc):
    from .faiss import _check as c
    c(rc)


def _to_device_sample(x, nmax: int, seed: int, device: int):
    """Random subsample (faiss: max_points_per_centroid * k) -> CUDA f32 tensor."""
    import torch
    dev = torch.device("cuda", device)
    n = x.shape[0]
    if n > nmax:
        rng = np.random.default_rng(seed)
        sel = np.sort(rng.choice(n, nmax, replace=False))
        if type(x).__module__.startswith("torch"):
            x = x[torch.as_tensor(sel, device=x.device)]
        else:
            x = x[sel]
    if type(x).__module__.startswith("torch"):
        return x.to(dev, torch.float32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)


def _assign_ip(x, c, device):
    """arg max_j <x_i, c_j> (ties: smallest j) -> int32 CUDA tensor."""
    import torch
    n = x.shape[0]
    a = torch.empty(n, dtype=torch.int32, device=x.device)
    stream = c_void_p(torch.cuda.current_stream().cuda_stream)
    _check(_lib().mi_ip_assign(device, n, c_void_p(x.data_ptr()), c.shape[0], c_void_p(c.data_ptr()),
                               x.shape[1], c_void_p(a.data_ptr()), c_void_p(0), stream))
    return a


def _cluster_means(x, assign, cent, device):
    """In-place k-means update on the library's deterministic kernel (mi_cluster_means):
    members summed in ascending row order; empty clusters keep their row.  Returns counts."""
    import torch
    k, d = cent.shape
    cnt = torch.empty(k, dtype=torch.int32, device=x.device)
    stream = c_void_p(torch.cuda.current_stream().cuda_stream)
    _check(_lib().mi_cluster_means(device, x.shape[0], c_void_p(x.data_ptr()), d, c_void_p(assign.data_ptr()), k,
                                   c_void_p(cent.data_ptr()), c_void_p(cnt.data_ptr()), stream))
    return cnt


def _neg_half_sqnorm(c, device):
    import torch
    out = torch.empty(c.shape[0], dtype=torch.float32, device=c.device)
    _check(_lib().mi_neg_half_sqnorm(device, c.shape[0], c_void_p(c.data_ptr()), c.shape[1], c_void_p(out.data_ptr()),
                                     c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out


def _renorm_rows(c, device):
    """faiss fvec_renorm_L2 on the centroid table, in place (ClusteringParameters.spherical): row *= 1 / sqrt(|row|^2) where
    |row|^2 > 0.  |row|^2 is the library's ascending-k fmaf chain (mi_neg_half_sqnorm x -2: exact); the k reciprocal roots
    are taken on the host in double (1.0 / sqrtf(nr), as faiss writes it) so that the oracle's numpy gives the same bits;
    the scaling is one f32 multiply per element."""
    import torch
    nr = (_neg_half_sqnorm(c, device) * -2.0).cpu().numpy()
    inv = np.ones_like(nr)
    pos = nr > 0
    inv[pos] = (1.0 / np.sqrt(nr[pos]).astype(np.float64)).astype(np.float32)
    c.mul_(torch.from_numpy(inv).to(c.device).unsqueeze(1))


def kmeans_l2(x, k: int, niter: int, seed: int, device: int, verbose: bool = False, spherical: bool = False):
    """Lloyd k-means, bit-reproducible (restated by oracle/train_oracle.py).
    arg min ||x-c||^2 = arg max (<x,c> - ||c||^2/2), evaluated by the inner-product kernel on
    vectors augmented with one column ([x, 1, 0...] . [c, -||c||^2/2, 0...]); the update step
    sums a cluster's members in ascending row order (mi_cluster_means).
    spherical (faiss ClusteringParameters.spherical; what faiss's index_factory sets for METRIC_INNER_PRODUCT [PRIOR]):
    the centroids are L2-normalised after the initial draw and after every update (faiss Clustering::post_process_centroids),
    and points go to the centroid of largest inner product (faiss assigns with the index being trained: an IndexFlatIP)."""
    import torch
    n, d = x.shape
    g = torch.Generator(device="cpu").manual_seed(seed)
    perm = torch.randperm(n, generator=g)[:k].to(x.device)
    c = x[perm].clone()
    if n <= k:  # degenerate: fewer points than centroids
        c = torch.cat([c, c[torch.randint(0, max(n, 1), (k - c.shape[0],), generator=g).to(x.device)]])
    c = c.contiguous()
    if spherical:
        _renorm_rows(c, device)
    # zero columns after the augmenting one add exact zeros to the end of every fmaf chain (the
    # scores do not change); padding to a multiple of 128 columns lets big problems take the
    # two-stage assignment (f16 MFMA scores + exact re-scoring, bit-identical arg max)
    pad = 127 if d % 128 == 0 and k >= 8192 else 3
    xa = None if spherical else torch.cat([x, torch.ones(n, 1, device=x.device), torch.zeros(n, pad, device=x.device)], 1).contiguous()
    for it in range(niter):
        if spherical:
            a = _assign_ip(x, c, device)
        else:
            ca = torch.cat([c, _neg_half_sqnorm(c, device).unsqueeze(1), torch.zeros(k, pad, device=x.device)], 1).contiguous()
            a = _assign_ip(xa, ca, device)
        cnt = _cluster_means(x, a, c, device)
        empty = cnt == 0
        ne = int(empty.sum())
        if ne:  # re-seed empty clusters from random points
            idx = torch.randint(0, n, (ne,), generator=g).to(x.device)
            c[empty] = x[idx]
        if spherical:
            _renorm_rows(c, device)
        if verbose:
            print(f"  kmeans iter {it}: {ne} empty clusters")
    return c.contiguous()


def _assign_list(x, c, device, metric):
    """List of every row: arg max <x, c> (inner product) or arg min |x - c|^2 (L2: the
    augmented inner product, as k-means assigns)."""
    import torch
    if metric == 0:
        return _assign_ip(x, c, device)
    n, k = x.shape[0], c.shape[0]
    xa = torch.cat([x, torch.ones(n, 1, device=x.device), torch.zeros(n, 3, device=x.device)], 1).contiguous()
    ca = torch.cat([c, _neg_half_sqnorm(c, device).unsqueeze(1), torch.zeros(k, 3, device=x.device)], 1).contiguous()
    return _assign_ip(xa, ca, device)


def train_ivfpq(x, nlist: int, M: int, by_residual: bool, cp, device: int, verbose: bool = False, centroids=None,
                metric: int = 0, pq_cp=None):
    """cp: the coarse quantiser's ClusteringParameters (index.cp); pq_cp: the product quantiser's (index.pq.cp; None: cp)."""
    pq_cp = pq_cp or cp
    import torch
    d = x.shape[1]
    dsub = d // M
    # coarse centroids (given: the caller's quantizer was already trained)
    if centroids is not None:
        cent = centroids.contiguous()
    else:
        xs = _to_device_sample(x, cp.max_points_per_centroid * nlist, cp.seed, device)
        if verbose:
            print(f"train: coarse k-means on {xs.shape[0]} points, k={nlist}")
        cent = kmeans_l2(xs, nlist, cp.niter, cp.seed, device, verbose, spherical=bool(getattr(cp, 'spherical', False)))
    # PQ codebooks on (residual) sub-vectors
    xp = _to_device_sample(x, pq_cp.max_points_per_centroid * 256, pq_cp.seed + 1, device)
    if by_residual:
        a = _assign_list(xp, cent, device, metric).long()
        xp = (xp - cent[a]).contiguous()
    n = xp.shape[0]
    g = torch.Generator(device="cpu").manual_seed(pq_cp.seed + 2)
    init = torch.randperm(n, generator=g)[:256]
    if init.numel() < 256:
        init = torch.cat([init, torch.randint(0, n, (256 - init.numel(),), generator=g)])
    cb = xp[init.to(xp.device)].view(256, M, dsub).permute(1, 0, 2).contiguous()  # [M,256,dsub]
    codes = torch.empty(n, M, dtype=torch.uint8, device=xp.device)
    stream = lambda: c_void_p(torch.cuda.current_stream().cuda_stream)
    offs = (torch.arange(M, device=xp.device, dtype=torch.int32) * 256).unsqueeze(0)
    rows = xp.view(n * M, dsub)                      # row i*M + m = sub-vector m of point i
    for it in range(pq_cp.niter):
        _check(_lib().mi_pq_encode(device, n, c_void_p(xp.data_ptr()), d, M, c_void_p(cb.data_ptr()),
                                   c_void_p(codes.data_ptr()), stream()))
        a = (codes.to(torch.int32) + offs).reshape(-1).contiguous()          # cluster = m * 256 + code
        _cluster_means(rows, a, cb.view(M * 256, dsub), device)              # in place; empty codewords keep their value
    torch.cuda.synchronize()
    return cent, cb
