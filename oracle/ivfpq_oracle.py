"""ctypes front end of oracle/ivfpq_oracle.c plus a numpy brute-force checker.

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py, never by the product package.

PARITY UNPINNED: see the header of ivfpq_oracle.c.  The reference calls this
arithmetic through `sidecar-search index {train,fill,tune}` (reference
Makefile:39,25,32); the arithmetic itself is faiss's, which is absent here.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libivfpq_oracle.so")
_lib = None

FLT_MAX = np.float32(np.finfo(np.float32).max)


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc, seconds)."""
    src = os.path.join(_HERE, "ivfpq_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libivfpq_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        so = os.environ.get("MI_ORACLE_SO") or _SO        # tests/test_sanitizers.py: an ASan + UBSan build of the same source
        if so == _SO and not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(so)
        _lib.oracle_num_threads.restype = ctypes.c_int
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def num_threads() -> int:
    return int(lib().oracle_num_threads())


def flat_ip(q, base, k):
    """IndexFlatIP.search restatement -> (D[nq,k] f32, I[nq,k] i64)."""
    q, base = _f32(q), _f32(base)
    nq, d = q.shape
    D = np.empty((nq, k), np.float32)
    I = np.empty((nq, k), np.int64)
    lib().oracle_flat_ip(ctypes.c_int64(nq), ctypes.c_int(d), _p(q, ctypes.c_float),
                         ctypes.c_int64(base.shape[0]), _p(base, ctypes.c_float),
                         ctypes.c_int(k), _p(D, ctypes.c_float), _p(I, ctypes.c_int64))
    return D, I


def lut(q1, codebook):
    """ADC table of one query: codebook [M,ksub,dsub] -> [M,ksub]."""
    q1, codebook = _f32(q1), _f32(codebook)
    M, ksub, dsub = codebook.shape
    out = np.empty((M, ksub), np.float32)
    lib().oracle_lut(ctypes.c_int(M * dsub), ctypes.c_int(M), ctypes.c_int(ksub),
                     _p(q1, ctypes.c_float), _p(codebook, ctypes.c_float),
                     _p(out, ctypes.c_float))
    return out


def encode(x, centroids, codebook, by_residual=True):
    """Index.add arithmetic -> (list_no[n] i32, codes[n,M] u8)."""
    x, centroids, codebook = _f32(x), _f32(centroids), _f32(codebook)
    n, d = x.shape
    M, ksub, _ = codebook.shape
    list_no = np.empty(n, np.int32)
    codes = np.empty((n, M), np.uint8)
    lib().oracle_encode(ctypes.c_int64(n), ctypes.c_int(d), _p(x, ctypes.c_float),
                        ctypes.c_int(centroids.shape[0]), _p(centroids, ctypes.c_float),
                        ctypes.c_int(M), ctypes.c_int(ksub), _p(codebook, ctypes.c_float),
                        ctypes.c_int(int(by_residual)), _p(list_no, ctypes.c_int32),
                        _p(codes, ctypes.c_uint8))
    return list_no, codes


def build_lists(list_no, codes, ids, nlist):
    """Group (codes, ids) by list, keeping insertion order inside a list
    (faiss ArrayInvertedLists appends) -> (list_off[nlist+1], codes, ids)."""
    order = np.argsort(list_no, kind="stable")
    counts = np.bincount(list_no, minlength=nlist)
    off = np.zeros(nlist + 1, np.int64)
    np.cumsum(counts, out=off[1:])
    return off, np.ascontiguousarray(codes[order]), np.ascontiguousarray(ids[order].astype(np.int64))


def search(q, centroids, codebook, list_off, codes, ids, nprobe, k, by_residual=True,
           return_coarse=False):
    """Index.search restatement -> (D[nq,k] f32, I[nq,k] i64)."""
    q, centroids, codebook = _f32(q), _f32(centroids), _f32(codebook)
    nq, d = q.shape
    M, ksub, _ = codebook.shape
    nlist = centroids.shape[0]
    nprobe = min(int(nprobe), nlist)
    list_off = np.ascontiguousarray(list_off, np.int64)
    codes = np.ascontiguousarray(codes, np.uint8)
    ids = np.ascontiguousarray(ids, np.int64)
    D = np.empty((nq, k), np.float32)
    I = np.empty((nq, k), np.int64)
    cI = np.empty((nq, nprobe), np.int32)
    cD = np.empty((nq, nprobe), np.float32)
    lib().oracle_search(ctypes.c_int64(nq), ctypes.c_int(d), _p(q, ctypes.c_float),
                        ctypes.c_int(nlist), _p(centroids, ctypes.c_float),
                        ctypes.c_int(M), ctypes.c_int(ksub), _p(codebook, ctypes.c_float),
                        ctypes.c_int(int(by_residual)), _p(list_off, ctypes.c_int64),
                        _p(codes, ctypes.c_uint8), _p(ids, ctypes.c_int64),
                        ctypes.c_int(nprobe), ctypes.c_int(k), _p(D, ctypes.c_float),
                        _p(I, ctypes.c_int64), _p(cI, ctypes.c_int32), _p(cD, ctypes.c_float))
    if return_coarse:
        return D, I, cI, cD
    return D, I


def search_preassigned(q, codebook, list_off, codes, ids, coarse_I, coarse_D, k, by_residual=True):
    """IndexIVF.search_preassigned restatement -> (D, I)."""
    q, codebook = _f32(q), _f32(codebook)
    nq, d = q.shape
    M, ksub, _ = codebook.shape
    coarse_I = np.ascontiguousarray(coarse_I, np.int32)
    coarse_D = _f32(coarse_D)
    nprobe = coarse_I.shape[1]
    list_off = np.ascontiguousarray(list_off, np.int64)
    codes = np.ascontiguousarray(codes, np.uint8)
    ids = np.ascontiguousarray(ids, np.int64)
    D = np.empty((nq, k), np.float32)
    I = np.empty((nq, k), np.int64)
    lib().oracle_search_preassigned(ctypes.c_int64(nq), ctypes.c_int(d), _p(q, ctypes.c_float), ctypes.c_int(M),
                                    ctypes.c_int(ksub), _p(codebook, ctypes.c_float), ctypes.c_int(int(by_residual)),
                                    _p(list_off, ctypes.c_int64), _p(codes, ctypes.c_uint8), _p(ids, ctypes.c_int64),
                                    ctypes.c_int(nprobe), _p(coarse_I, ctypes.c_int32), _p(coarse_D, ctypes.c_float),
                                    ctypes.c_int(k), _p(D, ctypes.c_float), _p(I, ctypes.c_int64))
    return D, I


def rerank(q, base, cand_I, k):
    """IndexRefineFlat re-ranking restatement: exact scores of the candidate ids
    (negative = empty), k best under (score desc, id asc) -> (D[nq,k], I[nq,k])."""
    q, base = _f32(q), _f32(base)
    cand_I = np.ascontiguousarray(cand_I, np.int64)
    nq, d = q.shape
    kc = cand_I.shape[1]
    D = np.empty((nq, k), np.float32)
    I = np.empty((nq, k), np.int64)
    lib().oracle_rerank(ctypes.c_int64(nq), ctypes.c_int(d), _p(q, ctypes.c_float), _p(base, ctypes.c_float),
                        ctypes.c_int(kc), _p(cand_I, ctypes.c_int64), ctypes.c_int(k),
                        _p(D, ctypes.c_float), _p(I, ctypes.c_int64))
    return D, I


def sq8_train(x):
    """faiss ScalarQuantizer(QT_8bit).train restatement -> trained [2 d] = vmin | vdiff."""
    x = _f32(x)
    n, d = x.shape
    tr = np.empty(2 * d, np.float32)
    lib().oracle_sq8_train(ctypes.c_int64(n), ctypes.c_int(d), _p(x, ctypes.c_float), _p(tr, ctypes.c_float))
    return tr


def sq8_encode(x, trained):
    x, trained = _f32(x), _f32(trained)
    n, d = x.shape
    codes = np.empty((n, d), np.uint8)
    lib().oracle_sq8_encode(ctypes.c_int64(n), ctypes.c_int(d), _p(x, ctypes.c_float), _p(trained, ctypes.c_float),
                            _p(codes, ctypes.c_uint8))
    return codes


def sq8_decode(codes, trained):
    codes, trained = np.ascontiguousarray(codes, np.uint8), _f32(trained)
    n, d = codes.shape
    x = np.empty((n, d), np.float32)
    lib().oracle_sq8_decode(ctypes.c_int64(n), ctypes.c_int(d), _p(codes, ctypes.c_uint8), _p(trained, ctypes.c_float),
                            _p(x, ctypes.c_float))
    return x


def rerank_sq8(q, codes, trained, cand_I, k):
    """IndexRefine over an IndexScalarQuantizer(QT_8bit) store: scores of the candidate ids against the
    decoded rows, k best under (score desc, id asc)."""
    q, trained = _f32(q), _f32(trained)
    codes = np.ascontiguousarray(codes, np.uint8)
    cand_I = np.ascontiguousarray(cand_I, np.int64)
    nq, d = q.shape
    kc = cand_I.shape[1]
    D = np.empty((nq, k), np.float32)
    I = np.empty((nq, k), np.int64)
    lib().oracle_rerank_sq8(ctypes.c_int64(nq), ctypes.c_int(d), _p(q, ctypes.c_float), _p(codes, ctypes.c_uint8),
                            _p(trained, ctypes.c_float), ctypes.c_int(kc), _p(cand_I, ctypes.c_int64), ctypes.c_int(k),
                            _p(D, ctypes.c_float), _p(I, ctypes.c_int64))
    return D, I


def merge(D_parts, I_parts):
    """k-way merge of per-shard results [nparts,nq,k] -> (D[nq,k], I[nq,k])."""
    D_parts = _f32(D_parts)
    I_parts = np.ascontiguousarray(I_parts, np.int64)
    nparts, nq, k = D_parts.shape
    D = np.empty((nq, k), np.float32)
    I = np.empty((nq, k), np.int64)
    lib().oracle_merge(ctypes.c_int(nparts), ctypes.c_int64(nq), ctypes.c_int(k),
                       _p(D_parts, ctypes.c_float), _p(I_parts, ctypes.c_int64),
                       _p(D, ctypes.c_float), _p(I, ctypes.c_int64))
    return D, I


# ----------------------------------------------------------------------
# numpy brute force: "decode every code, dot, argsort" in float64.  It shares
# no code and no evaluation order with the C restatement; it checks the
# *algorithm* (what is summed), the C file fixes the *order* (how).
# ----------------------------------------------------------------------

def brute_force_search(q, centroids, codebook, list_off, codes, ids, nprobe, k,
                       by_residual=True):
    q = np.asarray(q, np.float64)
    centroids = np.asarray(centroids, np.float64)
    cb = np.asarray(codebook, np.float64)
    M, ksub, dsub = cb.shape
    nlist = centroids.shape[0]
    nprobe = min(int(nprobe), nlist)
    list_of = np.repeat(np.arange(nlist), np.diff(list_off))
    # decoded residual (or vector) of every stored code
    dec = cb[np.arange(M)[None, :], codes.astype(np.int64)].reshape(len(codes), M * dsub)
    D = np.full((q.shape[0], k), -np.inf)
    I = np.full((q.shape[0], k), -1, np.int64)
    coarse = q @ centroids.T
    for qi in range(q.shape[0]):
        probe = np.lexsort((np.arange(nlist), -coarse[qi]))[:nprobe]
        sel = np.isin(list_of, probe)
        if not sel.any():
            continue
        sc = dec[sel] @ q[qi]
        if by_residual:
            sc = sc + coarse[qi][list_of[sel]]
        sid = ids[sel]
        o = np.lexsort((sid, -sc))[:k]
        D[qi, :len(o)] = sc[o]
        I[qi, :len(o)] = sid[o]
    return D, I


def cluster_means(x, assign, centroids):
    """In-place k-means update (mi_cluster_means): returns the member counts."""
    x = _f32(x)
    assign = np.ascontiguousarray(assign, np.int32)
    assert centroids.dtype == np.float32 and centroids.flags.c_contiguous
    k, d = centroids.shape
    counts = np.empty(k, np.int32)
    lib().oracle_cluster_means(ctypes.c_int64(x.shape[0]), ctypes.c_int(d), _p(x, ctypes.c_float),
                               _p(assign, ctypes.c_int32), ctypes.c_int(k), _p(centroids, ctypes.c_float),
                               _p(counts, ctypes.c_int32))
    return counts


def neg_half_sqnorm(x):
    x = _f32(x)
    out = np.empty(x.shape[0], np.float32)
    lib().oracle_neg_half_sqnorm(ctypes.c_int64(x.shape[0]), ctypes.c_int(x.shape[1]), _p(x, ctypes.c_float),
                                 _p(out, ctypes.c_float))
    return out


# ---- METRIC_L2 (see the header of the L2 section in ivfpq_oracle.c) ----
def flat_l2(q, base, k):
    """IndexFlatL2.search restatement -> (D squared distances ascending, I)."""
    q, base = _f32(q), _f32(base)
    nq, d = q.shape
    D = np.empty((nq, k), np.float32)
    I = np.empty((nq, k), np.int64)
    lib().oracle_flat_l2(ctypes.c_int64(nq), ctypes.c_int(d), _p(q, ctypes.c_float), ctypes.c_int64(base.shape[0]),
                         _p(base, ctypes.c_float), ctypes.c_int(k), _p(D, ctypes.c_float), _p(I, ctypes.c_int64))
    return D, I


def encode_l2(x, centroids, codebook, by_residual=True):
    """Index.add arithmetic, METRIC_L2 -> (list_no i32, codes u8 [n,M], tnorm f32 [n])."""
    x, centroids, codebook = _f32(x), _f32(centroids), _f32(codebook)
    n, d = x.shape
    M, ksub, _ = codebook.shape
    list_no = np.empty(n, np.int32)
    codes = np.empty((n, M), np.uint8)
    tnorm = np.empty(n, np.float32)
    lib().oracle_encode_l2(ctypes.c_int64(n), ctypes.c_int(d), _p(x, ctypes.c_float), ctypes.c_int(centroids.shape[0]),
                           _p(centroids, ctypes.c_float), ctypes.c_int(M), ctypes.c_int(ksub), _p(codebook, ctypes.c_float),
                           ctypes.c_int(int(by_residual)), _p(list_no, ctypes.c_int32), _p(codes, ctypes.c_uint8),
                           _p(tnorm, ctypes.c_float))
    return list_no, codes, tnorm


def build_lists_l2(list_no, codes, ids, tnorm, nlist):
    order = np.argsort(list_no, kind="stable")
    off, c, i = build_lists(list_no, codes, ids, nlist)
    return off, c, i, np.ascontiguousarray(tnorm[order])


def search_l2(q, centroids, codebook, list_off, codes, ids, tnorm, nprobe, k, by_residual=True, return_coarse=False):
    q, centroids, codebook = _f32(q), _f32(centroids), _f32(codebook)
    nq, d = q.shape
    M, ksub, _ = codebook.shape
    nlist = centroids.shape[0]
    npb = min(nprobe, nlist)
    D = np.empty((nq, k), np.float32)
    I = np.empty((nq, k), np.int64)
    cI = np.empty((nq, npb), np.int32)
    cD = np.empty((nq, npb), np.float32)
    list_off = np.ascontiguousarray(list_off, np.int64)
    codes = np.ascontiguousarray(codes, np.uint8)
    ids = np.ascontiguousarray(ids, np.int64)
    tnorm = _f32(tnorm)
    lib().oracle_search_l2(ctypes.c_int64(nq), ctypes.c_int(d), _p(q, ctypes.c_float), ctypes.c_int(nlist),
                           _p(centroids, ctypes.c_float), ctypes.c_int(M), ctypes.c_int(ksub), _p(codebook, ctypes.c_float),
                           ctypes.c_int(int(by_residual)), _p(list_off, ctypes.c_int64), _p(codes, ctypes.c_uint8),
                           _p(ids, ctypes.c_int64), _p(tnorm, ctypes.c_float), ctypes.c_int(npb), ctypes.c_int(k),
                           _p(D, ctypes.c_float), _p(I, ctypes.c_int64), _p(cI, ctypes.c_int32), _p(cD, ctypes.c_float))
    if return_coarse:
        return D, I, cI, cD
    return D, I


def brute_force_l2(q, centroids, codebook, list_off, codes, by_residual=True):
    """float64: exact squared distances |q - x^|^2 to every decoded vector, in storage order
    (independent check of the expansion search_l2 evaluates)."""
    centroids, codebook = np.asarray(centroids, np.float64), np.asarray(codebook, np.float64)
    M = codebook.shape[0]
    sizes = np.diff(list_off)
    dec = np.concatenate([codebook[m][codes[:, m]] for m in range(M)], axis=1)
    if by_residual:
        dec = dec + centroids[np.repeat(np.arange(len(sizes)), sizes)]
    qq = np.asarray(q, np.float64)
    return ((qq[:, None, :] - dec[None]) ** 2).sum(2)
