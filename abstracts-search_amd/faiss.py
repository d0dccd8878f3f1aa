"""faiss-shaped Python mirror of the MI355X IVF-PQ index.

Drop-in for the subset of the faiss Python API that the reference reaches
through ``sidecar-search index train|fill|tune`` (reference Makefile:39,
Makefile:25, Makefile:32) and its query-time ``app.py`` (reference
README.md:28): ``index_factory``, ``IndexIVFPQ.{train,add,add_with_ids,search,
reset}``, ``nprobe``, ``ntotal``, ``is_trained``, ``IndexFlatIP``,
``write_index`` / ``read_index`` (faiss's file format), ``ParameterSpace`` /
``OperatingPoints`` / ``IntersectionCriterion`` (autotune.py: the `tune` step),
``IndexRefineFlat``, ``SearchParametersIVF`` / ``IndexRefineSearchParameters``
(``search(x, k, params=...)``).

Everything numeric happens in HIP kernels behind the C ABI of
``include/mi_ivfpq.h``; this file only checks arguments, moves pointers and
raises Python exceptions from C status codes.  There is no CPU fallback.

``search``/``add`` take float32 C-contiguous numpy arrays like faiss (results
are fresh numpy arrays), or torch CUDA tensors (results are CUDA tensors and
the call only enqueues work on the current stream -- the path bench.py times).
"""
from __future__ import annotations

import os
import ctypes
import re
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint8, c_void_p

import numpy as np

from . import _native, faiss_io
from .autotune import (AutoTuneCriterion, IntersectionCriterion, OneRecallAtRCriterion,  # noqa: F401
                       OperatingPoint, OperatingPoints, ParameterRange, ParameterSpace)

METRIC_INNER_PRODUCT = 0
METRIC_L2 = 1

_c_f32p = POINTER(c_float)


class _Lib:
    """Lazy binding of libmi_ivfpq.so (one per process)."""

    _lib = None

    @classmethod
    def get(cls):
        if cls._lib is None:
            lib = _native.load("ivfpq")
            lib.mi_last_error.restype = c_char_p
            v = c_void_p
            sigs = {
                "mi_device_count": [POINTER(c_int)],
                "mi_index_create": [c_int, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(v)],
                "mi_index_destroy": [v],
                "mi_index_set_coarse": [v, v],
                "mi_index_set_codebook": [v, v],
                "mi_index_get_coarse": [v, v],
                "mi_index_get_codebook": [v, v],
                "mi_index_is_trained": [v, POINTER(c_int)],
                "mi_index_ntotal": [v, POINTER(c_int64)],
                "mi_index_reset": [v],
                "mi_index_seal": [v],
                "mi_index_add": [v, c_int64, v, v],
                "mi_index_encode": [v, c_int64, v, v, v],
                "mi_index_add_codes": [v, c_int64, v, v, v],
                "mi_index_list_size": [v, c_int, POINTER(c_int64)],
                "mi_index_get_list": [v, c_int, v, v],
                "mi_index_list_sizes": [v, v],
                "mi_index_export_lists": [v, c_int, c_int, v, v],
                "mi_index_reserve": [v, c_int64],
                "mi_index_save": [v, c_char_p, c_char_p],
                "mi_index_load": [c_char_p, c_int, POINTER(v)],
                "mi_index_load_at": [c_char_p, c_int64, c_int, POINTER(v)],
                "mi_index_get_params": [v, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int),
                                        POINTER(c_int), POINTER(c_int), POINTER(c_int)],
                "mi_index_set_nprobe": [v, c_int],
                "mi_index_search": [v, c_int64, v, c_int, c_int, v, v, v],
                "mi_index_coarse_lut": [v, c_int64, v, c_int, v, v, v],
                "mi_index_profile_scan": [v, c_int, v, POINTER(c_double), POINTER(c_int64)],
                "mi_index_prune_stats": [v, v, c_int],
                "mi_index_search_candidates": [v, c_int64, v, c_int, c_int, v, v],
                "mi_index_coarse_slice": [v, c_int64, v, c_int, c_int, c_int, v, v, v],
                "mi_index_search_preassigned": [v, c_int64, v, c_int, c_int, v, v, v, v, v],
                "mi_merge_topk": [c_int, c_int, c_int64, c_int, v, v, v, v, v],
                "mi_merge_topk_gathered": [c_int, c_int, c_int64, c_int, v, c_int64, c_int64, c_int64, c_int64,
                                           c_int64, c_int64, v, v, v],
                "mi_shards_unique_id": [c_char_p, v],
                "mi_shards_create": [v, v, c_int, c_int, c_int, v, c_char_p, c_int64, c_int64, c_int64, POINTER(v)],
                "mi_shards_search": [v, c_int64, v, c_int, c_int, v, v, v],
                "mi_shards_destroy": [v],
                "mi_flat_create": [c_int, c_int, POINTER(v)],
                "mi_flat_create_ex": [c_int, c_int, c_int, POINTER(v)],
                "mi_flat_create_metric": [c_int, c_int, c_int, POINTER(v)],
                "mi_flat_sq_train": [v, c_int64, v, c_int],
                "mi_flat_sq_get_trained": [v, v],
                "mi_flat_sq_set_trained": [v, v],
                "mi_flat_sq_is_trained": [v, POINTER(c_int)],
                "mi_flat_destroy": [v],
                "mi_flat_add": [v, c_int64, v],
                "mi_flat_reserve": [v, c_int64],
                "mi_flat_ntotal": [v, POINTER(c_int64)],
                "mi_flat_reconstruct_n": [v, c_int64, c_int64, v],
                "mi_flat_get_rows": [v, c_int64, v, v],
                "mi_flat_release_workspaces": [v],
                "mi_index_release_workspaces": [v],
                "mi_flat_reset": [v],
                "mi_flat_search": [v, c_int64, v, c_int, v, v, v],
                "mi_flat_rerank": [v, c_int64, v, c_int, v, c_int, v, v, v],
                "mi_ip_assign": [c_int, c_int64, v, c_int64, v, c_int, v, v, v],
                "mi_ip_gemm": [c_int, c_int64, v, c_int64, v, c_int, v, v, v],
                "mi_ivfpq_reload_env": [],
                "mi_pq_encode": [c_int, c_int64, v, c_int, c_int, v, v, v],
                "mi_cluster_means": [c_int, c_int64, v, c_int, v, c_int, v, v, v],
                "mi_neg_half_sqnorm": [c_int, c_int64, v, c_int, v, v],
            }
            for name, args in sigs.items():
                fn = getattr(lib, name)
                fn.argtypes = args
                fn.restype = c_int
            cls._lib = lib
        return cls._lib


def _check(rc: int):
    if rc != 0:
        raise RuntimeError("mi_ivfpq: " + _Lib.get().mi_last_error().decode())


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _current_stream():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _as_f32(x, d: int, what: str = "x"):
    """faiss semantics: 2-D, second dim == d (assert), converted to C-contiguous float32."""
    if _is_torch(x):
        import torch
        assert x.dim() == 2, f"{what} must be 2-D"
        assert x.shape[1] == d, f"{what}.shape[1] ({x.shape[1]}) != index.d ({d})"
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.to(torch.float32).contiguous()
        return x
    x = np.ascontiguousarray(x, dtype=np.float32)
    assert x.ndim == 2, f"{what} must be 2-D"
    assert x.shape[1] == d, f"{what}.shape[1] ({x.shape[1]}) != index.d ({d})"
    return x


def _ptr(x):
    if x is None:
        return c_void_p(0)
    if _is_torch(x):
        return c_void_p(x.data_ptr())
    return c_void_p(x.ctypes.data)


def reload_env() -> None:
    """re-read the library's MI_* knobs from the environment (tests / tools; the library reads them once otherwise).  Not a
    faiss call."""
    _check(_Lib.get().mi_ivfpq_reload_env())


def get_num_gpus() -> int:
    n = c_int(0)
    _check(_Lib.get().mi_device_count(ctypes.byref(n)))
    return n.value


def omp_set_num_threads(n: int) -> None:  # API compatibility; the index runs on the GPU
    pass


# ----------------------------------------------------------------------
# IndexFlatIP
# ----------------------------------------------------------------------

class IndexFlatIP:
    """faiss.IndexFlatIP: exact inner-product search (config #1's plumbing
    index; the same GEMM + top-k kernels as the coarse quantiser)."""

    metric_type = METRIC_INNER_PRODUCT
    is_trained = True

    def __init__(self, d: int, device: int = 0, _storage: int = 0, _metric: int = METRIC_INNER_PRODUCT):
        self.d = int(d)
        self.device = int(device)
        self._h = c_void_p()
        if _metric == METRIC_L2:
            _check(_Lib.get().mi_flat_create_metric(self.d, METRIC_L2, self.device, ctypes.byref(self._h)))
            self.metric_type = METRIC_L2
        else:
            _check(_Lib.get().mi_flat_create_ex(self.d, self.device, int(_storage), ctypes.byref(self._h)))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _Lib.get().mi_flat_destroy(h)
            except Exception:
                pass

    @property
    def ntotal(self) -> int:
        n = c_int64(0)
        _check(_Lib.get().mi_flat_ntotal(self._h, ctypes.byref(n)))
        return n.value

    def train(self, x):
        pass

    def add(self, x):
        x = _as_f32(x, self.d)
        _check(_Lib.get().mi_flat_add(self._h, x.shape[0], _ptr(x)))

    def reserve(self, n: int):
        """Room for n vectors in total (not in faiss's Python API; its C++ callers reserve the
        codes vector): a store filled in chunks then never holds two copies while growing."""
        _check(_Lib.get().mi_flat_reserve(self._h, int(n)))

    def reset(self):
        _check(_Lib.get().mi_flat_reset(self._h))

    def reconstruct_n(self, i0: int = 0, ni: int = -1) -> np.ndarray:
        """faiss Index.reconstruct_n: the stored vectors [i0, i0 + ni) (all of them by default)."""
        ni = self.ntotal - i0 if ni < 0 else ni
        out = np.empty((ni, self.d), np.float32)
        _check(_Lib.get().mi_flat_reconstruct_n(self._h, int(i0), int(ni), _ptr(out)))
        return out

    def release_workspaces(self) -> None:
        """Give the per-stream scratch buffers back to the allocator (they return on the next call)."""
        _check(_Lib.get().mi_flat_release_workspaces(self._h))

    def get_rows(self, ids) -> np.ndarray:
        """The stored bytes of the rows `ids`, uint8 [len(ids), row_bytes]: d float32 (IndexFlat), d IEEE halves
        (QT_fp16) or d codes (QT_8bit) per row -- what faiss exposes as IndexFlatCodes.codes; the hook that hands a
        sample of a store too large to export to the oracle."""
        ids = np.ascontiguousarray(ids.cpu().numpy() if _is_torch(ids) else ids, np.int64).ravel()
        qt = getattr(self, "qtype", None)
        elem = 1 if qt == ScalarQuantizer.QT_8bit else 2 if qt == ScalarQuantizer.QT_fp16 else 4
        width = self.d + (4 if getattr(self, "metric_type", METRIC_INNER_PRODUCT) == METRIC_L2 else 0)
        out = np.empty((ids.shape[0], width * elem), np.uint8)
        _check(_Lib.get().mi_flat_get_rows(self._h, ids.shape[0], _ptr(ids), _ptr(out)))
        return out

    def reconstruct(self, i: int) -> np.ndarray:
        return self.reconstruct_n(int(i), 1)[0]

    def search(self, x, k: int):
        x = _as_f32(x, self.d)
        assert k > 0
        nq = x.shape[0]
        if _is_torch(x) and x.is_cuda:
            import torch
            D = torch.empty((nq, k), dtype=torch.float32, device=x.device)
            I = torch.empty((nq, k), dtype=torch.int64, device=x.device)
            _check(_Lib.get().mi_flat_search(self._h, nq, _ptr(x), k, _ptr(D), _ptr(I), _current_stream()))
            return D, I
        if _is_torch(x):
            x = x.numpy()
        D = np.empty((nq, k), np.float32)
        I = np.empty((nq, k), np.int64)
        _check(_Lib.get().mi_flat_search(self._h, nq, _ptr(x), k, _ptr(D), _ptr(I), c_void_p(0)))
        return D, I


    def rerank(self, x, cand_I, k: int, D=None, I=None, stream: int | None = None):
        """Exact scores of the candidate ids cand_I [nq, kc] (kc a multiple of k,
        negative = empty) and the k best -- the second stage of IndexRefineFlat.
        numpy in -> numpy out; CUDA tensors in -> CUDA tensors (D / I may be given)."""
        x = _as_f32(x, self.d)
        nq, kc = x.shape[0], int(cand_I.shape[1])
        if _is_torch(x) and x.is_cuda:
            import torch
            if D is None:
                D = torch.empty((nq, k), dtype=torch.float32, device=x.device)
                I = torch.empty((nq, k), dtype=torch.int64, device=x.device)
            st = _current_stream() if stream is None else c_void_p(stream)
            _check(_Lib.get().mi_flat_rerank(self._h, nq, _ptr(x), kc, _ptr(cand_I), k, _ptr(D), _ptr(I), st))
            return D, I
        if _is_torch(x):
            x = x.numpy()
        cand_I = np.ascontiguousarray(cand_I, np.int64)
        D = np.empty((nq, k), np.float32)
        I = np.empty((nq, k), np.int64)
        _check(_Lib.get().mi_flat_rerank(self._h, nq, _ptr(x), kc, _ptr(cand_I), k, _ptr(D), _ptr(I), c_void_p(0)))
        return D, I


class IndexFlat(IndexFlatIP):
    """faiss.IndexFlat(d, metric=METRIC_L2).  METRIC_L2: squared L2 distances, ascending,
    +FLT_MAX in unfilled slots, evaluated through the expansion |q|^2 + |x|^2 - 2<q, x>
    (one ascending-k chain per pair over augmented rows: oracle `flat_l2`)."""

    def __init__(self, d: int, metric: int = METRIC_L2, device: int = 0):
        if metric not in (METRIC_INNER_PRODUCT, METRIC_L2):
            raise NotImplementedError("metric must be METRIC_INNER_PRODUCT or METRIC_L2")
        super().__init__(d, device, _metric=int(metric))
        self.metric_type = int(metric)


class IndexFlatL2(IndexFlat):
    def __init__(self, d: int, device: int = 0):
        super().__init__(d, METRIC_L2, device)


class ScalarQuantizer:
    """faiss.ScalarQuantizer's quantiser-type constants (QT_fp16 and QT_8bit are implemented)."""
    QT_8bit, QT_4bit, QT_8bit_uniform, QT_4bit_uniform, QT_fp16, QT_8bit_direct, QT_6bit = range(7)


class _SQView:
    """index.sq: faiss's ScalarQuantizer member (qtype, d, code_size, trained)."""

    def __init__(self, index):
        self._index = index
        self.qtype, self.d = index.qtype, index.d
        self.code_size = index.d * (2 if index.qtype == ScalarQuantizer.QT_fp16 else 1)

    @property
    def trained(self) -> np.ndarray:
        """QT_8bit: float32 [2 d] = vmin | vdiff (faiss's layout); QT_fp16: empty."""
        if self.qtype != ScalarQuantizer.QT_8bit:
            return np.empty(0, np.float32)
        out = np.empty(2 * self.d, np.float32)
        _check(_Lib.get().mi_flat_sq_get_trained(self._index._h, _ptr(out)))
        return out

    @trained.setter
    def trained(self, t):
        t = np.ascontiguousarray(t, np.float32)
        assert t.shape == (2 * self.d,)
        _check(_Lib.get().mi_flat_sq_set_trained(self._index._h, _ptr(t)))


class IndexScalarQuantizer(IndexFlatIP):
    """faiss.IndexScalarQuantizer(d, qtype, METRIC_INNER_PRODUCT), the refine index of
    "...,Refine(SQfp16)" / "...,Refine(SQ8)": rerank() and reconstruct_n() work, a full search() is
    not implemented.

    QT_fp16: the vectors are kept as IEEE half (round to nearest even at add(), no scaling) and
    scored as <q, (float)x16> in the same f32 chain as IndexFlatIP -- half the bytes per re-ranked
    candidate.

    QT_8bit: one byte per component with per-dimension ranges -- train(x) takes vmin / vdiff = the
    minimum and the span of every dimension over x (faiss's RS_minmax with rangestat_arg 0),
    add() stores (int)(255 * clip((x - vmin) / vdiff)), a candidate is scored against vmin +
    (code + 0.5) / 255 * vdiff.  A quarter of the f32 bytes: the whole 207 M-vector refine store
    of BASELINE.json configs[3] (212 GB) fits one MI355X beside the index."""

    def __init__(self, d: int, qtype: int = ScalarQuantizer.QT_fp16, metric: int = METRIC_INNER_PRODUCT, device: int = 0):
        if qtype not in (ScalarQuantizer.QT_fp16, ScalarQuantizer.QT_8bit):
            raise NotImplementedError("IndexScalarQuantizer: ScalarQuantizer.QT_fp16 and QT_8bit are implemented on the MI355X path")
        if metric != METRIC_INNER_PRODUCT:
            raise NotImplementedError("only METRIC_INNER_PRODUCT is implemented on the MI355X path")
        super().__init__(d, device, _storage=1 if qtype == ScalarQuantizer.QT_fp16 else 2)
        self.qtype = qtype
        self.sq = _SQView(self)

    @property
    def is_trained(self) -> bool:
        t = c_int(0)
        _check(_Lib.get().mi_flat_sq_is_trained(self._h, ctypes.byref(t)))
        return bool(t.value)

    def train(self, x, merge: bool = False):
        """ScalarQuantizer.train (a no-op for QT_fp16).  merge=True widens the trained ranges
        with another chunk of training rows instead of replacing them."""
        if self.qtype != ScalarQuantizer.QT_8bit:
            return
        x = _as_f32(x, self.d)
        if _is_torch(x) and not x.is_cuda:
            x = x.numpy()
        _check(_Lib.get().mi_flat_sq_train(self._h, x.shape[0], _ptr(x), int(merge)))

    def search(self, x, k: int):
        raise NotImplementedError("IndexScalarQuantizer.search: the scalar-quantised store serves re-ranking only")


# ----------------------------------------------------------------------
# IndexIVFPQ
# ----------------------------------------------------------------------

class ClusteringParameters:
    """faiss.ClusteringParameters: the fields faiss's Clustering reads.  What train() here honours: `niter`,
    `max_points_per_centroid` (the training set is subsampled to that many points per centroid, like faiss), `seed`,
    `verbose`, `spherical` (coarse k-means only: centroids L2-normalised after every update and points assigned by inner
    product -- faiss Clustering::post_process_centroids [PRIOR]; `index_factory(..., METRIC_INNER_PRODUCT)` sets it, as
    faiss's factory does [PRIOR]; the IndexIVFPQ constructor leaves it False, as faiss's does).  `nredo` > 1,
    `int_centroids`, `frozen_centroids`, `update_index` raise at train() instead of being ignored; `min_points_per_centroid` only warns in faiss and is unused here.

    train() here is NOT faiss's Clustering: it is a bit-reproducible Lloyd's k-means (oracle/train_oracle.py is its
    restatement) -- initial centroids from a seeded permutation (faiss: its own RandomGenerator permutation), empty clusters
    re-seeded from random points (faiss: split the largest cluster with a +-1/1024 perturbation), members summed in
    ascending row order.  An index TRAINED here is therefore a different (equally valid) index from the one faiss trains
    on the same data; a faiss-trained index is reproduced through set_centroids / set_codebook or read_index."""

    def __init__(self, niter: int = 25):
        self.niter = niter
        self.nredo = 1
        self.max_points_per_centroid = 256
        self.min_points_per_centroid = 39
        self.seed = 1234
        self.verbose = False
        self.spherical = False
        self.int_centroids = False
        self.update_index = False
        self.frozen_centroids = False

    def check_supported(self, what: str):
        bad = [k for k, v in (("nredo", self.nredo != 1), ("int_centroids", self.int_centroids),
                              ("update_index", self.update_index), ("frozen_centroids", self.frozen_centroids)) if v]
        if bad:
            raise NotImplementedError(f"{what}: ClusteringParameters.{', '.join(bad)} not implemented (would be silently ignored)")
        if self.niter < 0 or self.max_points_per_centroid < 1:
            raise ValueError(f"{what}: niter >= 0 and max_points_per_centroid >= 1 required")


class _PQ:
    """index.pq: ProductQuantizer parameters (read-only view) + its own ClusteringParameters `cp` (faiss: pq.cp, 25
    iterations by default -- the coarse quantiser's `index.cp` defaults to 10, as faiss's Level1Quantizer sets it)."""

    def __init__(self, d, M, nbits):
        self.d, self.M, self.nbits = d, M, nbits
        self.ksub = 1 << nbits
        self.dsub = d // M
        self.code_size = M
        self.cp = ClusteringParameters(25)


class IndexIVFPQ:
    """faiss.IndexIVFPQ (inner product, 8-bit codes) on one MI355X.

    Same call surface as faiss: ``train(x)``, ``add(x)``, ``add_with_ids(x, ids)``,
    ``search(x, k) -> (D, I)``, ``reset()``, attributes ``d ntotal is_trained
    nprobe nlist metric_type by_residual pq``.
    """

    def __init__(self, *args, **kw):
        """Two signatures: faiss's own ``IndexIVFPQ(quantizer, d, nlist, M, nbits_per_idx[, metric])``
        (first argument an IndexFlat; centroids it already holds are taken over; faiss's default
        metric there is METRIC_L2) and this
        mirror's ``IndexIVFPQ(d, nlist, M, nbits=8, metric=METRIC_INNER_PRODUCT, by_residual=True, device=0)``."""
        quantizer = None
        if args and isinstance(args[0], IndexFlatIP):
            quantizer, args = args[0], args[1:]
            names = ("d", "nlist", "M", "nbits", "metric")
            p = dict(nbits=8, metric=METRIC_L2, by_residual=True, device=quantizer.device)
        else:
            names = ("d", "nlist", "M", "nbits", "metric", "by_residual", "device")
            p = dict(nbits=8, metric=METRIC_INNER_PRODUCT, by_residual=True, device=0)
        if len(args) > len(names):
            raise TypeError("IndexIVFPQ: too many positional arguments")
        p.update(dict(zip(names, args)))
        p.update(kw)
        d, nlist, M, nbits, metric, by_residual, device = (p[k] for k in ("d", "nlist", "M", "nbits", "metric", "by_residual", "device"))
        if quantizer is not None:
            assert quantizer.d == int(d), f"quantizer.d ({quantizer.d}) != d ({d})"
        self.d, self.nlist, self.device = int(d), int(nlist), int(device)
        self.metric_type = int(metric)
        self.by_residual = bool(by_residual)
        self.nprobe = 1
        self.pq = _PQ(self.d, int(M), int(nbits))
        self.cp = ClusteringParameters(10)
        self.verbose = False
        self._h = c_void_p()
        _check(_Lib.get().mi_index_create(self.d, self.nlist, int(M), int(nbits), self.metric_type,
                                          int(self.by_residual), self.device, ctypes.byref(self._h)))
        self._coarse_set = False
        if quantizer is not None and quantizer.ntotal:
            # a quantizer that already holds centroids is a trained quantizer (faiss: quantizer.is_trained
            # and quantizer.ntotal == nlist -> train() only trains the PQ)
            assert quantizer.ntotal == self.nlist, f"quantizer holds {quantizer.ntotal} vectors, nlist is {self.nlist}"
            self.set_centroids(quantizer.reconstruct_n(0, self.nlist))

    @property
    def quantizer(self):
        """faiss's index.quantizer: an IndexFlat over the coarse centroids (a copy; empty
        until the index is trained)."""
        q = IndexFlat(self.d, self.metric_type, self.device)
        if self._coarse_set:
            q.add(self.get_centroids())
        return q

    @classmethod
    def _from_handle(cls, h, device: int = 0):
        """Wrap a handle created by the library (mi_index_load)."""
        vals = [c_int(0) for _ in range(7)]
        _check(_Lib.get().mi_index_get_params(h, *[ctypes.byref(x) for x in vals]))
        d, nlist, M, nbits, metric, by_res, nprobe = (x.value for x in vals)
        self = cls.__new__(cls)
        self.d, self.nlist, self.device = d, nlist, int(device)
        self.metric_type, self.by_residual, self.nprobe = metric, bool(by_res), max(1, nprobe)
        self.pq = _PQ(d, M, nbits)
        self.cp = ClusteringParameters(10)
        self.verbose = False
        self._h = h
        self._coarse_set = self.is_trained
        return self

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _Lib.get().mi_index_destroy(h)
            except Exception:
                pass

    # -- state ---------------------------------------------------------
    @property
    def ntotal(self) -> int:
        n = c_int64(0)
        _check(_Lib.get().mi_index_ntotal(self._h, ctypes.byref(n)))
        return n.value

    @property
    def is_trained(self) -> bool:
        t = c_int(0)
        _check(_Lib.get().mi_index_is_trained(self._h, ctypes.byref(t)))
        return bool(t.value)

    @property
    def code_size(self) -> int:
        return self.pq.M

    def set_centroids(self, centroids):
        """Coarse centroids [nlist, d] (what faiss keeps in index.quantizer)."""
        c = np.ascontiguousarray(centroids, np.float32) if not _is_torch(centroids) else centroids.contiguous()
        assert tuple(c.shape) == (self.nlist, self.d)
        _check(_Lib.get().mi_index_set_coarse(self._h, _ptr(c)))
        self._coarse_set = True

    def set_codebook(self, codebook):
        """PQ codebook [M, 256, d/M] (faiss: index.pq.centroids)."""
        c = np.ascontiguousarray(codebook, np.float32) if not _is_torch(codebook) else codebook.contiguous()
        assert tuple(c.shape) == (self.pq.M, self.pq.ksub, self.pq.dsub)
        _check(_Lib.get().mi_index_set_codebook(self._h, _ptr(c)))

    def get_centroids(self) -> np.ndarray:
        out = np.empty((self.nlist, self.d), np.float32)
        _check(_Lib.get().mi_index_get_coarse(self._h, _ptr(out)))
        return out

    def get_codebook(self) -> np.ndarray:
        out = np.empty((self.pq.M, self.pq.ksub, self.pq.dsub), np.float32)
        _check(_Lib.get().mi_index_get_codebook(self._h, _ptr(out)))
        return out

    # -- train (setup, not on the timed path) ---------------------------
    def train(self, x):
        """k-means for the coarse centroids, then k-means per sub-quantiser on
        the residuals (faiss IndexIVFPQ.train; reference Makefile:39).  The
        assignment steps run on the HIP kernels (mi_ip_assign / mi_pq_encode);
        the centroid means are torch scatter-adds on the device."""
        from . import _train
        x = _as_f32(x, self.d)
        # faiss: a quantizer that is already trained and holds nlist centroids keeps them
        # (IndexIVFPQ(quantizer, ...) with a filled quantizer): only the PQ is trained then
        given = None
        if self._coarse_set and not self.is_trained:
            import torch
            given = torch.from_numpy(self.get_centroids()).to(torch.device("cuda", self.device))
        self.cp.check_supported("IndexIVFPQ.train (index.cp)")
        self.pq.cp.check_supported("IndexIVFPQ.train (index.pq.cp)")
        if self.pq.cp.spherical:
            raise NotImplementedError("IndexIVFPQ.train: index.pq.cp.spherical (unit-norm PQ codewords) is not implemented; "
                                      "index.cp.spherical (the coarse quantiser) is")
        cent, cb = _train.train_ivfpq(x, self.nlist, self.pq.M, self.by_residual, self.cp,
                                      self.device, self.verbose or self.cp.verbose, centroids=given, metric=self.metric_type,
                                      pq_cp=self.pq.cp)
        self.set_centroids(cent)
        self.set_codebook(cb)

    # -- add -------------------------------------------------------------
    def add(self, x):
        self.add_with_ids(x, None)

    def add_with_ids(self, x, ids):
        x = _as_f32(x, self.d)
        n = x.shape[0]
        if ids is not None:
            if _is_torch(ids):
                import torch
                ids = ids.to(torch.int64).contiguous()
            else:
                ids = np.ascontiguousarray(ids, np.int64)
            assert ids.shape == (n,)
        if _is_torch(x) and not x.is_cuda:
            x = x.numpy()
        _check(_Lib.get().mi_index_add(self._h, n, _ptr(x), _ptr(ids)))

    def add_codes(self, list_no, codes, ids=None):
        """Append pre-encoded entries (read_index path)."""
        list_no = np.ascontiguousarray(list_no, np.int32)
        codes = np.ascontiguousarray(codes, np.uint8)
        n = list_no.shape[0]
        assert codes.shape == (n, self.pq.M)
        if ids is not None:
            ids = np.ascontiguousarray(ids, np.int64)
        _check(_Lib.get().mi_index_add_codes(self._h, n, _ptr(list_no), _ptr(codes), _ptr(ids)))

    def encode(self, x):
        """(list_no[n] int32, codes[n, M] uint8): the arithmetic of add()."""
        x = _as_f32(x, self.d)
        if _is_torch(x) and not x.is_cuda:
            x = x.numpy()
        n = x.shape[0]
        list_no = np.empty(n, np.int32)
        codes = np.empty((n, self.pq.M), np.uint8)
        _check(_Lib.get().mi_index_encode(self._h, n, _ptr(x), _ptr(list_no), _ptr(codes)))
        return list_no, codes

    def reset(self):
        _check(_Lib.get().mi_index_reset(self._h))

    def seal(self):
        """(not in faiss) The index is filled and will be searched: free the append log the lists were built from (80 B per
        vector: 16.6 GB of the 207 M-vector index) and keep the scan image.  add / add_with_ids / export / write_index after a
        seal() rebuild the log from the image first -- nothing changes but the HBM in use."""
        _check(_Lib.get().mi_index_seal(self._h))

    def list_size(self, list_no: int) -> int:
        n = c_int64(0)
        _check(_Lib.get().mi_index_list_size(self._h, int(list_no), ctypes.byref(n)))
        return n.value

    def get_list(self, list_no: int):
        n = self.list_size(list_no)
        codes = np.empty((n, self.pq.M), np.uint8)
        ids = np.empty(n, np.int64)
        _check(_Lib.get().mi_index_get_list(self._h, int(list_no), _ptr(codes), _ptr(ids)))
        return codes, ids

    def list_sizes(self) -> np.ndarray:
        """All list sizes (int64 [nlist]) in one call."""
        out = np.empty(self.nlist, np.int64)
        _check(_Lib.get().mi_index_list_sizes(self._h, _ptr(out)))
        return out

    def release_workspaces(self) -> None:
        """Give the per-stream scratch buffers (a set per stream that ever searched) back to the allocator; they return
        on the next call.  Not while another thread is inside a call on this index."""
        _check(_Lib.get().mi_index_release_workspaces(self._h))

    def export_lists(self, list_lo: int = 0, list_hi: int | None = None):
        """(codes [rows, M] uint8, ids [rows] int64) of the lists [list_lo, list_hi)
        concatenated in list order, each list in insertion order (host arrays)."""
        list_hi = self.nlist if list_hi is None else int(list_hi)
        rows = int(self.list_sizes()[list_lo:list_hi].sum())
        codes = np.empty((rows, self.pq.M), np.uint8)
        ids = np.empty(rows, np.int64)
        _check(_Lib.get().mi_index_export_lists(self._h, int(list_lo), list_hi, _ptr(codes), _ptr(ids)))
        return codes, ids

    def reserve(self, n: int):
        """Room for n vectors in total (the lists live in HBM; a fill of known size then
        never re-allocates while growing)."""
        _check(_Lib.get().mi_index_reserve(self._h, int(n)))

    # -- search ------------------------------------------------------------
    def search(self, x, k: int, nprobe: int | None = None, params=None):
        """faiss's search(x, k, params=SearchParametersIVF(nprobe=...)): per-call parameters
        override the index attribute (`nprobe=` is this mirror's shorthand for the same)."""
        x = _as_f32(x, self.d)
        assert k > 0
        if params is not None:
            if getattr(params, "sel", None) is not None or getattr(params, "max_codes", 0):
                raise NotImplementedError("SearchParametersIVF: only nprobe is implemented on the MI355X path")
            nprobe = getattr(params, "nprobe", nprobe)
        nprobe = int(self.nprobe if nprobe is None else nprobe)
        nq = x.shape[0]
        if _is_torch(x) and x.is_cuda:
            import torch
            D = torch.empty((nq, k), dtype=torch.float32, device=x.device)
            I = torch.empty((nq, k), dtype=torch.int64, device=x.device)
            _check(_Lib.get().mi_index_search(self._h, nq, _ptr(x), k, nprobe, _ptr(D), _ptr(I),
                                              _current_stream()))
            return D, I
        if _is_torch(x):
            x = x.numpy()
        D = np.empty((nq, k), np.float32)
        I = np.empty((nq, k), np.int64)
        _check(_Lib.get().mi_index_search(self._h, nq, _ptr(x), k, nprobe, _ptr(D), _ptr(I), c_void_p(0)))
        return D, I

    def search_into(self, x, k: int, D, I, nprobe: int | None = None, stream: int | None = None):
        """search() into caller-owned CUDA tensors (no allocation; what a
        steady-state serving loop uses).  `stream`: raw hipStream_t (int), default
        torch's current stream; the library keeps one workspace set per stream,
        so batches issued on different streams overlap on the GPU."""
        nprobe = self.nprobe if nprobe is None else nprobe
        st = _current_stream() if stream is None else c_void_p(stream)
        rc = _Lib.get().mi_index_search(self._h, x.shape[0], c_void_p(x.data_ptr()), k, nprobe,
                                        c_void_p(D.data_ptr()), c_void_p(I.data_ptr()), st)
        if rc:
            _check(rc)

    def search_candidates_into(self, x, kc: int, I, nprobe: int | None = None, stream: int | None = None):
        """The kc best entries of every query as an unordered SET of ids (CUDA tensors; -1 = unfilled): the first
        stage of an IndexRefine search, whose re-rank does not depend on the order of its candidates."""
        nprobe = self.nprobe if nprobe is None else nprobe
        st = _current_stream() if stream is None else c_void_p(stream)
        rc = _Lib.get().mi_index_search_candidates(self._h, x.shape[0], c_void_p(x.data_ptr()), int(kc), int(nprobe),
                                                   c_void_p(I.data_ptr()), st)
        if rc:
            _check(rc)

    def coarse_slice(self, x, nprobe: int, list_lo: int, list_hi: int, D_out=None):
        """quantizer.search restricted to centroids [list_lo, list_hi) with global
        list numbers -> (coarse_I int32, coarse_D f32) CUDA tensors [nq, nprobe].
        D_out: a contiguous f32 [nq, nprobe] CUDA tensor the scores are written into (the sharded search's send buffer)."""
        import torch
        x = _as_f32(x, self.d)
        nq = x.shape[0]
        cI = torch.empty((nq, nprobe), dtype=torch.int32, device=x.device)
        if D_out is not None:
            assert D_out.is_cuda and D_out.dtype == torch.float32 and D_out.is_contiguous() and tuple(D_out.shape) == (nq, nprobe)
        cD = D_out if D_out is not None else torch.empty((nq, nprobe), dtype=torch.float32, device=x.device)
        _check(_Lib.get().mi_index_coarse_slice(self._h, nq, _ptr(x), int(nprobe), int(list_lo), int(list_hi),
                                                _ptr(cI), _ptr(cD), _current_stream()))
        return cI, cD

    def search_preassigned(self, x, k: int, coarse_I, coarse_D, D=None, I=None):
        """faiss IndexIVF.search_preassigned (CUDA tensors): search with a given
        coarse assignment [nq, nprobe] (int32 list numbers, -1 = none).  D / I: contiguous f32 / int64 [nq, k] CUDA
        tensors to write into (the sharded search's send buffer) instead of fresh ones."""
        import torch
        x = _as_f32(x, self.d)
        nq = x.shape[0]
        coarse_I = coarse_I.to(torch.int32).contiguous()
        coarse_D = coarse_D.to(torch.float32).contiguous()
        assert coarse_I.shape == coarse_D.shape and coarse_I.shape[0] == nq
        if D is None:
            D = torch.empty((nq, k), dtype=torch.float32, device=x.device)
            I = torch.empty((nq, k), dtype=torch.int64, device=x.device)
        else:
            assert I is not None and D.is_cuda and I.is_cuda and D.dtype == torch.float32 and I.dtype == torch.int64
            assert D.is_contiguous() and I.is_contiguous() and tuple(D.shape) == (nq, k) and tuple(I.shape) == (nq, k)
        _check(_Lib.get().mi_index_search_preassigned(self._h, nq, _ptr(x), k, coarse_I.shape[1], _ptr(coarse_I),
                                                      _ptr(coarse_D), _ptr(D), _ptr(I), _current_stream()))
        return D, I

    def coarse_and_lut(self, x, nprobe: int | None = None, want_lut: bool = True):
        """Steps 1-2 of search for parity tests: (coarse_I, coarse_D, lut)."""
        x = _as_f32(x, self.d)
        if _is_torch(x) and not x.is_cuda:
            x = x.numpy()
        nprobe = min(int(self.nprobe if nprobe is None else nprobe), self.nlist)
        nq = x.shape[0]
        cI = np.empty((nq, nprobe), np.int32)
        cD = np.empty((nq, nprobe), np.float32)
        lut = np.empty((nq, self.pq.M, self.pq.ksub), np.float32) if want_lut else None
        _check(_Lib.get().mi_index_coarse_lut(self._h, nq, _ptr(x), nprobe, _ptr(cI), _ptr(cD), _ptr(lut)))
        if self.metric_type == METRIC_L2:       # the library hands back -|q - c|^2; faiss reports squared distances
            cD = np.where(cI < 0, np.float32(np.finfo(np.float32).max), -cD)
        return cI, cD, lut

    def prune_stats(self, reset: bool = False) -> dict:
        """Exact list pruning (mi_ivfpq.h, mi_index_prune_stats): what the pruned searches since the last reset skipped."""
        out = (ctypes.c_ulonglong * 3)()
        _check(_Lib.get().mi_index_prune_stats(self._h, out, int(reset)))
        g2, gall, nq = int(out[0]), int(out[1]), int(out[2])
        return {"queries": nq, "groups_all_probes": gall, "groups_second_phase": g2}

    # -- scan-kernel timing (HIP events on the launch stream) ------------
    def profile_scan(self, reps: int = 50, stream: int | None = None):
        """Replays the scan kernel of the last search on `stream` `reps` times
        between two HIP events on that stream -> {"scan_ms_avg", "scan_bytes"}."""
        ms, b = c_double(0), c_int64(0)
        st = _current_stream() if stream is None else c_void_p(stream)
        _check(_Lib.get().mi_index_profile_scan(self._h, int(reps), st,
                                                ctypes.byref(ms), ctypes.byref(b)))
        return {"scan_ms_avg": ms.value, "scan_bytes": b.value}


def merge_topk(D_parts, I_parts, device: int = 0):
    """k-way merge of per-shard results [nparts, nq, k] -> (D, I)  (the
    arithmetic after the all-gather; faiss IndexShards' merge)."""
    nparts, nq, k = D_parts.shape
    if _is_torch(D_parts) and D_parts.is_cuda:
        import torch
        D_parts, I_parts = D_parts.contiguous(), I_parts.contiguous()
        D = torch.empty((nq, k), dtype=torch.float32, device=D_parts.device)
        I = torch.empty((nq, k), dtype=torch.int64, device=D_parts.device)
        _check(_Lib.get().mi_merge_topk(D_parts.device.index or 0, nparts, nq, k, _ptr(D_parts),
                                        _ptr(I_parts), _ptr(D), _ptr(I), _current_stream()))
        return D, I
    D_parts = np.ascontiguousarray(D_parts, np.float32)
    I_parts = np.ascontiguousarray(I_parts, np.int64)
    D = np.empty((nq, k), np.float32)
    I = np.empty((nq, k), np.int64)
    _check(_Lib.get().mi_merge_topk(device, nparts, nq, k, _ptr(D_parts), _ptr(I_parts), _ptr(D),
                                    _ptr(I), c_void_p(0)))
    return D, I


def merge_topk_gathered(gathered, nparts: int, nq: int, k: int, blk_bytes: int, id_affine=(1, 0, 0),
                        q_lo: int = 0, nq_out: int | None = None, D=None, I=None):
    """The exchange step's merge straight on the receive buffer of one all-gather
    (uint8 CUDA tensor, `nparts` blocks of `blk_bytes` = [D f32 | pad 8 | I i64], see
    mi_merge_topk_gathered); ids are translated local -> global by the affine map."""
    import torch
    nq_out = nq if nq_out is None else nq_out
    if D is None:
        D = torch.empty((nq_out, k), dtype=torch.float32, device=gathered.device)
        I = torch.empty((nq_out, k), dtype=torch.int64, device=gathered.device)
    mul, add, step = id_affine
    _check(_Lib.get().mi_merge_topk_gathered(gathered.device.index or 0, int(nparts), int(nq), int(k), _ptr(gathered),
                                             int(blk_bytes), int(mul), int(add), int(step), int(q_lo), int(nq_out),
                                             _ptr(D), _ptr(I), _current_stream()))
    return D, I


# ----------------------------------------------------------------------
# module-level faiss functions
# ----------------------------------------------------------------------

class IndexRefine:
    """faiss.IndexRefine(base_index, refine_index) / IndexRefineFlat(base_index): the base index proposes
    ``k * k_factor`` candidates, an IndexFlat over the same vectors (in add()
    order: the base index must number its vectors sequentially, as faiss
    requires) recomputes their exact scores and the k best are returned.
    PQ64 ranks near-identical neighbours poorly (recall@10 0.38 on the bench
    corpus whatever nprobe); the refine stage is what reaches the north star's
    recall >= 0.95 (factory string "IVF4096,PQ64,RFlat")."""

    def __init__(self, base_index, refine_index=None):
        self.base_index = base_index
        self.d = base_index.d
        self.refine_index = refine_index if refine_index is not None else IndexFlat(self.d, base_index.metric_type, base_index.device)
        self.k_factor = 1.0
        self.metric_type = base_index.metric_type

    @property
    def ntotal(self) -> int:
        return self.base_index.ntotal

    @property
    def nprobe(self):
        return self.base_index.nprobe

    @nprobe.setter
    def nprobe(self, v):
        self.base_index.nprobe = v

    @property
    def is_trained(self) -> bool:
        return self.base_index.is_trained and self.refine_index.is_trained

    def train(self, x):
        """faiss IndexRefine::train: both indexes see the training set, always (an IVF-PQ base retrains its codebooks
        on a new sample exactly like faiss's, whose IndexRefine::train calls base_index->train unconditionally)."""
        self.base_index.train(x)
        self.refine_index.train(x)

    def add(self, x):
        if self.base_index.ntotal != self.refine_index.ntotal:
            raise RuntimeError("IndexRefineFlat: base and refine index are out of step")
        self.base_index.add(x)
        self.refine_index.add(x)

    def add_with_ids(self, x, ids):
        """faiss's IndexRefine keeps its refine index in add() order and looks candidates up by
        label, so labels must be positions: only ids that continue 0, 1, 2, ... are accepted
        (anything else would silently re-rank the wrong rows).  For a shard of a larger corpus
        keep positions here and translate at the exchange (ShardedIndex id_affine / id_map)."""
        ids_h = ids.cpu().numpy() if _is_torch(ids) else np.asarray(ids)
        n0 = self.ntotal
        if ids_h.shape != (len(x),) or not np.array_equal(ids_h, np.arange(n0, n0 + len(x))):
            raise NotImplementedError("IndexRefine.add_with_ids: ids must be the positions ntotal, ntotal + 1, ... "
                                      "(the refine index is addressed by position); use add(), and id_affine / id_map "
                                      "of ShardedIndex for a shard's global numbering")
        self.add(x)

    def reset(self):
        self.base_index.reset()
        self.refine_index.reset()

    def search(self, x, k: int, params=None):
        """params: faiss.IndexRefineSearchParameters(k_factor=..., base_index_params=...)."""
        k_factor = self.k_factor if params is None else getattr(params, "k_factor", self.k_factor)
        k_base = int(k * k_factor)
        k_base = max(k, k_base - k_base % k)      # the re-ranking step takes whole multiples of k
        bp = getattr(params, "base_index_params", None)
        if _is_torch(x) and x.is_cuda and hasattr(self.base_index, "search_candidates_into") and self.metric_type == METRIC_INNER_PRODUCT:
            import torch
            x = _as_f32(x, self.d)
            if bp is not None and (getattr(bp, "sel", None) is not None or getattr(bp, "max_codes", 0)):
                raise NotImplementedError("SearchParametersIVF: only nprobe is implemented on the MI355X path")
            cand = torch.empty((x.shape[0], k_base), dtype=torch.int64, device=x.device)
            self.base_index.search_candidates_into(x, k_base, cand, getattr(bp, "nprobe", None))
            return self.refine_index.rerank(x, cand, k)
        _, cand = self.base_index.search(x, k_base, params=bp)
        return self.refine_index.rerank(x, cand, k)

    def release_workspaces(self) -> None:
        for ix in (self.base_index, self.refine_index):
            if hasattr(ix, "release_workspaces"):
                ix.release_workspaces()

    def search_into(self, x, k: int, D, I, cand_D, cand_I, stream: int | None = None):
        """search() into caller-owned CUDA tensors (cand_I: [nq, k_base] scratch; cand_D is only written when the base
        index has no unordered-candidates entry point)."""
        if hasattr(self.base_index, "search_candidates_into") and self.metric_type == METRIC_INNER_PRODUCT:
            self.base_index.search_candidates_into(x, int(cand_I.shape[1]), cand_I, None, stream)
        else:
            self.base_index.search_into(x, int(cand_I.shape[1]), cand_D, cand_I, None, stream)
        self.refine_index.rerank(x, cand_I, k, D, I, stream)


class IndexRefineFlat(IndexRefine):
    """faiss.IndexRefineFlat: IndexRefine whose refine index is an IndexFlat (factory ",RFlat")."""


class SearchParameters:
    """faiss.SearchParameters (base class; `sel` = IDSelector is not implemented here)."""

    def __init__(self, sel=None):
        self.sel = sel


class SearchParametersIVF(SearchParameters):
    def __init__(self, sel=None, nprobe: int = 1, max_codes: int = 0, quantizer_params=None):
        super().__init__(sel)
        self.nprobe, self.max_codes, self.quantizer_params = int(nprobe), int(max_codes), quantizer_params


class IndexRefineSearchParameters(SearchParameters):
    def __init__(self, sel=None, k_factor: float = 1.0, base_index_params=None):
        super().__init__(sel)
        self.k_factor, self.base_index_params = float(k_factor), base_index_params


_FACTORY_RE = re.compile(r"^IVF(\d+)(?:_HNSW\d+)?,PQ(\d+)(?:x(\d+))?(,RFlat|,Refine\(SQfp16\)|,Refine\(SQ8\)|,Refine\(Flat\))?$")


def index_factory(d: int, description: str, metric: int = METRIC_L2, device: int = 0):
    """faiss.index_factory for the strings this path uses: "IVF{nlist},PQ{M}"
    (optionally "PQ{M}x8", ",RFlat", ",Refine(SQfp16)", ",Refine(SQ8)") and "Flat".  Like faiss the default
    metric is METRIC_L2 (the reference's normalised embeddings want METRIC_INNER_PRODUCT)."""
    description = description.replace(" ", "")
    if description == "Flat":
        return IndexFlat(d, metric, device)
    if metric not in (METRIC_INNER_PRODUCT, METRIC_L2):
        raise NotImplementedError("metric must be METRIC_INNER_PRODUCT or METRIC_L2")
    m = _FACTORY_RE.match(description)
    if not m:
        raise ValueError(f"index_factory: unsupported description {description!r} "
                         "(supported: 'Flat', 'IVF<nlist>,PQ<M>[x8][,RFlat | ,Refine(SQfp16) | ,Refine(SQ8)]')")
    nlist, M, nbits = int(m.group(1)), int(m.group(2)), int(m.group(3) or 8)
    index = IndexIVFPQ(d, nlist, M, nbits, metric, device=device)
    if metric == METRIC_INNER_PRODUCT:
        index.cp.spherical = True                # [PRIOR] faiss index_factory: "if (metric == METRIC_INNER_PRODUCT) index_ivf->cp.spherical = true"
    if m.group(4) == ",Refine(SQfp16)":          # half-precision refine store: half the HBM and half the bytes per candidate
        return IndexRefine(index, IndexScalarQuantizer(d, ScalarQuantizer.QT_fp16, metric, device))
    if m.group(4) == ",Refine(SQ8)":             # 8-bit refine store with per-dimension ranges: a quarter of the f32 bytes
        return IndexRefine(index, IndexScalarQuantizer(d, ScalarQuantizer.QT_8bit, metric, device))
    return IndexRefineFlat(index) if m.group(4) else index


def normalize_L2(x) -> None:
    """faiss.normalize_L2: rows of a float32 matrix scaled to unit length, in place (what a
    cosine-similarity pipeline does before an inner-product index; zero rows stay zero)."""
    if _is_torch(x):
        n = x.norm(dim=1, keepdim=True)
        x.div_(n.masked_fill_(n == 0, 1.0))
        return
    if not (isinstance(x, np.ndarray) and x.dtype == np.float32 and x.ndim == 2 and x.flags.c_contiguous):
        raise TypeError("normalize_L2 needs a C-contiguous float32 matrix")
    n = np.sqrt(np.einsum("ij,ij->i", x, x, dtype=np.float32))
    n[n == 0] = 1.0
    x /= n[:, None]


# ---- faiss-gpu's cloning entry points: the index already lives on the GPU, so they are
# identities (kept so that code written for faiss-cpu + faiss-gpu runs unchanged) ----
class StandardGpuResources:
    def noTempMemory(self): pass
    def setTempMemory(self, nbytes): pass
    def setDefaultNullStreamAllDevices(self): pass


class GpuClonerOptions:
    def __init__(self):
        self.useFloat16 = False
        self.useFloat16CoarseQuantizer = False
        self.usePrecomputed = False
        self.indicesOptions = 0
        self.reserveVecs = 0
        self.storeTransposed = False
        self.verbose = False


class GpuMultipleClonerOptions(GpuClonerOptions):
    def __init__(self):
        super().__init__()
        self.shard = False


def index_cpu_to_gpu(res, device: int, index, options=None):
    if getattr(index, "device", device) != device:
        raise NotImplementedError("the index lives on the device it was created on (device=%d)" % index.device)
    return index


def index_cpu_to_all_gpus(index, co=None, ngpu: int = -1):
    """One process drives one GPU here; multi-GPU search is `shards.ShardedIndex` (one
    process per GPU over RCCL), not an in-process replica set."""
    return index


def index_gpu_to_cpu(index):
    return index


def extract_index_ivf(index):
    """faiss.extract_index_ivf: the IndexIVF inside a wrapper (IndexRefineFlat -> its base index)."""
    while not isinstance(index, IndexIVFPQ):
        inner = getattr(index, "base_index", None) or getattr(index, "index", None)
        if inner is None:
            raise RuntimeError("extract_index_ivf: no IndexIVF inside this index")
        index = inner
    return index


def downcast_index(index):
    return index


_MAGIC = "mi355x-ivfpq-v1"


class IndexPreTransform:
    """faiss.IndexPreTransform over LinearTransforms (OPQMatrix, RandomRotationMatrix, LinearTransform): x -> A x + b
    applied (one exact-f32 GEMM per transform on the library's own kernel) in front of train / add / search of the wrapped
    index.  read_index returns one for an "IxPT" file and write_index writes one back; the matrices come from the file or
    from the caller -- OPQ TRAINING (faiss OPQMatrix.train) is not implemented, so `index_factory` does not accept "OPQ..."
    strings and train() needs a chain that is already trained."""

    def __init__(self, chain, index):
        self.chain = [(np.ascontiguousarray(A, np.float32), None if b is None else np.ascontiguousarray(b, np.float32)) for A, b in chain]
        self.index = index
        self.d = int(self.chain[0][0].shape[1]) if self.chain else index.d
        self.metric_type = index.metric_type
        self._dev = None

    ntotal = property(lambda self: self.index.ntotal)
    is_trained = property(lambda self: self.index.is_trained)
    nprobe = property(lambda self: self.index.nprobe, lambda self, v: setattr(self.index, "nprobe", v))

    def apply(self, x):
        """VectorTransform::apply of the whole chain (numpy in -> numpy out, CUDA tensor in -> CUDA tensor out): every
        transform is one exact-f32 GEMM on the library's own kernel (mi_ip_gemm: an ascending-k fmaf chain per output, + b)."""
        import torch
        was_np = not _is_torch(x)
        dev = torch.device("cuda", self.index.device)
        t = torch.as_tensor(np.ascontiguousarray(x, np.float32)) if was_np else x.float()
        if t.dim() != 2 or t.shape[1] != self.d:
            raise ValueError(f"IndexPreTransform: expected [n, {self.d}] vectors, got {tuple(t.shape)}")
        t = t.to(dev).contiguous()
        if self._dev is None:
            self._dev = [(torch.from_numpy(A).to(dev).contiguous(), None if b is None else torch.from_numpy(b).to(dev).contiguous())
                         for A, b in self.chain]
        stream = c_void_p(torch.cuda.current_stream(dev).cuda_stream)   # the stream of index.device, not of torch's current device
        for A, b in self._dev:
            if t.shape[0] == 0:
                t = torch.empty((0, A.shape[0]), dtype=torch.float32, device=dev)
                continue
            if A.shape[1] % 4:
                raise NotImplementedError("IndexPreTransform: transform input width must be a multiple of 4")
            out = torch.empty((t.shape[0], A.shape[0]), dtype=torch.float32, device=dev)
            _check(_Lib.get().mi_ip_gemm(self.index.device, t.shape[0], c_void_p(t.data_ptr()), A.shape[0], c_void_p(A.data_ptr()),
                                         A.shape[1], c_void_p(b.data_ptr() if b is not None else 0), c_void_p(out.data_ptr()), stream))
            t = out
        return t.cpu().numpy() if was_np else t

    def train(self, x):
        self.index.train(self.apply(x))

    def add(self, x):
        self.index.add(self.apply(x))

    def search(self, x, k, **kw):
        return self.index.search(self.apply(x), k, **kw)

    def reset(self):
        self.index.reset()


def _lists_of(index):
    sizes = index.list_sizes()
    codes, ids = index.export_lists()
    return sizes, codes, ids


def write_index(index, fname: str, ondisk_data: str | None = None) -> None:
    """``faiss.write_index``.  Writes faiss's binary format (``IwPQ`` with an
    ``IndexFlat`` quantiser; lists in the file, or in ``ondisk_data`` as
    ``OnDiskInvertedLists`` -- the reference's ``index.faiss`` +
    ``ondisk.ivfdata`` pair, Makefile:11-12) -- see faiss_io.py for the layout
    and its validation status.  A name ending in ``.npz`` selects the package's
    own numpy container instead."""
    if isinstance(index, IndexFlatIP):
        raise NotImplementedError("write_index: IndexFlatIP is not serialised")
    if isinstance(index, IndexPreTransform):
        if str(fname).endswith(".npz"):
            raise NotImplementedError("write_index: IndexPreTransform goes to faiss's binary format only (IxPT)")
        # every transform goes out as `LTra` (a plain LinearTransform: what faiss applies for OPQ / a random rotation at
        # search time -- the OPQm / rrot TYPE of a transform read from a file is not kept)
        tmp = str(fname) + ".sub.tmp"
        try:
            write_index(index.index, tmp, ondisk_data)
            faiss_io.dump_pretransform(fname, index.chain, index.d, index.ntotal, index.is_trained, index.metric_type, tmp)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
        return
    if isinstance(index, IndexRefine):
        raise NotImplementedError("write_index: IndexRefine / IndexRefineFlat (faiss's IxRF) is not serialised -- write "
                                  "index.base_index; the refine stage re-reads the raw vectors at load time")
    if getattr(index, "hnsw_quantizer", False):
        import warnings
        warnings.warn("write_index: this index was read from a file with an IndexHNSWFlat (IHNf) coarse quantiser; it is "
                      "written with a plain IndexFlat quantiser over the same centroids (the HNSW graph is not kept: "
                      "faiss will search the coarse level exactly, as this package does)", stacklevel=2)
    if not str(fname).endswith(".npz"):
        # the C ABI streams the lists from HBM to the file slab by slab (mi_index_save)
        _check(_Lib.get().mi_index_set_nprobe(index._h, max(1, int(index.nprobe))))
        _check(_Lib.get().mi_index_save(index._h, str(fname).encode(),
                                        None if ondisk_data is None else str(ondisk_data).encode()))
        return
    sizes, codes, ids = _lists_of(index)
    trained = index.is_trained
    cent = index.get_centroids() if trained else np.zeros((0,), np.float32)
    cb = index.get_codebook() if trained else np.zeros((0,), np.float32)
    with open(fname, "wb") as f:
        np.savez(f, magic=np.array(_MAGIC),
                 params=np.array([index.d, index.nlist, index.pq.M, index.pq.nbits, index.metric_type,
                                  int(index.by_residual), index.nprobe, int(trained)], np.int64),
                 centroids=cent, codebook=cb, sizes=sizes, codes=codes, ids=ids)


def read_index(fname: str, device: int = 0):
    """``faiss.read_index``: a faiss ``IwPQ`` file (in-file or on-disk lists) or
    the package's own ``.npz`` container, told apart by their first bytes."""
    with open(fname, "rb") as f:
        head = f.read(4)
    if head != b"PK\x03\x04":
        chain, offset = faiss_io.parse_pretransform(fname) if head == b"IxPT" else (None, 0)
        h = c_void_p()
        rc = _Lib.get().mi_index_load_at(str(fname).encode(), int(offset), int(device), ctypes.byref(h))
        if rc:
            msg = _Lib.get().mi_last_error().decode()
            if "no HIP device" in msg:
                raise RuntimeError("mi_ivfpq: " + msg)
            raise faiss_io.FaissFormatError(msg)
        index = IndexIVFPQ._from_handle(h, device)
        # an IndexHNSWFlat quantiser was read as its flat storage (exact coarse search: a superset of the graph's probes);
        # the index remembers it, and write_index says that what it writes is the IVF-Flat-quantiser form
        index.hnsw_quantizer = not faiss_io.parse_is_flat(fname, int(offset))
        return index if chain is None else IndexPreTransform(chain, index)
    z = np.load(fname, allow_pickle=False)
    if str(z["magic"]) != _MAGIC:
        raise ValueError(f"{fname}: not a {_MAGIC} file")
    d, nlist, M, nbits, metric, by_res, nprobe, trained = (int(v) for v in z["params"])
    index = IndexIVFPQ(d, nlist, M, nbits, metric, bool(by_res), device)
    index.nprobe = nprobe
    if trained:
        index.set_centroids(z["centroids"])
        index.set_codebook(z["codebook"])
    sizes = z["sizes"]
    if sizes.sum():
        index.add_codes(np.repeat(np.arange(nlist, dtype=np.int32), sizes), z["codes"], z["ids"])
    return index
