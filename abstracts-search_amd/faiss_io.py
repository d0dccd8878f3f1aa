"""faiss's binary index format for the one index family of the hot path:
``IndexIVFPQ`` over an ``IndexFlat`` (or ``IndexHNSWFlat``: read as its flat storage) coarse quantiser, with in-file
(``ArrayInvertedLists``) or on-disk (``OnDiskInvertedLists``) lists.

This is the artefact the reference pipeline publishes and consumes --
``index.faiss`` + ``ondisk.ivfdata`` (reference Makefile:11-12, README.md:10) --
SURVEY section 8(f) row 1.  The byte layout below restates faiss's documented
serialisation (``faiss/impl/index_write.cpp`` / ``index_read.cpp`` /
``invlists/OnDiskInvertedLists.cpp``); faiss itself is absent from this image,
so the layout is UNVALIDATED against files written by faiss: the tests pin it to
hand-assembled byte strings and to write -> read round trips only.

Pure host-side parsing (numpy + struct); the product arithmetic stays in the
HIP library -- this module only moves bytes into ``IndexIVFPQ.set_centroids /
set_codebook / add_codes``.

Layout (little endian; ``size_t``/``idx_t`` = 8 bytes, ``int`` = 4, ``bool`` = 1)::

    IwPQ                                   fourcc of IndexIVFPQ
      header: d:int ntotal:i64 dummy:i64 dummy:i64 is_trained:bool metric:int [metric_arg:f32 if metric > 1]
      nlist:size_t nprobe:size_t
      quantizer: IxFI | IxF2               IndexFlatIP / IndexFlatL2
        header (as above)
        n_floats:size_t  f32[n_floats]     the centroids, row-major [nlist][d]
      direct map: type:char  n:size_t i64[n]
      by_residual:bool code_size:size_t
      pq: d:size_t M:size_t nbits:size_t  n:size_t f32[n]    codebook [M][2^nbits][d/M]
      inverted lists:
        ilar nlist:size_t code_size:size_t
             full n:size_t size_t[nlist]            (sizes)   or
             sprs n:size_t size_t[2 * non_empty]    (list, size) pairs
             per non-empty list: u8[size * code_size] i64[size]
        ilod nlist:size_t code_size:size_t
             n:size_t {size, capacity, offset}:size_t[3][n]
             n:size_t {offset, capacity}:size_t[2][n]          (free slots)
             n:size_t char[n]                                  (data file name)
             totsize:size_t
             data file: per list at `offset`: u8[capacity * code_size] i64[capacity]
        il00                                   no lists
"""
from __future__ import annotations

import io
import os
import struct

import numpy as np

METRIC_INNER_PRODUCT = 0
METRIC_L2 = 1


class FaissFormatError(ValueError):
    pass


# ----------------------------------------------------------------------
# primitive readers / writers
# ----------------------------------------------------------------------
class _Reader:
    def __init__(self, f, name: str):
        self.f, self.name = f, name

    def raw(self, n: int) -> bytes:
        b = self.f.read(n)
        if len(b) != n:
            raise FaissFormatError(f"{self.name}: truncated file (wanted {n} bytes, got {len(b)})")
        return b

    def fourcc(self) -> str:
        return self.raw(4).decode("latin-1")

    def one(self, fmt: str):
        return struct.unpack("<" + fmt, self.raw(struct.calcsize("<" + fmt)))[0]

    def vector(self, dtype, count_scale: int = 1) -> np.ndarray:
        """size_t count followed by count*count_scale items of dtype."""
        n = self.one("Q") * count_scale
        dt = np.dtype(dtype)
        if n > (1 << 40):
            raise FaissFormatError(f"{self.name}: implausible vector length {n}")
        return np.frombuffer(self.raw(n * dt.itemsize), dtype=dt).copy()


def _w(f, fmt: str, *vals) -> None:
    f.write(struct.pack("<" + fmt, *vals))


def _wvec(f, a: np.ndarray, count_scale: int = 1) -> None:
    a = np.ascontiguousarray(a)
    assert a.size % count_scale == 0
    _w(f, "Q", a.size // count_scale)
    f.write(a.tobytes())


def _read_header(r: _Reader):
    d = r.one("i")
    ntotal = r.one("q")
    r.one("q")
    r.one("q")
    trained = bool(r.one("B"))
    metric = r.one("i")
    if metric > 1:
        r.one("f")
    return d, ntotal, trained, metric


def _write_header(f, d: int, ntotal: int, trained: bool, metric: int) -> None:
    _w(f, "iqqqBi", d, ntotal, 1 << 20, 1 << 20, int(trained), metric)


# ----------------------------------------------------------------------
# reading
# ----------------------------------------------------------------------
def parse_pretransform(fname: str):
    """[PRIOR: faiss index_write.cpp as remembered; no faiss-written file has been read on any box of this pool]
    An IndexPreTransform file: fourcc "IxPT", the index header, int32 chain length, then per VectorTransform a fourcc
    ("LTra": LinearTransform / OPQMatrix, "rrot": RandomRotationMatrix), uint8 have_bias, vector<float> A [d_out x d_in],
    vector<float> b, int32 d_in, int32 d_out, uint8 is_trained; then the sub-index.  Returns (chain, offset) with
    chain = [(A [d_out, d_in] float32, b [d_out] float32 or None), ...] and the byte offset of the sub-index, or
    (None, 0) when the file does not start with IxPT.  Other transforms (PCA with eigenvalues, ITQ, norm) are rejected."""
    with open(fname, "rb") as fh:
        r = _Reader(fh, fname)
        if r.fourcc() != "IxPT":
            return None, 0
        _read_header(r)
        nt = r.one("i")
        if not 0 <= nt <= 8:
            raise FaissFormatError(f"{fname}: implausible pre-transform chain length {nt}")
        chain = []
        for _ in range(nt):
            cc = r.fourcc()
            if cc not in ("LTra", "rrot"):
                raise FaissFormatError(f"{fname}: vector transform {cc!r} is not supported (LinearTransform / OPQMatrix / RandomRotationMatrix only)")
            have_bias = bool(r.one("B"))
            A = r.vector(np.float32)
            b = r.vector(np.float32)
            d_in, d_out = r.one("i"), r.one("i")
            r.one("B")
            # faiss's writer only guarantees A.size() >= d_in * d_out and b.size() >= d_out: longer vectors are legal
            if d_in <= 0 or d_out <= 0 or A.size < d_in * d_out or (have_bias and b.size < d_out):
                raise FaissFormatError(f"{fname}: transform {cc} is {d_out} x {d_in} but holds {A.size} + {b.size} floats")
            chain.append((A[:d_in * d_out].reshape(d_out, d_in).copy(), b[:d_out].copy() if have_bias else None))
        return chain, fh.tell()


def dump_pretransform(fname: str, chain, d: int, ntotal: int, trained: bool, metric: int, sub_index_file: str) -> None:
    """Write an IndexPreTransform file ([PRIOR] layout, the inverse of parse_pretransform): "IxPT", the index header (d = the
    chain's input width), the chain as "LTra" records, then the bytes of `sub_index_file` (the wrapped index, already
    written)."""
    with open(fname, "wb") as f:
        f.write(b"IxPT")
        _write_header(f, d, ntotal, trained, metric)
        _w(f, "i", len(chain))
        for A, b in chain:
            A = np.ascontiguousarray(A, np.float32)
            f.write(b"LTra")
            _w(f, "B", int(b is not None))
            _wvec(f, A.reshape(-1))
            _wvec(f, np.zeros(0, np.float32) if b is None else np.ascontiguousarray(b, np.float32))
            _w(f, "iiB", A.shape[1], A.shape[0], 1)
        with open(sub_index_file, "rb") as sf:
            while True:
                blk = sf.read(1 << 24)
                if not blk:
                    break
                f.write(blk)


def _skip_hnsw(r: "_Reader") -> None:
    """[PRIOR: faiss write_HNSW] the graph of an IndexHNSW record: five vectors (assign_probas f64, cum_nneighbor_per_level
    i32, levels i32, offsets u64, neighbors i32) and five ints (entry_point, max_level, efConstruction, efSearch, the
    deprecated upper_beam)."""
    for dt in (np.float64, np.int32, np.int32, np.uint64, np.int32):
        n = r.one("Q")
        if n > (1 << 40):
            raise FaissFormatError(f"{r.name}: implausible HNSW vector length {n}")
        r.f.seek(n * np.dtype(dt).itemsize, os.SEEK_CUR)
    r.raw(20)


def parse(fname: str) -> dict:
    """Parse an IndexIVFPQ file into plain arrays (no GPU needed):
    d nlist M nbits metric by_residual nprobe is_trained ntotal,
    centroids [nlist,d], codebook [M,ksub,dsub], sizes [nlist],
    codes [n,M] and ids [n] concatenated in list order."""
    with open(fname, "rb") as fh:
        r = _Reader(fh, fname)
        h = r.fourcc()
        if h in ("IvPQ", "IvQR", "IwQR"):
            raise FaissFormatError(f"{fname}: {h} (legacy / refined IVFPQ) is not supported, only IwPQ")
        if h != "IwPQ":
            raise FaissFormatError(f"{fname}: fourcc {h!r} is not an IndexIVFPQ (IwPQ)")
        d, ntotal, trained, metric = _read_header(r)
        nlist, nprobe = r.one("Q"), r.one("Q")
        qh = r.fourcc()
        hnsw_quantizer = qh == "IHNf"
        if hnsw_quantizer:
            # "IVF65536_HNSW32,...": an IndexHNSWFlat in front of the lists.  Its flat storage is the centroid table; it is
            # searched EXACTLY here (what the graph approximates), the graph is skipped
            _read_header(r)
            _skip_hnsw(r)
            qh = r.fourcc()
        elif qh in ("IHNp", "IHNs", "IHN2", "IHNc"):
            raise FaissFormatError(f"{fname}: coarse quantiser {qh!r} (HNSW over compressed storage) holds no exact centroid "
                                   "table and is not supported; IHNf (IndexHNSWFlat) is")
        if qh not in ("IxFI", "IxF2", "IxFl"):
            raise FaissFormatError(f"{fname}: coarse quantiser {qh!r} is not an IndexFlat")
        qd, qn, _qt, _qm = _read_header(r)
        cent = r.vector(np.float32)
        if qd != d or cent.size != qn * d:
            raise FaissFormatError(f"{fname}: quantiser shape mismatch ({qn} x {qd}, {cent.size} floats)")
        if trained and qn != nlist:
            raise FaissFormatError(f"{fname}: quantiser holds {qn} centroids, nlist is {nlist}")
        r.one("b")                    # direct map type
        dm = r.vector(np.int64)
        del dm
        by_residual = bool(r.one("B"))
        code_size = r.one("Q")
        pd, M, nbits = r.one("Q"), r.one("Q"), r.one("Q")
        cb = r.vector(np.float32)
        if pd != d or M == 0 or d % M or nbits != 8 or code_size != M:
            raise FaissFormatError(f"{fname}: unsupported PQ (d={pd}, M={M}, nbits={nbits}, code_size={code_size})")
        ksub, dsub = 1 << nbits, d // M
        if cb.size not in (0, M * ksub * dsub):
            raise FaissFormatError(f"{fname}: PQ codebook has {cb.size} floats, expected {M * ksub * dsub}")
        sizes, codes, ids = _read_invlists(r, fname, nlist, code_size)
    if int(sizes.sum()) != ntotal:
        raise FaissFormatError(f"{fname}: lists hold {int(sizes.sum())} vectors, header says {ntotal}")
    return dict(d=d, nlist=nlist, M=M, nbits=nbits, metric=metric, by_residual=by_residual, nprobe=nprobe,
                is_trained=trained, ntotal=ntotal, centroids=cent.reshape(qn, d), hnsw_quantizer=hnsw_quantizer,
                codebook=cb.reshape(M, ksub, dsub) if cb.size else cb, sizes=sizes, codes=codes, ids=ids)


def quantizer_fourcc(fname: str, offset: int = 0) -> str:
    """fourcc of the coarse quantiser of the IwPQ index that starts at byte `offset` of the file -- the headers only
    (nothing proportional to the index is read)."""
    with open(fname, "rb") as fh:
        fh.seek(offset)
        r = _Reader(fh, fname)
        if r.fourcc() != "IwPQ":
            raise FaissFormatError(f"{fname}: not an IndexIVFPQ (IwPQ) at byte {offset}")
        _read_header(r)
        r.one("Q"), r.one("Q")
        return r.fourcc()


def parse_is_flat(fname: str, offset: int = 0) -> bool:
    """whether the file's coarse quantiser is a plain IndexFlat (False: an IndexHNSWFlat read as its flat storage)"""
    return quantizer_fourcc(fname, offset) != "IHNf"


def _read_invlists(r: _Reader, fname: str, nlist: int, code_size: int):
    h = r.fourcc()
    if h == "il00":
        return np.zeros(nlist, np.int64), np.zeros((0, code_size), np.uint8), np.zeros(0, np.int64)
    if h == "ilar":
        nl, cs = r.one("Q"), r.one("Q")
        if nl != nlist or cs != code_size:
            raise FaissFormatError(f"{fname}: inverted lists are {nl} x {cs} B, index says {nlist} x {code_size} B")
        kind = r.fourcc()
        sizes = np.zeros(nlist, np.int64)
        if kind == "full":
            s = r.vector(np.uint64)
            if s.size != nlist:
                raise FaissFormatError(f"{fname}: {s.size} list sizes for {nlist} lists")
            sizes[:] = s
        elif kind == "sprs":
            s = r.vector(np.uint64).reshape(-1, 2)
            sizes[s[:, 0].astype(np.int64)] = s[:, 1]
        else:
            raise FaissFormatError(f"{fname}: list size encoding {kind!r}")
        n = int(sizes.sum())
        codes = np.empty((n, code_size), np.uint8)
        ids = np.empty(n, np.int64)
        o = 0
        for l in range(nlist):
            k = int(sizes[l])
            if k:
                codes[o:o + k] = np.frombuffer(r.raw(k * code_size), np.uint8).reshape(k, code_size)
                ids[o:o + k] = np.frombuffer(r.raw(k * 8), np.int64)
                o += k
        return sizes, codes, ids
    if h == "ilod":
        nl, cs = r.one("Q"), r.one("Q")
        if nl != nlist or cs != code_size:
            raise FaissFormatError(f"{fname}: inverted lists are {nl} x {cs} B, index says {nlist} x {code_size} B")
        lists = r.vector(np.uint64, 3).reshape(-1, 3)     # size, capacity, offset
        r.vector(np.uint64, 2)                            # free slots
        data_name = r.vector(np.uint8).tobytes().decode("utf-8", "replace")
        r.one("Q")                                        # totsize
        if lists.shape[0] != nlist:
            raise FaissFormatError(f"{fname}: {lists.shape[0]} on-disk list records for {nlist} lists")
        # faiss stores the path it was written with; the file travels next to the index
        cand = [data_name, os.path.join(os.path.dirname(os.path.abspath(fname)), os.path.basename(data_name))]
        path = next((c for c in cand if c and os.path.exists(c)), None)
        if path is None:
            raise FaissFormatError(f"{fname}: on-disk list data {data_name!r} not found (looked in {cand})")
        sizes = lists[:, 0].astype(np.int64)
        n = int(sizes.sum())
        codes = np.empty((n, code_size), np.uint8)
        ids = np.empty(n, np.int64)
        data = np.memmap(path, dtype=np.uint8, mode="r")
        o = 0
        for l in range(nlist):
            k, cap, off = int(lists[l, 0]), int(lists[l, 1]), int(lists[l, 2])
            if k:
                if off + cap * (code_size + 8) > data.size or k > cap:
                    raise FaissFormatError(f"{path}: list {l} ({k}/{cap} at {off}) runs past the end of the file")
                codes[o:o + k] = data[off:off + k * code_size].reshape(k, code_size)
                ids[o:o + k] = np.frombuffer(data[off + cap * code_size:off + cap * code_size + k * 8].tobytes(),
                                             np.int64)
                o += k
        return sizes, codes, ids
    raise FaissFormatError(f"{fname}: inverted lists {h!r} are not supported (ilar, ilod, il00)")


# ----------------------------------------------------------------------
# writing
# ----------------------------------------------------------------------
def dump(fname: str, *, d: int, nlist: int, M: int, nbits: int, metric: int, by_residual: bool, nprobe: int,
         is_trained: bool, centroids, codebook, sizes, codes, ids, ondisk_data: str | None = None) -> None:
    """Write an IndexIVFPQ file.  ``codes``/``ids`` are concatenated in list
    order, ``sizes[l]`` entries per list.  With ``ondisk_data`` the lists go to
    that file (OnDiskInvertedLists, capacity = size) and the index refers to it."""
    if nbits != 8:
        raise FaissFormatError("only nbits = 8 is supported")
    sizes = np.ascontiguousarray(sizes, np.int64)
    codes = np.ascontiguousarray(codes, np.uint8).reshape(-1, M)
    ids = np.ascontiguousarray(ids, np.int64)
    ntotal = int(sizes.sum())
    assert codes.shape[0] == ntotal and ids.shape[0] == ntotal and sizes.shape[0] == nlist
    cent = np.ascontiguousarray(centroids, np.float32).reshape(-1)
    cb = np.ascontiguousarray(codebook, np.float32).reshape(-1)
    qn = cent.size // d if d else 0
    buf = io.BytesIO()
    buf.write(b"IwPQ")
    _write_header(buf, d, ntotal, is_trained, metric)
    _w(buf, "QQ", nlist, nprobe)
    buf.write(b"IxFI" if metric == METRIC_INNER_PRODUCT else b"IxF2")
    _write_header(buf, d, qn, True, metric)
    _wvec(buf, cent)
    _w(buf, "b", 0)                       # DirectMap::NoMap
    _wvec(buf, np.zeros(0, np.int64))
    _w(buf, "BQ", int(by_residual), M)
    _w(buf, "QQQ", d, M, nbits)
    _wvec(buf, cb)
    offs = np.zeros(nlist + 1, np.int64)
    np.cumsum(sizes, out=offs[1:])
    if ondisk_data is None:
        buf.write(b"ilar")
        _w(buf, "QQ", nlist, M)
        nz = np.flatnonzero(sizes)
        if nz.size > nlist // 2:
            buf.write(b"full")
            _wvec(buf, sizes.astype(np.uint64))
        else:
            buf.write(b"sprs")
            _wvec(buf, np.stack([nz, sizes[nz]], 1).astype(np.uint64).reshape(-1))
        for l in nz:
            buf.write(codes[offs[l]:offs[l + 1]].tobytes())
            buf.write(ids[offs[l]:offs[l + 1]].tobytes())
    else:
        rec = np.zeros((nlist, 3), np.uint64)
        pos = 0
        with open(ondisk_data, "wb") as df:
            for l in range(nlist):
                k = int(sizes[l])
                rec[l] = (k, k, pos)
                df.write(codes[offs[l]:offs[l + 1]].tobytes())
                df.write(ids[offs[l]:offs[l + 1]].tobytes())
                pos += k * (M + 8)
        buf.write(b"ilod")
        _w(buf, "QQ", nlist, M)
        _wvec(buf, rec.reshape(-1), 3)
        _wvec(buf, np.zeros(0, np.uint64), 2)
        _wvec(buf, np.frombuffer(os.path.basename(ondisk_data).encode(), np.uint8))
        _w(buf, "Q", pos)
    with open(fname, "wb") as f:
        f.write(buf.getvalue())
