"""faiss binary index format, host-side parsing only: no GPU.

faiss is not importable here, so the byte layout is pinned to a hand-assembled
file that follows faiss's documented serialisation field by field
(faiss/impl/index_write.cpp: write_index_header, write_ivf_header,
write_ProductQuantizer, write_InvertedLists; invlists/OnDiskInvertedLists.cpp),
plus write -> read round trips of the package's own writer.
"""
import importlib
import struct

import numpy as np
import pytest

fio = importlib.import_module("abstracts_search_amd.faiss_io")


def tiny(seed=0, d=8, nlist=4, M=2, sizes=(3, 0, 2, 1)):
    rng = np.random.default_rng(seed)
    cent = rng.standard_normal((nlist, d)).astype(np.float32)
    cb = rng.standard_normal((M, 256, d // M)).astype(np.float32)
    sizes = np.array(sizes, np.int64)
    n = int(sizes.sum())
    codes = rng.integers(0, 256, (n, M)).astype(np.uint8)
    ids = rng.integers(0, 1 << 40, n).astype(np.int64)
    return dict(d=d, nlist=nlist, M=M, nbits=8, metric=fio.METRIC_INNER_PRODUCT, by_residual=True, nprobe=3,
                is_trained=True, centroids=cent, codebook=cb, sizes=sizes, codes=codes, ids=ids)


def header(d, ntotal, trained, metric):
    return struct.pack("<iqqqBi", d, ntotal, 1 << 20, 1 << 20, int(trained), metric)


def hand_assembled(t, lists="full"):
    """The file faiss's write_index would produce for `t`, field by field."""
    b = b"IwPQ" + header(t["d"], int(t["sizes"].sum()), True, 0)
    b += struct.pack("<QQ", t["nlist"], t["nprobe"])
    b += b"IxFI" + header(t["d"], t["nlist"], True, 0)
    b += struct.pack("<Q", t["centroids"].size) + t["centroids"].tobytes()
    b += struct.pack("<b", 0) + struct.pack("<Q", 0)                      # direct map: NoMap, empty array
    b += struct.pack("<BQ", 1, t["M"])                                    # by_residual, code_size
    b += struct.pack("<QQQ", t["d"], t["M"], 8)
    b += struct.pack("<Q", t["codebook"].size) + t["codebook"].tobytes()
    b += b"ilar" + struct.pack("<QQ", t["nlist"], t["M"])
    if lists == "full":
        b += b"full" + struct.pack("<Q", t["nlist"]) + t["sizes"].astype("<u8").tobytes()
    else:
        nz = np.flatnonzero(t["sizes"])
        b += b"sprs" + struct.pack("<Q", 2 * nz.size)
        for l in nz:
            b += struct.pack("<QQ", l, t["sizes"][l])
    o = 0
    for l in range(t["nlist"]):
        k = int(t["sizes"][l])
        b += t["codes"][o:o + k].tobytes() + t["ids"][o:o + k].tobytes()
        o += k
    return b


def same(z, t):
    for key in ("d", "nlist", "M", "nbits", "metric", "by_residual", "nprobe", "is_trained"):
        assert z[key] == t[key], key
    assert np.array_equal(z["centroids"], t["centroids"]) and np.array_equal(z["codebook"], t["codebook"])
    assert np.array_equal(z["sizes"], t["sizes"])
    assert np.array_equal(z["codes"], t["codes"]) and np.array_equal(z["ids"], t["ids"])


@pytest.mark.parametrize("lists", ["full", "sprs"])
def test_reader_follows_the_documented_layout(tmp_path, lists):
    t = tiny()
    f = tmp_path / "index.faiss"
    f.write_bytes(hand_assembled(t, lists))
    same(fio.parse(str(f)), t)


def test_writer_emits_the_documented_layout(tmp_path):
    t = tiny()                                  # 3 of 4 lists non-empty -> "full"
    f = tmp_path / "index.faiss"
    fio.dump(str(f), **t)
    assert f.read_bytes() == hand_assembled(t, "full")
    t = tiny(sizes=(5, 0, 0, 0))                # 1 of 4 -> "sprs"
    fio.dump(str(f), **t)
    assert f.read_bytes() == hand_assembled(t, "sprs")


def test_ondisk_lists_roundtrip(tmp_path):
    t = tiny(seed=3, nlist=6, sizes=(4, 0, 1, 7, 0, 2))
    f, data = tmp_path / "index.faiss", tmp_path / "ondisk.ivfdata"
    fio.dump(str(f), ondisk_data=str(data), **t)
    assert data.stat().st_size == int(t["sizes"].sum()) * (t["M"] + 8)
    raw = f.read_bytes()
    assert b"ilod" in raw and b"ondisk.ivfdata" in raw
    same(fio.parse(str(f)), t)
    # capacity > size and a non-zero first offset, as a grown on-disk list file has them
    recs, pos = [], 16
    blob = bytearray(b"\xee" * 16)
    o = 0
    for l in range(t["nlist"]):
        k = int(t["sizes"][l]); cap = k + 3
        recs += [k, cap, pos]
        blob += t["codes"][o:o + k].tobytes() + b"\x00" * (3 * t["M"])
        blob += t["ids"][o:o + k].tobytes() + b"\x00" * 24
        pos += cap * (t["M"] + 8); o += k
    data.write_bytes(bytes(blob))
    head = raw[:raw.index(b"ilod")]
    tail = b"ilod" + struct.pack("<QQ", t["nlist"], t["M"]) + struct.pack("<Q", t["nlist"])
    tail += struct.pack(f"<{len(recs)}Q", *recs) + struct.pack("<Q", 0)
    name = b"/some/other/box/ondisk.ivfdata"       # stale absolute path: the sibling file is used
    tail += struct.pack("<Q", len(name)) + name + struct.pack("<Q", pos)
    f.write_bytes(head + tail)
    same(fio.parse(str(f)), t)


def test_untrained_and_empty(tmp_path):
    t = tiny(sizes=(0, 0, 0, 0))
    f = tmp_path / "e.faiss"
    fio.dump(str(f), **t)
    z = fio.parse(str(f))
    assert z["ntotal"] == 0 and z["codes"].shape == (0, t["M"])


def test_rejects_what_it_does_not_understand(tmp_path):
    t = tiny()
    good = hand_assembled(t)
    f = tmp_path / "bad.faiss"
    f.write_bytes(b"IxFI" + good[4:])
    with pytest.raises(fio.FaissFormatError, match="not an IndexIVFPQ"):
        fio.parse(str(f))
    f.write_bytes(good[:100])
    with pytest.raises(fio.FaissFormatError, match="truncated"):
        fio.parse(str(f))
    f.write_bytes(good.replace(b"ilar", b"ilzz"))
    with pytest.raises(fio.FaissFormatError, match="not supported"):
        fio.parse(str(f))
    bad_total = b"IwPQ" + header(t["d"], 99, True, 0) + good[4 + len(header(0, 0, True, 0)):]
    f.write_bytes(bad_total)
    with pytest.raises(fio.FaissFormatError, match="header says 99"):
        fio.parse(str(f))


def test_pretransform_chain_is_parsed(tmp_path):
    """[PRIOR layout] an "IxPT" file -- header, chain of LinearTransforms (what OPQ is written as), then the IwPQ record:
    parse_pretransform returns the matrices and the sub-index's offset, and what follows at that offset parses as the
    index written alone.  (reference Makefile:12-13: whatever `sidecar-search index train` wrapped the index in.)"""
    import struct
    from abstracts_search_amd import faiss_io as fio
    rng = np.random.default_rng(5)
    d, d_in = 16, 24
    A = rng.standard_normal((d, d_in)).astype(np.float32)
    b = rng.standard_normal(d).astype(np.float32)
    sub = str(tmp_path / "sub.faiss")
    cent = rng.standard_normal((4, d)).astype(np.float32)
    cb = rng.standard_normal((4, 256, 4)).astype(np.float32)
    sizes = np.array([2, 0, 1, 0])
    codes = rng.integers(0, 256, (3, 4)).astype(np.uint8)
    ids = np.array([5, 6, 7], np.int64)
    fio.dump(sub, d=d, nlist=4, M=4, nbits=8, metric=0, by_residual=True, nprobe=2, is_trained=True, centroids=cent, codebook=cb,
             sizes=sizes, codes=codes, ids=ids)
    f = str(tmp_path / "opq.faiss")
    with open(f, "wb") as fh:
        fh.write(b"IxPT")
        fh.write(struct.pack("<iqqqBi", d_in, 3, 1 << 20, 1 << 20, 1, 0))
        fh.write(struct.pack("<i", 2))
        for M_, b_ in ((A, b), (np.eye(d, dtype=np.float32), None)):
            fh.write(b"LTra" if b_ is not None else b"rrot")
            fh.write(struct.pack("<B", b_ is not None))
            fh.write(struct.pack("<Q", M_.size)); fh.write(M_.tobytes())
            bb = b_ if b_ is not None else np.zeros(0, np.float32)
            fh.write(struct.pack("<Q", bb.size)); fh.write(bb.tobytes())
            fh.write(struct.pack("<iiB", M_.shape[1], M_.shape[0], 1))
        fh.write(open(sub, "rb").read())
    chain, off = fio.parse_pretransform(f)
    assert len(chain) == 2 and np.array_equal(chain[0][0], A) and np.array_equal(chain[0][1], b) and chain[1][1] is None
    assert open(f, "rb").read()[off:] == open(sub, "rb").read()
    assert fio.parse_pretransform(sub) == (None, 0)
    with open(f, "r+b") as fh:                                  # a transform this build does not implement: a named error
        fh.seek(4 + 33 + 4)
        fh.write(b"PcAm")
    with pytest.raises(fio.FaissFormatError, match="PcAm"):
        fio.parse_pretransform(f)


def hnsw_record(rng, nlist, levels=3):
    """[PRIOR: faiss write_HNSW] a plausible graph for nlist points: the five vectors and five ints, contents arbitrary --
    the reader skips them"""
    probas = np.array([0.9, 0.09, 0.01][:levels], np.float64)
    cum = np.array([0, 64, 96, 128][:levels + 1], np.int32)
    lev = rng.integers(1, levels + 1, nlist).astype(np.int32)
    offsets = np.concatenate([[0], np.cumsum(cum[lev])]).astype(np.uint64)
    neigh = rng.integers(-1, nlist, int(offsets[-1])).astype(np.int32)
    b = b""
    for v in (probas, cum, lev, offsets, neigh):
        b += struct.pack("<Q", v.size) + v.tobytes()
    return b + struct.pack("<iiiii", 3, levels - 1, 40, 16, 1)


def with_hnsw_quantizer(t, raw: bytes) -> bytes:
    """the IwPQ bytes of hand_assembled(t) with the IndexFlatIP quantiser wrapped in an IndexHNSWFlat record (IHNf: header,
    graph, then the flat storage index): what "IVF<n>_HNSW32,PQ<M>" writes"""
    flat_at = raw.index(b"IxFI")
    flat_len = 4 + 33 + 8 + t["centroids"].size * 4
    rng = np.random.default_rng(3)
    return (raw[:flat_at] + b"IHNf" + header(t["d"], t["nlist"], True, 0) + hnsw_record(rng, t["nlist"]) +
            raw[flat_at:flat_at + flat_len] + raw[flat_at + flat_len:])


def test_hnsw_coarse_quantiser_is_read_as_its_flat_storage(tmp_path):
    """[PRIOR layout] reference Makefile:39 passes no factory string to `index train`: if sidecar-search's default is
    "IVF65536_HNSW32,...", index.faiss holds an IndexHNSWFlat in front of the lists.  The reader takes its flat storage as
    the centroid table (searched exactly: a superset of what the graph would probe) and skips the graph."""
    t = tiny()
    f = tmp_path / "hnsw.faiss"
    f.write_bytes(with_hnsw_quantizer(t, hand_assembled(t, "full")))
    z = fio.parse(str(f))
    same(z, t)
    assert z["hnsw_quantizer"] is True and fio.parse_is_flat(str(f)) is False
    g = tmp_path / "flat.faiss"
    g.write_bytes(hand_assembled(t, "full"))
    assert fio.parse(str(g))["hnsw_quantizer"] is False
    bad = with_hnsw_quantizer(t, hand_assembled(t, "full")).replace(b"IHNf", b"IHNp")
    f.write_bytes(bad)
    with pytest.raises(fio.FaissFormatError, match="IHNp.*compressed storage"):
        fio.parse(str(f))


def test_linear_transform_longer_than_its_shape_is_accepted(tmp_path):
    """faiss's writer only asserts A.size() >= d_in * d_out and b.size() >= d_out: the surplus is ignored, not an error"""
    rng = np.random.default_rng(9)
    d_in, d_out = 12, 8
    A = rng.standard_normal(d_in * d_out + 5).astype(np.float32)
    b = rng.standard_normal(d_out + 3).astype(np.float32)
    f = tmp_path / "pt.faiss"
    with open(f, "wb") as fh:
        fh.write(b"IxPT" + header(d_in, 0, True, 0) + struct.pack("<i", 1))
        fh.write(b"LTra" + struct.pack("<B", 1) + struct.pack("<Q", A.size) + A.tobytes() + struct.pack("<Q", b.size) + b.tobytes())
        fh.write(struct.pack("<iiB", d_in, d_out, 1))
        fh.write(b"IwPQ")
    chain, off = fio.parse_pretransform(str(f))
    assert np.array_equal(chain[0][0], A[:d_in * d_out].reshape(d_out, d_in)) and np.array_equal(chain[0][1], b[:d_out])
    assert f.read_bytes()[off:] == b"IwPQ"
    # and the writer is the reader's inverse
    sub = tmp_path / "sub.bin"
    sub.write_bytes(b"IwPQ-and-the-rest")
    g = tmp_path / "out.faiss"
    fio.dump_pretransform(str(g), chain, d_in, 7, True, 0, str(sub))
    chain2, off2 = fio.parse_pretransform(str(g))
    assert np.array_equal(chain2[0][0], chain[0][0]) and np.array_equal(chain2[0][1], chain[0][1])
    assert g.read_bytes()[off2:] == b"IwPQ-and-the-rest"
