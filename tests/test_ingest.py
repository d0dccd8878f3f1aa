"""Parquet feed of `index fill`: row-group streaming (CPU) and the full fill (GPU)."""
import os

import numpy as np
import pytest


def _write_shards(tmp_path, x, ids, rows_per_file=700, row_group=256, half=False, fixed=True):
    import pyarrow as pa
    import pyarrow.parquet as pq
    d = x.shape[1]
    root = tmp_path / "data"
    root.mkdir()
    for s, lo in enumerate(range(0, len(x), rows_per_file)):
        part = x[lo:lo + rows_per_file]
        vals = pa.array(part.reshape(-1).astype(np.float16 if half else np.float32))
        emb = pa.FixedSizeListArray.from_arrays(vals, d) if fixed else pa.ListArray.from_arrays(
            pa.array(np.arange(0, len(part) * d + 1, d, dtype=np.int32)), vals)
        pq.write_table(pa.table({"id": ids[lo:lo + rows_per_file], "embedding": emb}),
                       str(root / f"shard-{s:03d}.parquet"), row_group_size=row_group)
    return str(root)


@pytest.mark.parametrize("half,fixed", [(False, True), (True, True), (False, False)])
def test_iter_row_groups_order_and_dtype(tmp_path, half, fixed):
    from abstracts_search_amd.ingest import iter_row_groups
    rng = np.random.default_rng(1)
    x = rng.standard_normal((1500, 16)).astype(np.float32)
    ids = [f"https://openalex.org/W{i}" for i in range(1500)]
    root = _write_shards(tmp_path, x, ids, half=half, fixed=fixed)
    got_ids, got = [], []
    ngroups = 0
    for i, e in iter_row_groups(root, d=16):
        assert e.dtype == np.float32 and e.shape[1] == 16 and len(i) == len(e) <= 256
        got_ids += i
        got.append(e)
        ngroups += 1
    assert ngroups == 7                       # files of 700, 700, 100 rows -> 3 + 3 + 1 row groups
    assert got_ids == ids
    ref = x.astype(np.float16).astype(np.float32) if half else x
    assert np.array_equal(np.concatenate(got), ref)
    with pytest.raises(ValueError):
        list(iter_row_groups(root, d=32))


@pytest.mark.gpu
def test_fill_from_parquet_matches_direct_add(tmp_path):
    import pyarrow.parquet as pq
    import abstracts_search_amd.faiss as faiss
    from abstracts_search_amd.ingest import fill_from_parquet
    rng = np.random.default_rng(2)
    d, nlist, M, n = 64, 16, 8, 3000
    cent = rng.standard_normal((nlist, d)).astype(np.float32)
    cb = (0.3 * rng.standard_normal((M, 256, d // M))).astype(np.float32)
    x = (cent[rng.integers(0, nlist, n)] + 0.3 * rng.standard_normal((n, d))).astype(np.float32)
    ids = [f"W{7 * i}" for i in range(n)]
    root = _write_shards(tmp_path, x, ids)

    def make():
        idx = faiss.IndexIVFPQ(d, nlist, M, 8, faiss.METRIC_INNER_PRODUCT)
        idx.set_centroids(cent)
        idx.set_codebook(cb)
        return idx

    a, b = make(), make()
    out = str(tmp_path / "ids.parquet")
    assert fill_from_parquet(a, root, ids_out=out) == n
    b.add(x)
    a.nprobe = b.nprobe = 4
    q = x[:20]
    Da, Ia = a.search(q, 10)
    Db, Ib = b.search(q, 10)
    assert np.array_equal(Ia, Ib) and np.array_equal(Da, Db)
    assert pq.read_table(out).column("id").to_pylist() == ids     # position -> id map
