"""Parquet feed for `index fill` (SURVEY 8(f) "next" row 2; reference
Makefile:24-25 `sidecar-search index ... fill`, README.md:60 `dump --shard-size
2097152 --row-group-size 65536`).

The reference's embedding store is a directory of parquet shards written by
`sidecar-search dump`; `fill` streams them into `Index.add` and records the
position -> OpenAlex id map as `ids.parquet` (reference Makefile:11).  The
column schema is sidecar-search's (not in the reference), so column names are
parameters; embeddings may be list<float32|float16> or fixed_size_list.

Row groups (65536 rows in the reference's settings) are read one at a time, so
host memory stays at one row group regardless of the corpus size; the add
itself (coarse assign + PQ encode) runs on the GPU through the index object.
"""
from __future__ import annotations

import glob
import os
from typing import Iterator

import numpy as np


def parquet_files(data_dir: str) -> list[str]:
    if os.path.isfile(data_dir):
        return [data_dir]
    files = sorted(glob.glob(os.path.join(data_dir, "**", "*.parquet"), recursive=True))
    if not files:
        raise FileNotFoundError(f"no parquet files under {data_dir!r}")
    return files


def iter_row_groups(data_dir: str, id_col: str = "id", emb_col: str = "embedding",
                    d: int | None = None) -> Iterator[tuple[list, np.ndarray]]:
    """Yields (ids, embeddings float32 [n, d]) per parquet row group, in file and
    row-group order (the order that defines positions)."""
    import pyarrow as pa
    import pyarrow.parquet as pq
    for f in parquet_files(data_dir):
        pf = pq.ParquetFile(f)
        for rg in range(pf.num_row_groups):
            t = pf.read_row_group(rg, columns=[id_col, emb_col])
            col = t.column(emb_col).combine_chunks()
            n = len(col)
            if pa.types.is_fixed_size_list(col.type):
                width = col.type.list_size
                flat = col.flatten()
            else:
                flat = col.flatten()
                width = len(flat) // max(n, 1)
                if n and width * n != len(flat):
                    raise ValueError(f"{f}: ragged embedding column {emb_col!r}")
            emb = np.asarray(flat.to_numpy(zero_copy_only=False), dtype=np.float32).reshape(n, width)
            if d is not None and n and width != d:
                raise ValueError(f"{f}: embedding width {width} != index.d {d}")
            yield t.column(id_col).to_pylist(), emb


def fill_from_parquet(index, data_dir: str, id_col: str = "id", emb_col: str = "embedding",
                      ids_out: str | None = None, progress=None) -> int:
    """`index fill`: add every embedding under data_dir to `index` (sequential
    positions, like faiss Index.add) and, if ids_out is given, write the
    position -> id table as a one-column parquet file.  Returns the number added."""
    import pyarrow as pa
    import pyarrow.parquet as pq
    if not index.is_trained:
        raise RuntimeError("fill_from_parquet: index is not trained")
    writer = None
    total = 0
    try:
        for ids, emb in iter_row_groups(data_dir, id_col, emb_col, index.d):
            if len(emb):
                index.add(emb)
                total += len(emb)
                if ids_out is not None:
                    # the position -> id table is streamed row group by row group: host memory
                    # stays at one row group whatever the corpus size (207 M ids never sit in a list)
                    t = pa.table({id_col: ids})
                    if writer is None:
                        writer = pq.ParquetWriter(ids_out, t.schema)
                    writer.write_table(t, row_group_size=65536)
            if progress:
                progress(total)
        if ids_out is not None and writer is None:                      # empty input: an empty table
            pq.write_table(pa.table({id_col: pa.array([], pa.string())}), ids_out)
    finally:
        if writer is not None:
            writer.close()
    return total
