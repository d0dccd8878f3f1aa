"""CPU tests of the drop-in boundary: the C-ABI libraries load and export every
symbol include/*.h declares; without a GPU they fail loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from conftest import HAS_GPU, ROOT

HEADERS = {"mi_ivfpq.h": "ivfpq", "mi_encoder.h": "encoder"}


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", txt)))


@pytest.mark.parametrize("header", sorted(HEADERS))
def test_library_exports_every_declared_symbol(header):
    if not os.path.exists(os.path.join(ROOT, "include", header)):
        pytest.skip(f"{header} not present yet")
    import abstracts_search_amd._native as nat
    nat.build(HEADERS[header])
    lib = ctypes.CDLL(nat.lib_path(HEADERS[header]))
    names = _declared(header)
    assert len(names) >= 5
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/{header} but not exported"


@pytest.mark.skipif(HAS_GPU, reason="checks the no-GPU failure mode")
def test_no_cpu_fallback_without_gpu():
    import abstracts_search_amd.faiss as faiss
    with pytest.raises(RuntimeError, match="no HIP device"):
        faiss.IndexFlatIP(64)
    with pytest.raises(RuntimeError, match="no HIP device"):
        faiss.index_factory(64, "IVF16,PQ8", faiss.METRIC_INNER_PRODUCT)


def test_index_factory_argument_errors():
    import abstracts_search_amd.faiss as faiss
    with pytest.raises(ValueError):
        faiss.index_factory(64, "HNSW32", faiss.METRIC_INNER_PRODUCT)
    with pytest.raises(NotImplementedError):
        faiss.index_factory(64, "IVF16,PQ8", 7)   # neither METRIC_INNER_PRODUCT nor METRIC_L2
    assert faiss.METRIC_INNER_PRODUCT == 0 and faiss.METRIC_L2 == 1


def test_product_does_not_import_oracle():
    """The product package must never route through oracle/ (parity would be void)."""
    pkg = os.path.join(ROOT, "abstracts-search_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "libivfpq_oracle" not in src and "oracle/_ref" not in src, f
                assert not re.search(r"#include\s+[<\"].*oracle", src), f
