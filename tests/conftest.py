import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


HAS_GPU = _has_gpu()


def pytest_collection_modifyitems(config, items):
    # -m gpu on a box without a GPU must fail loudly, not silently pass:
    # gpu tests are only skipped when they were not explicitly selected.
    selected_gpu = "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or "")
    if HAS_GPU or selected_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container (gpu tests run via gpurun)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import ivfpq_oracle
    ivfpq_oracle.build()
    return ivfpq_oracle


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def _reload_native_env():
    """the HIP libraries read their MI_* knobs once; tests that switch one call the libraries' reload hooks (only libraries
    this process has loaded already: a CPU test never builds or loads one through here)"""
    mods = sys.modules
    st = mods.get("abstracts_search_amd.sentence_transformers")
    fa = mods.get("abstracts_search_amd.faiss")
    if st is not None and getattr(st._Lib, "_lib", None) is not None:
        st.reload_env()
    if fa is not None and getattr(fa._Lib, "_lib", None) is not None:
        fa.reload_env()


class _KnobPatch:
    """pytest's monkeypatch, with setenv / delenv of an MI_* name followed by the libraries' reload_env()"""

    def __init__(self, mp):
        self._mp = mp

    def setenv(self, name, value, prepend=None):
        self._mp.setenv(name, value, prepend)
        if name.startswith("MI_"):
            _reload_native_env()

    def delenv(self, name, raising=True):
        self._mp.delenv(name, raising)
        if name.startswith("MI_"):
            _reload_native_env()

    def __getattr__(self, attr):
        return getattr(self._mp, attr)


@pytest.fixture
def monkeypatch(monkeypatch):
    yield _KnobPatch(monkeypatch)
    monkeypatch.undo()
    _reload_native_env()
