"""Import shim: the package directory is named ``abstracts-search_amd`` (not a
valid Python identifier); this module makes it importable as
``abstracts_search_amd`` (``import abstracts_search_amd.faiss as faiss``)."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "abstracts-search_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
del _f
