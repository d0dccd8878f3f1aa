"""Drop-in switch for the reference pipeline.

`sidecar-search` and the query-time `app.py` import `faiss` and `sentence_transformers` by
name (reference requirements.txt:1, README.md:28).  With this directory and the repository
root on PYTHONPATH and ABSTRACTS_SEARCH_BACKEND=mi355x in the environment (both set from
env.mk, which the reference Makefile includes at line 4), those names resolve to the
MI355X mirrors; without the variable nothing changes."""
import os
import sys

if os.environ.get("ABSTRACTS_SEARCH_BACKEND") == "mi355x":
    import abstracts_search_amd.faiss as _faiss
    import abstracts_search_amd.sentence_transformers as _st
    sys.modules["faiss"] = _faiss
    sys.modules["sentence_transformers"] = _st
