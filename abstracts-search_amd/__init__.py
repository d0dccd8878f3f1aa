"""MI355X-native embed-and-search hot path for colonelwatch/abstracts-search.

Two drop-in surfaces, both thin Python over C-ABI HIP libraries (no CPU
fallback -- without the HIP extension and a GPU every compute call raises):

* ``abstracts_search_amd.faiss`` -- the faiss subset the reference's
  ``sidecar-search index train|fill|tune`` and ``app.py`` use
  (reference Makefile:39,25,32; README.md:28).
* ``abstracts_search_amd.sentence_transformers`` -- ``SentenceTransformer.encode``
  as used by ``sidecar-search build`` (reference Makefile:65; README.md:28,60).
"""
__version__ = "0.1.0"
