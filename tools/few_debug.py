"""Debug driver of the query-time path: the TINY oracle model, a few tokens, MI_FEW_SYNC=1 names every stage."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import abstracts_search_amd.sentence_transformers as st
from oracle import encoder_oracle as E
W = E.synth_weights(E.TINY, 7)
model = st.SentenceTransformer(config=E.TINY.to_dict(), weights=W)
rng = np.random.default_rng(0)
lens = [int(v) for v in sys.argv[1:]] or [5]
toks = [rng.integers(0, E.TINY.vocab_size, L).tolist() for L in lens]
hs = model.last_hidden_state(toks)
cu = np.concatenate([[0], np.cumsum(lens)])
ref = E.stack_forward(E.TINY, W, np.concatenate(toks), cu).numpy()
cos = (hs * ref).sum(1) / (np.linalg.norm(hs, axis=1) * np.linalg.norm(ref, axis=1))
print("hidden cos min", cos.min(), "max rel", (np.linalg.norm(hs - ref, axis=1) / np.linalg.norm(ref, axis=1)).max())
