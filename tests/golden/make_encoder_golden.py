"""Generates tests/golden/encoder_tiny.npz: inputs and expected outputs of the
encode path on a tiny Qwen2-style configuration.

The expected tensors come from transformers.models.qwen2.Qwen2Model (a real
implementation of the stack, imported here in the build container) for the
last hidden state, both causal and bidirectional, and from
oracle/encoder_oracle.py for the pooled / Dense / normalised embedding (the
sentence-transformers modules are three lines of arithmetic).  Weights are not
stored: oracle.encoder_oracle.synth_weights(cfg, seed) regenerates them exactly
(integer hash, bf16-representable).  Run from the repository root:
    python tests/golden/make_encoder_golden.py
"""
import os
import sys
from dataclasses import replace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import encoder_oracle as E  # noqa: E402

rng = np.random.default_rng(20260928)
lens = [5, 64, 1, 33, 100, 8, 65]
cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
ids = rng.integers(0, E.TINY.vocab_size, int(cu[-1])).astype(np.int32)
out = dict(ids=ids, cu_seqlens=cu, seed=np.int64(7))
for causal in (False, True):
    cfg = replace(E.TINY, causal=causal)
    W = E.synth_weights(cfg, 7)
    hf = E.hf_last_hidden_state(cfg, W, ids, cu).numpy()
    mine = E.stack_forward(cfg, W, ids, cu).numpy()
    assert np.abs(hf - mine).max() < 2e-5, np.abs(hf - mine).max()
    tag = "causal" if causal else "bidir"
    out[f"hidden_{tag}"] = hf.astype(np.float32)
    out[f"embed_{tag}"] = E.encode(cfg, W, ids, cu, True).numpy()
    out[f"embed_raw_{tag}"] = E.encode(cfg, W, ids, cu, False).numpy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "encoder_tiny.npz"), **out)
print({k: getattr(v, "shape", v) for k, v in out.items()})
