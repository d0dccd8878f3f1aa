"""CPU tests of the encoder oracle: restatement vs transformers' Qwen2Model and
vs the committed golden tensors."""
import os
from dataclasses import replace

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "encoder_tiny.npz"))


@pytest.mark.parametrize("causal", [False, True])
def test_oracle_matches_golden_and_transformers(gold, causal):
    from oracle import encoder_oracle as E
    cfg = replace(E.TINY, causal=causal)
    W = E.synth_weights(cfg, int(gold["seed"]))
    tag = "causal" if causal else "bidir"
    hs = E.stack_forward(cfg, W, gold["ids"], gold["cu_seqlens"]).numpy()
    assert np.abs(hs - gold[f"hidden_{tag}"]).max() < 2e-5      # golden hidden state is Qwen2Model's
    hf = E.hf_last_hidden_state(cfg, W, gold["ids"], gold["cu_seqlens"]).numpy()
    assert np.abs(hs - hf).max() < 2e-5
    e = E.encode(cfg, W, gold["ids"], gold["cu_seqlens"], True).numpy()
    assert np.abs(e - gold[f"embed_{tag}"]).max() < 1e-5
    assert np.allclose(np.linalg.norm(e, axis=1), 1.0, atol=1e-5)


def test_synth_weights_are_bf16_exact_and_deterministic():
    from oracle import encoder_oracle as E
    W1, W2 = E.synth_weights(E.TINY, 3), E.synth_weights(E.TINY, 3)
    for k in W1:
        assert torch.equal(W1[k], W2[k])
        assert torch.equal(W1[k], W1[k].bfloat16().float()), k
    assert set(W1) == set(E.weight_shapes(E.TINY))


def test_bidirectional_differs_from_causal(gold):
    assert np.abs(gold["hidden_bidir"] - gold["hidden_causal"]).max() > 0.1
    # the last token of a sequence sees the same context either way only for 1-token sequences
    cu = gold["cu_seqlens"]
    one = [i for i in range(len(cu) - 1) if cu[i + 1] - cu[i] == 1][0]
    a = gold["hidden_bidir"][cu[one]]
    b = gold["hidden_causal"][cu[one]]
    assert np.abs(a - b).max() < 1e-5
