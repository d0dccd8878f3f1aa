"""CPU tests of the sentence-transformers facade's host logic: a hub id resolves the way the reference names its
model (reference README.md:28 MODEL_NAME="NovaSearch/stella_en_1.5B_v5", README.md:60 SIDECARSEARCH_MODEL=...), and
how `batch_size` / `token_budget` cut the length-sorted inputs into forward passes (reference Makefile:65 `-b 32`)."""
import json
import os

import pytest


@pytest.fixture()
def st():
    import abstracts_search_amd.sentence_transformers as m
    return m


def _fake_cache(root, repo="NovaSearch/stella_en_1.5B_v5", rev="0123456789abcdef0123456789abcdef01234567"):
    d = os.path.join(root, "models--" + repo.replace("/", "--"))
    snap = os.path.join(d, "snapshots", rev)
    os.makedirs(snap)
    os.makedirs(os.path.join(d, "refs"))
    with open(os.path.join(d, "refs", "main"), "w") as f:
        f.write(rev)
    with open(os.path.join(snap, "config.json"), "w") as f:
        json.dump({"hidden_size": 64}, f)
    return snap


def test_hub_id_resolves_from_the_local_cache(st, tmp_path, monkeypatch):
    snap = _fake_cache(str(tmp_path / "hub"))
    monkeypatch.setenv("HF_HUB_OFFLINE", "1")
    # cache_folder= (sentence-transformers' keyword)
    got = st.SentenceTransformer._resolve_hub_id("NovaSearch/stella_en_1.5B_v5", {"cache_folder": str(tmp_path / "hub")})
    assert os.path.samefile(got, snap) and os.path.exists(os.path.join(got, "config.json"))
    # a model that is nowhere: an error that says where it was looked for, not a fallback
    with pytest.raises(FileNotFoundError, match="neither a local model directory nor in the Hugging Face cache"):
        st.SentenceTransformer._resolve_hub_id("NovaSearch/no_such_model", {"cache_folder": str(tmp_path / "hub")})


def test_hub_id_honours_hf_home(st, tmp_path):
    # HF_HOME is read when huggingface_hub is imported: a fresh interpreter, like a user's process
    import subprocess
    import sys
    snap = _fake_cache(str(tmp_path / "home" / "hub"))
    code = ("import abstracts_search_amd.sentence_transformers as m, os;"
            "print(m.SentenceTransformer._resolve_hub_id('NovaSearch/stella_en_1.5B_v5', {}))")
    env = dict(os.environ, HF_HOME=str(tmp_path / "home"), HF_HUB_OFFLINE="1",
               PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env.pop("HF_HUB_CACHE", None)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert os.path.samefile(out.stdout.strip().splitlines()[-1], snap)


def test_passes_respect_batch_size_and_token_budget(st):
    m = object.__new__(st.SentenceTransformer)
    toks = [[0] * n for n in (30, 5, 12, 50, 7, 7, 41, 3)]
    order = sorted(range(len(toks)), key=lambda i: -len(toks[i]))
    m.token_budget = 32768
    assert m._passes(order, toks, None) == [order]                       # default: the budget alone
    p = m._passes(order, toks, 3)                                        # an explicit batch_size bounds a pass from above
    assert [len(x) for x in p] == [3, 3, 2] and [i for x in p for i in x] == order
    m.token_budget = 60
    p = m._passes(order, toks, 3)
    assert all(len(x) <= 3 and (sum(len(toks[i]) for i in x) <= 60 or len(x) == 1) for x in p)
    assert [i for x in p for i in x] == order and all(p)
    m.token_budget = None                                                # sentence-transformers' own rule
    assert [len(x) for x in m._passes(order, toks, 3)] == [3, 3, 2]
    assert [len(x) for x in m._passes(order, toks, None)] == [8]         # (default 32)


# config.json of a stella / gte-Qwen2 style checkpoint as published: Qwen2 fields + an `auto_map` to the repository's own
# modeling_qwen.py (hence TRUST_REMOTE_CODE in reference README.md:28,60).  Whether the real file also carries
# `is_causal` cannot be checked offline (SURVEY Appendix B.1 [PRIOR]) -- both spellings are covered.
STELLA_STYLE = dict(architectures=["Qwen2Model"], model_type="qwen2", hidden_size=1536, num_attention_heads=12,
                    num_key_value_heads=2, num_hidden_layers=28, intermediate_size=8960, vocab_size=151646,
                    rms_norm_eps=1e-6, rope_theta=1000000.0, max_position_embeddings=131072,
                    auto_map={"AutoModel": "modeling_qwen.Qwen2Model",
                              "AutoModelForCausalLM": "modeling_qwen.Qwen2ForCausalLM"})


def test_attention_mask_is_never_guessed(st, monkeypatch):
    monkeypatch.delenv("MI_ENCODER_CAUSAL", raising=False)
    base = dict(hidden_size=64, num_attention_heads=4, vocab_size=10, num_hidden_layers=1, intermediate_size=128)
    assert st._cfg_from_hf(base)["causal"] is True                         # plain Qwen2: transformers runs it causally
    assert st._cfg_from_hf(base, is_embedding_model=True)["causal"] is True   # ... under sentence-transformers too
    assert st._cfg_from_hf(dict(base, is_causal=False))["causal"] is False
    assert st._cfg_from_hf(dict(STELLA_STYLE, is_causal=False))["causal"] is False
    assert st._cfg_from_hf(dict(STELLA_STYLE, is_causal=True))["causal"] is True
    # remote modelling code and no key: the mask lives in code that is not run here.  The published stella / gte-Qwen2
    # signature (auto_map -> modeling_qwen.Qwen2Model + modules.json) is taken as bidirectional, loudly; without
    # modules.json, or with any other remote code, an error that says what to pass
    with pytest.warns(UserWarning, match="BIDIRECTIONAL"):
        assert st._cfg_from_hf(STELLA_STYLE, is_embedding_model=True)["causal"] is False
    with pytest.warns(UserWarning, match="BIDIRECTIONAL"):                  # hub-qualified spelling of the same class
        hub = dict(STELLA_STYLE, auto_map={"AutoModel": "NovaSearch/stella_en_1.5B_v5--modeling_qwen.Qwen2Model"})
        assert st._cfg_from_hf(hub, is_embedding_model=True)["causal"] is False
    with pytest.raises(ValueError, match="causal=False.*MI_ENCODER_CAUSAL"):
        st._cfg_from_hf(STELLA_STYLE)
    with pytest.raises(ValueError, match="causal=False.*MI_ENCODER_CAUSAL"):
        st._cfg_from_hf(dict(STELLA_STYLE, auto_map={"AutoModel": "modeling_other.OtherModel"}), is_embedding_model=True)
    for spelling, want_causal in (("no", False), ("off", False), ("False", False), ("yes", True), ("ON", True), ("true", True)):
        monkeypatch.setenv("MI_ENCODER_CAUSAL", spelling)
        assert st._cfg_from_hf(STELLA_STYLE)["causal"] is want_causal
    monkeypatch.setenv("MI_ENCODER_CAUSAL", "maybe")                         # an unknown value is an error, not True
    with pytest.raises(ValueError, match="MI_ENCODER_CAUSAL"):
        st._cfg_from_hf(STELLA_STYLE)
    monkeypatch.delenv("MI_ENCODER_CAUSAL")
    assert st._cfg_from_hf(STELLA_STYLE, causal=False)["causal"] is False    # constructor override
    monkeypatch.setenv("MI_ENCODER_CAUSAL", "0")                             # pipeline override (no keyword to pass)
    assert st._cfg_from_hf(STELLA_STYLE)["causal"] is False
    monkeypatch.setenv("MI_ENCODER_CAUSAL", "1")
    assert st._cfg_from_hf(STELLA_STYLE)["causal"] is True
    assert st._cfg_from_hf(dict(STELLA_STYLE, is_causal=False), causal=True)["causal"] is True   # explicit beats the file
    monkeypatch.delenv("MI_ENCODER_CAUSAL")
    full = st._cfg_from_hf(dict(STELLA_STYLE, is_causal=False))
    want = {k: v for k, v in st.STELLA_EN_1_5B_V5.items() if k not in ("dense_out", "max_seq_len")}
    assert {k: full[k] for k in want} == want                              # the [PRIOR] shape table == the parsed file
