"""Host tokenisation against the real HF fast-tokenizer wrapper that
sentence-transformers calls (`transformers.PreTrainedTokenizerFast`, importable here):
truncation keeps the post-processor's trailing special token; `add_eos_token` is honoured.
No GPU: HostTokenizer is pure host code."""
import json

import pytest

from abstracts_search_amd.sentence_transformers import HostTokenizer


def _tok(post_processor: bool):
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    vocab = {f"w{i}": i for i in range(50)}
    vocab.update({"[UNK]": 50, "<eos>": 51})
    tk = Tokenizer(models.WordLevel(vocab, unk_token="[UNK]"))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    if post_processor:
        tk.post_processor = processors.TemplateProcessing(single="$A <eos>", special_tokens=[("<eos>", 51)])
    return tk


TEXTS = ["w1 w2 w3", "w4 " * 40, "", "w7 nope w8 " * 9]


@pytest.mark.parametrize("max_len", [8, 16, 512])
def test_truncation_matches_hf_fast_tokenizer(max_len):
    from transformers import PreTrainedTokenizerFast
    tk = _tok(post_processor=True)
    hf = PreTrainedTokenizerFast(tokenizer_object=_tok(post_processor=True), eos_token="<eos>", unk_token="[UNK]")
    want = hf(TEXTS, truncation=True, max_length=max_len, padding=False)["input_ids"]
    got = HostTokenizer(tk)(TEXTS, max_len)
    for w, g in zip(want, got):
        assert g == (w if w else [0])
    long = got[1]
    assert len(long) == min(max_len, 41) and long[-1] == 51          # the trailing special token survives truncation


def test_add_eos_token_from_tokenizer_config(tmp_path):
    tk = _tok(post_processor=False)
    tk.save(str(tmp_path / "tokenizer.json"))
    json.dump({"add_eos_token": True, "eos_token": {"content": "<eos>"}}, open(tmp_path / "tokenizer_config.json", "w"))
    ht = HostTokenizer.from_dir(str(tmp_path))
    assert ht.eos_id == 51 and not ht.appends_eos
    got = ht(TEXTS, 16)
    assert got[0] == [1, 2, 3, 51]
    assert len(got[1]) == 16 and got[1][-1] == 51 and got[1][:15] == [4] * 15
    assert got[2] == [51]
    # a tokenizer.json that already appends EOS is not given a second one
    tk2 = _tok(post_processor=True)
    tk2.save(str(tmp_path / "tokenizer.json"))
    ht2 = HostTokenizer.from_dir(str(tmp_path))
    assert ht2.appends_eos and ht2(TEXTS, 16)[0] == [1, 2, 3, 51]
    # without add_eos_token nothing is appended
    json.dump({"eos_token": "<eos>"}, open(tmp_path / "tokenizer_config.json", "w"))
    tk.save(str(tmp_path / "tokenizer.json"))
    assert HostTokenizer.from_dir(str(tmp_path))(TEXTS, 16)[0] == [1, 2, 3]
