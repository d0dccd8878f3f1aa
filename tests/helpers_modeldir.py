"""A sentence-transformers model directory (config.json, safetensors, modules.json +
2_Dense_*, prompts, tokenizer.json) for the tiny oracle configuration -- the layout the
reference loads stella_en_1.5B_v5 from (README.md:28).  Test infrastructure."""
import json


def write_model_dir(d, cfg, W, max_seq_length=32, prompts=None, post_processor=False):
    """Returns the vocabulary; `d` is a pathlib.Path that will hold the model."""
    from safetensors.torch import save_file
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    dense = f"2_Dense_{cfg.dense_out}"
    (d / dense).mkdir(parents=True)
    json.dump(dict(hidden_size=cfg.hidden, num_attention_heads=cfg.n_heads, num_key_value_heads=cfg.n_kv_heads,
                   head_dim=cfg.head_dim, num_hidden_layers=cfg.n_layers, intermediate_size=cfg.intermediate,
                   vocab_size=cfg.vocab_size, rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta,
                   max_position_embeddings=cfg.max_seq_len, is_causal=False), open(d / "config.json", "w"))
    save_file({("model." + k): v.bfloat16() for k, v in W.items() if not k.startswith("dense.")},
              str(d / "model.safetensors"))
    save_file({"linear.weight": W["dense.weight"], "linear.bias": W["dense.bias"]}, str(d / dense / "model.safetensors"))
    json.dump(dict(in_features=cfg.hidden, out_features=cfg.dense_out, bias=True), open(d / dense / "config.json", "w"))
    json.dump([dict(idx=0, name="0", path="", type="sentence_transformers.models.Transformer"),
               dict(idx=1, name="1", path="1_Pooling", type="sentence_transformers.models.Pooling"),
               dict(idx=2, name="2", path=dense, type="sentence_transformers.models.Dense")],
              open(d / "modules.json", "w"))
    json.dump(dict(prompts=prompts if prompts is not None else {"s2p_query": "query: "}, default_prompt_name=None),
              open(d / "config_sentence_transformers.json", "w"))
    json.dump(dict(max_seq_length=max_seq_length), open(d / "sentence_bert_config.json", "w"))
    nspecial = 4 if post_processor else 3
    vocab = {f"w{i}": i for i in range(cfg.vocab_size - nspecial)}
    vocab.update({"query": cfg.vocab_size - 3, ":": cfg.vocab_size - 2, "[UNK]": cfg.vocab_size - 1})
    if post_processor:
        vocab["<eos>"] = cfg.vocab_size - 4
    tk = Tokenizer(models.WordLevel(vocab, unk_token="[UNK]"))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    if post_processor:   # what a tokenizer that appends an end-of-sequence token looks like (TemplateProcessing)
        tk.post_processor = processors.TemplateProcessing(single="$A <eos>", special_tokens=[("<eos>", vocab["<eos>"])])
    tk.save(str(d / "tokenizer.json"))
    return vocab
