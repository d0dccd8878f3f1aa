"""run-to-run determinism of a few-hundred-token pass (debugging aid): LENS="40,40,40" python tools/mid_debug.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import abstracts_search_amd.sentence_transformers as st
cfg = dict(st.STELLA_EN_1_5B_V5); cfg["vocab_size"] = 4096; cfg["n_layers"] = int(os.environ.get("ENC_LAYERS", 2))
g = torch.Generator(device="cuda").manual_seed(7)
rnd = lambda shape, scale: (torch.randn(shape, generator=g, device="cuda") * scale).bfloat16()
H, I = cfg["hidden"], cfg["intermediate"]; qc, kc = cfg["n_heads"] * cfg["head_dim"], cfg["n_kv_heads"] * cfg["head_dim"]
W = {"embed_tokens.weight": rnd((cfg["vocab_size"], H), 0.3), "norm.weight": torch.ones(H, device="cuda"),
     "dense.weight": rnd((cfg["dense_out"], H), H ** -0.5), "dense.bias": torch.zeros(cfg["dense_out"], device="cuda")}
for l in range(cfg["n_layers"]):
    p = f"layers.{l}."
    W.update({p + "input_layernorm.weight": torch.ones(H, device="cuda"), p + "post_attention_layernorm.weight": torch.ones(H, device="cuda"),
        p + "self_attn.q_proj.weight": rnd((qc, H), H ** -0.5), p + "self_attn.q_proj.bias": rnd((qc,), 0.1),
        p + "self_attn.k_proj.weight": rnd((kc, H), H ** -0.5), p + "self_attn.k_proj.bias": rnd((kc,), 0.1),
        p + "self_attn.v_proj.weight": rnd((kc, H), H ** -0.5), p + "self_attn.v_proj.bias": rnd((kc,), 0.1),
        p + "self_attn.o_proj.weight": rnd((H, qc), qc ** -0.5), p + "mlp.gate_proj.weight": rnd((I, H), H ** -0.5),
        p + "mlp.up_proj.weight": rnd((I, H), H ** -0.5), p + "mlp.down_proj.weight": rnd((H, I), I ** -0.5)})
lens = [int(x) for x in os.environ.get("LENS", "40,40,40").split(",")]
rng = np.random.default_rng(1)
toks = [rng.integers(0, cfg["vocab_size"], L).tolist() for L in lens]
outs = []
for r in range(6):
    m = st.SentenceTransformer(config=cfg, weights=W) if r % 2 == 0 else m
    outs.append(m.last_hidden_state(toks))
d = [float(np.abs(o - outs[0]).max()) for o in outs]
bad = [(int(i), int(j)) for i, j in zip(*np.nonzero(outs[1] != outs[0]))][:5]
print(os.environ.get("TAG", ""), "tokens", sum(lens), "max |run_r - run_0| of the hidden states:", d, "first differing (row, col):", bad, flush=True)
