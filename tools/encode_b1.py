"""Batch-1 query encode (stella shape, random-init) repeated: per-kernel profile target.
usage: python tools/encode_b1.py [ntokens] [reps] [nseq]   (run under tools/prof_cmd.sh on the GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import abstracts_search_amd.sentence_transformers as st

ntok = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
nseq = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg = dict(st.STELLA_EN_1_5B_V5)
cfg["n_layers"] = int(os.environ.get("ENC_LAYERS", cfg["n_layers"]))   # ENC_LAYERS=2: both layers' weights (187 MB) stay in the 256 MB Infinity Cache
model = st.SentenceTransformer(config=cfg)
g = torch.Generator(device="cuda").manual_seed(7)
def rnd(shape, scale): return (torch.randn(shape, generator=g, device="cuda") * scale).bfloat16()
H, I = cfg["hidden"], cfg["intermediate"]; qc, kc = cfg["n_heads"] * cfg["head_dim"], cfg["n_kv_heads"] * cfg["head_dim"]
model.load_weights({"embed_tokens.weight": rnd((cfg["vocab_size"], H), 0.3), "norm.weight": torch.ones(H, device="cuda"),
                    "dense.weight": rnd((cfg["dense_out"], H), H ** -0.5), "dense.bias": torch.zeros(cfg["dense_out"], device="cuda")})
for l in range(cfg["n_layers"]):
    p = f"layers.{l}."
    model.load_weights({p + "input_layernorm.weight": torch.ones(H, device="cuda"), p + "post_attention_layernorm.weight": torch.ones(H, device="cuda"),
        p + "self_attn.q_proj.weight": rnd((qc, H), H ** -0.5), p + "self_attn.q_proj.bias": rnd((qc,), 0.1),
        p + "self_attn.k_proj.weight": rnd((kc, H), H ** -0.5), p + "self_attn.k_proj.bias": rnd((kc,), 0.1),
        p + "self_attn.v_proj.weight": rnd((kc, H), H ** -0.5), p + "self_attn.v_proj.bias": rnd((kc,), 0.1),
        p + "self_attn.o_proj.weight": rnd((H, qc), qc ** -0.5), p + "mlp.gate_proj.weight": rnd((I, H), H ** -0.5),
        p + "mlp.up_proj.weight": rnd((I, H), H ** -0.5), p + "mlp.down_proj.weight": rnd((H, I), I ** -0.5)})
rng = np.random.default_rng(1)
toks = [rng.integers(0, cfg["vocab_size"], ntok).tolist() for _ in range(nseq)]
for _ in range(5):
    model.encode_tokens(toks, batch_size=nseq, normalize_embeddings=True, as_tensor=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    model.encode_tokens(toks, batch_size=nseq, normalize_embeddings=True, as_tensor=True)
torch.cuda.synchronize()
print(f"batch {nseq}, {ntok} tokens each: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per encode")
