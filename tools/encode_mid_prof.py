"""One query batch (ENC_NQ queries of 16-48 tokens) through the stella-shape encoder, ENC_REPS times: run under
rocprofv3 --kernel-trace --stats to see where a few-hundred-token forward pass spends its time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import abstracts_search_amd.sentence_transformers as st
cfg = dict(st.STELLA_EN_1_5B_V5)
cfg["n_layers"] = int(os.environ.get("ENC_LAYERS", cfg["n_layers"]))   # ENC_LAYERS=2: both layers' weights (187 MB) stay in the 256 MB Infinity Cache; cfg["vocab_size"] = 8192
g = torch.Generator(device="cuda").manual_seed(7)
rnd = lambda shape, scale: (torch.randn(shape, generator=g, device="cuda") * scale).bfloat16()
H, I = cfg["hidden"], cfg["intermediate"]; qc, kc = cfg["n_heads"] * cfg["head_dim"], cfg["n_kv_heads"] * cfg["head_dim"]
model = st.SentenceTransformer(config=cfg)
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
nq, reps = int(os.environ.get("ENC_NQ", 16)), int(os.environ.get("ENC_REPS", 30))
rng = np.random.default_rng(1)
toks = None
if nq in (1, 16, 256):                                   # bench.py's cfg5 batches, drawn in its order (1, 16, 256): 31 / 570 / 8 097 tokens
    for b in (1, 16, 256):
        toks = [rng.integers(0, cfg["vocab_size"], int(rng.integers(16, 49))).tolist() for _ in range(b)]
        if b == nq:
            break
else:
    toks = [rng.integers(0, cfg["vocab_size"], int(rng.integers(16, 49))).tolist() for _ in range(nq)]
for _ in range(3): model.encode_tokens(toks, batch_size=nq, normalize_embeddings=True, as_tensor=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): model.encode_tokens(toks, batch_size=nq, normalize_embeddings=True, as_tensor=True)
torch.cuda.synchronize()
print(f"nq {nq} tokens {sum(map(len, toks))}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per forward pass", flush=True)
