import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import abstracts_search_amd.sentence_transformers as st
import abstracts_search_amd.faiss as faiss
cfg = dict(st.STELLA_EN_1_5B_V5)
model = st.SentenceTransformer(config=cfg)
g = torch.Generator(device="cuda").manual_seed(7)
rnd = lambda shape, scale: (torch.randn(shape, generator=g, device="cuda") * scale).bfloat16()
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
toks = [rng.integers(0, cfg["vocab_size"], 31).tolist()]
N, nlist, M, d = 4_000_000, 65536, 64, 1024
cent = rng.standard_normal((nlist, d), dtype=np.float32); cent /= np.linalg.norm(cent, axis=1, keepdims=True)
idx = faiss.IndexIVFPQ(d, nlist, M, 8, faiss.METRIC_INNER_PRODUCT)
idx.set_centroids(cent); idx.set_codebook((0.05 * rng.standard_normal((M, 256, d // M))).astype(np.float32))
idx.add_codes(rng.integers(0, nlist, N, dtype=np.int32), rng.integers(0, 256, (N, M), dtype=np.uint8), np.arange(N, dtype=np.int64))
idx.nprobe = 64
def enc(): return model.encode_tokens(toks, batch_size=1, normalize_embeddings=True, as_tensor=True)
e = enc(); torch.cuda.synchronize()
D = torch.empty((1, 10), device="cuda"); I = torch.empty((1, 10), dtype=torch.int64, device="cuda")
def srch(): idx.search_into(e, 10, D, I)
def lat(f, n=300):
    for _ in range(20): f()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0))
    a = np.array(ts) * 1e3
    return np.median(a[:, 0]), np.median(a[:, 1])
def thr(f, n=300):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("encode: host return %.3f ms, latency %.3f ms; back to back %.3f ms" % (*lat(enc), thr(enc)))
print("search: host return %.3f ms, latency %.3f ms; back to back %.3f ms" % (*lat(srch), thr(srch)))
def both(): 
    x = enc(); idx.search_into(x, 10, D, I)
print("both  : host return %.3f ms, latency %.3f ms; back to back %.3f ms" % (*lat(both), thr(both)))
def both2():
    x = enc(); return idx.search(x, 10)
print("both (index.search allocating): host return %.3f ms, latency %.3f ms" % lat(both2))
