"""BASELINE.json configs[4] on one GPU: end-to-end encode + search for mixed query
batches 1 / 16 / 256 (latency and throughput), stella-shape random-init encoder,
cfg2 index (or, E2E_N / E2E_NLIST, one shard of the 207 M index).  Queries are short (prompt + question, 16-48 tokens).  GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import abstracts_search_amd.faiss as faiss
import abstracts_search_amd.synth as synth
import abstracts_search_amd.sentence_transformers as st
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

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
# index: cfg2 by default; E2E_N=25875000 E2E_NLIST=65536 is one shard of the 207 M configuration
N, NLIST = int(os.environ.get("E2E_N", 1_000_000)), int(os.environ.get("E2E_NLIST", 4096))
idx = faiss.IndexIVFPQ(1024, NLIST, 64, 8, faiss.METRIC_INNER_PRODUCT)
idx.cp.niter = idx.pq.cp.niter = 6 if NLIST <= 4096 else 4
x = synth.corpus_cuda(min(N, max(1_000_000, 64 * NLIST)), 1024)
idx.train(x)
if N <= x.shape[0]:
    idx.add(x)
else:
    del x
    CH = 65536 * 16
    for c0 in range(0, N, CH):
        idx.add(synth.corpus_cuda(min(CH, N - c0), 1024, row0=c0))
idx.nprobe = int(os.environ.get("E2E_NPROBE", 16))
print(f"index: {N} x 1024, IVF{NLIST},PQ64, nprobe {idx.nprobe}", flush=True)
rng = np.random.default_rng(1)
for batch in (1, 16, 256):
    toks = [rng.integers(0, cfg["vocab_size"], int(rng.integers(16, 49))).tolist() for _ in range(batch)]
    def once():
        e = model.encode_tokens(toks, batch_size=batch, normalize_embeddings=True, as_tensor=True)
        return idx.search(e, 10)
    for _ in range(3): once()
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps): D, I_ = once()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps): e = model.encode_tokens(toks, batch_size=batch, normalize_embeddings=True, as_tensor=True)
    torch.cuda.synchronize()
    de = (time.perf_counter() - t0) / reps
    print(f"batch {batch:4d}: encode+search {dt*1e3:8.3f} ms ({batch/dt:9.0f} queries/s), encode alone {de*1e3:8.3f} ms", flush=True)
