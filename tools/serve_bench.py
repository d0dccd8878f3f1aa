"""A served query path (reference README.md:18-29: app.py encodes one query and searches the index per request):
T host threads, each on its own HIP stream, issue single-query encode + search calls against ONE encoder handle and ONE
index handle (the C ABI's search-type calls are thread-safe: a workspace set is leased per stream).  Reports requests/s and
the median latency per request for T = 1, 2, 4, 8.  GPU box; SERVE_N / SERVE_NLIST size the index."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import abstracts_search_amd.faiss as faiss
import abstracts_search_amd.synth as synth
import abstracts_search_amd.sentence_transformers as st

cfg = dict(st.STELLA_EN_1_5B_V5); cfg["vocab_size"] = 8192
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
N, NLIST = int(os.environ.get("SERVE_N", 8 * 1048576)), int(os.environ.get("SERVE_NLIST", 4096))
idx = faiss.IndexIVFPQ(1024, NLIST, 64, 8, faiss.METRIC_INNER_PRODUCT)
idx.cp.niter = idx.pq.cp.niter = 4
idx.train(synth.corpus_cuda(1048576, 1024))
for c0 in range(0, N, 1048576):
    idx.add(synth.corpus_cuda(1048576, 1024, row0=c0))
idx.nprobe = 16
rng = np.random.default_rng(3)
queries = [[rng.integers(0, cfg["vocab_size"], int(rng.integers(16, 49))).tolist()] for _ in range(64)]

def request(i):
    e = model.encode_tokens(queries[i % 64], batch_size=1, normalize_embeddings=True, as_tensor=True)
    D, I = idx.search(e, 10)
    return I.cpu()                                    # the response leaves the GPU: the request's one synchronisation

for i in range(8): request(i)
for T in (1, 2, 4, 8):
    lat, stop = [[] for _ in range(T)], time.perf_counter() + 2.0
    streams = [torch.cuda.Stream() for _ in range(T)]
    def worker(t):
        i = t * 1000
        with torch.cuda.stream(streams[t]):
            while time.perf_counter() < stop:
                t0 = time.perf_counter(); request(i); lat[t].append(time.perf_counter() - t0); i += 1
    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    dt = time.perf_counter() - t0
    allv = sorted(v for l in lat for v in l)
    print(f"threads {T}: {len(allv) / dt:8.1f} requests/s, latency p50 {allv[len(allv) // 2] * 1e3:.2f} ms p95 {allv[int(0.95 * len(allv))] * 1e3:.2f} ms", flush=True)
