"""HISTORY (rounds 1-3): kept for the record of how a number in DESIGN_HISTORY.md / profiles/ was produced.  NOT maintained: knobs it
sets may no longer exist (MI_RERANK, MI_REFINE_DEBUG, MI_SCAN_DEBUG ... are silent no-ops now) and paths may have moved."""
"""Encoder throughput on synthetic abstracts (cfg3 shape: stella_en_1.5B_v5
architecture, random-init bf16 weights, clipped log-normal lengths).  GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import abstracts_search_amd.sentence_transformers as st

nabs = int(os.environ.get("NABS", 4096)); bs = int(os.environ.get("BS", 128)); nlayers = int(os.environ.get("LAYERS", 28))
cfg = dict(st.STELLA_EN_1_5B_V5); cfg["n_layers"] = nlayers
model = st.SentenceTransformer(config=cfg)
g = torch.Generator(device="cuda").manual_seed(7)
def rnd(shape, scale): return (torch.randn(shape, generator=g, device="cuda") * scale).bfloat16()
H, I = cfg["hidden"], cfg["intermediate"]; qc, kc = cfg["n_heads"] * cfg["head_dim"], cfg["n_kv_heads"] * cfg["head_dim"]
W = {"embed_tokens.weight": rnd((cfg["vocab_size"], H), 0.3), "norm.weight": torch.ones(H, device="cuda"),
     "dense.weight": rnd((cfg["dense_out"], H), H ** -0.5), "dense.bias": torch.zeros(cfg["dense_out"], device="cuda")}
model.load_weights(W)
for l in range(nlayers):
    p = f"layers.{l}."
    model.load_weights({p + "input_layernorm.weight": torch.ones(H, device="cuda"), p + "post_attention_layernorm.weight": torch.ones(H, device="cuda"),
        p + "self_attn.q_proj.weight": rnd((qc, H), H ** -0.5), p + "self_attn.q_proj.bias": rnd((qc,), 0.1),
        p + "self_attn.k_proj.weight": rnd((kc, H), H ** -0.5), p + "self_attn.k_proj.bias": rnd((kc,), 0.1),
        p + "self_attn.v_proj.weight": rnd((kc, H), H ** -0.5), p + "self_attn.v_proj.bias": rnd((kc,), 0.1),
        p + "self_attn.o_proj.weight": rnd((H, qc), qc ** -0.5), p + "mlp.gate_proj.weight": rnd((I, H), H ** -0.5),
        p + "mlp.up_proj.weight": rnd((I, H), H ** -0.5), p + "mlp.down_proj.weight": rnd((H, I), I ** -0.5)})
rng = np.random.default_rng(7)
lens = np.clip(np.exp(rng.normal(np.log(220), 0.45, nabs)), 8, 512).astype(int)
toks = [rng.integers(0, cfg["vocab_size"], L).tolist() for L in lens]
ntok = int(lens.sum())
model.encode_tokens(toks[:bs], batch_size=bs, as_tensor=True); torch.cuda.synchronize()
t0 = time.perf_counter(); e = model.encode_tokens(toks, batch_size=bs, normalize_embeddings=True, as_tensor=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
flops = ntok * 2 * (H * (qc + 2 * kc) + qc * H + 3 * H * I) * nlayers
print(f"layers {nlayers} bs {bs}: {nabs/dt:9.1f} abstracts/s {ntok/dt:11.0f} tokens/s  {flops/dt/1e12:7.1f} TFLOP/s (GEMM flops only, wall)  median len {int(np.median(lens))}", flush=True)
model.profile(True)
model.encode_tokens(toks[:4 * bs], batch_size=bs, as_tensor=True); torch.cuda.synchronize()
p = model.profile_read(); model.profile(False)
print(f"GEMM kernels: {p['gemm_ms']:.2f} ms for {p['gemm_flops']/1e12:.2f} TFLOP -> {p['gemm_flops']/p['gemm_ms']/1e9:.1f} TFLOP/s", flush=True)
print("finite:", bool(torch.isfinite(e).all()), "norms", e.norm(dim=1)[:3].tolist())
