"""oracle/encoder_oracle.py -- CPU (torch fp32) restatement of the encode half of
the hot path: Qwen2-style decoder stack -> mean pooling -> Dense -> L2-normalise,
i.e. what `SentenceTransformer("NovaSearch/stella_en_1.5B_v5").encode(...)` runs
for the reference (`sidecar-search build`, reference Makefile:65; query-time
`app.py`, reference README.md:28).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product package.

Pinning.  The reference holds none of this arithmetic (it lives in
sentence-transformers / transformers / the model's remote code, all
un-vendored), and neither the stella weights nor sentence-transformers are
available here.  What IS available is `transformers.models.qwen2.Qwen2Model`
(transformers 5.15; the reference pins <= 4.49), so this restatement is checked
against it on a tiny random-weight configuration, causal (2-D padding mask) and
bidirectional (4-D additive mask), see tests/test_oracle_encoder.py and
tests/golden/make_encoder_golden.py.  Facts about stella itself (bidirectional
attention, Dense 1536->1024 with bias, prompts) are SURVEY.md Appendix B.1
[PRIOR]; the encoder is configuration-driven so none is hard-coded.

Layout: sequences are PACKED (no padding tokens): ids[T], cu_seqlens[nseq+1].
Weight names follow HF Qwen2 (`layers.N.self_attn.q_proj.weight`, ...), plus
`dense.weight` / `dense.bias` for the sentence-transformers Dense module.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, asdict

import numpy as np
import torch


@dataclass
class EncoderConfig:
    vocab_size: int = 151646
    hidden: int = 1536
    n_layers: int = 28
    n_heads: int = 12
    n_kv_heads: int = 2
    head_dim: int = 128
    intermediate: int = 8960
    rms_eps: float = 1e-6
    rope_theta: float = 1e6
    causal: bool = False          # stella / gte-Qwen2 run the stack bidirectionally [PRIOR]
    dense_out: int = 1024         # 0 = no Dense module
    dense_bias: bool = True
    max_seq_len: int = 512

    def to_dict(self):
        return asdict(self)


STELLA_1_5B = EncoderConfig()
TINY = EncoderConfig(vocab_size=64, hidden=256, n_layers=2, n_heads=4, n_kv_heads=2, head_dim=64,
                     intermediate=384, dense_out=64, max_seq_len=128)


def weight_shapes(cfg: EncoderConfig) -> dict:
    H, I, hd = cfg.hidden, cfg.intermediate, cfg.head_dim
    s = {"embed_tokens.weight": (cfg.vocab_size, H), "norm.weight": (H,)}
    for l in range(cfg.n_layers):
        p = f"layers.{l}."
        s[p + "input_layernorm.weight"] = (H,)
        s[p + "post_attention_layernorm.weight"] = (H,)
        s[p + "self_attn.q_proj.weight"] = (cfg.n_heads * hd, H)
        s[p + "self_attn.q_proj.bias"] = (cfg.n_heads * hd,)
        s[p + "self_attn.k_proj.weight"] = (cfg.n_kv_heads * hd, H)
        s[p + "self_attn.k_proj.bias"] = (cfg.n_kv_heads * hd,)
        s[p + "self_attn.v_proj.weight"] = (cfg.n_kv_heads * hd, H)
        s[p + "self_attn.v_proj.bias"] = (cfg.n_kv_heads * hd,)
        s[p + "self_attn.o_proj.weight"] = (H, cfg.n_heads * hd)
        s[p + "mlp.gate_proj.weight"] = (I, H)
        s[p + "mlp.up_proj.weight"] = (I, H)
        s[p + "mlp.down_proj.weight"] = (H, I)
    if cfg.dense_out:
        s["dense.weight"] = (cfg.dense_out, H)
        if cfg.dense_bias:
            s["dense.bias"] = (cfg.dense_out,)
    return s


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    return z ^ (z >> np.uint64(31))


def to_bf16_exact(a: np.ndarray) -> np.ndarray:
    """round float32 to the nearest bf16-representable float32 (RNE)"""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) & np.uint64(0xFFFF0000)
    return u.astype(np.uint32).view(np.float32).reshape(a.shape)


def synth_weights(cfg: EncoderConfig, seed: int = 7) -> dict:
    """Deterministic, platform-independent random-init weights (integer hash ->
    uniform), already rounded to bf16-representable values so that the bf16 GPU
    model and the fp32 oracle hold identical parameters.  Scales follow the
    usual 1/sqrt(fan_in) initialisation; norm weights are near 1."""
    out = {}
    with np.errstate(over="ignore"):
        for i, (name, shape) in enumerate(weight_shapes(cfg).items()):
            n = int(np.prod(shape))
            idx = np.arange(n, dtype=np.uint64) + np.uint64((seed * 1000003 + i) << 32)
            u = (_splitmix64(idx) >> np.uint64(40)).astype(np.float64) / float(1 << 24)  # [0,1)
            u = 2.0 * u - 1.0
            if name.endswith("norm.weight") or name.endswith("layernorm.weight"):
                w = 1.0 + 0.1 * u
            elif name.endswith(".bias"):
                w = 0.1 * u
            elif name == "embed_tokens.weight":
                w = 0.5 * u
            else:
                w = u * math.sqrt(3.0 / shape[-1])
            out[name] = torch.from_numpy(to_bf16_exact(w.astype(np.float32).reshape(shape)))
    return out


# ----------------------------------------------------------------------
# the restatement
# ----------------------------------------------------------------------

def rmsnorm(x, w, eps):
    var = x.pow(2).mean(-1, keepdim=True)
    return x * torch.rsqrt(var + eps) * w


def rope_tables(cfg: EncoderConfig, n: int):
    hd = cfg.head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float64) / hd))
    ang = torch.arange(n, dtype=torch.float64)[:, None] * inv[None, :]          # [n, hd/2]
    return torch.cos(ang).float(), torch.sin(ang).float()


def apply_rope(x, cos, sin):
    """x [L, nh, hd]; HF rotate_half convention: pairs (i, i + hd/2)."""
    hd = x.shape[-1]
    x1, x2 = x[..., : hd // 2], x[..., hd // 2:]
    c, s = cos[:, None, :], sin[:, None, :]
    return torch.cat([x1 * c - x2 * s, x2 * c + x1 * s], -1)


def stack_forward(cfg: EncoderConfig, W: dict, ids, cu_seqlens):
    """Packed token ids [T] -> last hidden state after the final norm [T, H] (fp32)."""
    ids = torch.as_tensor(ids, dtype=torch.long)
    cu = [int(v) for v in cu_seqlens]
    hd, nh, nkv = cfg.head_dim, cfg.n_heads, cfg.n_kv_heads
    x = W["embed_tokens.weight"][ids].float()
    cos, sin = rope_tables(cfg, max(b - a for a, b in zip(cu[:-1], cu[1:])) if len(cu) > 1 else 1)
    for l in range(cfg.n_layers):
        p = f"layers.{l}."
        h = rmsnorm(x, W[p + "input_layernorm.weight"], cfg.rms_eps)
        q = h @ W[p + "self_attn.q_proj.weight"].T + W[p + "self_attn.q_proj.bias"]
        k = h @ W[p + "self_attn.k_proj.weight"].T + W[p + "self_attn.k_proj.bias"]
        v = h @ W[p + "self_attn.v_proj.weight"].T + W[p + "self_attn.v_proj.bias"]
        att = torch.empty_like(q)
        for a, b in zip(cu[:-1], cu[1:]):
            L = b - a
            qs = apply_rope(q[a:b].view(L, nh, hd), cos[:L], sin[:L])
            ks = apply_rope(k[a:b].view(L, nkv, hd), cos[:L], sin[:L])
            vs = v[a:b].view(L, nkv, hd)
            ks = ks.repeat_interleave(nh // nkv, dim=1)
            vs = vs.repeat_interleave(nh // nkv, dim=1)
            sc = torch.einsum("qhd,khd->hqk", qs, ks) / math.sqrt(hd)
            if cfg.causal:
                sc = sc + torch.full((L, L), float("-inf")).triu(1)
            pr = torch.softmax(sc, -1)
            att[a:b] = torch.einsum("hqk,khd->qhd", pr, vs).reshape(L, nh * hd)
        x = x + att @ W[p + "self_attn.o_proj.weight"].T
        h = rmsnorm(x, W[p + "post_attention_layernorm.weight"], cfg.rms_eps)
        g = h @ W[p + "mlp.gate_proj.weight"].T
        u = h @ W[p + "mlp.up_proj.weight"].T
        x = x + (torch.nn.functional.silu(g) * u) @ W[p + "mlp.down_proj.weight"].T
    return rmsnorm(x, W["norm.weight"], cfg.rms_eps)


def encode(cfg: EncoderConfig, W: dict, ids, cu_seqlens, normalize: bool = True):
    """sentence-transformers pipeline: Transformer -> mean Pooling -> Dense ->
    (normalize_embeddings) -> float32 [nseq, out_dim]."""
    hs = stack_forward(cfg, W, ids, cu_seqlens)
    cu = [int(v) for v in cu_seqlens]
    pooled = torch.stack([hs[a:b].mean(0) for a, b in zip(cu[:-1], cu[1:])])
    if cfg.dense_out:
        pooled = pooled @ W["dense.weight"].T
        if cfg.dense_bias:
            pooled = pooled + W["dense.bias"]
    if normalize:
        pooled = torch.nn.functional.normalize(pooled, p=2, dim=1)
    return pooled


# ----------------------------------------------------------------------
# cross-check against transformers.Qwen2Model (tiny configs only)
# ----------------------------------------------------------------------

def hf_last_hidden_state(cfg: EncoderConfig, W: dict, ids, cu_seqlens):
    """The same packed batch through transformers' Qwen2Model (padded), as an
    independent implementation of the stack.  Returns packed [T, H]."""
    from transformers import Qwen2Config, Qwen2Model
    hc = Qwen2Config(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden, intermediate_size=cfg.intermediate,
                     num_hidden_layers=cfg.n_layers, num_attention_heads=cfg.n_heads,
                     num_key_value_heads=cfg.n_kv_heads, head_dim=cfg.head_dim, rms_norm_eps=cfg.rms_eps,
                     rope_theta=cfg.rope_theta, max_position_embeddings=cfg.max_seq_len,
                     attention_dropout=0.0, use_sliding_window=False, tie_word_embeddings=False)
    hc._attn_implementation = "eager"
    m = Qwen2Model(hc).eval().float()
    sd = {k: v.clone() for k, v in W.items() if not k.startswith("dense.")}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("rotary" in k or "inv_freq" in k for k in missing), missing
    cu = [int(v) for v in cu_seqlens]
    lens = [b - a for a, b in zip(cu[:-1], cu[1:])]
    B, Lm = len(lens), max(lens)
    ids = torch.as_tensor(ids, dtype=torch.long)
    pad = torch.zeros((B, Lm), dtype=torch.long)
    mask = torch.zeros((B, Lm), dtype=torch.long)
    for i, (a, b) in enumerate(zip(cu[:-1], cu[1:])):
        pad[i, : b - a] = ids[a:b]
        mask[i, : b - a] = 1
    with torch.no_grad():
        if cfg.causal:
            out = m(input_ids=pad, attention_mask=mask).last_hidden_state
        else:
            add = torch.zeros((B, 1, Lm, Lm))
            add.masked_fill_(mask[:, None, None, :] == 0, torch.finfo(torch.float32).min)
            out = m(input_ids=pad, attention_mask=add).last_hidden_state
    return torch.cat([out[i, : lens[i]] for i in range(B)])
