"""GPU parity tests of the encoder: HIP kernels (through the C ABI) vs plain
PyTorch fp32 references of the same ops and vs the oracle / golden tensors.
Tolerances: bf16 operands with f32 accumulation -> relative 1e-2 on GEMM
outputs; embeddings within 1e-3 cosine (BASELINE.json north star).

The file name sorts AFTER test_ivfpq_gpu.py on purpose: the encoder must give the same
embeddings in a process that has searched an index first (a serving process does both).  Its
workspaces then come out of recycled device memory; an attention-output buffer whose padding
rows were never written turned embeddings into NaN in exactly that order (0 x NaN in the
masked part of P.V) while the suite ran the encoder tests first and stayed green."""
import os
from dataclasses import replace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def st():
    import abstracts_search_amd.sentence_transformers as m
    return m


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "encoder_tiny.npz"))


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 192), (1000, 520, 1536), (4096, 2048, 1536),
                                   (4096, 1536, 8960), (4096, 17920, 1536), (29312, 1536, 8960)])   # down / gate-up shapes of stella; the bench's token count
def test_gemm_bf16_vs_torch_fp32(st, M, N, K):
    import torch
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn((M, K), generator=g, device="cuda").bfloat16()
    W = (torch.randn((N, K), generator=g, device="cuda") / K ** 0.5).bfloat16()
    # asymmetric operands: a transposed or permuted fragment layout cannot pass
    A[:, 0] += 3.0
    W[0, :] += 0.5
    C = st.gemm_bf16(A, W).float()
    ref = A.float() @ W.float().T
    err = (C - ref).abs().max().item()
    assert err <= 1e-2 * ref.abs().max().item() + 1e-3, err


@pytest.mark.parametrize("M,N,K", [(29312, 1536, 8960),    # down: 115 x 6 = 690 tiles -> two rounds + 22 / 23 tiles per XCD as K ranges of 140 K tiles
                                   (28000, 2048, 1536),    # QKV shape: 880 tiles -> three rounds + 14 per XCD, several pieces per tile
                                   (27958, 1536, 1536),    # O shape, ragged M
                                   (20000, 17920, 1536),   # gate/up shape: 79 x 70 tiles, order 0 (per-XCD eighths), a short remainder
                                   (16640, 4096, 1024)])   # 65 x 16 = 1040 tiles, K tiles = 16: the smallest K that is split
def test_persistent_stream_k_gemm_vs_one_tile_per_workgroup(st, M, N, K, monkeypatch):
    """The many-token GEMMs as ONE persistent workgroup per CU with a stream-K tail (MI_GEMM_PERSIST=15; round 6's experiment: it
    lost its A/B and is off by default -- DESIGN 6.1) against the same slab kernel launched one tile per workgroup (the
    default) and against torch fp32: whole tiles are bit-identical (the same K
    walk), tiles whose K range was dealt out over several workgroups differ by the rounding of the partial sums only; twice in
    a row the persistent launch gives the same bits (partials are added in a fixed order); no head gave up waiting."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn((M, K), generator=g, device="cuda").bfloat16()
    W = (torch.randn((N, K), generator=g, device="cuda") / K ** 0.5).bfloat16()
    A[:, 0] += 3.0
    W[0, :] += 0.5
    monkeypatch.delenv("MI_GEMM_PERSIST", raising=False)         # the default: one tile per workgroup
    st.reload_env()
    p0 = st.debug_counter("persist_launches")
    C0 = st.gemm_bf16(A, W).float()
    assert st.debug_counter("persist_launches") == p0
    monkeypatch.setenv("MI_GEMM_PERSIST", "15")
    st.reload_env()
    C1 = st.gemm_bf16(A, W).float()
    assert st.debug_counter("persist_launches") == p0 + 1
    C2 = st.gemm_bf16(A, W).float()
    monkeypatch.delenv("MI_GEMM_PERSIST")
    st.reload_env()
    assert torch.equal(C1, C2)
    assert st.debug_counter("sk_giveups") == 0
    ref = A.float() @ W.float().T
    scale = ref.abs().max().item()
    assert (C1 - ref).abs().max().item() <= 1e-2 * scale + 1e-3
    d = (C1 - C0).abs()
    assert d.max().item() <= scale * 2 ** -7 + 1e-3              # a bf16 rounding step of the largest value, at most
    frac_same = (d == 0).float().mean().item()
    assert frac_same > 0.6, frac_same                            # the whole-tile rounds are the same arithmetic


def test_persistent_stream_k_encoder_pass_vs_default(st, monkeypatch):
    """A 24 k-token pass through a 2-layer stella-width model with all four projections on the persistent stream-K launches
    (fused RMSNorm / rotary / SwiGLU epilogues behind heads that first add their tiles' partial sums) against the default
    dispatch: the same embeddings to bf16 rounding, twice the same bits, no head gave up."""
    cfg = dict(st.STELLA_EN_1_5B_V5)
    cfg["vocab_size"], cfg["n_layers"] = 2048, 2
    W = _rand_weights_gpu(cfg, 31)
    rng = np.random.default_rng(31)
    lens = np.clip(np.exp(rng.normal(np.log(220), 0.45, 400)), 8, 512).astype(int)
    lens = lens[: int(np.searchsorted(np.cumsum(lens), 24000))]
    toks = [rng.integers(0, cfg["vocab_size"], int(L)).tolist() for L in lens]
    outs = {}
    for name, env in (("default", None), ("persist", "15"), ("persist2", "15")):
        monkeypatch.delenv("MI_GEMM_PERSIST", raising=False)
        if env:
            monkeypatch.setenv("MI_GEMM_PERSIST", env)
        model = st.SentenceTransformer(config=cfg, weights=W)      # (the constructor re-reads the knobs)
        model.token_budget = None
        p0 = st.debug_counter("persist_launches")
        outs[name] = model.encode_tokens(toks, batch_size=len(toks), normalize_embeddings=True)
        took = st.debug_counter("persist_launches") - p0
        assert (took >= 2 * cfg["n_layers"]) if env else (took == 0), (name, took)   # at least QKV and gate/up of every layer
    monkeypatch.delenv("MI_GEMM_PERSIST", raising=False)
    st.reload_env()
    assert st.debug_counter("sk_giveups") == 0
    assert np.array_equal(outs["persist"], outs["persist2"])
    cos = (outs["default"] * outs["persist"]).sum(1)
    assert cos.min() > 1 - 2e-4 and np.abs(outs["default"] - outs["persist"]).max() < 4e-3, cos.min()


def _split(ids, cu):
    return [ids[cu[i]:cu[i + 1]].tolist() for i in range(len(cu) - 1)]


def test_recycled_device_memory_does_not_poison_padding_rows(st, gold):
    """Workspaces come from recycled device memory, not from zeroed pages: device buffers full
    of NaN are released by an index of the same process, then a new encoder allocates.  Padding
    rows (between packed sequences) must still hold finite values everywhere a real token's
    masked attention weights -- exact zeros -- multiply them."""
    import gc
    import abstracts_search_amd.faiss as faiss
    from oracle import encoder_oracle as E
    import torch
    # what an index that lived in the same process leaves behind: ids of -1, scores of -FLT_MAX
    # (as bf16 pairs: NaN and -inf patterns) in buffers that go back to the allocator
    rng = np.random.default_rng(0)
    d, nlist, n = 64, 32, 12000
    cent = rng.standard_normal((nlist, d)).astype(np.float32)
    cb = (0.3 * rng.standard_normal((8, 256, 8))).astype(np.float32)
    x = (cent[rng.integers(0, nlist, n)] + 0.3 * rng.standard_normal((n, d))).astype(np.float32)
    base = faiss.IndexIVFPQ(d, nlist, 8, 8, faiss.METRIC_INNER_PRODUCT)
    base.set_centroids(cent)
    base.set_codebook(cb)
    idx = faiss.IndexRefineFlat(base)
    idx.add(x)
    for nprobe, kf in ((1, 1), (4, 4), (16, 4)):
        idx.nprobe, idx.k_factor = nprobe, kf
        idx.search(x[:64], 10)
        idx.search(torch.from_numpy(x[:64]).cuda(), 10)
    torch.cuda.synchronize()
    del idx, base
    gc.collect()
    W = E.synth_weights(E.TINY, int(gold["seed"]))
    model = st.SentenceTransformer(config=E.TINY.to_dict(), weights=W)
    toks = _split(gold["ids"], gold["cu_seqlens"])
    hs = model.last_hidden_state(toks)
    ref = gold["hidden_bidir"]
    assert np.isfinite(hs).all()
    cos = (hs * ref).sum(1) / (np.linalg.norm(hs, axis=1) * np.linalg.norm(ref, axis=1))
    assert cos.min() > 1 - 1e-3, cos.min()


@pytest.mark.parametrize("causal", [False, True])
def test_tiny_model_vs_golden(st, gold, causal):
    from oracle import encoder_oracle as E
    cfg = replace(E.TINY, causal=causal)
    W = E.synth_weights(cfg, int(gold["seed"]))
    model = st.SentenceTransformer(config=cfg.to_dict(), weights=W)
    toks = _split(gold["ids"], gold["cu_seqlens"])
    tag = "causal" if causal else "bidir"
    hs = model.last_hidden_state(toks)
    ref = gold[f"hidden_{tag}"]
    assert hs.shape == ref.shape
    cos = (hs * ref).sum(1) / (np.linalg.norm(hs, axis=1) * np.linalg.norm(ref, axis=1))
    assert cos.min() > 1 - 1e-3, cos.min()
    assert np.abs(hs - ref).max() < 0.08 * np.abs(ref).max()
    model.token_budget = None
    e = model.encode_tokens(toks, batch_size=3, normalize_embeddings=True)   # several batches, re-ordered by length
    model.token_budget = 32768
    cos = (e * gold[f"embed_{tag}"]).sum(1)
    assert cos.min() > 1 - 1e-3, cos
    assert np.allclose(np.linalg.norm(e, axis=1), 1.0, atol=1e-4)
    raw = model.encode_tokens(toks, batch_size=32, normalize_embeddings=False)
    ref_raw = gold[f"embed_raw_{tag}"]
    assert np.abs(raw - ref_raw).max() < 0.03 * np.abs(ref_raw).max() + 1e-3


def test_batching_invariance(st, gold):
    """an embedding does not depend on which other sequences share its batch"""
    from oracle import encoder_oracle as E
    W = E.synth_weights(E.TINY, 7)
    model = st.SentenceTransformer(config=E.TINY.to_dict(), weights=W)
    toks = _split(gold["ids"], gold["cu_seqlens"])
    a = model.encode_tokens(toks, batch_size=32, normalize_embeddings=True)     # one pass (token budget 32 768)
    model.token_budget = None                                                   # passes cut by batch_size alone
    b = model.encode_tokens(toks, batch_size=1, normalize_embeddings=True)
    # (passes of <= 48 tokens take the query-time kernels of csrc/encoder_few.h, whose bf16 rounding points differ from the
    # general path's -- 1/rms applied to the accumulator instead of the operand, f32 RoPE: two bf16 computations of the same
    # embedding, each within the 1e-3 cosine budget of the fp32 oracle, agree to ~3e-3 per component on this 64-d model)
    assert np.abs(a - b).max() < 4e-3 and ((a * b).sum(1)).min() > 1 - 1e-4
    c = model.encode_tokens(toks[::-1], batch_size=2, normalize_embeddings=True)[::-1]
    assert np.abs(a - c).max() < 4e-3 and ((a * c).sum(1)).min() > 1 - 1e-4
    # passes cut by a token budget: a few sequences each, never an empty pass, every sequence exactly once
    model.token_budget = 40
    order = sorted(range(len(toks)), key=lambda i: -len(toks[i]))
    passes = model._passes(order, toks, 32)
    assert sorted(i for p in passes for i in p) == list(range(len(toks))) and len(passes) > 1 and all(passes)
    assert all(sum(len(toks[i]) for i in p) <= 40 or len(p) == 1 for p in passes)
    e = model.encode_tokens(toks, batch_size=32, normalize_embeddings=True)
    assert np.abs(a - e).max() < 4e-3


def test_errors(st):
    from oracle import encoder_oracle as E
    model = st.SentenceTransformer(config=E.TINY.to_dict())
    with pytest.raises(RuntimeError, match="not loaded"):
        model.encode_tokens([[1, 2, 3]])
    model.load_weights(E.synth_weights(E.TINY, 7))
    with pytest.raises(RuntimeError, match="longer than max_seq_len"):
        model.encode_tokens([[1] * (E.TINY.max_seq_len + 1)])
    with pytest.raises(RuntimeError, match="out of range"):
        model.encode_tokens([[E.TINY.vocab_size]])
    with pytest.raises(RuntimeError, match="no tokenizer"):
        model.encode("hello")


def test_model_directory_and_text_encode(st, tmp_path):
    """the sentence-transformers directory layout + tokenizer + prompts path"""
    from oracle import encoder_oracle as E
    from helpers_modeldir import write_model_dir
    cfg = E.TINY
    W = E.synth_weights(cfg, 11)
    d = tmp_path / "model"
    write_model_dir(d, cfg, W, max_seq_length=32)

    model = st.SentenceTransformer(str(d), trust_remote_code=True)
    assert model.get_sentence_embedding_dimension() == cfg.dense_out and model.max_seq_length == 32
    docs = ["w1 w2 w3 w4", "w9", "w5 w5 w7 unknownword w8 " * 20]
    e = model.encode(docs, batch_size=2, normalize_embeddings=True)
    assert e.shape == (3, cfg.dense_out) and e.dtype == np.float32
    one = model.encode(docs[0], normalize_embeddings=True)
    assert one.shape == (cfg.dense_out,) and np.abs(one - e[0]).max() < 2e-3
    ids = [model.tokenize([s])[0] for s in docs]
    assert len(ids[2]) == 32                                      # truncated to max_seq_length
    cu = np.concatenate([[0], np.cumsum([len(i) for i in ids])])
    ref = E.encode(replace(cfg, max_seq_len=32), W, np.concatenate(ids), cu, True).numpy()
    assert ((e * ref).sum(1) > 1 - 1e-3).all()
    qp = model.encode(docs[0], prompt_name="s2p_query", normalize_embeddings=True)
    qm = model.encode("query: " + docs[0], normalize_embeddings=True)
    assert np.abs(qp - qm).max() < 1e-6 and np.abs(qp - one).max() > 1e-3
    with pytest.raises(ValueError, match="not found"):
        model.encode(docs[0], prompt_name="nope")


def test_model_named_by_its_hub_id_loads_from_the_cache(st, tmp_path, monkeypatch):
    """SentenceTransformer("NovaSearch/stella_en_1.5B_v5") the way the reference names its model (README.md:28
    MODEL_NAME=..., README.md:60 SIDECARSEARCH_MODEL=...): the id resolves through the Hugging Face cache layout
    (models--org--name/snapshots/<rev>, refs/main) with the hub offline, and encodes exactly like the directory itself."""
    from oracle import encoder_oracle as E
    from helpers_modeldir import write_model_dir
    cfg = E.TINY
    W = E.synth_weights(cfg, 12)
    rev = "89ab" * 10
    repo = tmp_path / "hub" / "models--NovaSearch--stella_en_1.5B_v5"
    snap = repo / "snapshots" / rev
    write_model_dir(snap, cfg, W, max_seq_length=32)
    (repo / "refs").mkdir()
    (repo / "refs" / "main").write_text(rev)
    monkeypatch.setenv("HF_HUB_OFFLINE", "1")
    by_id = st.SentenceTransformer("NovaSearch/stella_en_1.5B_v5", trust_remote_code=True, cache_folder=str(tmp_path / "hub"))
    by_dir = st.SentenceTransformer(str(snap), trust_remote_code=True)
    docs = ["w1 w2 w3 w4", "w9 w8"]
    a = by_id.encode(docs, prompt_name="s2p_query", normalize_embeddings=True)
    b = by_dir.encode(docs, prompt_name="s2p_query", normalize_embeddings=True)
    assert a.shape == (2, cfg.dense_out) and np.array_equal(a, b)
    with pytest.raises(FileNotFoundError, match="Hugging Face cache"):
        st.SentenceTransformer("NovaSearch/not_there", cache_folder=str(tmp_path / "hub"))


def _rand_weights_gpu(cfg, seed):
    """random-init bf16 weights of a given architecture, generated on the GPU"""
    import torch
    g = torch.Generator(device="cuda").manual_seed(seed)

    def rnd(shape, scale):
        return (torch.randn(shape, generator=g, device="cuda") * scale).bfloat16()

    H, I = cfg["hidden"], cfg["intermediate"]
    qc, kc = cfg["n_heads"] * cfg["head_dim"], cfg["n_kv_heads"] * cfg["head_dim"]
    W = {"embed_tokens.weight": rnd((cfg["vocab_size"], H), 0.3), "norm.weight": 1 + rnd((H,), 0.1),
         "dense.weight": rnd((cfg["dense_out"], H), H ** -0.5), "dense.bias": rnd((cfg["dense_out"],), 0.1)}
    for l in range(cfg["n_layers"]):
        p = f"layers.{l}."
        W.update({p + "input_layernorm.weight": 1 + rnd((H,), 0.1), p + "post_attention_layernorm.weight": 1 + rnd((H,), 0.1),
                  p + "self_attn.q_proj.weight": rnd((qc, H), H ** -0.5), p + "self_attn.q_proj.bias": rnd((qc,), 0.1),
                  p + "self_attn.k_proj.weight": rnd((kc, H), H ** -0.5), p + "self_attn.k_proj.bias": rnd((kc,), 0.1),
                  p + "self_attn.v_proj.weight": rnd((kc, H), H ** -0.5), p + "self_attn.v_proj.bias": rnd((kc,), 0.1),
                  p + "self_attn.o_proj.weight": rnd((H, qc), qc ** -0.5), p + "mlp.gate_proj.weight": rnd((I, H), H ** -0.5),
                  p + "mlp.up_proj.weight": rnd((I, H), H ** -0.5), p + "mlp.down_proj.weight": rnd((H, I), I ** -0.5)})
    return W


def test_stella_shape_full_depth_vs_oracle(st):
    """BASELINE.json's encoder at its real shape (1536 hidden, 28 layers, 12/2
    heads of 128, 8960 MLP, Dense 1024; vocabulary cut to 4096 rows to bound the
    test's memory) with random-init weights: embeddings within 1e-3 cosine of the
    fp32 CPU oracle, the north star's tolerance."""
    import torch
    from oracle import encoder_oracle as E
    cfg = dict(st.STELLA_EN_1_5B_V5)
    cfg["vocab_size"] = 4096
    W = _rand_weights_gpu(cfg, 5)
    model = st.SentenceTransformer(config=cfg, weights=W)
    rng = np.random.default_rng(5)
    lens = [3, 40, 130, 64, 17]
    toks = [rng.integers(0, cfg["vocab_size"], L).tolist() for L in lens]
    e = model.encode_tokens(toks, batch_size=8, normalize_embeddings=True)
    ocfg = E.EncoderConfig(**cfg)
    Wc = {k: v.float().cpu() for k, v in W.items()}
    cu = np.concatenate([[0], np.cumsum(lens)])
    with torch.no_grad():
        ref = E.encode(ocfg, Wc, np.concatenate(toks), cu, True).numpy()
    cos = (e * ref).sum(1)
    assert cos.min() > 1 - 1e-3, cos


def test_bulk_encode_path_at_stella_widths_vs_oracle(st):
    """The forward pass bench.py times, against the fp32 oracle: stella's widths (hidden 1536, 12 / 2 heads of 128, MLP
    8960, Dense 1024; 3 layers and a 4096-row vocabulary bound the oracle's time) on the bench's own kind of batch -- 128
    abstracts with clipped log-normal lengths (median 220), several of them forced to the 512-token maximum, ~29 k
    tokens -- under the DEFAULT dispatch: the hand-ordered 256x256 slab kernels on every projection (K = 1536 and
    K = 8960), sequences of up to 8 attention chunks, pooling through the GEMM path.  Hidden states and embeddings within
    1e-3 cosine of the oracle (the north star's tolerance).  (reference call site: Makefile:65 `build -b 32`, README.md:60)"""
    import torch
    from oracle import encoder_oracle as E
    cfg = dict(st.STELLA_EN_1_5B_V5)
    cfg["vocab_size"], cfg["n_layers"] = 4096, 3
    W = _rand_weights_gpu(cfg, 23)
    rng = np.random.default_rng(7)
    lens = np.clip(np.exp(rng.normal(np.log(220), 0.45, 128)), 8, 512).astype(int)
    lens[:5] = 512                                            # full-length sequences: 8 chunks of 64 keys
    lens[5] = 8
    toks = [rng.integers(0, cfg["vocab_size"], int(L)).tolist() for L in lens]
    ntok = int(lens.sum())
    assert 25000 < ntok < 32768
    model = st.SentenceTransformer(config=cfg, weights=W)
    model.token_budget = None                                 # one pass of 128 abstracts, as `bench.py --encode-batch 128` issues it
    fused = st.debug_counter("fused_norm_launches"), st.debug_counter("fused_rope_launches")
    e = model.encode_tokens(toks, batch_size=128, normalize_embeddings=True)
    # no rmsnorm_kernel / rope_kernel between the GEMMs of this pass: every RMSNorm but the first layer's first (and the final
    # one) is split over the slab epilogues either side of it, the rotary embedding is in the QKV epilogue
    assert st.debug_counter("fused_norm_launches") - fused[0] == 2 * cfg["n_layers"] - 1
    assert st.debug_counter("fused_rope_launches") - fused[1] == cfg["n_layers"]
    order = sorted(range(128), key=lambda i: -len(toks[i]))   # the pass's own order (longest first)
    hs = model.last_hidden_state([toks[i] for i in order])
    Wc = {k: v.float().cpu() for k, v in W.items()}
    ocfg = E.EncoderConfig(**cfg)
    with torch.no_grad():
        cu = np.concatenate([[0], np.cumsum(lens)])
        ref = E.encode(ocfg, Wc, np.concatenate(toks), cu, True).numpy()
        cu_o = np.concatenate([[0], np.cumsum([len(toks[i]) for i in order])])
        ref_hs = E.stack_forward(ocfg, Wc, np.concatenate([toks[i] for i in order]), cu_o).numpy()
    cos = (e * ref).sum(1)
    assert cos.min() > 1 - 1e-3, (cos.min(), int(cos.argmin()), int(lens[cos.argmin()]))
    ch = (hs * ref_hs).sum(1) / (np.linalg.norm(hs, axis=1) * np.linalg.norm(ref_hs, axis=1))
    assert ch.min() > 1 - 1e-3, ch.min()
    # and through the library's own batching (token budget 32 768: the same abstracts plus a few more in one pass)
    model.token_budget = 32768
    more = toks + [rng.integers(0, cfg["vocab_size"], 300).tolist() for _ in range(9)]
    e2 = model.encode_tokens(more, batch_size=32, normalize_embeddings=True)
    assert np.abs(e2[:128] - e).max() < 2e-3 and ((e2[:128] * ref).sum(1)).min() > 1 - 1e-3


def test_fused_epilogues_agree_with_the_standalone_kernels(st, monkeypatch):
    """The many-token pass with the RMSNorms and the rotary embedding inside the slab GEMM epilogues (default) against the same
    pass through rmsnorm_kernel / rope_kernel (MI_NO_BULK_FUSE=1), each half of the fusion alone too: same embeddings to bf16
    rounding.  Also a width whose residual GEMMs take the 256 x 192 tiles (16 slots of sums of squares per row)."""
    cfg = dict(st.STELLA_EN_1_5B_V5)
    cfg["vocab_size"], cfg["n_layers"] = 2048, 2
    W = _rand_weights_gpu(cfg, 29)
    rng = np.random.default_rng(29)
    for ntok_target in (28000, 6400):                         # 110 row tiles: 256-column residual tiles; 25: 256 x 192 (n192_pays)
        lens = np.clip(np.exp(rng.normal(np.log(220), 0.45, 400)), 8, 512).astype(int)
        lens = lens[: int(np.searchsorted(np.cumsum(lens), ntok_target))]
        toks = [rng.integers(0, cfg["vocab_size"], int(L)).tolist() for L in lens]
        outs = {}
        for name, env in (("fused", {}), ("plain", {"MI_NO_BULK_FUSE": "1"}), ("norm_only", {"MI_NO_ROPE_FUSE": "1"}),
                          ("rope_only", {"MI_NO_NORM_FUSE": "1"})):
            for k in ("MI_NO_BULK_FUSE", "MI_NO_ROPE_FUSE", "MI_NO_NORM_FUSE"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            model = st.SentenceTransformer(config=cfg, weights=W)
            model.token_budget = None
            c0 = st.debug_counter("fused_norm_launches"), st.debug_counter("fused_rope_launches"), st.debug_counter("n192_launches")
            outs[name] = model.encode_tokens(toks, batch_size=len(toks), normalize_embeddings=True)
            took = (st.debug_counter("fused_norm_launches") - c0[0], st.debug_counter("fused_rope_launches") - c0[1])
            assert took == {"fused": (3, 2), "plain": (0, 0), "norm_only": (3, 0), "rope_only": (0, 2)}[name], (name, took)
            assert (st.debug_counter("n192_launches") - c0[2] > 0) == (ntok_target == 6400)
        for name in ("plain", "norm_only", "rope_only"):
            cos = (outs["fused"] * outs[name]).sum(1)
            assert cos.min() > 1 - 2e-4 and np.abs(outs["fused"] - outs[name]).max() < 4e-3, (ntok_target, name, cos.min())


@pytest.mark.parametrize("nq,layers", [(1, 28), (5, 28), (16, 28), (64, 4), (132, 3), (200, 2), (256, 3)])
def test_query_batches_at_stella_widths_vs_oracle(st, nq, layers):
    """BASELINE.json configs[4]'s query batches (1 / 16 / 256 prompted queries of 16-48 tokens: ~30, ~570 and ~8 200
    tokens) at stella's widths against the fp32 oracle -- the few-token path (fragment-major weights, skinny / split-K
    tiles), the few-hundred-token path (128-row tiles, K-split down projection) and the first sizes that take the
    256x256 slab kernel.  Full depth for 1 and 16 queries, 3 layers for 256 (the oracle's time).  (reference
    README.md:28: the query-time app encodes with prompt_name s2p_query.)"""
    import torch
    from oracle import encoder_oracle as E
    cfg = dict(st.STELLA_EN_1_5B_V5)
    cfg["vocab_size"], cfg["n_layers"] = 4096, layers
    W = _rand_weights_gpu(cfg, 31 + nq)
    rng = np.random.default_rng(nq)
    lens = rng.integers(16, 49, nq)
    toks = [rng.integers(0, cfg["vocab_size"], int(L)).tolist() for L in lens]
    model = st.SentenceTransformer(config=cfg, weights=W)
    before, before_n, before_w = st.debug_counter("splitk_launches"), st.debug_counter("reduce_norm_launches"), st.debug_counter("n192_launches")
    before_m = st.debug_counter("mid_launches")
    e = model.encode_tokens(toks, batch_size=nq, normalize_embeddings=True)
    # ~100 to ~5000 tokens: the down projection of every layer runs K-split through the workspace (16 / 64 / 132 queries:
    # 18 x 10, 54 x 4 and 102 x 2 workgroups), and its reduction pass writes the next layer's first RMSNorm
    # ~100 to ~3000 / ~4000 tokens: the QKV and O projections are ONE launch each (encoder_mid.h: whole-K 64 x 64 .. 128 x 128
    # tiles, bias + RoPE resp. residual + RMSNorm partials in the epilogue: no planes, no reduction pass, no rope_kernel)
    # (132 queries and more: the pooled Dense GEMM -- M = queries > 64, K = 1536 -- is such a projection too)
    assert st.debug_counter("mid_launches") - before_m == {5: 2 * layers, 16: 2 * layers, 64: 2 * layers}.get(nq, 1 if nq > 64 else 0)
    assert st.debug_counter("splitk_launches") - before == {5: layers, 16: layers, 64: layers, 132: layers}.get(nq, 0)
    assert st.debug_counter("reduce_norm_launches") - before_n == {5: layers - 1, 16: layers - 1, 64: layers - 1, 132: layers - 1}.get(nq, 0)
    # ~5 400 to 8 192 tokens: O and down projection on 256 x 192 tiles (8 tile columns: one full round of workgroups);
    # 4 100 to 5 400: the O projection alone (the down projection is K-split there)
    assert st.debug_counter("n192_launches") - before_w == {200: 2 * layers, 132: layers}.get(nq, 0)
    Wc = {k: v.float().cpu() for k, v in W.items()}
    with torch.no_grad():
        ref = E.encode(E.EncoderConfig(**cfg), Wc, np.concatenate(toks), np.concatenate([[0], np.cumsum(lens)]), True).numpy()
    cos = (e * ref).sum(1)
    assert cos.min() > 1 - 1e-3, (nq, cos.min())


def test_192_row_tiles_vs_oracle(st, monkeypatch):
    """Token counts at which 256-row GEMM tiles would multiply mostly padding (bench.py's 16-query batch: ~570 tokens = three
    row tiles either way, 576 rows instead of 768): the gate/up and down projections take the slab kernel's 192-row variant.
    Same embeddings as the 256-row tiles (MI_NO_M192=1) to rounding, and within 1e-3 cosine of the fp32 oracle."""
    import torch
    from oracle import encoder_oracle as E
    cfg = dict(st.STELLA_EN_1_5B_V5)
    cfg["vocab_size"], cfg["n_layers"] = 4096, 3
    W = _rand_weights_gpu(cfg, 57)
    rng = np.random.default_rng(57)
    for lens in ([31] * 18, [48] * 6 + [17] * 3):            # 558 tokens -> 576 rows; 339 -> 352 rows: two tiles of 192 against two of 256
        toks = [rng.integers(0, cfg["vocab_size"], int(L)).tolist() for L in lens]
        outs = {}
        for name in ("m192", "m256"):
            monkeypatch.delenv("MI_NO_M192", raising=False)
            if name == "m256":
                monkeypatch.setenv("MI_NO_M192", "1")
            model = st.SentenceTransformer(config=cfg, weights=W)
            c0 = st.debug_counter("m192_launches")
            outs[name] = model.encode_tokens(toks, batch_size=len(toks), normalize_embeddings=True)
            assert st.debug_counter("m192_launches") - c0 == (2 * cfg["n_layers"] if name == "m192" else 0), (name, len(lens))
        Wc = {k: v.float().cpu() for k, v in W.items()}
        with torch.no_grad():
            ref = E.encode(E.EncoderConfig(**cfg), Wc, np.concatenate(toks), np.concatenate([[0], np.cumsum(lens)]), True).numpy()
        assert ((outs["m192"] * ref).sum(1)).min() > 1 - 1e-3
        assert ((outs["m192"] * outs["m256"]).sum(1)).min() > 1 - 2e-4 and np.abs(outs["m192"] - outs["m256"]).max() < 4e-3


@pytest.mark.parametrize("lens", [[40] * 3, [31] * 18, [48] * 6 + [17] * 3, [33] * 40, [16, 48] * 40, [512, 300, 77, 8]])
def test_one_launch_projections_vs_the_k_split_path_and_oracle(st, lens, monkeypatch):
    """encoder_mid.h (QKV and O projections of ~100 .. ~4000 tokens as one launch each: whole-K tiles, RoPE / residual +
    RMSNorm partials in the epilogue) against round 4's K-split planes + reduction passes (MI_NO_MID_GEMM=1) and against the
    fp32 oracle, over token counts that pick every tile shape (64 x 64 .. 128 x 128), ragged last row tiles, the fused and
    the standalone post-attention RMSNorm (<= 256 tokens: gate/up is not on the slab kernel), and long sequences."""
    import torch
    from oracle import encoder_oracle as E
    cfg = dict(st.STELLA_EN_1_5B_V5)
    cfg["vocab_size"], cfg["n_layers"] = 4096, 2
    W = _rand_weights_gpu(cfg, 91)
    rng = np.random.default_rng(len(lens))
    toks = [rng.integers(0, cfg["vocab_size"], int(L)).tolist() for L in lens]
    outs = {}
    for name in ("mid", "ksplit"):
        monkeypatch.delenv("MI_NO_MID_GEMM", raising=False)
        if name == "ksplit":
            monkeypatch.setenv("MI_NO_MID_GEMM", "1")
        model = st.SentenceTransformer(config=cfg, weights=W)
        c0 = st.debug_counter("mid_launches")
        outs[name] = model.encode_tokens(toks, batch_size=len(toks), normalize_embeddings=True)
        # (QKV + O per layer; from 64 sequences up the pooled Dense is a GEMM of K = hidden too and takes the same tiles)
        assert st.debug_counter("mid_launches") - c0 == (2 * cfg["n_layers"] + (len(lens) >= 64) if name == "mid" else 0), name
    Wc = {k: v.float().cpu() for k, v in W.items()}
    with torch.no_grad():
        ref = E.encode(E.EncoderConfig(**cfg), Wc, np.concatenate(toks), np.concatenate([[0], np.cumsum(lens)]), True).numpy()
    assert ((outs["mid"] * ref).sum(1)).min() > 1 - 1e-3
    assert ((outs["mid"] * outs["ksplit"]).sum(1)).min() > 1 - 2e-4 and np.abs(outs["mid"] - outs["ksplit"]).max() < 4e-3
    monkeypatch.delenv("MI_NO_MID_GEMM", raising=False)
    again = st.SentenceTransformer(config=cfg, weights=W).encode_tokens(toks, batch_size=len(toks), normalize_embeddings=True)
    assert np.array_equal(again, outs["mid"])                  # no atomics: run-to-run identical


@pytest.mark.parametrize("env", [{"MI_MID_TILE": "128x128"}, {"MI_MID_TILE": "128x64"}, {"MI_MID_TILE": "96x64"}, {"MI_MID_TILE": "64x64"},
                                 {"MI_SPLITK": "4"}, {"MI_SPLITK": "14"}, {"MI_NO_FEW": "1"}, {"MI_NO_SHORT_ATTN": "1"}, {"MI_NO_KROT": "1"}, {"MI_KROT": "-1"}])
def test_dispatch_knobs_keep_the_embeddings(st, env, monkeypatch):
    """The tool knobs that force a tile shape of encoder_mid.h (every instantiation at one token count), the K-slice count of
    the down projection's all-tiles split, and the general path for a handful of tokens (MI_NO_FEW): same embeddings as the
    default dispatch to bf16 rounding, within 1e-3 cosine of the fp32 oracle.  (The libraries read MI_* once: the fixture
    calls reload_env.)"""
    import torch
    from oracle import encoder_oracle as E
    cfg = dict(st.STELLA_EN_1_5B_V5)
    cfg["vocab_size"], cfg["n_layers"] = 4096, 2
    W = _rand_weights_gpu(cfg, 17)
    rng = np.random.default_rng(17)
    lens = [7, 22] if "MI_NO_FEW" in env else [37] * 9 + [48] * 5
    toks = [rng.integers(0, cfg["vocab_size"], int(L)).tolist() for L in lens]
    base = st.SentenceTransformer(config=cfg, weights=W).encode_tokens(toks, batch_size=len(toks), normalize_embeddings=True)
    few0, sa0 = st.debug_counter("few_passes"), st.debug_counter("short_attn_launches")
    assert sa0 > 0 or "MI_NO_FEW" in env                       # (every sequence <= 48 tokens: one wave per head and query tile)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    out = st.SentenceTransformer(config=cfg, weights=W).encode_tokens(toks, batch_size=len(toks), normalize_embeddings=True)
    if "MI_NO_FEW" in env:
        assert st.debug_counter("few_passes") == few0          # the query-time path stayed out
    if "MI_NO_SHORT_ATTN" in env:
        assert st.debug_counter("short_attn_launches") == sa0  # the persistent flash-attention kernel took them
    Wc = {k: v.float().cpu() for k, v in W.items()}
    with torch.no_grad():
        ref = E.encode(E.EncoderConfig(**cfg), Wc, np.concatenate(toks), np.concatenate([[0], np.cumsum(lens)]), True).numpy()
    assert ((out * ref).sum(1)).min() > 1 - 1e-3
    assert ((out * base).sum(1)).min() > 1 - 2e-4 and np.abs(out - base).max() < 4e-3


@pytest.mark.parametrize("ntok", [70, 130, 193, 257, 385, 449, 577, 700, 830, 1000, 1290, 1700])
def test_k_split_tile_widths_and_k_offsets_over_token_counts(st, ntok, monkeypatch):
    """The down projection of ~70 .. 1 700 tokens: the launcher picks the K-split tile width (256 / 192 / 128 columns) and slice
    count by its cost model, slabs request the tile's rows only, and the row tiles of one-round GEMMs walk K from offsets 3 K tiles
    apart -- against the same pass on 256-column tiles without the offsets (MI_DOWN_BN=256, MI_NO_KROT=1: round 4's launches) to
    bf16 rounding, over token counts either side of every row-tile and dispatch border.  Run-to-run identical (no atomics)."""
    cfg = dict(st.STELLA_EN_1_5B_V5)
    cfg["vocab_size"], cfg["n_layers"] = 4096, 2
    W = _rand_weights_gpu(cfg, 23)
    rng = np.random.default_rng(ntok)
    lens, left = [], ntok
    while left > 0:
        L = int(min(left, rng.integers(9, 49)))
        lens.append(L)
        left -= L
    toks = [rng.integers(0, cfg["vocab_size"], L).tolist() for L in lens]
    new = st.SentenceTransformer(config=cfg, weights=W).encode_tokens(toks, batch_size=len(toks), normalize_embeddings=True)
    again = st.SentenceTransformer(config=cfg, weights=W).encode_tokens(toks, batch_size=len(toks), normalize_embeddings=True)
    assert np.array_equal(new, again)
    monkeypatch.setenv("MI_DOWN_BN", "256")
    monkeypatch.setenv("MI_NO_KROT", "1")
    old = st.SentenceTransformer(config=cfg, weights=W).encode_tokens(toks, batch_size=len(toks), normalize_embeddings=True)
    assert ((new * old).sum(1)).min() > 1 - 2e-4 and np.abs(new - old).max() < 4e-3, (ntok, ((new * old).sum(1)).min())


@pytest.mark.parametrize("lens", [[1], [3], [16], [17], [32], [33], [48], [5, 9, 20], [1] * 7, [16, 16, 16], [2, 46]])
def test_few_token_path_at_stella_widths_vs_oracle(st, lens, monkeypatch):
    """The query-time path (csrc/encoder_few.h: RMSNorm in the GEMM prologue, RoPE / SwiGLU / residual atomics in the
    epilogues, weights as 1-KiB pieces) for every token-tile count it serves (1..48 tokens = 1, 2 or 3 tiles of 16, the
    tile edges, several sequences in one pass) at stella's widths, 3 layers deep: last hidden state per token and the
    embedding against the fp32 oracle.  (reference README.md:28: one prompted query per call.)"""
    import torch
    from oracle import encoder_oracle as E
    if len(lens) % 2 == 0:                                       # half the layouts through the in-launch reduction of the down projection
        monkeypatch.setenv("MI_FEW_D_FUSE", "1")
    cfg = dict(st.STELLA_EN_1_5B_V5)
    cfg["vocab_size"], cfg["n_layers"] = 4096, 3
    W = _rand_weights_gpu(cfg, 77)
    rng = np.random.default_rng(sum(lens) * 31 + len(lens))
    toks = [rng.integers(0, cfg["vocab_size"], int(L)).tolist() for L in lens]
    model = st.SentenceTransformer(config=cfg, weights=W)
    before, ao0, gu0 = st.debug_counter("few_passes"), st.debug_counter("few_ao_passes"), st.debug_counter("few_gu8_passes")
    hs = model.last_hidden_state(toks)
    e = model.encode_tokens(toks, batch_size=len(lens), normalize_embeddings=True)
    assert st.debug_counter("few_passes") - before == 2
    # one sequence of <= 32 tokens: the attention ran inside the O projection's workgroups (few_ao_kernel)
    assert st.debug_counter("few_ao_passes") - ao0 == (2 if len(lens) == 1 and lens[0] <= 32 else 0)
    assert st.debug_counter("few_gu8_passes") - gu0 == 2         # gate/up on 8-feature units (few_gu8_kernel) at these widths
    Wc = {k: v.float().cpu() for k, v in W.items()}
    ids, cu = np.concatenate(toks), np.concatenate([[0], np.cumsum(lens)])
    with torch.no_grad():
        ref_h = E.stack_forward(E.EncoderConfig(**cfg), Wc, ids, cu).numpy()
        ref_e = E.encode(E.EncoderConfig(**cfg), Wc, ids, cu, True).numpy()
    assert hs.shape == ref_h.shape and np.isfinite(hs).all()
    cos = (hs * ref_h).sum(1) / (np.linalg.norm(hs, axis=1) * np.linalg.norm(ref_h, axis=1))
    assert cos.min() > 1 - 1e-3, (cos.min(), int(cos.argmin()))
    rel = np.linalg.norm(hs - ref_h, axis=1) / np.linalg.norm(ref_h, axis=1)
    assert rel.max() < 3e-2, rel.max()
    assert ((e * ref_e).sum(1)).min() > 1 - 1e-3


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("ntok", [1, 7, 16, 17, 25, 31, 32])
def test_attention_inside_the_o_projection_vs_the_two_launches(st, ntok, causal, monkeypatch):
    """One prompted query (reference README.md:28) of <= 32 tokens: few_ao_kernel -- the attention computed by the O
    projection's own workgroups from Q / K / V^T pieces the QKV epilogue writes, its output registers the B operand of the
    projection -- against few_attn_kernel + few_o_kernel (MI_NO_FEW_AO=1: rows of Q | K, V^T rows, fragments through memory)
    and against the fp32 oracle, bidirectional and causal, at stella's widths, at the token-tile and key-tile edges.  The second
    model also takes the gate/up projection on 16-feature unit pairs (MI_NO_FEW_GU8=1: few_gemm_kernel<FEW_GU>) where the
    first runs few_gu8_kernel, and the QKV projection with its fragments staged through LDS (MI_NO_FEW_QKV8=1) where the first
    keeps them in registers (few_qkv8_kernel)."""
    import torch
    from oracle import encoder_oracle as E
    cfg = dict(st.STELLA_EN_1_5B_V5)
    cfg["vocab_size"], cfg["n_layers"], cfg["causal"] = 4096, 3, causal
    W = _rand_weights_gpu(cfg, 91)
    rng = np.random.default_rng(ntok * 7 + causal)
    toks = [rng.integers(0, cfg["vocab_size"], ntok).tolist()]
    ao0, gu0 = st.debug_counter("few_ao_passes"), st.debug_counter("few_gu8_passes")
    model = st.SentenceTransformer(config=cfg, weights=W)
    hs = model.last_hidden_state(toks)
    e = model.encode_tokens(toks, batch_size=1, normalize_embeddings=True)
    e2 = model.encode_tokens(toks, batch_size=1, normalize_embeddings=True)
    assert st.debug_counter("few_ao_passes") - ao0 == 3
    assert np.array_equal(e, e2)                                  # no atomics: run to run identical
    assert st.debug_counter("few_gu8_passes") - gu0 == 3
    monkeypatch.setenv("MI_NO_FEW_AO", "1")
    monkeypatch.setenv("MI_NO_FEW_GU8", "1")
    monkeypatch.setenv("MI_NO_FEW_QKV8", "1")
    q0 = st.debug_counter("few_qkv8_passes")
    model2 = st.SentenceTransformer(config=cfg, weights=W)
    hs2 = model2.last_hidden_state(toks)
    e_two = model2.encode_tokens(toks, batch_size=1, normalize_embeddings=True)
    assert st.debug_counter("few_ao_passes") - ao0 == 3           # the two launches took these
    assert st.debug_counter("few_gu8_passes") - gu0 == 3 and st.debug_counter("few_qkv8_passes") == q0
    cos2 = (hs * hs2).sum(1) / (np.linalg.norm(hs, axis=1) * np.linalg.norm(hs2, axis=1))
    assert cos2.min() > 1 - 2e-4 and ((e * e_two).sum(1)).min() > 1 - 2e-4, (cos2.min(), (e * e_two).sum(1))
    Wc = {k: v.float().cpu() for k, v in W.items()}
    ids, cu = np.asarray(toks[0]), np.asarray([0, ntok])
    with torch.no_grad():
        ref_h = E.stack_forward(E.EncoderConfig(**cfg), Wc, ids, cu).numpy()
        ref_e = E.encode(E.EncoderConfig(**cfg), Wc, ids, cu, True).numpy()
    cos = (hs * ref_h).sum(1) / (np.linalg.norm(hs, axis=1) * np.linalg.norm(ref_h, axis=1))
    assert cos.min() > 1 - 1e-3, (cos.min(), int(cos.argmin()))
    assert ((e * ref_e).sum(1)).min() > 1 - 1e-3


@pytest.mark.parametrize("tile", ["big", "slab8", "slab4", "mid", "mid64", "small", "tiny", "128"])
def test_gemm_tile_configs_vs_torch(st, tile, monkeypatch):
    """every GEMM tile configuration (forced through MI_GEMM_TILE) against torch fp32"""
    import torch
    monkeypatch.setenv("MI_GEMM_TILE", tile)
    g = torch.Generator(device="cuda").manual_seed(3)
    # (777, 260, 128): N % 8 != 0 sends the slab configurations to the ring kernel; (300, 264, 64) and (513, 520, 192)
    # are ragged in M and N on the slab kernel itself, with 2 and 6 K steps (shorter than its pipeline)
    for M, N, K in ((1024, 512, 1536), (40, 1536, 256), (777, 260, 128), (300, 264, 64), (513, 520, 192)):
        A = torch.randn((M, K), generator=g, device="cuda").bfloat16()
        W = (torch.randn((N, K), generator=g, device="cuda") / K ** 0.5).bfloat16()
        A[:, 0] += 3.0
        W[0, :] += 0.5
        C = st.gemm_bf16(A, W).float()
        ref = A.float() @ W.float().T
        err = (C - ref).abs().max().item()
        assert err <= 1e-2 * ref.abs().max().item() + 1e-3, (tile, M, N, K, err)


@pytest.mark.parametrize("tile", ["slab8", "slab4"])
def test_tiny_model_vs_golden_on_the_slab_kernel(st, gold, tile, monkeypatch):
    """every fused epilogue of the hand-ordered 256x256 kernel (QKV + bias + V^T, residual, SwiGLU, plain
    store), forced onto the tiny golden model (tensors from transformers.Qwen2Model)"""
    from oracle import encoder_oracle as E
    monkeypatch.setenv("MI_GEMM_TILE", tile)
    W = E.synth_weights(E.TINY, int(gold["seed"]))
    model = st.SentenceTransformer(config=E.TINY.to_dict(), weights=W)
    toks = _split(gold["ids"], gold["cu_seqlens"])
    hs = model.last_hidden_state(toks)
    ref = gold["hidden_bidir"]
    cos = (hs * ref).sum(1) / (np.linalg.norm(hs, axis=1) * np.linalg.norm(ref, axis=1))
    assert cos.min() > 1 - 1e-3, cos.min()
    e = model.encode_tokens(toks, batch_size=3, normalize_embeddings=True)
    assert ((e * gold["embed_bidir"]).sum(1)).min() > 1 - 1e-3


def test_large_batch_slab_kernel_with_tail_split_vs_small_tiles(st, monkeypatch):
    """33 280 tokens through a 2-layer model whose GEMMs take the default big-tile path: the slab kernel on
    every projection, 260 output tiles on 256 CUs for the residual GEMMs -- the last 8 are split along K
    with f32 atomics (the wave-quantisation tail).  Same embeddings as the 128x128 ring tiles and as the
    old 256x256 ring kernel."""
    from oracle import encoder_oracle as E
    cfg = E.EncoderConfig(vocab_size=64, hidden=512, n_layers=2, n_heads=4, n_kv_heads=2, head_dim=128,
                          intermediate=1024, dense_out=64, max_seq_len=128)
    W = E.synth_weights(cfg, 11)
    rng = np.random.default_rng(11)
    toks = [rng.integers(0, 64, 128).tolist() for _ in range(260)]
    outs = {}
    for name, env in (("default", {"MI_TAIL_SPLIT_FORCE": "1"}), ("mid", {"MI_GEMM_TILE": "mid"}), ("ring", {"MI_GEMM_RING": "1"}),
                      ("nosplit", {})):
        for k in ("MI_GEMM_TILE", "MI_GEMM_RING", "MI_TAIL_SPLIT_FORCE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        model = st.SentenceTransformer(config=cfg.to_dict(), weights=W)
        model.token_budget = None                    # ONE pass of 260 sequences = 33 280 tokens: 130 x 2 = 260 tiles, 4 in the tail
        before = st.debug_counter("tail_split_launches")
        outs[name] = model.encode_tokens(toks, batch_size=260, normalize_embeddings=True)
        took = st.debug_counter("tail_split_launches") - before
        # both residual GEMMs (O, down) of both layers split their tail: the slab kernel under MI_TAIL_SPLIT_FORCE (its cost
        # model would not: short K), the old ring kernel by its unpriced rule; 128x128 tiles and the unforced default never
        assert took == (4 if name in ("default", "ring") else 0), (name, took)
    # (the cost model of the launcher would not split these short-K tails: MI_TAIL_SPLIT_FORCE exercises the path)
    for name in ("mid", "ring", "nosplit"):
        cos = (outs["default"] * outs[name]).sum(1)
        assert cos.min() > 1 - 1e-3, (name, cos.min())
        assert np.abs(outs["default"] - outs[name]).max() < 1e-2, name


def test_many_sequences_pool_through_the_gemm_path(st, monkeypatch):
    """>= 64 sequences: final norm over all tokens + mean-pool kernel + Dense as one GEMM + row normalisation; the same
    embeddings as the per-sequence pool kernel (MI_POOL_GEMM=0) and as the fp32 oracle, raw and normalised."""
    import torch
    from oracle import encoder_oracle as E
    cfg = E.EncoderConfig(vocab_size=64, hidden=256, n_layers=2, n_heads=4, n_kv_heads=2, head_dim=64,
                          intermediate=384, dense_out=64, max_seq_len=128)
    W = E.synth_weights(cfg, 21)
    rng = np.random.default_rng(21)
    lens = rng.integers(1, 60, 90)
    toks = [rng.integers(0, 64, int(L)).tolist() for L in lens]
    model = st.SentenceTransformer(config=cfg.to_dict(), weights=W)
    outs = {}
    for norm in (True, False):
        for mode in ("1", "0"):
            monkeypatch.setenv("MI_POOL_GEMM", mode)
            outs[(norm, mode)] = model.encode_tokens(toks, batch_size=90, normalize_embeddings=norm)
    cu = np.concatenate([[0], np.cumsum(lens)])
    Wc = {k: v.float().cpu() if hasattr(v, "float") else v for k, v in W.items()}
    with torch.no_grad():
        ref = E.encode(cfg, Wc, np.concatenate(toks), cu, True).numpy()
    for mode in ("1", "0"):
        assert ((outs[(True, mode)] * ref).sum(1)).min() > 1 - 1e-3, mode
    # the two pooling paths round at the same points (bf16 normalised hidden states, bf16 pooled vector) and sum a
    # column in the same order: what is left between them is the f32 summation order of the Dense dot products
    assert np.abs(outs[(True, "1")] - outs[(True, "0")]).max() < 2e-6
    a, b = outs[(False, "1")], outs[(False, "0")]
    assert np.abs(a - b).max() < 1e-5 * np.abs(b).max() + 1e-6




@pytest.mark.parametrize("shape", ["hd64_pair", "hd128_pair", "hd128_single", "hd64_single"])
def test_fuzz_packed_batches_vs_oracle(st, shape):
    """seeded fuzz over what the packed layout can look like: sequences of 1 .. max_seq_len tokens back to back with no
    alignment (every start offset mod 8, lengths across the 64-key chunk and 64-query block boundaries), 1 .. 60 sequences
    per pass, both head dims, paired and single query heads per workgroup, bidirectional and causal -- last hidden state
    and embedding against the fp32 oracle.  (reference Makefile:65: the batch encode of arbitrary abstracts.)"""
    import os as _os
    import torch
    from oracle import encoder_oracle as E
    seed = int(_os.environ.get("MI_FUZZ_SEED", "0"))
    hd = 64 if shape.startswith("hd64") else 128
    heads, kv = (4, 2) if shape.endswith("pair") else (2, 2)
    rng = np.random.default_rng(4242 + seed + (hd + heads))
    for trial in range(6):
        cfg = replace(E.TINY, head_dim=hd, n_heads=heads, n_kv_heads=kv, max_seq_len=200, causal=bool(trial % 2))
        W = E.synth_weights(cfg, 100 + trial)
        model = st.SentenceTransformer(config=cfg.to_dict(), weights=W)
        nseq = int(rng.choice([1, 2, 5, 17, 60]))
        kind = trial % 3
        if kind == 0:
            lens = rng.integers(1, 12, nseq)                     # many tiny sequences: several per 64-row block
        elif kind == 1:
            lens = rng.integers(1, 201, nseq)                    # anything
        else:
            lens = rng.choice([1, 7, 8, 9, 63, 64, 65, 127, 128, 129, 191, 192, 193, 200], nseq)
        toks = [rng.integers(0, cfg.vocab_size, int(L)).tolist() for L in lens]
        ids = np.concatenate(toks)
        cu = np.concatenate([[0], np.cumsum(lens)])
        with torch.no_grad():
            ref_h = E.stack_forward(cfg, W, ids, cu).numpy()
            ref_e = E.encode(cfg, W, ids, cu, True).numpy()
        ctx = dict(shape=shape, trial=trial, nseq=nseq, lens=[int(v) for v in lens][:20])
        hs = model.last_hidden_state(toks)
        assert hs.shape == ref_h.shape and np.isfinite(hs).all(), ctx
        cos = (hs * ref_h).sum(1) / (np.linalg.norm(hs, axis=1) * np.linalg.norm(ref_h, axis=1))
        assert cos.min() > 1 - 1e-3, (cos.min(), int(cos.argmin()), ctx)
        model.token_budget = None
        e = model.encode_tokens(toks, batch_size=int(rng.choice([1, 4, 64])), normalize_embeddings=True)
        assert ((e * ref_e).sum(1)).min() > 1 - 1e-3, ctx
