"""GPU parity tests at the shapes of BASELINE.json configs[3] and configs[4]:

cfg4  IVF65536,PQ64 over >= 2 M clustered 1024-d vectors, batch-1024 queries: the full search
      (two-stage coarse quantiser -> LUT -> sliced scan -> top-k) against the oracle, bit for
      bit; the same index dealt into 2 / 8 vector shards whose per-shard top-k go through the
      exchange step's packed buffer + merge == the unsharded result.
cfg1  BASELINE configs[0] on the HIP path: ONE prompted query -> SentenceTransformer.encode -> IndexFlatIP(1024)
      over 10 000 seeded unit vectors, against encoder-oracle -> oracle.flat_ip.
cfg5  end to end: SentenceTransformer.encode(texts, prompt_name="s2p_query") -> index.search at
      batches 1 / 16 / 256 against encoder-oracle -> ivfpq-oracle.

Everything goes through the C ABI (ctypes mirrors); the oracle is the checker only."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def faiss():
    import abstracts_search_amd.faiss as f
    return f


@pytest.fixture(scope="module")
def cfg4(faiss):
    """IVF65536,PQ64 over 2 M rows of the bench corpus, trained briefly (setup)."""
    import torch
    import abstracts_search_amd.synth as synth
    n, d, nlist, M = 2 * 1024 * 1024, 1024, 65536, 64
    x = synth.corpus_cuda(n, d)
    idx = faiss.IndexIVFPQ(d, nlist, M, 8, faiss.METRIC_INNER_PRODUCT)
    idx.cp.niter = idx.pq.cp.niter = 2
    idx.train(x)
    idx.add(x)
    q = synth.queries_cuda(x, 1024, seed=77)
    torch.cuda.synchronize()
    return dict(x=x, idx=idx, q=q, n=n, d=d, nlist=nlist, M=M)


def test_cfg4_full_search_vs_oracle(faiss, oracle, cfg4):
    """batch 1024 x nprobe {16, 64} x k 10 on IVF65536,PQ64: ids and f32 scores equal the oracle's"""
    import torch
    idx, q, n, nlist = cfg4["idx"], cfg4["q"], cfg4["n"], cfg4["nlist"]
    assert idx.ntotal == n
    sizes = idx.list_sizes()
    assert sizes.sum() == n
    off = np.zeros(nlist + 1, np.int64)
    np.cumsum(sizes, out=off[1:])
    codes, ids = idx.export_lists()
    assert np.array_equal(np.sort(ids), np.arange(n))                       # every row in exactly one list
    cent, cb = idx.get_centroids(), idx.get_codebook()
    qh = q.cpu().numpy()
    for nprobe in (16, 64):
        idx.nprobe = nprobe
        D, I = idx.search(q, 10)
        De, Ie, cIe, cDe = oracle.search(qh, cent, cb, off, codes, ids, nprobe, 10, return_coarse=True)
        cI, cD, _ = idx.coarse_and_lut(q, nprobe, want_lut=False)
        assert np.array_equal(cI, cIe) and np.array_equal(bits(cD), bits(cDe))   # two-stage coarse == exact chain
        assert np.array_equal(I.cpu().numpy(), Ie), nprobe
        assert np.array_equal(bits(D.cpu().numpy()), bits(De)), nprobe
    # the stored codes are the oracle's encoding of the same rows (sample), in insertion order
    rows = np.arange(0, n, 32771)[:64]
    ln, cs = oracle.encode(cfg4["x"][torch.as_tensor(rows, device="cuda")].cpu().numpy(), cent, cb)
    pos = np.empty(n, np.int64)
    pos[ids] = np.arange(n)
    for r, l, c in zip(rows, ln, cs):
        p = pos[r]
        assert off[l] <= p < off[l + 1] and np.array_equal(codes[p], c)
    for l in np.flatnonzero(sizes > 1)[:50]:
        assert (np.diff(ids[off[l]:off[l + 1]]) > 0).all()                  # add() order inside a list


@pytest.mark.parametrize("nshards", [2, 8])
def test_cfg4_vector_shards_equal_unsharded(faiss, cfg4, nshards):
    """rows i mod S -> S shard indexes numbered by position; their (D, I) in the exchange step's
    packed layout, merged with the closed-form id map == the unsharded search, bit for bit"""
    import torch
    idx, q, x = cfg4["idx"], cfg4["q"], cfg4["x"]
    nq, k, nprobe = q.shape[0], 10, 16
    idx.nprobe = nprobe
    D0, I0 = idx.search(q, k)
    cent = torch.from_numpy(idx.get_centroids()).cuda()
    cb = torch.from_numpy(idx.get_codebook()).cuda()
    dbytes = (nq * k * 4 + 7) // 8 * 8
    blk = dbytes + nq * k * 8
    buf = torch.zeros(nshards * blk, dtype=torch.uint8, device="cuda")
    shards = []
    for p in range(nshards):
        sh = faiss.IndexIVFPQ(cfg4["d"], cfg4["nlist"], cfg4["M"], 8, faiss.METRIC_INNER_PRODUCT)
        sh.set_centroids(cent)
        sh.set_codebook(cb)
        sh.add(x[p::nshards].contiguous())
        sh.nprobe = nprobe
        part = buf[p * blk:(p + 1) * blk]
        sh.search_into(q, k, part[:nq * k * 4].view(torch.float32).view(nq, k), part[dbytes:].view(torch.int64).view(nq, k))
        shards.append(sh)
    D, I = faiss.merge_topk_gathered(buf, nshards, nq, k, blk, (nshards, 0, 1))
    assert torch.equal(I, I0) and torch.equal(D.view(torch.int32), D0.view(torch.int32))
    # a rank that brought its own slice of the batch merges that slice only
    Ds, Is = faiss.merge_topk_gathered(buf, nshards, nq, k, blk, (nshards, 0, 1), q_lo=256, nq_out=100)
    assert torch.equal(Is, I0[256:356]) and torch.equal(Ds, D0[256:356])


@pytest.mark.parametrize("nparts,k", [(8, 64), (8, 1000), (3, 4096), (8, 4096)])
def test_merge_tiers(faiss, oracle, nparts, k):
    """the k-way merge in every LDS tier (<= 64 KiB, <= 160 KiB, rows of pairs + select) against
    the oracle's merge: ties across parts, empty slots, k larger than what the parts hold"""
    import torch
    rng = np.random.default_rng(nparts * 10000 + k)
    nq = 5
    D = np.round(rng.standard_normal((nparts, nq, k)), 2).astype(np.float32)      # many exact ties
    I = rng.permutation(nparts * nq * k).reshape(nparts, nq, k).astype(np.int64)
    D = -np.sort(-D, axis=2) + np.float32(0)      # no -0.0: a chain of fmafs from +0 never produces one either
    I[:, 1, k // 2:] = -1                                                         # half-empty lists
    I[1:, 2, :] = -1
    I[:, 3, :] = -1                                                               # nothing at all
    De, Ie = oracle.merge(D, I)
    Dg, Ig = faiss.merge_topk(torch.from_numpy(D).cuda(), torch.from_numpy(I).cuda())
    assert np.array_equal(Ig.cpu().numpy(), Ie)
    assert np.array_equal(bits(Dg.cpu().numpy()), bits(De))


def test_write_read_index_through_the_c_abi(faiss, cfg4, tmp_path):
    """mi_index_save / mi_index_load (faiss's IwPQ + on-disk lists) at IVF65536 scale: the file
    parses with the independent Python reader, and the re-loaded index answers identically"""
    import torch
    import abstracts_search_amd.faiss_io as fio
    idx, q = cfg4["idx"], cfg4["q"]
    idx.nprobe = 16
    f, data = str(tmp_path / "index.faiss"), str(tmp_path / "ondisk.ivfdata")
    faiss.write_index(idx, f, ondisk_data=data)
    assert os.path.getsize(data) == idx.ntotal * (cfg4["M"] + 8)
    z = fio.parse(f)
    assert z["ntotal"] == idx.ntotal and z["nprobe"] == 16 and z["nlist"] == cfg4["nlist"]
    assert np.array_equal(z["sizes"], idx.list_sizes())
    codes, ids = idx.export_lists()
    assert np.array_equal(z["codes"], codes) and np.array_equal(z["ids"], ids)
    back = faiss.read_index(f)
    assert back.ntotal == idx.ntotal and back.nprobe == 16
    D0, I0 = idx.search(q[:256].contiguous(), 10)
    D1, I1 = back.search(q[:256].contiguous(), 10)
    assert torch.equal(I0, I1) and torch.equal(D0, D1)


# ----------------------------------------------------------------------
# cfg1: one encoded query -> IndexFlatIP over 10 k x 1024 (BASELINE.json configs[0], the reference's plumbing case)
# ----------------------------------------------------------------------
def test_cfg1_one_query_flat_ip_10k_chain(faiss, oracle, tmp_path):
    import dataclasses
    import torch
    import abstracts_search_amd.sentence_transformers as st
    from oracle import encoder_oracle as E
    from helpers_modeldir import write_model_dir
    cfg = dataclasses.replace(E.TINY, dense_out=1024, max_seq_len=64)           # the reference's embedding width
    W = E.synth_weights(cfg, 31)
    mdir = tmp_path / "model"
    write_model_dir(mdir, cfg, W, max_seq_length=64, prompts={"s2p_query": "query : "})
    model = st.SentenceTransformer(str(mdir), trust_remote_code=True)
    d, n, k = 1024, 10000, 10
    assert model.get_sentence_embedding_dimension() == d
    rng = np.random.default_rng(1234)
    base = rng.standard_normal((n, d)).astype(np.float32)
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    idx = faiss.IndexFlatIP(d)
    idx.add(base)
    assert idx.ntotal == n
    query = " ".join(f"w{int(t)}" for t in rng.integers(0, cfg.vocab_size - 3, 17))
    qe = model.encode(query, prompt_name="s2p_query", normalize_embeddings=True)    # ONE string, as app.py calls it
    assert qe.shape == (d,) and qe.dtype == np.float32
    D, I = idx.search(qe[None, :], k)
    # (1) the search on the embedding the HIP encoder produced: ids and score bits equal the oracle's IndexFlatIP
    De, Ie = oracle.flat_ip(qe[None, :], base, k)
    assert np.array_equal(I, Ie) and np.array_equal(bits(D), bits(De))
    # (2) the embedding against the fp32 oracle on the same token ids: the north star's tolerance
    toks = model.tokenize(["query : " + query])
    cu = np.array([0, len(toks[0])])
    with torch.no_grad():
        qr = E.encode(cfg, W, np.asarray(toks[0]), cu, True).numpy()
    cos = float((qe * qr[0]).sum())
    assert cos > 1 - 1e-3, cos
    # (3) the whole chain on the oracle side: every rank whose gap to its neighbours exceeds the score error bound
    # |<qe - qr, x>| <= |qe - qr| (unit rows) must hold the same id
    kk = 256
    Dr, Ir = oracle.flat_ip(qr, base, kk)
    eps = float(np.linalg.norm(qe - qr[0])) * 1.01 + 1e-6
    gaps = Dr[0, :-1] - Dr[0, 1:]
    same = [r for r in range(k) if gaps[r] > 2 * eps and (r == 0 or gaps[r - 1] > 2 * eps)]
    assert all(I[0, r] == Ir[0, r] for r in same), (I, Ir)
    # ... and as sets: everything the oracle chain scores above its k-th by more than the bound is in the HIP top-k, and
    # nothing in the HIP top-k scores below the oracle chain's k-th by more than the bound
    kth = Dr[0, k - 1]
    assert Dr[0, kk - 1] < kth - 2 * eps, "widen kk"
    must = set(Ir[0, Dr[0] > kth + 2 * eps].tolist())
    may = set(Ir[0, Dr[0] >= kth - 2 * eps].tolist())
    got = set(I[0].tolist())
    assert must <= got <= may, (must - got, got - may)
    print(f"cfg1 chain: cosine {cos:.6f}; {len(same)} of {k} ranks separated by more than the bound, all agree")


# ----------------------------------------------------------------------
# cfg5: encode + search, end to end
# ----------------------------------------------------------------------
def test_cfg5_encode_then_search_vs_oracle_chain(faiss, oracle, tmp_path):
    import torch
    import abstracts_search_amd.sentence_transformers as st
    from oracle import encoder_oracle as E
    from helpers_modeldir import write_model_dir
    cfg = E.TINY
    W = E.synth_weights(cfg, 21)
    mdir = tmp_path / "model"
    write_model_dir(mdir, cfg, W, max_seq_length=48, prompts={"s2p_query": "query : "})
    model = st.SentenceTransformer(str(mdir), trust_remote_code=True)
    d = cfg.dense_out
    rng = np.random.default_rng(21)
    nwords = cfg.vocab_size - 3

    def text(lo, hi):
        return " ".join(f"w{int(t)}" for t in rng.integers(0, nwords, int(rng.integers(lo, hi))))

    docs = [text(5, 40) for _ in range(6000)]
    emb = model.encode(docs, batch_size=256, normalize_embeddings=True)          # documents are encoded bare
    nlist, M, k, nprobe = 64, 8, 10, 8
    idx = faiss.IndexIVFPQ(d, nlist, M, 8, faiss.METRIC_INNER_PRODUCT)
    idx.cp.niter = idx.pq.cp.niter = 8
    idx.train(emb)
    idx.add(emb)
    idx.nprobe = nprobe
    cent, cb = idx.get_centroids(), idx.get_codebook()
    sizes = idx.list_sizes()
    off = np.zeros(nlist + 1, np.int64)
    np.cumsum(sizes, out=off[1:])
    codes, ids = idx.export_lists()
    # largest norm of a decoded vector: |<dq, x^>| <= |dq| * this
    dec = cent[np.repeat(np.arange(nlist), sizes)] + np.concatenate([cb[m][codes[:, m]] for m in range(M)], axis=1)
    xnorm = float(np.linalg.norm(dec, axis=1).max())
    cnorm = float(np.linalg.norm(cent, axis=1).max())
    compared = agreed = total = overlap = 0
    for batch in (1, 16, 256):
        queries = [text(3, 20) for _ in range(batch)]
        qe = model.encode(queries, prompt_name="s2p_query", batch_size=64, normalize_embeddings=True)
        assert qe.shape == (batch, d)
        D, I = idx.search(qe, k)
        # (1) the search half alone, on the embeddings the HIP encoder produced: bit-exact
        De, Ie = oracle.search(qe, cent, cb, off, codes, ids, nprobe, k)
        assert np.array_equal(I, Ie) and np.array_equal(bits(D), bits(De)), batch
        # (2) the whole chain on the oracle side: fp32 encoder on the same token ids -> oracle search
        toks = model.tokenize(["query : " + s for s in queries])
        cu = np.concatenate([[0], np.cumsum([len(t) for t in toks])])
        with torch.no_grad():
            qr = E.encode(E.EncoderConfig(**{**cfg.to_dict(), "max_seq_len": 48}), W, np.concatenate(toks), cu, True).numpy()
        cos = (qe * qr).sum(1)
        assert cos.min() > 1 - 1e-3, cos.min()                                   # the north star's tolerance
        Dr, Ir, cIr, cDr = oracle.search(qr, cent, cb, off, codes, ids, nprobe + 1, k + 1, return_coarse=True)
        dq = np.linalg.norm(qe - qr, axis=1)
        for j in range(batch):
            eps_s, eps_c = dq[j] * xnorm * 1.01 + 1e-6, dq[j] * cnorm * 1.01 + 1e-6
            total += k
            # (the (nprobe+1)-list run above only supplies the coarse margin; ranks come from an nprobe-list run)
            Dj, Ij = oracle.search(qr[j:j + 1], cent, cb, off, codes, ids, nprobe, k + 1)
            overlap += len(set(I[j].tolist()) & set(Ij[0, :k].tolist()))
            if cDr[j, nprobe - 1] - cDr[j, nprobe] <= 2 * eps_c:
                continue                                                          # the probe set itself is within the error
            gaps = Dj[0, :-1] - Dj[0, 1:]                                         # gap below rank r, r = 0..k-1
            for r in range(k):
                lo_ok = gaps[r] > 2 * eps_s
                hi_ok = r == 0 or gaps[r - 1] > 2 * eps_s
                if lo_ok and hi_ok and Ij[0, r] >= 0:
                    compared += 1
                    agreed += int(I[j, r] == Ij[0, r])
    print(f"cfg5 chain: {compared} of {total} ranks separated by more than the bf16 score error bound; {agreed} agree; "
          f"top-{k} overlap of the two chains {overlap / total:.4f}")
    # the bound is Cauchy-Schwarz on the measured embedding difference (conservative): few ranks of a
    # random-init model's near-collinear embeddings clear it, every one that does must agree
    assert compared >= 100 and agreed == compared, (compared, agreed, total)
    assert overlap / total > 0.9


def test_native_exchange_step_rccl_world1(faiss, cfg4):
    """mi_shards_search (local search -> ncclAllGather -> merge inside the C ABI) on the real RCCL
    at world size 1, plain and with a half-precision refine stage and an affine id map: equal to
    the same search without the exchange."""
    import torch
    from abstracts_search_amd.shards import NativeShardedIndex
    idx, q, x = cfg4["idx"], cfg4["q"], cfg4["x"]
    idx.nprobe = 16
    D0, I0 = idx.search(q, 10)
    sh = NativeShardedIndex(idx, rank=0, world=1)
    D1, I1 = sh.search_replicated(q, 10)
    torch.cuda.synchronize()
    assert torch.equal(I1, I0) and torch.equal(D1, D0)
    # refine shard numbered by position, global id = 3 * local + 1
    n = 200_000
    base = faiss.IndexIVFPQ(cfg4["d"], cfg4["nlist"], cfg4["M"], 8, faiss.METRIC_INNER_PRODUCT)
    base.set_centroids(torch.from_numpy(idx.get_centroids()).cuda())
    base.set_codebook(torch.from_numpy(idx.get_codebook()).cuda())
    ref = faiss.IndexRefine(base, faiss.IndexScalarQuantizer(cfg4["d"]))
    ref.add(x[:n].contiguous())
    base.nprobe, ref.k_factor = 8, 16
    De, Ie = ref.search(q[:128].contiguous(), 10)
    shr = NativeShardedIndex(ref, id_affine=(3, 1, 0), rank=0, world=1)
    Dr, Ir = shr.search_replicated(q[:128].contiguous(), 10)
    torch.cuda.synchronize()
    assert torch.equal(Dr, De) and torch.equal(Ir, torch.where(Ie < 0, Ie, Ie * 3 + 1))


@pytest.mark.parametrize("nslices", [2, 4, 8])
def test_cfg4_coarse_quantiser_split_by_centroid_range(faiss, cfg4, nslices):
    """shard_coarse: every rank quantises against its slice of the 65536 centroids (the two-stage
    quantiser on the slice for 2 / 4 slices, the exact GEMM for 8), the per-slice top-nprobe lists
    merge to exactly the full quantiser's lists and scores"""
    import torch
    idx, q = cfg4["idx"], cfg4["q"]
    nprobe, nlist = 64, cfg4["nlist"]
    cI0, cD0, _ = idx.coarse_and_lut(q, nprobe, want_lut=False)
    per = nlist // nslices
    parts_I, parts_D = [], []
    for s_ in range(nslices):
        cI, cD = idx.coarse_slice(q, nprobe, s_ * per, (s_ + 1) * per)
        assert int(cI.min()) >= s_ * per and int(cI.max()) < (s_ + 1) * per
        parts_I.append(cI.to(torch.int64))
        parts_D.append(cD)
    D, I = faiss.merge_topk(torch.stack(parts_D), torch.stack(parts_I))
    assert np.array_equal(I.cpu().numpy(), cI0.astype(np.int64))
    assert np.array_equal(bits(D.cpu().numpy()), bits(cD0))
