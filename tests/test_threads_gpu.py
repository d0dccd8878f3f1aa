"""Concurrent readers on one handle ("search thread-safe for concurrent readers,
add/train exclusive"; the reference's query-time app.py, README.md:18-29, is a served web app).

Four host threads, each on its own HIP stream, hammer ONE index / refine / encoder handle with
different batches; ctypes drops the GIL inside a C-ABI call, so the calls really overlap on the
host and on the GPU.  Every result must equal, bit for bit, what the same call returns when it
runs alone (the serial pass is also what the oracle-parity tests of test_ivfpq_gpu.py check)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NT = 4


def _run_threads(work):
    """work(t) -> result, on NT threads released together; returns the list of results"""
    out, err = [None] * NT, []
    gate = threading.Barrier(NT)

    def body(t):
        try:
            gate.wait()
            out[t] = work(t)
        except BaseException as e:                      # noqa: BLE001 -- reported by the caller's assert
            err.append((t, repr(e)))

    th = [threading.Thread(target=body, args=(t,)) for t in range(NT)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not err, err
    return out


def _problem(seed, d, M, nlist, n, nq):
    rng = np.random.default_rng(seed)
    cent = rng.standard_normal((nlist, d)).astype(np.float32)
    cb = (0.3 * rng.standard_normal((M, 256, d // M))).astype(np.float32)
    x = (cent[rng.integers(0, nlist, n)] + 0.3 * rng.standard_normal((n, d))).astype(np.float32)
    q = (x[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, d))).astype(np.float32)
    return cent, cb, x, q


def test_concurrent_index_search_equals_serial():
    import torch
    import abstracts_search_amd.faiss as faiss
    d, M, nlist, n = 128, 16, 256, 60000
    cent, cb, x, q = _problem(3, d, M, nlist, n, NT * 6 * 96)
    idx = faiss.IndexIVFPQ(d, nlist, M, 8, faiss.METRIC_INNER_PRODUCT)
    idx.set_centroids(cent)
    idx.set_codebook(cb)
    idx.add(x)
    qd = torch.from_numpy(q).cuda().view(NT, 6, 96, d)
    shapes = [(10, 8), (70, 16), (10, 32), (200, 8), (1, 1), (10, 64)]       # (k, nprobe): in-scan top-k, all-pairs path, ...
    # the first search after an add builds the scan image: let the THREADS race for it
    streams = [torch.cuda.Stream() for _ in range(NT)]

    def work(t):
        res = []
        with torch.cuda.stream(streams[t]):
            for rep in range(3):
                for j, (k, nprobe) in enumerate(shapes):
                    D, I = idx.search(qd[t, j], k, nprobe=nprobe)
                    res.append((D, I))
            streams[t].synchronize()
        return [(D.cpu().numpy(), I.cpu().numpy()) for D, I in res]

    got = _run_threads(work)
    # host-pointer calls from threads share the NULL stream's workspace set: they take turns
    host = _run_threads(lambda t: idx.search(q[t * 96:(t + 1) * 96], 10, nprobe=8))
    for t in range(NT):
        for rep in range(3):
            for j, (k, nprobe) in enumerate(shapes):
                D, I = idx.search(qd[t, j], k, nprobe=nprobe)
                Dg, Ig = got[t][rep * len(shapes) + j]
                assert np.array_equal(I.cpu().numpy(), Ig), (t, rep, j)
                assert np.array_equal(D.cpu().numpy().view(np.uint32), Dg.view(np.uint32)), (t, rep, j)
        D, I = idx.search(q[t * 96:(t + 1) * 96], 10, nprobe=8)
        assert np.array_equal(I, host[t][1]) and np.array_equal(D.view(np.uint32), host[t][0].view(np.uint32))


@pytest.mark.parametrize("store", ["flat", "sqfp16", "sq8"])
def test_concurrent_refine_and_flat_search_equal_serial(store):
    import torch
    import abstracts_search_amd.faiss as faiss
    d, M, nlist, n = 64, 8, 64, 20000
    cent, cb, x, q = _problem(5, d, M, nlist, n, NT * 64)
    base = faiss.IndexIVFPQ(d, nlist, M, 8, faiss.METRIC_INNER_PRODUCT)
    base.set_centroids(cent)
    base.set_codebook(cb)
    if store == "flat":
        idx = faiss.IndexRefineFlat(base)
    else:
        qt = faiss.ScalarQuantizer.QT_fp16 if store == "sqfp16" else faiss.ScalarQuantizer.QT_8bit
        ref = faiss.IndexScalarQuantizer(d, qt, faiss.METRIC_INNER_PRODUCT)
        ref.train(x)
        idx = faiss.IndexRefine(base, ref)
    idx.add(x)
    idx.nprobe, idx.k_factor = 8, 6
    flat = faiss.IndexFlatIP(d)
    flat.add(x[:5000])
    qd = torch.from_numpy(q).cuda().view(NT, 64, d)
    streams = [torch.cuda.Stream() for _ in range(NT)]

    def work(t):
        with torch.cuda.stream(streams[t]):
            r = [idx.search(qd[t], 10) for _ in range(4)] + [flat.search(qd[t], 7) for _ in range(4)]
            streams[t].synchronize()
        return [(D.cpu().numpy(), I.cpu().numpy()) for D, I in r]

    got = _run_threads(work)
    for t in range(NT):
        D, I = idx.search(qd[t], 10)
        Df, If = flat.search(qd[t], 7)
        for r in range(4):
            assert np.array_equal(got[t][r][1], I.cpu().numpy()) and np.array_equal(got[t][r][0].view(np.uint32), D.cpu().numpy().view(np.uint32))
            assert np.array_equal(got[t][4 + r][1], If.cpu().numpy()) and np.array_equal(got[t][4 + r][0].view(np.uint32), Df.cpu().numpy().view(np.uint32))


def test_concurrent_encode_equals_serial():
    import torch
    import abstracts_search_amd.sentence_transformers as st
    from oracle import encoder_oracle as E
    W = E.synth_weights(E.TINY, 7)
    model = st.SentenceTransformer(config=E.TINY.to_dict(), weights=W)
    rng = np.random.default_rng(9)
    # a single query (the few-token path builds its fragment-major weight copies on first use: raced for),
    # a handful, and a pass that pools through the GEMM path
    sets = [[rng.integers(0, E.TINY.vocab_size, int(rng.integers(3, 60))).tolist() for _ in range(ns)]
            for ns in (1, 7, 70, 3)]
    streams = [torch.cuda.Stream() for _ in range(NT)]

    def work(t):
        with torch.cuda.stream(streams[t]):
            r = [model.encode_tokens(sets[(t + j) % 4], normalize_embeddings=True, as_tensor=True) for j in range(8)]
            streams[t].synchronize()
        return [e.cpu().numpy() for e in r]

    got = _run_threads(work)
    serial = [model.encode_tokens(s, normalize_embeddings=True) for s in sets]
    for t in range(NT):
        for j in range(8):
            # (the few-token path splits K with f32 atomics: run-to-run the sums differ in their last bits, threads or not)
            assert np.abs(got[t][j] - serial[(t + j) % 4]).max() < 1e-5, (t, j)
