"""GPU parity tests: the HIP path (through the C ABI) vs the oracle, bit-exact
for codes / indices / f32 scores, on golden fixtures, seeded random indexes
and edge cases."""
import importlib
import os

import numpy as np
import pytest

# extended fuzz runs: MI_FUZZ_SEED=n shifts every fuzz test's seed (the committed suite runs n = 0)
FUZZ_SEED = int(__import__("os").environ.get("MI_FUZZ_SEED", "0"))

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def faiss():
    import abstracts_search_amd.faiss as f
    assert f.get_num_gpus() >= 1
    return f


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "ivfpq_tiny.npz"))


def make_index(faiss, cent, cb, by_residual=True):
    idx = faiss.IndexIVFPQ(cent.shape[1], cent.shape[0], cb.shape[0], 8, faiss.METRIC_INNER_PRODUCT,
                           by_residual=by_residual)
    assert not idx.is_trained
    idx.set_centroids(cent)
    idx.set_codebook(cb)
    assert idx.is_trained
    return idx


def random_problem(seed, d, M, nlist, n, nq, scale=0.3):
    rng = np.random.default_rng(seed)
    cent = rng.standard_normal((nlist, d)).astype(np.float32)
    cb = (scale * rng.standard_normal((M, 256, d // M))).astype(np.float32)
    x = (cent[rng.integers(0, nlist, n)] + scale * rng.standard_normal((n, d))).astype(np.float32)
    q = (x[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, d))).astype(np.float32)
    return cent, cb, x, q


@pytest.mark.parametrize("nq,nprobe", [(1, 64), (5, 17), (16, 1), (64, 33)])
def test_sliced_coarse_selection_of_a_few_long_rows(faiss, oracle, monkeypatch, nq, nprobe):
    """A few queries against many centroids (16 384 = 4 slices of 4 096): the coarse selection runs as one workgroup per slice + a
    second pass over the slices' results (launch_select, SelSlices) -- same probes, same coarse scores (bits) and the same search
    results as one workgroup per row (MI_NO_SLICED_SELECT=1) and as the oracle, with centroids DUPLICATED across slices so that
    equal scores meet at the cut (ties by the smaller list number)."""
    d, M, nlist, n = 64, 8, 16384, 40000
    cent, cb, x, q = random_problem(61, d, M, nlist, n, nq)
    cent[4096:4096 + 300] = cent[:300]                           # the same centroid in two slices ...
    cent[12288 + 7] = cent[5000]                                 # ... and in two others
    idx = make_index(faiss, cent, cb)
    idx.add(x)
    idx.nprobe = nprobe
    cI, cD, _ = idx.coarse_and_lut(q, nprobe, want_lut=False)
    D, I = idx.search(q, 10)
    monkeypatch.setenv("MI_NO_SLICED_SELECT", "1")
    cI0, cD0, _ = idx.coarse_and_lut(q, nprobe, want_lut=False)
    D0, I0 = idx.search(q, 10)
    assert np.array_equal(cI, cI0) and np.array_equal(bits(cD), bits(cD0))
    assert np.array_equal(I, I0) and np.array_equal(bits(D), bits(D0))
    eD, eI = oracle.flat_ip(q, cent, nprobe)                     # the coarse quantiser IS a flat inner-product search over the centroids
    assert np.array_equal(cI, eI.astype(np.int32)) and np.array_equal(bits(cD), bits(eD))


def test_golden_fixture(faiss, gold):
    cent, cb, x, q, ids = gold["centroids"], gold["codebook"], gold["x"], gold["q"], gold["ids"]
    k = int(gold["k"])
    idx = make_index(faiss, cent, cb)
    ln, codes = idx.encode(x)
    assert np.array_equal(ln, gold["list_no"])
    assert np.array_equal(codes, gold["codes"])
    idx.add_with_ids(x, ids)
    assert idx.ntotal == x.shape[0]
    _, _, lut = idx.coarse_and_lut(q[:1], 1)
    assert np.array_equal(bits(lut[0]), bits(gold["lut_q0"]))
    for nprobe in (1, 4, 16):
        cI, cD, _ = idx.coarse_and_lut(q, nprobe, want_lut=False)
        assert np.array_equal(cI, gold[f"cI_np{nprobe}"])
        assert np.array_equal(bits(cD), bits(gold[f"cD_np{nprobe}"]))
        idx.nprobe = nprobe
        D, I = idx.search(q, k)
        assert np.array_equal(I, gold[f"I_np{nprobe}"])
        assert np.array_equal(bits(D), bits(gold[f"D_np{nprobe}"]))
    flat = faiss.IndexFlatIP(x.shape[1])
    flat.add(x)
    D, I = flat.search(q, k)
    assert np.array_equal(I, gold["flat_I"])
    assert np.array_equal(bits(D), bits(gold["flat_D"]))


@pytest.mark.parametrize("d,M,nlist,n,nq", [
    (64, 8, 16, 3000, 33),       # small, ragged nq
    (128, 16, 64, 20000, 130),   # second GEMM tile config (nq > 128)
    (256, 64, 32, 5000, 17),     # M = 64, dsub = 4
    (1024, 64, 256, 30000, 64),  # the north-star shape (d=1024, PQ64), reduced N
    (96, 4, 7, 900, 5),          # odd nlist, M = 4, dsub = 24 -> unsupported? (d/M must be pow2)
])
@pytest.mark.parametrize("by_residual", [True, False])
def test_search_matches_oracle(faiss, oracle, d, M, nlist, n, nq, by_residual):
    if (d // M) not in (1, 2, 4, 8, 16, 32, 64):
        with pytest.raises(RuntimeError, match="unsupported d/M"):
            faiss.IndexIVFPQ(d, nlist, M, 8, faiss.METRIC_INNER_PRODUCT)
        return
    cent, cb, x, q = random_problem(d * 7 + M, d, M, nlist, n, nq)
    idx = make_index(faiss, cent, cb, by_residual)
    idx.add(x[: n // 2])
    idx.add(x[n // 2:])          # two add() calls: sequential ids continue from ntotal
    assert idx.ntotal == n
    ln, codes = oracle.encode(x, cent, cb, by_residual)
    off, lc, li = oracle.build_lists(ln, codes, np.arange(n), nlist)
    # inverted lists hold exactly the oracle's codes in insertion order
    for l in (0, nlist // 2, nlist - 1):
        c, i = idx.get_list(l)
        assert np.array_equal(c, lc[off[l]:off[l + 1]]) and np.array_equal(i, li[off[l]:off[l + 1]])
    for nprobe, k in ((1, 10), (5, 10), (nlist, 1), (min(nlist, 33), 64)):
        idx.nprobe = nprobe
        D, I = idx.search(q, k)
        De, Ie, cIe, cDe = oracle.search(q, cent, cb, off, lc, li, nprobe, k, by_residual, return_coarse=True)
        cI, cD, _ = idx.coarse_and_lut(q, nprobe, want_lut=False)
        assert np.array_equal(cI, cIe)
        assert np.array_equal(bits(cD), bits(cDe))
        assert np.array_equal(I, Ie), (nprobe, k, np.argwhere(I != Ie)[:5])
        assert np.array_equal(bits(D), bits(De))


@pytest.mark.parametrize("env", [{}, {"MI_SCAN_NW": "16"}, {"MI_NO_FUSED_MERGE": "1"}, {"MI_NSLICE": "1"},
                                 {"MI_NSLICE": "7"}])
def test_probe_table_paths_and_launch_variants(faiss, oracle, monkeypatch, env):
    """The scan kernel locates code groups three ways -- nprobe <= 64 (tables in
    registers), <= 256 (register prefix search over LDS tables), > 256 (LDS walk) --
    and has launch variants behind tuning knobs; all must give the oracle's bits."""
    for key, val in env.items():
        monkeypatch.setenv(key, val)
    d, M, nlist, n, nq = 64, 8, 600, 24000, 37
    cent, cb, x, q = random_problem(99, d, M, nlist, n, nq)
    idx = make_index(faiss, cent, cb)
    idx.add(x)
    ln, codes = oracle.encode(x, cent, cb, True)
    off, lc, li = oracle.build_lists(ln, codes, np.arange(n), nlist)
    for nprobe, k in ((64, 10), (65, 10), (200, 33), (256, 10), (257, 10), (600, 100)):
        idx.nprobe = nprobe
        D, I = idx.search(q, k)
        De, Ie = oracle.search(q, cent, cb, off, lc, li, nprobe, k, True)
        assert np.array_equal(I, Ie), (env, nprobe, k, np.argwhere(I != Ie)[:5])
        assert np.array_equal(bits(D), bits(De)), (env, nprobe, k)


def test_refine_flat_matches_oracle(faiss, oracle):
    """IndexRefineFlat: base IVF-PQ candidates (k * k_factor) re-ranked with exact inner
    products.  The re-ranking scores are bit-identical to IndexFlatIP.search's for the same
    ids (same kernel), the result equals the oracle's re-ranking of the oracle's candidates,
    and on clustered data recall against the exact search goes up."""
    d, M, nlist, n, nq, k = 128, 16, 64, 20000, 50, 10
    cent, cb, x, q = random_problem(5, d, M, nlist, n, nq)
    base = make_index(faiss, cent, cb)
    idx = faiss.IndexRefineFlat(base)
    idx.add(x)
    assert idx.ntotal == n and idx.refine_index.ntotal == n
    idx.nprobe = 8
    ln, codes = oracle.encode(x, cent, cb, True)
    off, lc, li = oracle.build_lists(ln, codes, np.arange(n), nlist)
    Dx, Ix = oracle.flat_ip(q, x, k)
    recalls = []
    for kf in (1, 4, 10):
        idx.k_factor = kf
        D, I = idx.search(q, k)
        _, cand = oracle.search(q, cent, cb, off, lc, li, 8, k * kf, True)
        De, Ie = oracle.rerank(q, x, cand, k)
        assert np.array_equal(I, Ie), (kf, np.argwhere(I != Ie)[:5])
        assert np.array_equal(bits(D), bits(De)), kf
        recalls.append(np.mean([len(set(I[i]) & set(Ix[i])) / k for i in range(nq)]))
    assert recalls[2] >= recalls[0] and recalls[2] > 0.5, recalls
    # scores of the kept ids are the flat index's scores, bit for bit
    flat = faiss.IndexFlatIP(d)
    flat.add(x)
    Df, If = flat.search(q, 200)
    for i in range(nq):
        pos = {int(v): j for j, v in enumerate(If[i])}
        for j in range(k):
            if int(I[i, j]) in pos:
                assert bits(D[i, j:j + 1])[0] == bits(Df[i, pos[int(I[i, j])]:pos[int(I[i, j])] + 1])[0]
    # empty slots (a base index that cannot fill k * k_factor) and the torch path
    import torch
    idx.nprobe = 1
    idx.k_factor = 50
    D, I = idx.search(q[:5], k)
    _, cand = oracle.search(q[:5], cent, cb, off, lc, li, 1, k * 50, True)
    assert (cand < 0).any()
    De, Ie = oracle.rerank(q[:5], x, cand, k)
    assert np.array_equal(I, Ie) and np.array_equal(bits(D), bits(De))
    Dt, It = idx.search(torch.from_numpy(q[:5]).cuda(), k)
    assert np.array_equal(It.cpu().numpy(), Ie) and np.array_equal(bits(Dt.cpu().numpy()), bits(De))
    # per-call parameters (faiss.SearchParametersIVF / IndexRefineSearchParameters)
    idx.nprobe, idx.k_factor = 1, 1.0
    p = faiss.IndexRefineSearchParameters(k_factor=4, base_index_params=faiss.SearchParametersIVF(nprobe=8))
    D, I = idx.search(q, k, params=p)
    _, cand = oracle.search(q, cent, cb, off, lc, li, 8, k * 4, True)
    De, Ie = oracle.rerank(q, x, cand, k)
    assert np.array_equal(I, Ie) and np.array_equal(bits(D), bits(De))
    assert idx.nprobe == 1 and idx.k_factor == 1.0            # the index attributes are untouched
    Db, Ib = idx.base_index.search(q, k, params=faiss.SearchParametersIVF(nprobe=8))
    Dbe, Ibe = oracle.search(q, cent, cb, off, lc, li, 8, k, True)
    assert np.array_equal(Ib, Ibe) and np.array_equal(bits(Db), bits(Dbe))
    with pytest.raises(NotImplementedError):
        idx.base_index.search(q, k, params=faiss.SearchParametersIVF(nprobe=8, max_codes=100))
    f = faiss.index_factory(d, "IVF64,PQ16,RFlat", faiss.METRIC_INNER_PRODUCT)
    assert isinstance(f, faiss.IndexRefineFlat) and f.base_index.nlist == 64


def test_autotune_on_the_hip_index(faiss, oracle, tmp_path):
    """The `tune` step (reference Makefile:32) over the real index: every experiment is a
    search on the HIP path; each reported recall equals the oracle's result scored the same
    way, the front is a Pareto front, and the written params reproduce the chosen point."""
    at = importlib.import_module("abstracts_search_amd.autotune")
    d, M, nlist, n, nq, k = 64, 8, 32, 12000, 64, 10
    cent, cb, x, q = random_problem(11, d, M, nlist, n, nq)
    idx = faiss.IndexRefineFlat(make_index(faiss, cent, cb))
    idx.add(x)
    _, gt = oracle.flat_ip(q, x, k)
    ln, codes = oracle.encode(x, cent, cb, True)
    off, lc, li = oracle.build_lists(ln, codes, np.arange(n), nlist)
    ps = faiss.ParameterSpace()
    ps.initialize(idx)
    assert [pr.name for pr in ps.parameter_ranges] == ["k_factor_rf", "nprobe"]
    ps.parameter_ranges[0].values = [1.0, 4.0]
    ps.parameter_ranges[1].values = [1.0, 4.0, 16.0]
    ps.n_experiments = 0
    ps.verbose = 0
    crit = faiss.IntersectionCriterion(nq, k)
    crit.set_groundtruth(None, gt)
    ops = ps.explore(idx, q, crit)
    assert len(ops.all_pts) == 6
    for p in ops.all_pts:
        kf, nprobe = (int(float(t.split("=")[1])) for t in p.key.split(","))
        _, cand = oracle.search(q, cent, cb, off, lc, li, nprobe, k * kf, True)
        _, Ie = oracle.rerank(q, x, cand, k)
        assert p.perf == crit.evaluate(None, Ie), p
    by_key = {p.key: p.perf for p in ops.all_pts}
    assert by_key["k_factor_rf=4,nprobe=16"] >= by_key["k_factor_rf=1,nprobe=1"]
    front = ops.optimal_pts
    assert all(a.perf < b.perf and a.t < b.t for a, b in zip(front, front[1:]))
    doc = at.write_params(str(tmp_path / "params.json"), ops, min_perf=front[-1].perf)
    idx.nprobe, idx.k_factor = 1, 1.0
    at.read_params(str(tmp_path / "params.json"), idx)
    assert crit.evaluate(*idx.search(q, k)) == doc["perf"] == front[-1].perf
    # torch queries take the device path; same recalls
    import torch
    ops_t = ps.explore(idx, torch.from_numpy(q).cuda(), crit)
    assert [p.perf for p in ops_t.all_pts] == [p.perf for p in ops.all_pts]
    # a plain IVF-PQ index only exposes nprobe
    ps.initialize(idx.base_index)
    assert [pr.name for pr in ps.parameter_ranges] == ["nprobe"] and ps.n_combinations() == 5
    with pytest.raises(ValueError):
        ps.set_index_parameter(idx.base_index, "k_factor_rf", 2)


def test_two_stage_coarse_is_bit_identical(faiss, oracle, monkeypatch):
    """Large batches take the two-stage coarse quantiser (f16 MFMA scores, then the exact f32
    chain for every centroid within the proven error margin of the cut).  Its result must be
    the one-stage result bit for bit -- list numbers, scores, and therefore the search --
    also on adversarial inputs: large and tiny norms, duplicated centroids (exact ties broken
    by index), rows whose f16 image overflows, a NaN query."""
    rng = np.random.default_rng(314)
    d, M, nlist, nq = 128, 16, 1024, 300
    cb = (0.3 * rng.standard_normal((M, 256, d // M))).astype(np.float32)

    def run(cent, q, nprobes, x=None):
        out = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("MI_TWO_STAGE", mode)
            idx = make_index(faiss, cent, cb)
            if x is not None:
                idx.add(x)
            res = []
            for nprobe in nprobes:
                cI, cD, _ = idx.coarse_and_lut(q, nprobe, want_lut=False)
                res.append((cI, bits(cD)))
                if x is not None:
                    idx.nprobe = nprobe
                    D, I = idx.search(q, 10)
                    res.append((I, bits(D)))
            out[mode] = res
        for a, b in zip(out["0"], out["1"]):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        return out["1"]

    # clustered centroids (many near-equal scores around the cut), unit-ish norms
    base = rng.standard_normal((64, d)).astype(np.float32)
    cent = (base[rng.integers(0, 64, nlist)] + 0.02 * rng.standard_normal((nlist, d))).astype(np.float32)
    q = (cent[rng.integers(0, nlist, nq)] + 0.1 * rng.standard_normal((nq, d))).astype(np.float32)
    x = (cent[rng.integers(0, nlist, 20000)] + 0.1 * rng.standard_normal((20000, d))).astype(np.float32)
    got = run(cent, q, (1, 16, 64, 300, 1024), x)
    De, Ie = oracle.flat_ip(q, cent, 16)
    assert np.array_equal(got[2][0], Ie) and np.array_equal(got[2][1], bits(De))
    # duplicated centroids: exact ties, the lower index wins
    cent2 = cent.copy()
    cent2[1::2] = cent2[0::2]
    run(cent2, q, (1, 7, 64))
    # large and tiny norms (the margin scales with |q| max|c|), mixed in one batch
    scale = np.where(np.arange(nq) % 3 == 0, 100.0, np.where(np.arange(nq) % 3 == 1, 1e-3, 1.0)).astype(np.float32)
    run((cent * 30).astype(np.float32), (q * scale[:, None]).astype(np.float32), (5, 64))
    run((cent * 1e-3).astype(np.float32), q, (5, 64))
    # far below f16's range, and rows whose components span 12 orders of magnitude
    run((cent * 1e-9).astype(np.float32), (q * 1e-12).astype(np.float32), (5, 64))
    wide = np.exp(rng.uniform(-14, 14, (1, d))).astype(np.float32)
    run((cent * wide).astype(np.float32), (q / wide).astype(np.float32), (5, 64))
    run((cent * wide).astype(np.float32), q, (5, 64))
    # f16 overflow in some centroids / one query, and a NaN query: exact fallback rows
    cent3 = cent.copy()
    cent3[10] *= 1e5
    cent3[500, 3] = 7e4
    q3 = q.copy()
    q3[5] *= 1e6
    run(cent3, q3, (3, 32))
    q4 = q.copy()
    q4[7, 0] = np.nan
    res = run(cent, q4, (3,))
    assert (res[0][0][7] == -1).all()
    # the default dispatch (no forcing): 2048 queries x 8192 centroids takes the two-stage path
    cent8 = rng.standard_normal((8192, d)).astype(np.float32)
    q8 = (cent8[rng.integers(0, 8192, 2048)] + 0.5 * rng.standard_normal((2048, d))).astype(np.float32)
    idx = make_index(faiss, cent8, cb)
    monkeypatch.delenv("MI_TWO_STAGE")
    cI1, cD1, _ = idx.coarse_and_lut(q8, 16, want_lut=False)
    monkeypatch.setenv("MI_TWO_STAGE", "0")
    cI0, cD0, _ = idx.coarse_and_lut(q8, 16, want_lut=False)
    assert np.array_equal(cI0, cI1) and np.array_equal(bits(cD0), bits(cD1))
    De, Ie = oracle.flat_ip(q8[:64], cent8, 16)
    assert np.array_equal(cI1[:64], Ie) and np.array_equal(bits(cD1[:64]), bits(De))
    # 1024 x 32768: the 256 x 256 tiles of the approximate GEMM, rows of eight tiles in the
    # second stage; nprobe 200 takes the block-wide descent
    cent32 = rng.standard_normal((32768, d)).astype(np.float32)
    q32 = (cent32[rng.integers(0, 32768, 1024)] + 0.5 * rng.standard_normal((1024, d))).astype(np.float32)
    idx = make_index(faiss, cent32, cb)
    for nprobe in (8, 200, 300):
        if nprobe <= 128:
            monkeypatch.delenv("MI_TWO_STAGE")             # default dispatch
        else:
            monkeypatch.setenv("MI_TWO_STAGE", "1")        # (the default wants 2^26 scores for nprobe > 128)
        cI1, cD1, _ = idx.coarse_and_lut(q32, nprobe, want_lut=False)
        monkeypatch.setenv("MI_TWO_STAGE", "0")
        cI0, cD0, _ = idx.coarse_and_lut(q32, nprobe, want_lut=False)
        assert np.array_equal(cI0, cI1) and np.array_equal(bits(cD0), bits(cD1)), nprobe
    # the same size with rows that overflow f16 (a centroid x 1e5, a query x 1e6), a NaN query and duplicated
    # centroids: here the approximate GEMM is the slab kernel, whose epilogue hands the second stage the maxima
    # of every 64-column group -- a group with a non-finite score must send its row to the exact fallback
    cent33 = cent32.copy()
    cent33[77] *= 1e5
    cent33[20001, 5] = 7e4
    cent33[1::2][:4000] = cent33[0::2][:4000]
    q33 = q32.copy()
    q33[5] *= 1e6
    q33[9, 0] = np.nan
    idx = make_index(faiss, cent33, cb)
    for gm in ("1", "0"):
        monkeypatch.setenv("MI_REFINE_GMAX", gm)
        for nprobe in (1, 8, 64):
            monkeypatch.setenv("MI_TWO_STAGE", "1")
            cI1, cD1, _ = idx.coarse_and_lut(q33, nprobe, want_lut=False)
            monkeypatch.setenv("MI_TWO_STAGE", "0")
            cI0, cD0, _ = idx.coarse_and_lut(q33, nprobe, want_lut=False)
            assert np.array_equal(cI0, cI1) and np.array_equal(bits(cD0), bits(cD1)), (gm, nprobe)
            assert (cI1[9] == -1).all()
    monkeypatch.delenv("MI_REFINE_GMAX")
    # the adversarial norms again at the size where the group maxima steer the second stage (batched 16-byte reads of the
    # surviving groups, quad-loaded exact chains): large and tiny norms mixed in one batch, far below f16's range,
    # components that span 12 orders of magnitude, near-ties around the cut (clustered centroids: hundreds of
    # candidates within the margin -- several rounds of chains)
    scale = np.where(np.arange(1024) % 3 == 0, 100.0, np.where(np.arange(1024) % 3 == 1, 1e-3, 1.0)).astype(np.float32)
    wide = np.exp(rng.uniform(-14, 14, (1, d))).astype(np.float32)
    base32 = rng.standard_normal((512, d)).astype(np.float32)
    clustered = (base32[rng.integers(0, 512, 32768)] + 0.02 * rng.standard_normal((32768, d))).astype(np.float32)
    qcl = (clustered[rng.integers(0, 32768, 1024)] + 0.1 * rng.standard_normal((1024, d))).astype(np.float32)
    for cc, qq in (((cent32 * 30).astype(np.float32), (q32 * scale[:, None]).astype(np.float32)),
                   ((cent32 * 1e-9).astype(np.float32), (q32 * 1e-12).astype(np.float32)),
                   ((cent32 * wide).astype(np.float32), (q32 / wide).astype(np.float32)),
                   (clustered, qcl)):
        idx = make_index(faiss, cc, cb)
        for nprobe in (5, 64):
            monkeypatch.setenv("MI_TWO_STAGE", "1")
            cI1, cD1, _ = idx.coarse_and_lut(qq, nprobe, want_lut=False)
            monkeypatch.setenv("MI_TWO_STAGE", "0")
            cI0, cD0, _ = idx.coarse_and_lut(qq, nprobe, want_lut=False)
            assert np.array_equal(cI0, cI1) and np.array_equal(bits(cD0), bits(cD1)), nprobe


def test_two_stage_coarse_long_rows(faiss, oracle, monkeypatch):
    """Rows longer than 1024 floats take the second stage's `<4096>` instantiation (three workgroups per CU); three, twelve and
    sixteen 128-float sets in the quad-loaded exact chains (an odd count ends on a half round); with and without the group maxima (32 768 centroids x 1024
    queries is the slab GEMM, 8192 x 512 the ring GEMM).  Same lists and score bits as the one-stage quantiser and as the oracle's flat search."""
    rng = np.random.default_rng(2718)
    for d, M, nlist, nq in ((384, 48, 8192, 512), (1536, 96, 8192, 512), (2048, 64, 32768, 1024)):   # (the last: 512 slab tiles, group maxima)
        cb = (0.3 * rng.standard_normal((M, 256, d // M))).astype(np.float32)
        base = rng.standard_normal((256, d)).astype(np.float32)
        cent = (base[rng.integers(0, 256, nlist)] + 0.05 * rng.standard_normal((nlist, d))).astype(np.float32)
        q = (cent[rng.integers(0, nlist, nq)] + 0.2 * rng.standard_normal((nq, d))).astype(np.float32)
        idx = make_index(faiss, cent, cb)
        for gm in ("1", "0"):
            monkeypatch.setenv("MI_REFINE_GMAX", gm)
            for nprobe in (1, 24, 100):
                monkeypatch.setenv("MI_TWO_STAGE", "1")
                cI1, cD1, _ = idx.coarse_and_lut(q, nprobe, want_lut=False)
                monkeypatch.setenv("MI_TWO_STAGE", "0")
                cI0, cD0, _ = idx.coarse_and_lut(q, nprobe, want_lut=False)
                assert np.array_equal(cI0, cI1) and np.array_equal(bits(cD0), bits(cD1)), (d, gm, nprobe)
        monkeypatch.delenv("MI_REFINE_GMAX")
        De, Ie = oracle.flat_ip(q[:32], cent, 24)
        monkeypatch.setenv("MI_TWO_STAGE", "1")
        cI1, cD1, _ = idx.coarse_and_lut(q[:32].repeat(16, axis=0), 24, want_lut=False)
        assert np.array_equal(cI1[::16], Ie) and np.array_equal(bits(cD1[::16]), bits(De)), d
    monkeypatch.delenv("MI_TWO_STAGE")


def test_lut_matches_oracle(faiss, oracle):
    cent, cb, x, q = random_problem(3, 1024, 64, 8, 64, 9)
    idx = make_index(faiss, cent, cb)
    _, _, lut = idx.coarse_and_lut(q, 1)
    for i in range(q.shape[0]):
        assert np.array_equal(bits(lut[i]), bits(oracle.lut(q[i], cb)))


def test_gemm_tile_configs_bit_exact(faiss, oracle):
    """every ip_gemm tile configuration (chosen by nq) gives the oracle's fmaf chain"""
    rng = np.random.default_rng(11)
    d, nb = 1024, 777
    base = rng.standard_normal((nb, d)).astype(np.float32)
    flat = faiss.IndexFlatIP(d)
    flat.add(base)
    for nq in (1, 16, 17, 128, 129, 512, 513, 700):
        q = rng.standard_normal((nq, d)).astype(np.float32)
        D, I = flat.search(q, 5)
        De, Ie = oracle.flat_ip(q, base, 5)
        assert np.array_equal(I, Ie), nq
        assert np.array_equal(bits(D), bits(De)), nq


def test_gemm_awkward_shapes_bit_exact(faiss, oracle):
    """The f32 score GEMM over awkward shapes -- every tile configuration (1..600 query
    rows), partial and single K chunks (d = 4 .. 260, 1024), ragged and tiny bases --
    against the oracle's flat search, bit for bit."""
    rng = np.random.default_rng(3)
    for d in (4, 8, 36, 60, 64, 68, 100, 128, 132, 192, 260, 1024):
        for nb in (1, 15, 64, 777):
            base = rng.standard_normal((nb, d)).astype(np.float32)
            ix = faiss.IndexFlatIP(d)
            ix.add(base)
            for na in (1, 5, 16, 17, 64, 100, 128, 129, 300, 600):
                q = rng.standard_normal((na, d)).astype(np.float32)
                k = min(7, nb)
                D, I = ix.search(q, k)
                De, Ie = oracle.flat_ip(q, base, k)
                assert np.array_equal(I, Ie), (d, nb, na)
                assert np.array_equal(bits(D), bits(De)), (d, nb, na)


@pytest.mark.parametrize("multipass", [False, True])
def test_large_k(faiss, oracle, monkeypatch, multipass):
    """k > 64: one scan that stores every (score, id) + select_pairs_kernel (default), or one
    extraction pass per 64 results (MI_NO_ALLSCORES=1); both equal the oracle bit for bit."""
    if multipass:
        monkeypatch.setenv("MI_NO_ALLSCORES", "1")
    cent, cb, x, q = random_problem(21, 64, 8, 8, 4000, 12)
    idx = make_index(faiss, cent, cb)
    idx.add(x)
    ln, codes = oracle.encode(x, cent, cb)
    off, lc, li = oracle.build_lists(ln, codes, np.arange(len(x)), 8)
    for k, nprobe in ((65, 8), (200, 3), (1000, 8), (1024, 1), (2000, 8), (4096, 8)):
        idx.nprobe = nprobe
        D, I = idx.search(q, k)
        De, Ie = oracle.search(q, cent, cb, off, lc, li, nprobe, k)
        assert np.array_equal(I, Ie), k
        assert np.array_equal(bits(D), bits(De)), k
    # rows longer than one tile (> 4096 pairs per query), ragged lists, more lists than probes
    cent, cb, x, q = random_problem(22, 32, 4, 40, 30000, 9)
    idx = make_index(faiss, cent, cb)
    ids = np.random.default_rng(5).permutation(1 << 20)[:len(x)].astype(np.int64)
    idx.add_with_ids(x, ids)
    ln, codes = oracle.encode(x, cent, cb)
    off, lc, li = oracle.build_lists(ln, codes, ids, 40)
    # 4096 < k <= 8192: the 8192-slot instantiation of select_pairs_kernel (the refine stage's candidate lists at the
    # whole-index recall >= 0.95 point); k = 8192 of 30 000 at nprobe 40, and k larger than what 17 probes hold
    big = () if multipass else ((4097, 40), (6000, 40), (8192, 40), (8000, 17))
    for k, nprobe in ((100, 17), (640, 40), (70, 1)) + big:
        idx.nprobe = nprobe
        D, I = idx.search(q, k)
        De, Ie = oracle.search(q, cent, cb, off, lc, li, nprobe, k)
        assert np.array_equal(I, Ie), (k, nprobe)
        assert np.array_equal(bits(D), bits(De)), (k, nprobe)
    with pytest.raises(RuntimeError, match="8192"):
        idx.search(q, 8193)
    # masses of tied scores: 7000 copies of one vector (one list, identical codes) next to
    # 500 others, first with distinct ids (the cut falls inside the tie: lowest ids win),
    # then with every id used twice
    rng = np.random.default_rng(6)
    cent, cb, x, q = random_problem(23, 32, 4, 4, 500, 6)
    x = np.concatenate([np.repeat(x[:1], 7000, axis=0), x])
    q[0] = x[0]
    id_sets = [rng.permutation(1 << 16)[:len(x)].astype(np.int64)]
    if not multipass:   # extraction passes continue "strictly after (score, id)": identical pairs would be skipped
        id_sets.append((np.arange(len(x)) // 2).astype(np.int64))
    for ids in id_sets:
        idx = make_index(faiss, cent, cb)
        idx.add_with_ids(x, ids)
        ln, codes = oracle.encode(x, cent, cb)
        off, lc, li = oracle.build_lists(ln, codes, ids, 4)
        for k in (65, 300, 1024) + (() if multipass else (5000, 7100, 8000)):   # 7100: the cut falls inside the 7000-way tie
            idx.nprobe = 4
            D, I = idx.search(q, k)
            De, Ie = oracle.search(q, cent, cb, off, lc, li, 4, k)
            assert np.array_equal(I, Ie), k
            assert np.array_equal(bits(D), bits(De)), k
    # flat index / coarse quantiser with K > 64
    cent, cb, x, q = random_problem(21, 64, 8, 8, 4000, 12)
    flat = faiss.IndexFlatIP(64)
    flat.add(x)
    D, I = flat.search(q, 130)
    De, Ie = oracle.flat_ip(q, x, 130)
    assert np.array_equal(I, Ie) and np.array_equal(bits(D), bits(De))
    idx = make_index(faiss, cent, cb)
    cI, cD, _ = idx.coarse_and_lut(q, 8, want_lut=False)
    assert cI.shape == (12, 8)


@pytest.mark.parametrize("big_from", [None, "257", "100000"])
def test_select_paths(faiss, oracle, monkeypatch, big_from):
    """flat search exercises the selection kernels: select_kernel's threshold paths
    (K <= 64; K <= 256 with MI_SELECT_BIG_FROM=257), select_big_kernel (threshold + bitonic
    sort, 64 < K <= 4096), the insertion path (K > 4096, or everything above 256 with
    MI_SELECT_BIG_FROM=100000) and the tie-overflow fallbacks (more equal scores than
    survivor slots)."""
    if big_from:
        monkeypatch.setenv("MI_SELECT_BIG_FROM", big_from)
    rng = np.random.default_rng(77)
    d = 32
    base = rng.standard_normal((5000, d)).astype(np.float32)
    q = rng.standard_normal((9, d)).astype(np.float32)
    flat = faiss.IndexFlatIP(d)
    flat.add(base)
    for k in (1, 10, 64, 65, 200, 256, 257, 300, 1000, 1024, 2500, 4096):
        D, I = flat.search(q, k)
        De, Ie = oracle.flat_ip(q, base, k)
        assert np.array_equal(I, Ie), k
        assert np.array_equal(bits(D), bits(De)), k
    # 6000 identical vectors + a few distinct ones: every score ties
    dup = np.repeat(base[:1], 6000, axis=0)
    dup = np.concatenate([dup, base[1:40]])
    flat2 = faiss.IndexFlatIP(d)
    flat2.add(dup)
    for k in (5, 100, 300, 1024):
        D, I = flat2.search(q, k)
        De, Ie = oracle.flat_ip(q, dup, k)
        assert np.array_equal(I, Ie), k
        assert np.array_equal(bits(D), bits(De)), k
    # the coarse quantiser takes K = nprobe up to nlist: 5000 centroids (two row tiles)
    cent = base
    cb = rng.standard_normal((4, 256, d // 4)).astype(np.float32)
    idx = make_index(faiss, cent, cb)
    for nprobe in (65, 512, 2048, 4096, 4097, 5000):
        cI, cD, _ = idx.coarse_and_lut(q, nprobe, want_lut=False)
        De, Ie = oracle.flat_ip(q, cent, nprobe)
        assert np.array_equal(cI, Ie), nprobe
        assert np.array_equal(bits(cD), bits(De)), nprobe
    # reserve(): capacity hint only -- chunked adds after it give the same store
    flat4 = faiss.IndexFlatIP(d)
    flat4.reserve(len(base))
    for c0 in range(0, len(base), 1300):
        flat4.add(base[c0:c0 + 1300])
    flat4.reserve(10)                      # smaller than what is there: a no-op
    assert flat4.ntotal == len(base)
    D4, I4 = flat4.search(q, 10)
    D0, I0 = flat.search(q, 10)
    assert np.array_equal(I4, I0) and np.array_equal(bits(D4), bits(D0))
    # fewer rows than k, NaN-free tiny inputs
    flat3 = faiss.IndexFlatIP(d)
    flat3.add(base[:7])
    D, I = flat3.search(q, 12)
    De, Ie = oracle.flat_ip(q, base[:7], 12)
    assert np.array_equal(I, Ie) and np.array_equal(bits(D), bits(De))


def test_edge_cases(faiss, oracle):
    cent, cb, x, q = random_problem(4, 64, 8, 16, 40, 6)
    idx = make_index(faiss, cent, cb)
    # empty index: all -1
    D, I = idx.search(q, 5)
    assert (I == -1).all() and (D == -np.finfo(np.float32).max).all()
    ids = np.arange(40, dtype=np.int64)[::-1] * 3 + 5     # descending ids
    idx.add_with_ids(x, ids)
    # k > ntotal -> tail padded with -1 / -FLT_MAX; nprobe > nlist clamps
    idx.nprobe = 1000
    D, I = idx.search(q, 50)
    assert (I[:, 40:] == -1).all() and (D[:, 40:] == -np.finfo(np.float32).max).all()
    assert all(sorted(r[:40].tolist()) == sorted(ids.tolist()) for r in I)
    ln, codes = oracle.encode(x, cent, cb)
    off, lc, li = oracle.build_lists(ln, codes, ids, 16)
    De, Ie = oracle.search(q, cent, cb, off, lc, li, 16, 50)
    assert np.array_equal(I, Ie) and np.array_equal(bits(D), bits(De))
    # exact duplicates -> ties broken by ascending id, regardless of storage order
    idx.add_with_ids(x[:10], np.arange(10, dtype=np.int64) + 1)
    ids2 = np.concatenate([ids, np.arange(10) + 1])
    x2 = np.concatenate([x, x[:10]])
    ln, codes = oracle.encode(x2, cent, cb)
    off, lc, li = oracle.build_lists(ln, codes, ids2, 16)
    D, I = idx.search(x[:10], 4)
    De, Ie = oracle.search(x[:10], cent, cb, off, lc, li, 16, 4)
    assert np.array_equal(I, Ie) and np.array_equal(bits(D), bits(De))
    # reset
    idx.reset()
    assert idx.ntotal == 0
    D, I = idx.search(q, 3)
    assert (I == -1).all()
    # argument errors surface as Python exceptions
    with pytest.raises(AssertionError):
        idx.search(q[:, :32], 3)
    with pytest.raises(RuntimeError):
        idx.search(q, 9000)                      # k <= 8192
    untrained = faiss.IndexIVFPQ(64, 16, 8, 8, faiss.METRIC_INNER_PRODUCT)
    with pytest.raises(RuntimeError, match="not trained"):
        untrained.add(x)
    with pytest.raises(RuntimeError, match="metric"):
        faiss.IndexIVFPQ(64, 16, 8, 8, 5)           # neither METRIC_INNER_PRODUCT nor METRIC_L2 (L2: tests/test_l2_gpu.py)


def test_torch_device_path_equals_host_path(faiss):
    import torch
    cent, cb, x, q = random_problem(8, 128, 16, 32, 8000, 70)
    idx = make_index(faiss, cent, cb)
    idx.add(torch.from_numpy(x).cuda())        # device-pointer add
    idx.nprobe = 6
    D, I = idx.search(q, 10)
    Dt, It = idx.search(torch.from_numpy(q).cuda(), 10)
    torch.cuda.synchronize()
    assert np.array_equal(I, It.cpu().numpy()) and np.array_equal(bits(D), bits(Dt.cpu().numpy()))
    # determinism: same call twice, bit-identical
    Dt2, It2 = idx.search(torch.from_numpy(q).cuda(), 10)
    assert torch.equal(It, It2) and torch.equal(Dt, Dt2)


def test_merge_topk_and_shard_invariance(faiss, oracle):
    cent, cb, x, q = random_problem(9, 64, 8, 16, 6000, 40)
    n = len(x)
    whole = make_index(faiss, cent, cb)
    whole.add(x)
    whole.nprobe = 5
    D, I = whole.search(q, 10)
    for nshard in (2, 3, 8):
        Dp, Ip = [], []
        for s in range(nshard):                       # shard s holds rows i = s (mod nshard)
            sh = make_index(faiss, cent, cb)
            sh.add_with_ids(x[s::nshard], np.arange(n, dtype=np.int64)[s::nshard])
            sh.nprobe = 5
            d_, i_ = sh.search(q, 10)
            Dp.append(d_), Ip.append(i_)
        Dp, Ip = np.stack(Dp), np.stack(Ip)
        Dm, Im = faiss.merge_topk(Dp, Ip)
        assert np.array_equal(Im, I) and np.array_equal(bits(Dm), bits(D))
        Do, Io = oracle.merge(Dp, Ip)
        assert np.array_equal(Im, Io) and np.array_equal(bits(Dm), bits(Do))
        perm = np.random.default_rng(nshard).permutation(nshard)
        Dm2, Im2 = faiss.merge_topk(Dp[perm], Ip[perm])
        assert np.array_equal(Im2, Im)


def test_coarse_slices_and_search_preassigned(faiss, oracle):
    """the coarse quantiser split by centroid range (what each GPU of a node
    computes) + merge + search_preassigned reproduces the full search bit for bit"""
    import torch
    cent, cb, x, q = random_problem(12, 128, 16, 40, 9000, 37)
    idx = make_index(faiss, cent, cb)
    idx.add(x)
    nprobe, k = 7, 10
    idx.nprobe = nprobe
    D, I = idx.search(q, k)
    qd = torch.from_numpy(q).cuda()
    for nparts in (1, 3, 8):
        per = (40 + nparts - 1) // nparts
        cIs, cDs = [], []
        for r in range(nparts):
            lo, hi = min(r * per, 40), min((r + 1) * per, 40)
            if hi <= lo:
                continue
            cI, cD = idx.coarse_slice(qd, nprobe, lo, hi)
            cIs.append(cI.to(torch.int64)), cDs.append(cD)
        mD, mI = faiss.merge_topk(torch.stack(cDs), torch.stack(cIs))
        cIe, cDe, _ = idx.coarse_and_lut(q, nprobe, want_lut=False)
        assert np.array_equal(mI.cpu().numpy(), cIe) and np.array_equal(bits(mD.cpu().numpy()), bits(cDe))
        Dp, Ip = idx.search_preassigned(qd, k, mI.to(torch.int32), mD)
        assert np.array_equal(Ip.cpu().numpy(), I) and np.array_equal(bits(Dp.cpu().numpy()), bits(D))
    # oracle's search_preassigned agrees too (incl. -1 entries in the assignment)
    ln, codes = oracle.encode(x, cent, cb)
    off, lc, li = oracle.build_lists(ln, codes, np.arange(len(x)), 40)
    cI2 = cIe.copy()
    cI2[:, -2:] = -1
    De, Ie = oracle.search_preassigned(q, cb, off, lc, li, cI2, cDe, k)
    Dp, Ip = idx.search_preassigned(qd, k, torch.from_numpy(cI2).cuda(), torch.from_numpy(cDe).cuda())
    assert np.array_equal(Ip.cpu().numpy(), Ie) and np.array_equal(bits(Dp.cpu().numpy()), bits(De))
    # k > 64 with a caller's assignment that names the longest list in every probe slot
    # (rows longer than the nprobe longest distinct lists: the all-pairs buffers must hold them)
    longest = int(np.argmax(np.diff(off)))
    cI3 = np.full_like(cIe, longest)
    cI3[:, 0] = cIe[:, 0]
    De, Ie = oracle.search_preassigned(q, cb, off, lc, li, cI3, cDe, 200)
    Dp, Ip = idx.search_preassigned(qd, 200, torch.from_numpy(cI3).cuda(), torch.from_numpy(cDe).cuda())
    assert np.array_equal(Ip.cpu().numpy(), Ie) and np.array_equal(bits(Dp.cpu().numpy()), bits(De))


@pytest.mark.parametrize("kind", ["faiss", "faiss-ondisk", "npz"])
def test_write_read_roundtrip(faiss, tmp_path, kind):
    cent, cb, x, q = random_problem(10, 64, 8, 16, 2000, 20)
    idx = make_index(faiss, cent, cb)
    idx.add_with_ids(x, np.arange(2000, dtype=np.int64) + 77)
    idx.nprobe = 4
    D, I = idx.search(q, 10)
    f = str(tmp_path / ("index.npz" if kind == "npz" else "index.faiss"))
    faiss.write_index(idx, f, ondisk_data=str(tmp_path / "ondisk.ivfdata") if kind == "faiss-ondisk" else None)
    idx2 = faiss.read_index(f)
    assert idx2.ntotal == 2000 and idx2.nprobe == 4 and idx2.is_trained
    D2, I2 = idx2.search(q, 10)
    assert np.array_equal(I, I2) and np.array_equal(bits(D), bits(D2))


def test_read_index_of_a_pretransform_file(faiss, tmp_path):
    """[PRIOR layout] read_index of an "IxPT" file (an OPQ-style rotation in front of the IVF-PQ index, what
    index_factory(d, "OPQ..,IVF..,PQ..") writes): the rotation is applied before the search, the IwPQ record is loaded from
    its offset -- results equal the bare index searched with the rotated queries.  (reference Makefile:12-13.)"""
    import struct
    import torch
    d, M, nlist = 64, 8, 16
    cent, cb, x, q = random_problem(31, d, M, nlist, 3000, 12)
    rng = np.random.default_rng(2)
    R = np.linalg.qr(rng.standard_normal((d, d)))[0].astype(np.float32)
    idx = make_index(faiss, cent, cb)
    idx.add(x @ R.T)
    idx.nprobe = 5
    sub, f = str(tmp_path / "sub.faiss"), str(tmp_path / "opq.faiss")
    faiss.write_index(idx, sub)
    with open(f, "wb") as fh:
        fh.write(b"IxPT" + struct.pack("<iqqqBi", d, idx.ntotal, 1 << 20, 1 << 20, 1, 0) + struct.pack("<i", 1))
        fh.write(b"LTra" + struct.pack("<B", 0) + struct.pack("<Q", R.size) + R.tobytes() + struct.pack("<Q", 0) + struct.pack("<iiB", d, d, 1))
        fh.write(open(sub, "rb").read())
    pt = faiss.read_index(f)
    assert isinstance(pt, faiss.IndexPreTransform) and pt.ntotal == idx.ntotal and pt.d == d
    pt.nprobe = 5
    D, I = pt.search(q, 10)
    De, Ie = idx.search(pt.apply(q), 10)
    assert np.array_equal(I, Ie) and np.array_equal(bits(D), bits(De))
    Dn, In = idx.search((torch.from_numpy(q).cuda() @ torch.from_numpy(R).cuda().T).cpu().numpy(), 10)
    assert (I == In).mean() > 0.99                              # (the same rotation by another GEMM: rounding ties at most)


def test_pretransform_apply_is_the_exact_gemm_and_the_file_round_trips(faiss, oracle, tmp_path):
    """IndexPreTransform.apply runs on the library's own exact-f32 GEMM (mi_ip_gemm; no BLAS call in the product package):
    every output element is the ascending-k fmaf chain the oracle's IndexFlatIP computes, + b; write_index(IndexPreTransform)
    -> read_index gives the same chain and the same search results.  (reference Makefile:12-13, 39: whatever factory string
    sidecar-search's `index train` defaults to -- an "OPQ..," prefix is an IndexPreTransform.)"""
    d_in, d, M, nlist = 96, 64, 8, 16
    cent, cb, x, q = random_problem(41, d, M, nlist, 2500, 40)
    rng = np.random.default_rng(4)
    A = rng.standard_normal((d, d_in)).astype(np.float32) / np.float32(np.sqrt(d_in))   # (a float64 scalar would promote the matrix)
    b = rng.standard_normal(d).astype(np.float32)
    R = np.linalg.qr(rng.standard_normal((d, d)))[0].astype(np.float32)
    idx = make_index(faiss, cent, cb)
    pt = faiss.IndexPreTransform([(A, b), (R, None)], idx)
    xin = rng.standard_normal((x.shape[0], d_in)).astype(np.float32)
    qin = xin[:40] + 0.01 * rng.standard_normal((40, d_in)).astype(np.float32)
    # (1) the chain, element by element, against the oracle's dot product (flat_ip scores of the rows of A, all d of them)
    y = pt.apply(qin)
    D1, I1 = oracle.flat_ip(qin, A, d)
    step1 = np.empty((40, d), np.float32)
    np.put_along_axis(step1, I1, D1, axis=1)
    step1 = step1 + b[None, :]                                   # one f32 add per element, as add_row_bias_kernel does
    D2, I2 = oracle.flat_ip(step1, R, d)
    want = np.empty((40, d), np.float32)
    np.put_along_axis(want, I2, D2, axis=1)
    assert np.array_equal(bits(y), bits(want))
    import torch
    yt = pt.apply(torch.from_numpy(qin).cuda())                  # CUDA tensor in -> CUDA tensor out, same bits
    assert yt.is_cuda and np.array_equal(bits(yt.cpu().numpy()), bits(want))
    assert pt.apply(np.zeros((0, d_in), np.float32)).shape == (0, d)
    with pytest.raises(ValueError):
        pt.apply(np.zeros((3, d), np.float32))
    # (2) add / search through the chain, then the file
    pt.add(xin)
    pt.nprobe = 6
    D, I = pt.search(qin, 10)
    f = str(tmp_path / "opq.faiss")
    faiss.write_index(pt, f)
    assert open(f, "rb").read(4) == b"IxPT"
    back = faiss.read_index(f)
    assert isinstance(back, faiss.IndexPreTransform) and back.d == d_in and back.ntotal == pt.ntotal and len(back.chain) == 2
    assert np.array_equal(back.chain[0][0], A) and np.array_equal(back.chain[0][1], b) and back.chain[1][1] is None
    back.nprobe = 6
    Db, Ib = back.search(qin, 10)
    assert np.array_equal(I, Ib) and np.array_equal(bits(D), bits(Db))


@pytest.mark.parametrize("env", [{}, {"MI_SCAN_PRUNE_P1": "1"}, {"MI_SCAN_PRUNE_P1": "7"}, {"MI_NO_FUSED_MERGE": "1"}, {"MI_NSLICE": "3"},
                                 {"MI_SCAN_PRUNE_MODE": "2"}, {"MI_SCAN_PRUNE_MODE": "2", "MI_NSLICE": "3"},
                                 {"MI_SCAN_PRUNE_MODE": "2", "MI_NO_FUSED_MERGE": "1"}, {"MI_SCAN_PRUNE_MODE": "1"}])
def test_exact_list_pruning_is_bit_identical(faiss, oracle, monkeypatch, env):
    """Exact list pruning (mi_ivfpq.h: mi_index_prune_stats), both forms -- two scan launches (prune_tables_kernel: the best lists
    of every query, then only the lists whose score bound reaches the k-th score found) and the early stop inside the scan
    kernel (MI_SCAN_PRUNE_MODE=2 here: the default takes it from 512 queries up) -- must give the exhaustive scan's and the
    oracle's bits, whatever P1, the merge route, the slicing or the probe-table path (nprobe <= 64, <= 256, > 256);
    by_residual = False, k > 64 and METRIC_L2 keep the exhaustive scan.  The threshold that makes a batch 'large enough' for the
    two launches is lowered to 0 here (the default only lets query batches through)."""
    d, M, nlist, n, nq = 64, 8, 600, 24000, 37
    cent, cb, x, q = random_problem(99, d, M, nlist, n, nq)
    x[5000:5400] = x[5000]                                       # 400 identical vectors: ties at the cut, ids decide
    idx = make_index(faiss, cent, cb)
    idx.add(x)
    flat = make_index(faiss, cent, cb, by_residual=False)
    flat.add(x)
    ln, codes = oracle.encode(x, cent, cb, True)
    off, lc, li = oracle.build_lists(ln, codes, np.arange(n), nlist)
    for nprobe, k in ((8, 1), (16, 10), (64, 10), (64, 64), (65, 10), (256, 10), (257, 33), (600, 64), (64, 100)):
        idx.nprobe = flat.nprobe = nprobe
        monkeypatch.setenv("MI_SCAN_PRUNE", "0")
        D0, I0 = idx.search(q, k)
        F0 = flat.search(q, k)
        monkeypatch.setenv("MI_SCAN_PRUNE", "1")
        monkeypatch.setenv("MI_SCAN_PRUNE_MIN_GROUPS", "0")
        for key, val in env.items():
            monkeypatch.setenv(key, val)
        idx.prune_stats(reset=True)
        D1, I1 = idx.search(q, k)
        st = idx.prune_stats()
        F1 = flat.search(q, k)
        for key in list(env) + ["MI_SCAN_PRUNE_MIN_GROUPS"]:
            monkeypatch.delenv(key)
        assert np.array_equal(I0, I1), (env, nprobe, k, np.argwhere(I0 != I1)[:5])
        assert np.array_equal(bits(D0), bits(D1)), (env, nprobe, k)
        assert np.array_equal(F0[1], F1[1]) and np.array_equal(bits(F0[0]), bits(F1[0]))
        De, Ie = oracle.search(q, cent, cb, off, lc, li, nprobe, k, True)
        assert np.array_equal(I1, Ie) and np.array_equal(bits(D1), bits(De)), (env, nprobe, k)
        mode = env.get("MI_SCAN_PRUNE_MODE", os.environ.get("MI_SCAN_PRUNE_MODE", "0"))
        if k <= 64 and (mode != "2" or nprobe <= 64):   # (the early stop: probe tables in registers only)
            # this corpus sits tightly around its centroids: nearly every list beyond the first few is provably out
            assert st["queries"] == nq and 0 < st["groups_all_probes"], st
            assert st["groups_second_phase"] <= st["groups_all_probes"], st
            # (two launches; k = 64 > a list's 40 codes: the 64th score comes from another cluster and bounds little.  The early stop
            # gives little HERE: a list is one group, a wave has a threshold after three and two more in flight)
            if k <= 10 and nprobe >= 16 and mode != "2":
                assert st["groups_second_phase"] < 0.5 * st["groups_all_probes"], (nprobe, k, st)
        else:
            assert st["queries"] == 0, st                        # k > 64: the all-scores path, not pruned
    assert flat.prune_stats()["queries"] == 0                    # by_residual = False: dis0 = 0 for every list, nothing to bound
    # the default dispatch of a large batch (>= 512 queries, nprobe <= 64): the early stop
    if not env:
        rng = np.random.default_rng(5)
        qb = (x[rng.integers(0, n, 600)] + 0.05 * rng.standard_normal((600, d))).astype(np.float32)
        idx.nprobe = 64
        monkeypatch.setenv("MI_SCAN_PRUNE", "0")
        Db0, Ib0 = idx.search(qb, 10)
        monkeypatch.delenv("MI_SCAN_PRUNE")
        idx.prune_stats(reset=True)
        Db1, Ib1 = idx.search(qb, 10)
        st = idx.prune_stats()
        assert np.array_equal(Ib0, Ib1) and np.array_equal(bits(Db0), bits(Db1))
        if os.environ.get("MI_SCAN_PRUNE_MODE", "0") == "0":     # (a run of the suite with one form forced: the dispatch differs)
            assert st["queries"] == 600 and st["groups_second_phase"] < st["groups_all_probes"], st
    # caller-assigned lists (search_preassigned): in coarse order the early stop runs, shuffled the per-row check turns it off --
    # the same k best either way (and through the two launches when the early stop is not forced)
    import torch
    idx.nprobe = 33
    Dn, In = idx.search(q, 10)
    cI, cD, _ = idx.coarse_and_lut(q, 33, want_lut=False)
    perm = np.random.default_rng(3).permutation(33)
    qd = torch.from_numpy(q).cuda()
    for force in ("2", None):
        if force:
            monkeypatch.setenv("MI_SCAN_PRUNE_MODE", force)
        monkeypatch.setenv("MI_SCAN_PRUNE_MIN_GROUPS", "0")
        for ci, cd in ((cI, cD), (cI[:, perm], cD[:, perm])):
            idx.prune_stats(reset=True)
            Dp, Ip = idx.search_preassigned(qd, 10, torch.from_numpy(np.ascontiguousarray(ci)).cuda(), torch.from_numpy(np.ascontiguousarray(cd)).cuda())
            assert np.array_equal(Ip.cpu().numpy(), In) and np.array_equal(bits(Dp.cpu().numpy()), bits(Dn)), force
            assert idx.prune_stats()["queries"] == nq
        if force:
            monkeypatch.delenv("MI_SCAN_PRUNE_MODE")
        monkeypatch.delenv("MI_SCAN_PRUNE_MIN_GROUPS")
    # a query whose best lists hold fewer than k codes: no threshold, nothing pruned, same result
    tiny = make_index(faiss, cent, cb)
    tiny.add(x[:300])
    tiny.nprobe = 64
    monkeypatch.setenv("MI_SCAN_PRUNE", "0")
    Dt0, It0 = tiny.search(q, 64)
    monkeypatch.setenv("MI_SCAN_PRUNE", "1")
    monkeypatch.setenv("MI_SCAN_PRUNE_MIN_GROUPS", "0")
    Dt1, It1 = tiny.search(q, 64)
    assert np.array_equal(It0, It1) and np.array_equal(bits(Dt0), bits(Dt1)) and (It1 == -1).any()


@pytest.mark.parametrize("metric", ["ip", "l2"])
def test_seal_frees_the_log_and_changes_nothing(faiss, tmp_path, metric):
    """IndexIVFPQ.seal() (mi_index_seal; bench.py calls it once the 207 M-vector index is filled: 16.6 GB of HBM back): searches
    before and after give the same bits; export, write_index / read_index and further adds after a seal -- which rebuild the
    log from the scan image, in list order -- give what an index that was never sealed gives; seal twice, seal an empty index."""
    import torch
    d, M, nlist = 64, 16, 40
    cent, cb, x, q = random_problem(61, d, M, nlist, 9000, 32)
    mt = faiss.METRIC_INNER_PRODUCT if metric == "ip" else faiss.METRIC_L2

    def fresh():
        i = faiss.IndexIVFPQ(d, nlist, M, 8, mt)
        i.set_centroids(cent)
        i.set_codebook(cb)
        i.nprobe = 9
        return i
    ids = np.arange(9000, dtype=np.int64) * 3 + 7
    a, b = fresh(), fresh()
    a.seal()                                                     # empty: a no-op that must not break the first add
    for idx in (a, b):
        idx.add_with_ids(x[:6000], ids[:6000])
    D0, I0 = b.search(q, 10)
    free0 = torch.cuda.mem_get_info()[0]
    a.search(q, 10)
    a.seal()
    a.seal()
    assert torch.cuda.mem_get_info()[0] >= free0                 # (the log is gone; small indexes: allocator granularity hides the size)
    D1, I1 = a.search(q, 10)
    assert np.array_equal(I0, I1) and np.array_equal(bits(D0), bits(D1))
    ca, ia = a.export_lists()                                    # rebuilds the log from the image
    cb_, ib = b.export_lists()
    assert np.array_equal(ca, cb_) and np.array_equal(ia, ib) and np.array_equal(a.list_sizes(), b.list_sizes())
    a.seal()
    for idx in (a, b):
        idx.add_with_ids(x[6000:], ids[6000:])                   # add after a seal: the new entries land behind the old ones of every list
    D2, I2 = a.search(q, 10)
    D3, I3 = b.search(q, 10)
    assert np.array_equal(I2, I3) and np.array_equal(bits(D2), bits(D3))
    ca, ia = a.export_lists()
    cb_, ib = b.export_lists()
    assert np.array_equal(ca, cb_) and np.array_equal(ia, ib)
    a.seal()
    f = str(tmp_path / "sealed.faiss")
    faiss.write_index(a, f)                                      # save of a sealed index
    back = faiss.read_index(f)
    back.nprobe = 9
    D4, I4 = back.search(q, 10)
    assert back.ntotal == 9000 and np.array_equal(I4, I3) and np.array_equal(bits(D4), bits(D3))
    a.reset()
    assert a.ntotal == 0
    a.add_with_ids(x[:100], ids[:100])
    assert a.ntotal == 100 and a.search(q, 3)[1].max() >= 0


def test_read_index_with_an_hnsw_coarse_quantiser(faiss, tmp_path):
    """[PRIOR layout] "IVF<n>_HNSW32,PQ<M>": index.faiss holds an IndexHNSWFlat (IHNf) in front of the lists.  mi_index_load
    takes its flat storage as the centroid table and skips the graph: the loaded index equals the flat-quantiser one."""
    import struct
    d, M, nlist = 64, 8, 32
    cent, cb, x, q = random_problem(53, d, M, nlist, 4000, 16)
    idx = make_index(faiss, cent, cb)
    idx.add(x)
    idx.nprobe = 7
    D, I = idx.search(q, 10)
    flat = str(tmp_path / "flat.faiss")
    faiss.write_index(idx, flat)
    raw = open(flat, "rb").read()
    at = raw.index(b"IxFI")
    flen = 4 + 33 + 8 + nlist * d * 4
    rng = np.random.default_rng(1)
    lev = rng.integers(1, 4, nlist).astype(np.int32)
    cum = np.array([0, 64, 96, 128], np.int32)
    offsets = np.concatenate([[0], np.cumsum(cum[lev])]).astype(np.uint64)
    neigh = rng.integers(-1, nlist, int(offsets[-1])).astype(np.int32)
    graph = b""
    for v in (np.array([0.9, 0.09, 0.01]), cum, lev, offsets, neigh):
        graph += struct.pack("<Q", v.size) + v.tobytes()
    graph += struct.pack("<iiiii", 3, 2, 40, 16, 1)
    hdr = struct.pack("<iqqqBi", d, nlist, 1 << 20, 1 << 20, 1, 0)
    f = str(tmp_path / "hnsw.faiss")
    with open(f, "wb") as fh:
        fh.write(raw[:at] + b"IHNf" + hdr + graph + raw[at:at + flen] + raw[at + flen:])
    back = faiss.read_index(f)
    assert back.ntotal == idx.ntotal and back.nprobe == 7 and back.is_trained
    assert np.array_equal(back.get_centroids(), cent)
    D2, I2 = back.search(q, 10)
    assert np.array_equal(I, I2) and np.array_equal(bits(D), bits(D2))
    # the loaded index remembers the graph it dropped; writing it back says that the file changes type (IHNf -> IxFI)
    assert back.hnsw_quantizer is True and not getattr(faiss.read_index(flat), "hnsw_quantizer", True)
    with pytest.warns(UserWarning, match="IndexHNSWFlat"):
        faiss.write_index(back, str(tmp_path / "rewritten.faiss"))
    again = faiss.read_index(str(tmp_path / "rewritten.faiss"))
    assert again.hnsw_quantizer is False and np.array_equal(again.search(q, 10)[1], I)
    with open(f, "wb") as fh:                                    # HNSW over compressed storage: no exact centroid table
        fh.write(raw[:at] + b"IHNs" + hdr + graph + raw[at:])
    with pytest.raises(Exception, match="IHNs"):
        faiss.read_index(f)


def test_crosscheck_tool_runs_end_to_end_against_itself():
    """tools/crosscheck_faiss.py (first contact with the real faiss as one command) with this package standing in for the
    real library: both directions of the file format, every rank identical; without --self on a box that has no faiss it
    says so and exits 3."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "crosscheck_faiss.py")
    out = subprocess.run([sys.executable, tool, "--self", "--n", "30000", "--nlist", "64", "--d", "64", "--M", "8", "--nq", "100"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rep = json.loads(out.stdout)
    assert len(rep["cases"]) == 2 and all(c["id_match_rate"] == 1.0 and c["real"] == 0 and c["max_score_ulps"] == 0 for c in rep["cases"])
    try:
        import faiss as real                                    # noqa: F401
        has_real = not getattr(real, "__file__", "").startswith(root)
    except Exception:
        has_real = False
    if not has_real:
        out = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=600)
        assert out.returncode == 3 and json.loads(out.stdout)["faiss"] == "absent"


def test_sparse_lists_through_the_c_abi_match_the_documented_layout(faiss, tmp_path):
    """A non-empty index with at most half of its lists filled takes faiss's 'sprs' size encoding:
    a vector of 2 * non_empty words ({list, size} pairs flattened).  The C ABI writer
    (mi_index_save) must emit the bytes the independent pure-Python writer (faiss_io.dump, pinned
    to a hand-assembled file in tests/test_faiss_io.py) emits, and mi_index_load must read both."""
    import importlib
    fio = importlib.import_module("abstracts_search_amd.faiss_io")
    rng = np.random.default_rng(5)
    d, nlist, M = 64, 16, 8
    cent = rng.standard_normal((nlist, d)).astype(np.float32)
    cb = (0.3 * rng.standard_normal((M, 256, d // M))).astype(np.float32)
    idx = make_index(faiss, cent, cb)
    lists = np.repeat(np.array([2, 5, 11], np.int32), [7, 1, 4])        # 3 of 16 lists non-empty -> "sprs"
    codes = rng.integers(0, 256, (len(lists), M)).astype(np.uint8)
    ids = rng.integers(0, 1 << 40, len(lists)).astype(np.int64)
    idx.add_codes(lists, codes, ids)
    idx.nprobe = 16
    q = rng.standard_normal((5, d)).astype(np.float32)
    D, I = idx.search(q, 10)
    f, g = str(tmp_path / "abi.faiss"), str(tmp_path / "py.faiss")
    faiss.write_index(idx, f)
    raw = open(f, "rb").read()
    assert b"sprs" in raw
    sizes = np.bincount(lists, minlength=nlist).astype(np.int64)
    fio.dump(g, d=d, nlist=nlist, M=M, nbits=8, metric=fio.METRIC_INNER_PRODUCT, by_residual=True, nprobe=16,
             is_trained=True, centroids=cent, codebook=cb, sizes=sizes, codes=codes, ids=ids)
    assert raw == open(g, "rb").read()
    z = fio.parse(f)
    assert np.array_equal(z["sizes"], sizes) and np.array_equal(z["codes"], codes) and np.array_equal(z["ids"], ids)
    for path in (f, g):
        idx2 = faiss.read_index(path)
        assert idx2.ntotal == len(lists)
        D2, I2 = idx2.search(q, 10)
        assert np.array_equal(I, I2) and np.array_equal(bits(D), bits(D2))
    # a file this library wrote before round 3's fix (the count was the number of PAIRS): recognised by which reading of the
    # count adds up to ntotal, and loaded
    open(f, "wb").write(raw.replace(b"sprs" + np.uint64(6).tobytes(), b"sprs" + np.uint64(3).tobytes()))
    idx3 = faiss.read_index(f)
    D3, I3 = idx3.search(q, 10)
    assert idx3.ntotal == len(lists) and np.array_equal(I, I3) and np.array_equal(bits(D), bits(D3))
    # a count that fits neither reading: a named error
    bad = raw.replace(b"sprs" + np.uint64(6).tobytes(), b"sprs" + np.uint64(5).tobytes())
    open(f, "wb").write(bad)
    with pytest.raises(Exception):
        faiss.read_index(f)


def test_train_builds_a_usable_index(faiss):
    """train() is setup (not bit-pinned): check it yields a working quantiser
    with good recall on clustered data."""
    import abstracts_search_amd.synth as synth
    x = synth.corpus_rows(0, 20000, d=64, ncentres=64, cos=0.8)
    q = synth.queries_from(x, 50, cos=0.8)
    idx = faiss.index_factory(64, "IVF32,PQ16", faiss.METRIC_INNER_PRODUCT)
    idx.cp.niter = idx.pq.cp.niter = 8
    idx.train(x)
    assert idx.is_trained
    idx.add(x)
    idx.nprobe = 8
    D, I = idx.search(q, 10)
    flat = faiss.IndexFlatIP(64)
    flat.add(x)
    Df, If = flat.search(q, 10)
    recall = np.mean([len(set(a) & set(b)) / 10 for a, b in zip(I, If)])
    assert recall > 0.5, recall
    assert (D[:, :-1] >= D[:, 1:]).all()


def test_cfg2_full_size_properties(faiss, oracle):
    """BASELINE.json configs[1] at full size (1M x 1024, IVF4096,PQ64, batch 64):
    size-independent properties + oracle parity on the full index."""
    import torch
    import abstracts_search_amd.synth as synth
    n, d, nlist, M = 1_000_000, 1024, 4096, 64
    x = synth.corpus_cuda(n, d)
    idx = faiss.IndexIVFPQ(d, nlist, M, 8, faiss.METRIC_INNER_PRODUCT)
    idx.cp.niter = idx.pq.cp.niter = 4
    idx.train(x)
    idx.add(x)
    assert idx.ntotal == n
    sizes = np.array([idx.list_size(l) for l in range(nlist)])
    assert sizes.sum() == n                                   # every vector is in exactly one list
    q = synth.queries_cuda(x, 64)
    idx.nprobe = 16
    D, I = idx.search(q, 10)
    D2, I2 = idx.search(q, 10)
    assert torch.equal(I, I2) and torch.equal(D, D2)           # idempotent / deterministic
    Dn, In = D.cpu().numpy(), I.cpu().numpy()
    assert (Dn[:, :-1] >= Dn[:, 1:]).all()                     # sorted best first
    assert (In >= 0).all() and (In < n).all()
    assert all(len(set(r.tolist())) == 10 for r in In)         # no duplicate ids
    # the k=10 result is a prefix of the k=64 result and of the multi-pass k=100 result
    D64, I64 = idx.search(q, 64)
    D100, I100 = idx.search(q, 100)
    assert torch.equal(I64[:, :10], I) and torch.equal(I100[:, :64], I64)
    # a larger nprobe can only improve the k-th score
    idx.nprobe = 64
    Dw, _ = idx.search(q, 10)
    assert (Dw[:, -1] >= D[:, -1]).all()
    # permuting the queries permutes the rows
    perm = torch.randperm(64, device="cuda")
    idx.nprobe = 16
    Dp, Ip = idx.search(q[perm].contiguous(), 10)
    assert torch.equal(Ip, I[perm]) and torch.equal(Dp, D[perm])
    # oracle parity on the full index (codes pulled back from the library)
    cent, cb = idx.get_centroids(), idx.get_codebook()
    off = np.zeros(nlist + 1, np.int64)
    np.cumsum(sizes, out=off[1:])
    codes = np.empty((n, M), np.uint8)
    ids = np.empty(n, np.int64)
    for l in range(nlist):
        if sizes[l]:
            codes[off[l]:off[l + 1]], ids[off[l]:off[l + 1]] = idx.get_list(l)
    qh = q.cpu().numpy()
    De, Ie = oracle.search(qh, cent, cb, off, codes, ids, 16, 10)
    assert np.array_equal(In, Ie)
    assert np.array_equal(bits(Dn), bits(De))
    # and the stored codes are the oracle's encoding of the same rows (sample)
    rows = np.arange(0, n, 9973)[:64]
    ln, cs = oracle.encode(x[torch.as_tensor(rows, device="cuda")].cpu().numpy(), cent, cb)
    pos = {int(i): p for p, i in enumerate(ids)}
    for r, l, c in zip(rows, ln, cs):
        p = pos[int(r)]
        assert off[l] <= p < off[l + 1] and np.array_equal(codes[p], c)


def test_fuzz_against_oracle(faiss, oracle):
    """seeded fuzz over shapes / parameters / degenerate inputs: HIP == oracle, bit for bit"""
    rng = np.random.default_rng(2026 + FUZZ_SEED)
    for trial in range(40):
        M = int(rng.choice([4, 8, 16]))
        d = M * int(rng.choice([4, 8, 16]))
        nlist = int(rng.integers(1, 40))
        n = int(rng.choice([0, 1, 7, 63, 64, 65, 500, 3000]))
        nq = int(rng.integers(1, 40))
        k = int(rng.choice([1, 2, 10, 33, 64, 65, 100]))
        nprobe = int(rng.integers(1, nlist + 3))
        by_residual = bool(rng.integers(0, 2))
        cent = rng.standard_normal((nlist, d)).astype(np.float32)
        cb = (0.3 * rng.standard_normal((M, 256, d // M))).astype(np.float32)
        x = (cent[rng.integers(0, nlist, n)] + 0.3 * rng.standard_normal((n, d))).astype(np.float32)
        if n >= 64 and trial % 3 == 0:
            x[n // 2:] = x[: n - n // 2]                    # many exact duplicates -> ties
        if trial % 5 == 0:
            x = np.round(x, 1)                              # coarse values -> score ties across lists
            cent = np.round(cent, 1)
        q = rng.standard_normal((nq, d)).astype(np.float32)
        ids = rng.permutation(n).astype(np.int64) * 3 + 1
        idx = make_index(faiss, cent, cb, by_residual)
        if n:
            idx.add_with_ids(x, ids)
        idx.nprobe = nprobe
        D, I = idx.search(q, k)
        ln, codes = oracle.encode(x, cent, cb, by_residual) if n else (np.zeros(0, np.int32), np.zeros((0, M), np.uint8))
        off, lc, li = oracle.build_lists(ln, codes, ids, nlist)
        De, Ie = oracle.search(q, cent, cb, off, lc, li, nprobe, k, by_residual)
        ctx = dict(trial=trial, d=d, M=M, nlist=nlist, n=n, nq=nq, k=k, nprobe=nprobe, res=by_residual)
        assert np.array_equal(I, Ie), ctx
        assert np.array_equal(bits(D), bits(De)), ctx


def test_fuzz_refine_stores_against_oracle(faiss, oracle):
    """seeded fuzz over the refine stage: store (f32 / half / SQ8), shapes, candidate-list lengths from a few dozen to
    beyond 4096, duplicates (ties at the cut of the candidate list and in the final ranking), host and device queries --
    the device path goes through the unordered candidate sets, the host path through the sorted lists; both == oracle."""
    import torch
    rng = np.random.default_rng(909 + FUZZ_SEED)
    for trial in range(18):
        M = int(rng.choice([4, 8, 16]))
        d = M * int(rng.choice([8, 16]))
        nlist = int(rng.integers(4, 24))
        n = int(rng.choice([300, 2000, 9000]))
        nq = int(rng.integers(1, 24))
        k = int(rng.choice([1, 5, 10]))
        kf = int(rng.choice([3, 16, 100, 450, 700]))
        kf = min(kf, 8192 // k)
        nprobe = int(rng.integers(1, nlist + 1))
        store = ("flat", "sqfp16", "sq8")[trial % 3]
        cent = rng.standard_normal((nlist, d)).astype(np.float32)
        cb = (0.3 * rng.standard_normal((M, 256, d // M))).astype(np.float32)
        x = (cent[rng.integers(0, nlist, n)] + 0.3 * rng.standard_normal((n, d))).astype(np.float32)
        if trial % 2 == 0:
            x[n // 2:] = x[: n - n // 2]                    # every vector twice: ties everywhere
        q = (x[rng.integers(0, n, nq)] + 0.1 * rng.standard_normal((nq, d))).astype(np.float32)
        base = make_index(faiss, cent, cb)
        if store == "flat":
            idx = faiss.IndexRefineFlat(base)
        else:
            qt = faiss.ScalarQuantizer.QT_fp16 if store == "sqfp16" else faiss.ScalarQuantizer.QT_8bit
            r = faiss.IndexScalarQuantizer(d, qt, faiss.METRIC_INNER_PRODUCT)
            r.train(x[: n // 2 + 1])
            idx = faiss.IndexRefine(base, r)
        idx.add(x)
        idx.nprobe, idx.k_factor = nprobe, kf
        ln, codes = oracle.encode(x, cent, cb)
        off, lc, li = oracle.build_lists(ln, codes, np.arange(n), nlist)
        _, cand = oracle.search(q, cent, cb, off, lc, li, nprobe, k * kf)
        if store == "flat":
            De, Ie = oracle.rerank(q, x, cand, k)
        elif store == "sqfp16":
            De, Ie = oracle.rerank(q, x.astype(np.float16).astype(np.float32), cand, k)
        else:
            tr = oracle.sq8_train(x[: n // 2 + 1])
            De, Ie = oracle.rerank_sq8(q, oracle.sq8_encode(x, tr), tr, cand, k)
        ctx = dict(trial=trial, store=store, d=d, M=M, nlist=nlist, n=n, nq=nq, k=k, kf=kf, nprobe=nprobe)
        Dh, Ih = idx.search(q, k)
        assert np.array_equal(Ih, Ie) and np.array_equal(bits(Dh), bits(De)), ("host", ctx)
        Dd, Id = idx.search(torch.from_numpy(q).cuda(), k)
        assert np.array_equal(Id.cpu().numpy(), Ie) and np.array_equal(bits(Dd.cpu().numpy()), bits(De)), ("device", ctx)


def test_fuzz_large_paths(faiss, oracle, monkeypatch):
    """seeded fuzz over the large-parameter paths: nprobe in the hundreds (select_big_kernel /
    the block-wide descent), k > 64 (all-pairs pass + select_pairs_kernel), the two-stage
    coarse quantiser forced on half of the trials (search and add), duplicated / rounded
    data for ties: HIP == oracle, bit for bit"""
    rng = np.random.default_rng(777 + FUZZ_SEED)
    for trial in range(24):
        d = int(rng.choice([128, 256]))
        M = int(rng.choice([d // 16, d // 8, d // 4]))
        nlist = int(rng.choice([100, 257, 640, 1500]))
        n = int(rng.choice([300, 5000, 20000]))
        nq = int(rng.integers(1, 24))
        k = int(rng.choice([10, 64, 65, 130, 700, 2000]))
        nprobe = int(rng.choice([1, 17, 64, 65, 200, 257, 600, nlist]))
        nprobe = min(nprobe, nlist)
        by_residual = bool(rng.integers(0, 2))
        two = trial % 2 == 1 and nlist % 4 == 0
        monkeypatch.setenv("MI_TWO_STAGE", "1" if two else "0")
        cent = rng.standard_normal((nlist, d)).astype(np.float32)
        if trial % 4 == 2:
            cent[nlist // 2:] = cent[: nlist - nlist // 2]      # duplicated centroids: coarse ties
        cb = (0.3 * rng.standard_normal((M, 256, d // M))).astype(np.float32)
        x = (cent[rng.integers(0, nlist, n)] + 0.3 * rng.standard_normal((n, d))).astype(np.float32)
        if trial % 3 == 0:
            x[n // 2:] = x[: n - n // 2]
        if trial % 5 == 0:
            x = np.round(x, 1)
        q = (x[rng.integers(0, n, nq)] + 0.2 * rng.standard_normal((nq, d))).astype(np.float32)
        ids = rng.permutation(n).astype(np.int64) * 5 + 2
        idx = make_index(faiss, cent, cb, by_residual)
        idx.add_with_ids(x, ids)
        idx.nprobe = nprobe
        D, I = idx.search(q, k)
        ln, codes = oracle.encode(x, cent, cb, by_residual)
        off, lc, li = oracle.build_lists(ln, codes, ids, nlist)
        for l in (0, nlist - 1):                                  # add(): the oracle's lists
            c, i = idx.get_list(l)
            assert np.array_equal(c, lc[off[l]:off[l + 1]]) and np.array_equal(i, li[off[l]:off[l + 1]])
        De, Ie = oracle.search(q, cent, cb, off, lc, li, nprobe, k, by_residual)
        ctx = dict(trial=trial, d=d, M=M, nlist=nlist, n=n, nq=nq, k=k, nprobe=nprobe, res=by_residual, two=two)
        assert np.array_equal(I, Ie), ctx
        assert np.array_equal(bits(D), bits(De)), ctx


def test_sharded_index_with_refine_and_id_map_rccl_world1(faiss, oracle):
    """ShardedIndex over an IndexRefineFlat shard on the real RCCL path (world size 1: the
    collectives and the HIP merge run, the shard is the whole index): local result ids are
    positions; id_map turns them into the caller's global numbering before the exchange."""
    import socket
    import torch
    import torch.distributed as dist
    from abstracts_search_amd.shards import ShardedIndex
    d, M, nlist, n, nq, k = 64, 8, 32, 6000, 24, 10
    cent, cb, x, q = random_problem(41, d, M, nlist, n, nq)
    idx = faiss.IndexRefineFlat(make_index(faiss, cent, cb))
    idx.add(x)
    faiss.ParameterSpace().set_index_parameters(idx, "nprobe=6,k_factor_rf=4")
    rows = torch.arange(n, dtype=torch.int64) * 3 + 1            # the "global" numbering of this shard
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        sh = ShardedIndex(idx, id_map=rows.cuda())
        qd = torch.from_numpy(q).cuda()
        D, I = sh.search(qd, k)
        Dr, Ir = sh.search_replicated(qd, k)
        torch.cuda.synchronize()
        # batches issued round-robin on two streams (bench.py's sharded loop): one buffer set per stream, the
        # collectives ordered by issue order -- every batch still gets its own answer
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        qs = [qd, qd.flip(0).contiguous(), (qd * 0.5).contiguous(), qd.roll(3, 0).contiguous()]
        outs = []
        for rep in range(3):
            for b, qb in enumerate(qs):
                with torch.cuda.stream(streams[b % 2]):
                    outs.append((b, sh.search_replicated(qb, k)))
        torch.cuda.synchronize()
        want = [sh.search_replicated(qb, k) for qb in qs]
        torch.cuda.synchronize()
        for b, (Db, Ib) in outs:
            assert torch.equal(Ib, want[b][1]) and torch.equal(Db, want[b][0]), b
        # the same with the coarse quantiser split by centroid range (two exchanges per batch: probe lists, then results),
        # bench.py's layout from 4 ranks: on two streams, every batch equals the plain search of the base index
        base = idx.base_index
        base.nprobe = 6
        shc = ShardedIndex(base, shard_coarse=True)
        outs = []
        for rep in range(3):
            for b, qb in enumerate(qs):
                with torch.cuda.stream(streams[b % 2]):
                    outs.append((b, shc.search_replicated(qb, k)))
        torch.cuda.synchronize()
        for b, (Db, Ib) in outs:
            D0, I0 = base.search(qs[b], k)
            assert torch.equal(Ib, I0) and torch.equal(Db, D0), b
    finally:
        dist.destroy_process_group()
    ln, codes = oracle.encode(x, cent, cb, True)
    off, lc, li = oracle.build_lists(ln, codes, np.arange(n), nlist)
    _, cand = oracle.search(q, cent, cb, off, lc, li, 6, k * 4, True)
    De, Ie = oracle.rerank(q, x, cand, k)
    Ie = np.where(Ie >= 0, Ie * 3 + 1, -1)
    for Dg, Ig in ((D, I), (Dr, Ir)):
        assert np.array_equal(Ig.cpu().numpy(), Ie)
        assert np.array_equal(bits(Dg.cpu().numpy()), bits(De))


def test_two_stage_coarse_full_cfg4_size(faiss, monkeypatch):
    """BASELINE configs[3]'s coarse quantiser at full size -- 1024 queries x 65536 centroids x
    1024 dimensions, unit-norm clustered data -- through the default (two-stage) dispatch and
    through the exact f32 GEMM: the same lists and the same scores, bit for bit."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(5)
    d, nlist, nq = 1024, 65536, 1024
    base = torch.randn((4096, d), generator=g, device="cuda")
    cent = base[torch.randint(0, 4096, (nlist,), generator=g, device="cuda")] + 0.6 * torch.randn((nlist, d), generator=g, device="cuda")
    cent = (cent / cent.norm(dim=1, keepdim=True)).contiguous()
    q = cent[torch.randint(0, nlist, (nq,), generator=g, device="cuda")] + 0.3 / 32 * torch.randn((nq, d), generator=g, device="cuda")
    q = (q / q.norm(dim=1, keepdim=True)).contiguous()
    cb = torch.randn((64, 256, 16), generator=g, device="cuda") * 0.05
    idx = faiss.IndexIVFPQ(d, nlist, 64, 8, faiss.METRIC_INNER_PRODUCT)
    idx.set_centroids(cent)
    idx.set_codebook(cb)
    qh = q.cpu().numpy()
    out = {}
    for mode in (None, "0"):
        if mode is None:
            monkeypatch.delenv("MI_TWO_STAGE", raising=False)
        else:
            monkeypatch.setenv("MI_TWO_STAGE", mode)
        out[mode] = [idx.coarse_and_lut(qh, nprobe, want_lut=False)[:2] for nprobe in (1, 16, 64)]
    for (cI1, cD1), (cI0, cD0) in zip(out[None], out["0"]):
        assert np.array_equal(cI1, cI0) and np.array_equal(bits(cD1), bits(cD0))
        assert (cI1 >= 0).all() and (np.diff(cD1, axis=1) <= 0).all()


def test_refine_sqfp16_matches_oracle(faiss, oracle):
    """factory "IVF..,PQ..,Refine(SQfp16)": the refine store holds IEEE halves (round to nearest
    even, no scaling); re-ranked scores are <q, (float)x16> in the f32 chain -> bit-equal to the
    oracle's rerank over the half-rounded vectors; half the bytes per candidate."""
    d, M, nlist, n, nq, k = 128, 16, 32, 9000, 50, 10
    cent, cb, x, q = random_problem(77, d, M, nlist, n, nq)
    x16 = x.astype(np.float16).astype(np.float32)
    idx = faiss.index_factory(d, f"IVF{nlist},PQ{M},Refine(SQfp16)", faiss.METRIC_INNER_PRODUCT)
    assert isinstance(idx, faiss.IndexRefine) and isinstance(idx.refine_index, faiss.IndexScalarQuantizer)
    idx.base_index.set_centroids(cent)
    idx.base_index.set_codebook(cb)
    idx.add(x[: n // 3])
    idx.add(x[n // 3:])
    assert np.array_equal(bits(idx.refine_index.reconstruct_n(0, n)), bits(x16))     # the stored halves, widened
    ln, codes = oracle.encode(x, cent, cb, True)
    off, lc, li = oracle.build_lists(ln, codes, np.arange(n), nlist)
    for nprobe, kf in ((4, 4), (8, 16)):
        faiss.ParameterSpace().set_index_parameters(idx, f"nprobe={nprobe},k_factor_rf={kf}")
        D, I = idx.search(q, k)
        _, cand = oracle.search(q, cent, cb, off, lc, li, nprobe, k * kf, True)
        De, Ie = oracle.rerank(q, x16, cand, k)
        assert np.array_equal(I, Ie) and np.array_equal(bits(D), bits(De)), (nprobe, kf)
    import torch
    Dt, It = idx.search(torch.from_numpy(q).cuda(), k)                                # device path
    assert np.array_equal(It.cpu().numpy(), Ie) and np.array_equal(bits(Dt.cpu().numpy()), bits(De))
    with pytest.raises(NotImplementedError):
        idx.refine_index.search(q, k)


def test_search_candidates_is_the_topk_set(faiss, oracle):
    """mi_index_search_candidates (the first stage of a refine search): the same kc entries as search(x, kc) -- the
    oracle's top kc under (score desc, id asc) -- as an unordered set of ids; ties at the cut are decided by id, rows
    with fewer than kc entries are padded with -1.  And IndexRefine over it equals IndexRefine over the sorted list."""
    import torch
    cent, cb, x, q = random_problem(31, 32, 4, 40, 30000, 11)
    x[20000:24000] = x[5]                                      # 4001 identical vectors: cuts that fall inside a tie
    q[0] = x[5]
    ids = np.random.default_rng(8).permutation(1 << 20)[:len(x)].astype(np.int64)
    idx = make_index(faiss, cent, cb)
    idx.add_with_ids(x, ids)
    ln, codes = oracle.encode(x, cent, cb)
    off, lc, li = oracle.build_lists(ln, codes, ids, 40)
    qd = torch.from_numpy(q).cuda()
    for kc, nprobe in ((65, 40), (640, 17), (2000, 40), (4096, 40), (5120, 40), (8192, 40), (3000, 2), (50, 8)):
        I = torch.empty((len(q), kc), dtype=torch.int64, device="cuda")
        idx.search_candidates_into(qd, kc, I, nprobe)
        _, Ie = oracle.search(q, cent, cb, off, lc, li, nprobe, kc)
        got = I.cpu().numpy()
        for r in range(len(q)):
            a, b = np.sort(got[r]), np.sort(Ie[r])
            assert np.array_equal(a, b), (kc, nprobe, r, int((a != b).sum()))
    # the refine result through the unordered candidates (device path) == through the sorted list (host path) == oracle
    ref = faiss.IndexRefineFlat(make_index(faiss, cent, cb))
    ref.add(x)
    ref.nprobe, ref.k_factor = 40, 300
    Dh, Ih = ref.search(q, 10)
    Dd, Id = ref.search(qd, 10)
    assert np.array_equal(Id.cpu().numpy(), Ih) and np.array_equal(bits(Dd.cpu().numpy()), bits(Dh))
    ln, codes = oracle.encode(x, cent, cb)
    off, lc, li = oracle.build_lists(ln, codes, np.arange(len(x)), 40)
    _, cand = oracle.search(q, cent, cb, off, lc, li, 40, 3000)
    De, Ie = oracle.rerank(q, x, cand, 10)
    assert np.array_equal(Ih, Ie) and np.array_equal(bits(Dh), bits(De))


def test_search_candidates_rows_longer_than_the_resident_tiles(faiss, oracle):
    """The set selection keeps up to 4 tiles of 8192 scores of a row in registers; a longer row (here ~45 k scores: every
    list of a 45 000-vector index probed) is streamed twice instead.  Same contract as above, at kc = 3000 (the 4096-slot
    kernel hands long rows to the 8192-slot one), 5120 and 8192, with a run of identical vectors across the cut."""
    import torch
    cent, cb, x, q = random_problem(57, 32, 4, 24, 45000, 9)
    x[30000:33000] = x[7]
    q[0] = x[7]
    idx = make_index(faiss, cent, cb)
    idx.add(x)
    ln, codes = oracle.encode(x, cent, cb)
    off, lc, li = oracle.build_lists(ln, codes, np.arange(len(x)), 24)
    qd = torch.from_numpy(q).cuda()
    for kc in (3000, 5120, 8192):
        I = torch.empty((len(q), kc), dtype=torch.int64, device="cuda")
        idx.search_candidates_into(qd, kc, I, 24)
        _, Ie = oracle.search(q, cent, cb, off, lc, li, 24, kc)
        got = I.cpu().numpy()
        for r in range(len(q)):
            a, b = np.sort(got[r]), np.sort(Ie[r])
            assert np.array_equal(a, b), (kc, r, int((a != b).sum()))


@pytest.mark.parametrize("d,M", [(128, 16), (192, 48), (1024, 64)])
def test_refine_sq8_matches_oracle(faiss, oracle, d, M):
    """factory "IVF..,PQ..,Refine(SQ8)" (faiss IndexScalarQuantizer QT_8bit: per-dimension ranges trained as min /
    span, one byte per component): trained ranges, code bytes, decoded rows and re-ranked (D, I) all bit-equal to the
    oracle's restatement.  d = 128 / 1024 take the streaming kernel (whole 128-byte pieces), d = 192 the simple one."""
    import torch
    nlist, n, nq, k = 32, 6000, 40, 10
    cent, cb, x, q = random_problem(78 + d, d, M, nlist, n, nq)
    x[:, 7] = 0.5                                              # a constant dimension: vdiff == 0
    idx = faiss.index_factory(d, f"IVF{nlist},PQ{M},Refine(SQ8)", faiss.METRIC_INNER_PRODUCT)
    sqi = idx.refine_index
    assert isinstance(sqi, faiss.IndexScalarQuantizer) and sqi.qtype == faiss.ScalarQuantizer.QT_8bit and not idx.is_trained
    idx.base_index.set_centroids(cent)
    idx.base_index.set_codebook(cb)
    with pytest.raises(RuntimeError, match="not trained"):
        sqi.add(x[:4])
    # training in two chunks (host rows, then device rows merged in) = training on all rows at once
    tr_rows = x[: n // 2]
    sqi.train(tr_rows[:1000])
    sqi.train(torch.from_numpy(tr_rows[1000:]).cuda(), merge=True)
    tr = oracle.sq8_train(tr_rows)
    assert np.array_equal(bits(sqi.sq.trained), bits(tr)) and idx.is_trained
    idx.add(x[: n // 3])                                       # rows outside the trained ranges clip
    idx.add(torch.from_numpy(x[n // 3:]).cuda())
    codes = oracle.sq8_encode(x, tr)
    xd = oracle.sq8_decode(codes, tr)
    assert np.array_equal(bits(sqi.reconstruct_n(0, n)), bits(xd))
    some = np.array([n - 1, 0, 1, 2, 77, 76, 5000, 5001], np.int64)       # the stored bytes of any rows (bench parity hook)
    assert np.array_equal(sqi.get_rows(some), codes[some]) and np.array_equal(sqi.get_rows(torch.from_numpy(some).cuda()), codes[some])
    ln, pq = oracle.encode(x, cent, cb, True)
    off, lc, li = oracle.build_lists(ln, pq, np.arange(n), nlist)
    for nprobe, kf in ((4, 4), (8, 16), (32, 30)):
        faiss.ParameterSpace().set_index_parameters(idx, f"nprobe={nprobe},k_factor_rf={kf}")
        D, I = idx.search(q, k)
        _, cand = oracle.search(q, cent, cb, off, lc, li, nprobe, k * kf, True)
        De, Ie = oracle.rerank_sq8(q, codes, tr, cand, k)
        assert np.array_equal(I, Ie) and np.array_equal(bits(D), bits(De)), (nprobe, kf)
    Dt, It = idx.search(torch.from_numpy(q).cuda(), k)
    assert np.array_equal(It.cpu().numpy(), Ie) and np.array_equal(bits(Dt.cpu().numpy()), bits(De))
    # the ranges travel: a second store given the trained vector encodes the same bytes
    other = faiss.IndexScalarQuantizer(d, faiss.ScalarQuantizer.QT_8bit, faiss.METRIC_INNER_PRODUCT)
    other.sq.trained = tr
    other.add(x[:100])
    assert np.array_equal(bits(other.reconstruct_n(0, 100)), bits(xd[:100]))
    with pytest.raises(RuntimeError, match="already holds"):
        other.train(x[:10])


@pytest.mark.parametrize("kc,k", [(1000, 10), (5120, 10), (8192, 32), (300, 1), (2560, 20)])
def test_rerank_topk_of_long_candidate_lists(faiss, oracle, kc, k, monkeypatch):
    """IndexFlat.rerank with candidate lists of 256..8192 entries and k <= 32 (the last step of the recall >= 0.95 point:
    10 of 5120): topk_rows_kernel against the oracle's re-rank and against the general merge route (MI_NO_TOPK_ROWS=1) --
    empty slots (-1), a caller's duplicate ids (both copies are returned, like the oracle), fewer valid candidates than k."""
    import torch
    d, n, nq = 64, 6000, 23
    rng = np.random.default_rng(kc + k)
    x = rng.standard_normal((n, d)).astype(np.float32)
    x[100:140] = x[100]                                        # equal scores: ids decide
    q = rng.standard_normal((nq, d)).astype(np.float32)
    cand = rng.integers(0, n, (nq, kc)).astype(np.int64)       # (with replacement: duplicates)
    cand[rng.random((nq, kc)) < 0.1] = -1
    cand[0, :40] = np.arange(100, 140)
    cand[1] = -1
    cand[1, 5] = 77                                            # one valid candidate: the rest of the row is padding
    cand[2] = -1
    idx = faiss.IndexFlatIP(d)
    idx.add(x)
    De, Ie = oracle.rerank(q, x, cand, k)
    outs = {}
    for name in ("rows", "merge"):
        monkeypatch.delenv("MI_NO_TOPK_ROWS", raising=False)
        if name == "merge":
            monkeypatch.setenv("MI_NO_TOPK_ROWS", "1")
        D, I = idx.rerank(torch.from_numpy(q).cuda(), torch.from_numpy(cand).cuda(), k)
        outs[name] = (D.cpu().numpy(), I.cpu().numpy())
        assert np.array_equal(outs[name][1], Ie) and np.array_equal(bits(outs[name][0]), bits(De)), name
    assert (outs["rows"][1][2] == -1).all() and outs["rows"][1][1, 0] == 77 and (outs["rows"][1][1, 1:] == -1).all()


def test_release_workspaces_gives_the_scratch_back(faiss):
    """mi_index_release_workspaces / mi_flat_release_workspaces: searches on several streams leave a scratch set per stream
    behind; releasing them returns the memory and the next search (which allocates again) gives the same bits."""
    import torch
    d, M, nlist, n, k = 128, 16, 64, 30000, 10
    cent, cb, x, q = random_problem(93, d, M, nlist, n, 512)
    idx = faiss.index_factory(d, f"IVF{nlist},PQ{M},Refine(SQ8)", faiss.METRIC_INNER_PRODUCT)
    idx.base_index.set_centroids(cent)
    idx.base_index.set_codebook(cb)
    idx.refine_index.train(x)
    idx.add(x)
    faiss.ParameterSpace().set_index_parameters(idx, "nprobe=16,k_factor_rf=64")
    qd = torch.from_numpy(q).cuda()
    D0, I0 = idx.search(qd, k)
    streams = [torch.cuda.Stream() for _ in range(4)]
    outs = []
    for s_ in streams:
        D, I = torch.empty_like(D0), torch.empty_like(I0)
        cI = torch.empty((512, 640), dtype=torch.int64, device="cuda")
        idx.search_into(qd, k, D, I, None, cI, int(s_.cuda_stream))
        outs.append((D, I))
    torch.cuda.synchronize()
    for D, I in outs:
        assert torch.equal(I, I0) and torch.equal(D.view(torch.int32), D0.view(torch.int32))
    free0 = torch.cuda.mem_get_info()[0]
    idx.release_workspaces()
    free1 = torch.cuda.mem_get_info()[0]
    assert free1 > free0 + 5 * 512 * 640 * 4                   # at least the five re-rank score buffers came back
    D1, I1 = idx.search(qd, k)
    assert torch.equal(I1, I0) and torch.equal(D1.view(torch.int32), D0.view(torch.int32))
    idx.base_index.release_workspaces()
    idx.release_workspaces()                                   # nothing left: a no-op


def test_refine_sq8_at_the_timed_shape_matches_oracle(faiss, oracle):
    """The shape bench.py's at_recall_095 point times on the 207 M index -- d 1024, k 10, k_factor 512: 5120 candidates
    per query through the unordered-candidates scan, select_pairs_kernel<16,true,512> and the streaming SQ8 re-rank -- for a
    batch of 256 queries against the oracle: candidate SETS equal, re-ranked ids and score bits equal (duplicated rows:
    ties at every rank).  (reference Makefile:32 `tune` -> the (nprobe, k_factor) operating point.)"""
    import torch
    d, M, nlist, n, nq, k, kf, nprobe = 1024, 64, 64, 40000, 256, 10, 512, 24
    cent, cb, x, q = random_problem(4242, d, M, nlist, n, nq)
    x[n // 2:] = x[: n - n // 2]                                        # every vector twice
    base = make_index(faiss, cent, cb)
    r = faiss.IndexScalarQuantizer(d, faiss.ScalarQuantizer.QT_8bit, faiss.METRIC_INNER_PRODUCT)
    r.train(x[:20000])
    idx = faiss.IndexRefine(base, r)
    idx.add(torch.from_numpy(x).cuda())
    idx.nprobe, idx.k_factor = nprobe, kf
    qd = torch.from_numpy(q).cuda()
    cI = torch.empty((nq, k * kf), dtype=torch.int64, device="cuda")
    base.search_candidates_into(qd, k * kf, cI)
    D, I = idx.search(qd, k)
    tr = r.sq.trained
    codes = oracle.sq8_encode(x, tr)
    ln, pq = oracle.encode(x, cent, cb, True)
    off, lc, li = oracle.build_lists(ln, pq, np.arange(n), nlist)
    _, cand = oracle.search(q, cent, cb, off, lc, li, nprobe, k * kf, True)
    assert (cand >= 0).all()                                            # the lists hold more than 5120 codes per query
    assert np.array_equal(np.sort(cI.cpu().numpy(), axis=1), np.sort(cand, axis=1))
    De, Ie = oracle.rerank_sq8(q, codes, tr, cand, k)
    assert np.array_equal(I.cpu().numpy(), Ie) and np.array_equal(bits(D.cpu().numpy()), bits(De))
    # the bench's parity leg: the oracle fed with nothing but the candidate rows' stored bytes (ids remapped order-preservingly)
    sub = slice(0, 32)
    cs = cI[sub].cpu().numpy()
    uniq, inv = np.unique(cs, return_inverse=True)
    D2, I2 = oracle.rerank_sq8(q[sub], r.get_rows(uniq), tr, inv.reshape(cs.shape), k)
    assert np.array_equal(uniq[I2], Ie[sub]) and np.array_equal(bits(D2), bits(De[sub]))


def test_refine_add_with_ids_only_accepts_positions(faiss):
    d, M, nlist = 64, 8, 16
    cent, cb, x, q = random_problem(3, d, M, nlist, 300, 4)
    idx = faiss.IndexRefineFlat(make_index(faiss, cent, cb))
    idx.add_with_ids(x[:100], np.arange(100))               # positions: fine
    idx.add_with_ids(x[100:200], np.arange(100, 200))
    assert idx.ntotal == 200 and idx.refine_index.ntotal == 200
    with pytest.raises(NotImplementedError, match="positions"):
        idx.add_with_ids(x[200:], np.arange(100) * 3)
