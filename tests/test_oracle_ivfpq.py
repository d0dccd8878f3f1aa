"""CPU tests of the oracle itself: C restatement vs the committed golden
vectors, vs the independent numpy brute force, and algebraic properties.
(PARITY UNPINNED -- SURVEY.md 8(c): no reference tests / golden vectors exist.)"""
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "ivfpq_tiny.npz"))


def test_oracle_matches_golden(oracle, gold):
    cent, cb, x, q, ids = gold["centroids"], gold["codebook"], gold["x"], gold["q"], gold["ids"]
    k = int(gold["k"])
    list_no, codes = oracle.encode(x, cent, cb, True)
    assert np.array_equal(list_no, gold["list_no"])
    assert np.array_equal(codes, gold["codes"])
    assert np.array_equal(oracle.lut(q[0], cb), gold["lut_q0"])
    off, lc, li = oracle.build_lists(list_no, codes, ids, cent.shape[0])
    for nprobe in (1, 4, 16):
        D, I, cI, cD = oracle.search(q, cent, cb, off, lc, li, nprobe, k, True, return_coarse=True)
        assert np.array_equal(I, gold[f"I_np{nprobe}"])
        assert np.array_equal(D.view(np.uint32), gold[f"D_np{nprobe}"].view(np.uint32))
        assert np.array_equal(cI, gold[f"cI_np{nprobe}"])
        assert np.array_equal(cD.view(np.uint32), gold[f"cD_np{nprobe}"].view(np.uint32))
    D, I = oracle.flat_ip(q, x, k)
    assert np.array_equal(I, gold["flat_I"])
    assert np.array_equal(D.view(np.uint32), gold["flat_D"].view(np.uint32))


def test_golden_ties_ordered_by_id(gold):
    # rows 500..511 duplicate rows 0..11: equal scores must appear in ascending id order
    D, I = gold["D_np16"], gold["I_np16"]
    for d_row, i_row in zip(D, I):
        for a in range(len(d_row) - 1):
            if d_row[a] == d_row[a + 1] and i_row[a + 1] >= 0:
                assert i_row[a] < i_row[a + 1]
    assert (D[:, :-1] >= D[:, 1:]).all()


def _random_index(rng, d, M, nlist, n):
    cent = rng.standard_normal((nlist, d)).astype(np.float32)
    cb = (0.3 * rng.standard_normal((M, 256, d // M))).astype(np.float32)
    x = (cent[rng.integers(0, nlist, n)] + 0.3 * rng.standard_normal((n, d))).astype(np.float32)
    return cent, cb, x


@pytest.mark.parametrize("by_residual", [True, False])
def test_oracle_vs_numpy_bruteforce(oracle, by_residual):
    rng = np.random.default_rng(5)
    d, M, nlist, n, nq, k = 32, 4, 12, 700, 24, 7
    cent, cb, x = _random_index(rng, d, M, nlist, n)
    q = x[:nq] + 0.05 * rng.standard_normal((nq, d)).astype(np.float32)
    ln, codes = oracle.encode(x, cent, cb, by_residual)
    off, lc, li = oracle.build_lists(ln, codes, np.arange(n), nlist)
    for nprobe in (1, 3, 12, 40):
        D, I = oracle.search(q, cent, cb, off, lc, li, nprobe, k, by_residual)
        D2, I2 = oracle.brute_force_search(q, cent, cb, off, lc, li, nprobe, k, by_residual)
        assert np.array_equal(I >= 0, I2 >= 0)
        m = I >= 0
        assert np.allclose(D[m], D2[m], atol=1e-4, rtol=0)
        assert (I[m] == I2[m]).mean() > 0.98  # float64 may reorder near-ties only


def test_oracle_edge_cases(oracle):
    rng = np.random.default_rng(6)
    d, M, nlist, n = 16, 4, 8, 5
    cent, cb, x = _random_index(rng, d, M, nlist, n)
    ln, codes = oracle.encode(x, cent, cb)
    off, lc, li = oracle.build_lists(ln, codes, np.arange(n) + 10, nlist)
    # k larger than the number of stored vectors -> -1 / -FLT_MAX padding
    D, I = oracle.search(x[:2], cent, cb, off, lc, li, nlist, 9)
    assert (I[:, n:] == -1).all() and (D[:, n:] == -oracle.FLT_MAX).all()
    assert sorted(I[0, :n].tolist()) == list(range(10, 10 + n))
    # empty index
    off0 = np.zeros(nlist + 1, np.int64)
    D, I = oracle.search(x[:2], cent, cb, off0, lc[:0], li[:0], 3, 4)
    assert (I == -1).all()


@settings(max_examples=25, deadline=None)
@given(st.integers(1, 4), st.integers(1, 6), st.integers(1, 8), st.integers(0, 2 ** 31 - 1))
def test_merge_equals_global_topk(nparts, nq, k, seed):
    from oracle import ivfpq_oracle as O
    rng = np.random.default_rng(seed)
    # unique ids across parts, some duplicated scores, some -1 padding
    D = np.round(rng.standard_normal((nparts, nq, k)), 1).astype(np.float32)
    I = rng.permutation(nparts * nq * k).reshape(nparts, nq, k).astype(np.int64)
    D = -np.sort(-D, axis=2)
    pad = rng.random((nparts, nq, k)) < 0.2
    pad = np.maximum.accumulate(pad, axis=2)  # padding only at the tail
    D[pad], I[pad] = -O.FLT_MAX, -1
    # each part must itself be sorted under (score desc, id asc)
    for p in range(nparts):
        for qi in range(nq):
            o = np.lexsort((I[p, qi], -D[p, qi]))
            keep = I[p, qi][o] >= 0
            oo = np.concatenate([o[keep], o[~keep]])
            D[p, qi], I[p, qi] = D[p, qi][oo], I[p, qi][oo]
    Dm, Im = O.merge(D, I)
    for qi in range(nq):
        s, i = D[:, qi].ravel(), I[:, qi].ravel()
        v = i >= 0
        o = np.lexsort((i[v], -s[v]))[:k]
        exp_i = np.full(k, -1, np.int64)
        exp_i[:len(o)] = i[v][o]
        assert np.array_equal(Im[qi], exp_i)
    # permuting the parts does not change the result
    perm = rng.permutation(nparts)
    Dm2, Im2 = O.merge(D[perm], I[perm])
    assert np.array_equal(Im, Im2) and np.array_equal(Dm, Dm2)


def test_rerank_restatement():
    """oracle.rerank: exact scores of candidate ids, k best under (score desc, id asc);
    with the true top-k among unique candidates it reproduces flat_ip, empty slots are
    skipped, short candidate lists pad with -1 / -FLT_MAX."""
    from oracle import ivfpq_oracle as o
    rng = np.random.default_rng(0)
    base = rng.standard_normal((500, 32)).astype(np.float32)
    q = rng.standard_normal((7, 32)).astype(np.float32)
    Df, If = o.flat_ip(q, base, 10)
    cand = np.empty((7, 40), np.int64)
    for i in range(7):
        rest = np.setdiff1d(np.arange(500), If[i])
        cand[i] = rng.permutation(np.concatenate([If[i], rng.permutation(rest)[:30]]))
    D, I = o.rerank(q, base, cand, 10)
    assert np.array_equal(I, If) and np.array_equal(D.view(np.uint32), Df.view(np.uint32))
    cand[:, 10:] = -1
    D, I = o.rerank(q, base, cand, 20)
    assert (I[:, 10:] == -1).all() and (D[:, 10:] == -np.finfo(np.float32).max).all()
    for i in range(7):
        assert set(I[i, :10]) == set(cand[i, :10]) and (np.diff(D[i, :10]) <= 0).all()
    # ties: equal vectors rank by ascending id
    base2 = np.repeat(base[:1], 6, 0)
    D, I = o.rerank(q[:1], base2, np.array([[5, 2, 4, 0, -1, 3]], np.int64), 3)
    assert I.tolist() == [[0, 2, 3]]


def test_l2_restatement_against_float64(oracle):
    """METRIC_L2: the expansion the oracle evaluates agrees with exact float64 squared distances
    to the decoded vectors (values to rounding, result sets except near-ties)."""
    rng = np.random.default_rng(11)
    d, M, nlist, n, nq, k = 32, 4, 8, 1500, 20, 10
    cent = rng.standard_normal((nlist, d)).astype(np.float32)
    cb = (0.4 * rng.standard_normal((M, 256, d // M))).astype(np.float32)
    x = (cent[rng.integers(0, nlist, n)] + 0.4 * rng.standard_normal((n, d))).astype(np.float32)
    q = (x[:nq] + 0.1 * rng.standard_normal((nq, d))).astype(np.float32)
    for by_res in (True, False):
        ln, codes, t = oracle.encode_l2(x, cent, cb, by_res)
        # the list is the L2-nearest centroid
        d2 = ((x[:, None, :].astype(np.float64) - cent[None].astype(np.float64)) ** 2).sum(2)
        assert (ln == d2.argmin(1)).mean() > 0.999
        off, lc, li, lt = oracle.build_lists_l2(ln, codes, np.arange(n), t, nlist)
        D, I = oracle.search_l2(q, cent, cb, off, lc, li, lt, nlist, k, by_res)     # all lists: exhaustive
        exact = oracle.brute_force_l2(q, cent, cb, off, lc, by_res)
        order = np.argsort(exact, axis=1, kind="stable")[:, :k]
        want = np.take_along_axis(exact, order, 1)
        assert np.allclose(D, want, rtol=1e-4, atol=1e-4)
        assert np.mean([len(set(a) & set(li[b])) / k for a, b in zip(I.tolist(), order.tolist())]) > 0.98
        assert (np.diff(D, axis=1) >= 0).all()
    Df, If = oracle.flat_l2(q, x, k)
    ex = ((q[:, None, :].astype(np.float64) - x[None].astype(np.float64)) ** 2).sum(2)
    assert np.allclose(Df, np.sort(ex, axis=1)[:, :k], rtol=1e-4, atol=1e-4)
    assert (If[:, 0] == np.arange(nq)).all()
    # unfilled slots
    D, I = oracle.search_l2(q, cent, cb, np.zeros(nlist + 1, np.int64), np.zeros((0, M), np.uint8), np.zeros(0, np.int64),
                            np.zeros(0, np.float32), 3, k)
    assert (I == -1).all() and (D == np.finfo(np.float32).max).all()


def test_sq8_restatement_against_float64_and_the_documented_formulas(oracle):
    """ScalarQuantizer QT_8bit (faiss ',Refine(SQ8)'): per-dimension min / max training, code =
    (int)(255 * clip((x - vmin) / vdiff)), decode = vmin + (code + 0.5) / 255 * vdiff, inner-product
    re-rank over the decoded rows.  Pinned here against numpy restatements of those published formulas
    (float64 for the values, float32 step by step for the bits of the codes)."""
    rng = np.random.default_rng(12)
    n, d, nq, kc, k = 3000, 64, 9, 40, 10
    x = (rng.standard_normal((n, d)) * rng.uniform(0.01, 2.0, d)).astype(np.float32)
    x[:, 5] = 0.25                                   # a constant dimension: vdiff == 0 -> code 0, decodes to vmin
    tr = oracle.sq8_train(x)
    assert np.array_equal(tr[:d], x.min(0)) and np.array_equal(tr[d:], x.max(0) - x.min(0))
    # encode: float32 step by step, as faiss's scalar code does; out-of-range inputs clip
    y = np.concatenate([x[:500], x[:50] * 3.0])
    codes = oracle.sq8_encode(y, tr)
    vmin, vdiff = tr[:d], tr[d:]
    with np.errstate(divide="ignore", invalid="ignore"):
        xi = ((y - vmin).astype(np.float32) / vdiff).astype(np.float32)
    xi = np.where(vdiff == 0, np.float32(0), np.clip(xi, np.float32(0), np.float32(1))).astype(np.float32)
    want = (np.float32(255) * xi).astype(np.float32).astype(np.int64)
    assert np.array_equal(codes, want.astype(np.uint8))
    assert codes.min() == 0 and codes.max() == 255 and (codes[:, 5] == 0).all()
    # decode: within an ulp or two of the float64 value of the published formula, and the quantisation error is <= half a step
    xd = oracle.sq8_decode(codes, tr)
    ref = vmin.astype(np.float64) + (codes.astype(np.float64) + 0.5) / 255.0 * vdiff.astype(np.float64)
    assert np.abs(xd - ref).max() <= 4e-7 * max(1.0, np.abs(ref).max())
    inside = np.abs(xd[:500] - x[:500])
    assert (inside <= 0.5 * vdiff / 255.0 * (1 + 1e-4) + 2e-6 * np.abs(x).max()).all()
    # re-rank: scores against float64 dot products of the decoded rows; order = (score desc, id asc); -1 slots skipped
    allc = oracle.sq8_encode(x, tr)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    cand = rng.integers(0, n, (nq, kc))
    cand[:, 3] = cand[:, 7]                          # a duplicated candidate: both copies rank, adjacent
    cand[0, 10:] = -1
    D, I = oracle.rerank_sq8(q, allc, tr, cand, k)
    dec = oracle.sq8_decode(allc, tr).astype(np.float64)
    for qi in range(nq):
        ids = cand[qi][cand[qi] >= 0]
        sc = dec[ids] @ q[qi].astype(np.float64)
        order = sorted(range(len(ids)), key=lambda j: (-sc[j], ids[j]))[:k]
        got = I[qi][I[qi] >= 0]
        # float64 may order near-ties differently from the f32 chain: compare as sets of (id) and the scores closely
        assert sorted(got.tolist()) == sorted(ids[order].tolist()) or np.abs(np.sort(sc[order])[::-1] - D[qi][:len(order)]).max() < 1e-4
        assert np.abs(np.sort(sc[order])[::-1] - D[qi][:len(order)]).max() < 1e-4 * max(1.0, np.abs(sc).max())
        assert all(D[qi][j] > D[qi][j + 1] or (D[qi][j] == D[qi][j + 1] and I[qi][j] <= I[qi][j + 1]) for j in range(len(got) - 1))
    assert sorted(I[0].tolist()) == sorted(cand[0][:10].tolist())      # 10 live candidates (one id twice): all of them, no -1
    cand[1, 4:] = -1
    D, I = oracle.rerank_sq8(q, allc, tr, cand, k)
    assert (I[1][4:] == -1).all() and (D[1][4:] == -np.finfo(np.float32).max).all() and (I[1][:4] >= 0).all()


@given(seed=st.integers(0, 2**31 - 1), M=st.sampled_from([4, 16, 64]), spread=st.sampled_from([1e-3, 1.0, 1e4, 1e20]),
       dis_scale=st.sampled_from([0.0, 1.0, 1e6, 1e-30]))
@settings(max_examples=60, deadline=None)
def test_chain_of_row_maxima_bounds_every_code_in_floating_point(seed, M, spread, dis_scale):
    """The exactness argument of the exact list pruning (csrc/ivfpq_kernels.h above prune_tables_kernel), on the oracle's
    arithmetic: a code scores fl(dis0 + acc), acc = the f32 chain 0 + LUT[0][c0] + LUT[1][c1] + ... (m ascending: ivfpq_oracle.c,
    "ADC sum"); with A = the same chain over the row maxima, acc <= A and fl(dis0 + acc) <= fl(dis0 + A) for the COMPUTED values,
    because rounded addition is monotone in both operands -- no slack term, whatever the magnitudes (mixed signs, a dynamic range
    that makes most additions inexact, subnormals)."""
    rng = np.random.default_rng(seed)
    lut = (rng.standard_normal((M, 256)) * spread * 10.0 ** rng.integers(-8, 8, (M, 1))).astype(np.float32)
    if seed % 3 == 0:
        lut[rng.integers(0, M), rng.integers(0, 256)] = np.float32(1e-42)      # a subnormal entry
    codes = rng.integers(0, 256, (500, M))
    dis0 = np.float32(rng.standard_normal() * dis_scale)
    acc = np.zeros(500, np.float32)
    A = np.float32(0.0)
    for m in range(M):                                                          # one rounded addition per step, as the kernel does
        acc = (acc + lut[m, codes[:, m]]).astype(np.float32)
        A = np.float32(A + lut[m].max())
        assert (acc <= A).all()
    s, U = (dis0 + acc).astype(np.float32), np.float32(dis0 + A)
    assert (s <= U).all()
    # and the bound is attained by the code that takes every row's maximum: nothing tighter holds for a whole list
    best = lut.argmax(axis=1)
    accb = np.float32(0.0)
    for m in range(M):
        accb = np.float32(accb + lut[m, best[m]])
    assert np.float32(dis0 + accb) == U
