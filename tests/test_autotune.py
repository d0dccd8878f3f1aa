"""The `tune` step's host logic (reference Makefile:32 -> params.json): faiss's
ParameterSpace / OperatingPoints / criteria as restated in autotune.py.  CPU only:
the index behind the explorer is a small numpy stand-in with the same duck type
(`nlist`, `nprobe`, `search`), so nothing here launches a kernel."""
import importlib
import itertools

import numpy as np
import pytest

at = importlib.import_module("abstracts_search_amd.autotune")


def brute_front(pts):
    """Pareto front by definition, with faiss's tie rules: a point goes if another one is
    at least as accurate and strictly faster, or is the same (perf, t) added earlier;
    zero-perf points are never kept."""
    keep = []
    for i, (p, t) in enumerate(pts):
        if p == 0:
            continue
        dom = any(j != i and p2 > 0 and ((p2 >= p and t2 < t) or (p2 == p and t2 == t and j < i))
                  for j, (p2, t2) in enumerate(pts))
        if not dom:
            keep.append((p, t))
    return sorted(set(keep))


def test_operating_points_pareto_front_random():
    rng = np.random.default_rng(0)
    for trial in range(200):
        n = int(rng.integers(1, 30))
        # few distinct values so that ties in perf happen; distinct times
        perf = rng.integers(0, 8, n) / 8.0
        t = rng.permutation(n * 3)[:n] / 10.0 + 0.1
        ops = at.OperatingPoints()
        for i in range(n):
            ops.add(perf[i], t[i], f"k{i}", i)
        assert len(ops.all_pts) == n
        front = [(p.perf, p.t) for p in ops.optimal_pts]
        assert front[0] == (0.0, 0.0) and ops.optimal_pts[0].cno == -1
        assert front[1:] == brute_front(list(zip(perf.tolist(), t.tolist()))), trial
        # both coordinates strictly increase along the front
        assert all(a[0] < b[0] and a[1] < b[1] for a, b in zip(front, front[1:]))
        for target in (0.0, 0.3, 0.5, 0.99, 1.0):
            reach = [tt for pp, tt in front if pp >= target]
            assert ops.t_for_perf(target) == (min(reach) if reach else 1e50)


def test_operating_points_ties_and_merge():
    ops = at.OperatingPoints()
    assert ops.add(0.5, 1.0, "a", 0)
    assert not ops.add(0.5, 1.0, "b", 1)          # same point again: first one stays
    assert not ops.add(0.4, 1.0, "c", 2)          # less accurate, not faster
    assert ops.add(0.5, 0.9, "d", 3)              # same accuracy, faster: replaces
    assert [p.key for p in ops.optimal_pts] == ["", "d"]
    assert not ops.add(0.0, 0.1, "zero", 4)
    assert ops.add(0.9, 0.5, "e", 5)              # dominates d
    assert [p.key for p in ops.optimal_pts] == ["", "e"]
    other = at.OperatingPoints()
    other.add(0.95, 2.0, "x", 0)
    other.add(0.2, 3.0, "y", 1)
    assert ops.merge_with(other, "o/") == 1
    assert [p.key for p in ops.optimal_pts] == ["", "e", "o/x"]
    assert len(ops.all_pts) == 8


def test_criteria_match_their_definitions():
    rng = np.random.default_rng(1)
    nq, R, ngt = 40, 10, 20
    gt = np.stack([rng.permutation(1000)[:ngt] for _ in range(nq)]).astype(np.int64)
    I = np.stack([rng.permutation(1000)[:R] for _ in range(nq)]).astype(np.int64)
    I[::2, 3] = gt[::2, 0]                         # plant the true NN in half of the rows
    I[1::4, :4] = gt[1::4, 2:6]                    # and part of the top-10 in others
    I[5, :] = -1                                   # an unfilled result row
    one = at.OneRecallAtRCriterion(nq, R)
    one.set_groundtruth(None, gt)
    want1 = np.mean([gt[q, 0] in I[q] for q in range(nq)])
    assert one.evaluate(None, I) == pytest.approx(want1) and want1 >= 0.45
    inter = at.IntersectionCriterion(nq, R)
    inter.set_groundtruth(np.zeros((nq, ngt), np.float32), gt)
    want = np.mean([len(set(I[q]) & set(gt[q, :R])) / R for q in range(nq)])
    assert inter.evaluate(None, I) == pytest.approx(want)
    assert inter.evaluate(None, gt[:, :R]) == 1.0
    with pytest.raises(ValueError):
        inter.evaluate(None, I[:, :5])
    short = at.IntersectionCriterion(nq, R)
    short.set_groundtruth(None, gt[:, :5])
    with pytest.raises(RuntimeError):
        short.evaluate(None, I)
    with pytest.raises(RuntimeError):
        at.IntersectionCriterion(nq, R).evaluate(None, I)


class ToyIVF:
    """Exact search restricted to the nprobe best of `nlist` random buckets."""

    def __init__(self, x, nlist, seed=0):
        rng = np.random.default_rng(seed)
        self.x, self.nlist, self.nprobe = x, nlist, 1
        self.cent = x[rng.choice(len(x), nlist, replace=False)]
        self.assign = np.argmax(x @ self.cent.T, axis=1)
        self.calls = []

    def search(self, q, k):
        self.calls.append((self.nprobe, len(q)))
        S = q @ self.x.T
        probe = np.argsort(-(q @ self.cent.T), axis=1, kind="stable")[:, :self.nprobe]
        for i in range(len(q)):
            S[i, ~np.isin(self.assign, probe[i])] = -np.inf
        I = np.argsort(-S, axis=1, kind="stable")[:, :k]
        D = np.take_along_axis(S, I, 1)
        I[np.isinf(D)] = -1
        return D.astype(np.float32), I.astype(np.int64)


class ToyRefine:
    def __init__(self, base):
        self.base_index, self.k_factor = base, 1.0

    def search(self, q, k):
        _, cand = self.base_index.search(q, int(k * self.k_factor))
        return np.zeros((len(q), k), np.float32), cand[:, :k]


def test_parameter_space_ranges_names_and_setting():
    x = np.random.default_rng(2).standard_normal((300, 8)).astype(np.float32)
    ivf = ToyIVF(x, 20)
    ps = at.ParameterSpace()
    ps.initialize(ivf)
    assert [pr.name for pr in ps.parameter_ranges] == ["nprobe"]
    assert ps.parameter_ranges[0].values == [1, 2, 4, 8, 16]          # powers of two below nlist
    big = ToyIVF(x, 20)
    big.nlist = 65536
    ps.initialize(big)
    assert ps.parameter_ranges[0].values[-1] == 4096 and ps.n_combinations() == 13
    ref = ToyRefine(ivf)
    ps.initialize(ref)
    assert [pr.name for pr in ps.parameter_ranges] == ["k_factor_rf", "nprobe"]
    assert ps.n_combinations() == 7 * 5
    assert ps.combination_name(0) == "k_factor_rf=1,nprobe=1"
    assert ps.combination_name(7 * 5 - 1) == "k_factor_rf=64,nprobe=16"
    assert ps.combination_name(3 + 7 * 2) == "k_factor_rf=8,nprobe=4"   # first parameter varies fastest
    for c1, c2 in itertools.product(range(35), repeat=2):
        ge = (c1 % 7 >= c2 % 7) and (c1 // 7 >= c2 // 7)
        assert ps.combination_ge(c1, c2) == ge
    ps.set_index_parameters(ref, "nprobe=8, k_factor_rf=4")
    assert ivf.nprobe == 8 and ref.k_factor == 4.0
    ps.set_index_parameters(ref, 3 + 7 * 2)
    assert ivf.nprobe == 4 and ref.k_factor == 8.0
    ps.set_index_parameter(ivf, "nprobe", 2.0)
    assert ivf.nprobe == 2 and isinstance(ivf.nprobe, int)
    for bad in ("ht=3", "nprobe", "nprobe=0", "k_factor_rf=0.5"):
        with pytest.raises(ValueError):
            ps.set_index_parameters(ref, bad)
    with pytest.raises(ValueError):
        ps.set_index_parameter(ivf, "k_factor_rf", 2)                   # not a refine index


def test_explore_exhaustive_and_pruned():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2000, 16)).astype(np.float32)
    q = x[rng.choice(2000, 30)] + 0.1 * rng.standard_normal((30, 16)).astype(np.float32)
    gt = np.argsort(-(q @ x.T), axis=1, kind="stable")[:, :10]
    ivf = ToyIVF(x, 32)
    crit = at.IntersectionCriterion(30, 10)
    crit.set_groundtruth(None, gt)
    ps = at.ParameterSpace()
    ps.initialize(ivf)
    ps.verbose = 0
    ps.n_experiments = 0
    ps.batchsize = 16
    ops = ps.explore(ivf, q, crit)
    assert [p.cno for p in ops.all_pts] == list(range(5))
    assert ivf.calls[:3] == [(1, 16), (1, 16), (1, 14)]                # a warm-up batch, then every query once
    perf = [p.perf for p in ops.all_pts]
    assert perf == sorted(perf) and perf[-1] > perf[0]                  # more probes never hurt here
    # every reported perf is reproducible by setting the key
    for p in ops.all_pts:
        ps.set_index_parameters(ivf, p.key)
        assert crit.evaluate(*ivf.search(q, 10)) == p.perf
    # pruned exploration: first and last combination always run, implied ones are skipped
    ps.n_experiments = 500
    ops2 = at.OperatingPoints()
    # pretend nprobe=2 already reached the accuracy of the full index in less time than nprobe=1
    ops2.add(1.0, 1e-9, ps.combination_name(1), 1)
    ps.explore(ivf, q, crit, ops2)
    ran = [p.cno for p in ops2.all_pts[1:]]
    assert ran == [0]
    assert 2 not in ran and 3 not in ran and 4 not in ran               # >= nprobe=2 cannot beat it
    with pytest.raises(ValueError):
        ps.explore(ivf, q[:5], crit)
    with pytest.raises(RuntimeError):
        at.ParameterSpace().explore(ivf, q, crit)


def test_tune_and_params_file(tmp_path):
    rng = np.random.default_rng(4)
    x = rng.standard_normal((1500, 16)).astype(np.float32)
    q = x[:25] + 0.05 * rng.standard_normal((25, 16)).astype(np.float32)
    gt = np.argsort(-(q @ x.T), axis=1, kind="stable")[:, :10]
    ref = ToyRefine(ToyIVF(x, 16))
    ops = at.tune(ref, q, gt, k=10)
    assert ops.optimal_pts[-1].perf == max(p.perf for p in ops.all_pts)
    doc = at.write_params(str(tmp_path / "params.json"), ops, min_perf=0.5)
    chosen = at.params_for(ops, 0.5)
    assert doc["index_parameters"] == chosen.key and chosen.perf >= 0.5
    assert all(a["perf"] < b["perf"] and a["t"] < b["t"]
               for a, b in zip(doc["optimal_points"], doc["optimal_points"][1:]))
    ref.base_index.nprobe, ref.k_factor = 1, 1.0
    back = at.read_params(str(tmp_path / "params.json"), ref)
    assert back == doc
    want = dict(tok.split("=") for tok in chosen.key.split(","))
    assert ref.base_index.nprobe == int(float(want["nprobe"])) and ref.k_factor == float(want["k_factor_rf"])
    assert at.params_for(ops, 2.0) is ops.optimal_pts[-1]
