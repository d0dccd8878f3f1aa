"""faiss's auto-tuning objects for the `tune` step (reference Makefile:32,
``sidecar-search index … tune`` -> ``params.json``): ``ParameterSpace``,
``OperatingPoints`` and the two recall criteria.

faiss itself is not in the reference tree; this restates the
published behaviour of faiss's AutoTune: a parameter space is the cartesian
product of per-parameter value lists (``nprobe`` for an IVF index: powers of
two below nlist; ``k_factor_rf`` for an IndexRefine: 1..64), a combination is
numbered with the first parameter varying fastest, every experiment is one
timed ``index.search`` scored by a criterion, and the set of (perf, time)
pairs is reduced to its Pareto front.  Experiments whose outcome is implied
by earlier ones (a combination that is >= in every parameter cannot be
faster, one that is <= cannot be more accurate) are skipped.

Host logic only: the searches it times run on the HIP path through the index
object it is given.  ``ht`` (polysemous filtering) is not offered because the
MI355X scan does not implement it.
"""
from __future__ import annotations

import json
import time

import numpy as np


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _host_i64(I) -> np.ndarray:
    if _is_torch(I):
        I = I.detach().cpu().numpy()
    return np.ascontiguousarray(I, dtype=np.int64)


# ----------------------------------------------------------------------
# criteria
# ----------------------------------------------------------------------

class AutoTuneCriterion:
    """Scores a search result against a ground truth; ``nnn`` neighbours are
    requested from the index, ``gt_nnn`` are known per query."""

    def __init__(self, nq: int, nnn: int):
        self.nq, self.nnn = int(nq), int(nnn)
        self.gt_nnn = 0
        self.gt_D = None
        self.gt_I = None

    def set_groundtruth(self, gt_D, gt_I):
        gt_I = _host_i64(gt_I)
        if gt_I.ndim != 2 or gt_I.shape[0] != self.nq:
            raise ValueError(f"ground truth must be [nq={self.nq}, gt_nnn], got {gt_I.shape}")
        self.gt_nnn = int(gt_I.shape[1])
        self.gt_I = gt_I
        self.gt_D = None if gt_D is None else np.asarray(gt_D.detach().cpu() if _is_torch(gt_D) else gt_D, np.float32)

    def _result(self, I) -> np.ndarray:
        I = _host_i64(I)
        if I.shape != (self.nq, self.nnn):
            raise ValueError(f"result must be [{self.nq}, {self.nnn}], got {I.shape}")
        if self.gt_I is None:
            raise RuntimeError("set_groundtruth() was not called")
        return I

    def evaluate(self, D, I) -> float:
        raise NotImplementedError


class OneRecallAtRCriterion(AutoTuneCriterion):
    """1-recall@R: the fraction of queries whose true nearest neighbour is
    among the first R results."""

    def __init__(self, nq: int, R: int):
        super().__init__(nq, R)
        self.R = int(R)

    def evaluate(self, D, I) -> float:
        I = self._result(I)
        if self.gt_nnn < 1:
            raise RuntimeError("ground truth holds no neighbours")
        hit = (I[:, :self.R] == self.gt_I[:, :1]).any(axis=1)
        return float(hit.sum()) / self.nq


class IntersectionCriterion(AutoTuneCriterion):
    """recall@R in the set sense: |result[:R] ∩ truth[:R]| / R, averaged over
    the queries (the recall@10 of BASELINE.json's metric with R = 10)."""

    def __init__(self, nq: int, R: int):
        super().__init__(nq, R)
        self.R = int(R)

    def evaluate(self, D, I) -> float:
        I = self._result(I)
        if self.gt_nnn < self.R:
            raise RuntimeError(f"ground truth has {self.gt_nnn} < R={self.R} neighbours per query")
        tot = 0
        for q in range(self.nq):
            tot += np.intersect1d(I[q, :self.R], self.gt_I[q, :self.R]).size
        return tot / float(self.nq * self.R)


# ----------------------------------------------------------------------
# operating points
# ----------------------------------------------------------------------

class OperatingPoint:
    __slots__ = ("perf", "t", "key", "cno")

    def __init__(self, perf: float, t: float, key: str, cno: int):
        self.perf, self.t, self.key, self.cno = float(perf), float(t), str(key), int(cno)

    def __repr__(self):
        return f"OperatingPoint(perf={self.perf:.4f}, t={self.t:.6f}, key={self.key!r}, cno={self.cno})"


class OperatingPoints:
    """All experiments plus their Pareto front (``optimal_pts``: perf and t
    both strictly increasing; it starts with "do nothing": perf 0 in time 0)."""

    def __init__(self):
        self.clear()

    def clear(self):
        self.all_pts: list[OperatingPoint] = []
        self.optimal_pts: list[OperatingPoint] = [OperatingPoint(0.0, 0.0, "", -1)]

    def add(self, perf: float, t: float, key: str, cno: int = 0) -> bool:
        """Record an experiment; True if it joins the Pareto front."""
        op = OperatingPoint(perf, t, key, cno)
        self.all_pts.append(op)
        if op.perf == 0:
            return False                      # nothing beats doing nothing at zero accuracy
        front = self.optimal_pts
        # position of the first front point that is at least as accurate
        pos = next((i for i, p in enumerate(front) if p.perf >= op.perf), len(front))
        if pos < len(front) and front[pos].t <= op.t:
            return False                      # dominated: as accurate or better, and no slower
        if pos < len(front) and front[pos].perf == op.perf:
            front[pos] = op
        else:
            front.insert(pos, op)
        # the newcomer may dominate slower, less accurate points before it
        while pos > 0 and front[pos - 1].t > op.t:
            del front[pos - 1]
            pos -= 1
        return True

    def merge_with(self, other: "OperatingPoints", prefix: str = "") -> int:
        n = 0
        for op in other.all_pts:
            n += self.add(op.perf, op.t, prefix + op.key, op.cno)
        return n

    def t_for_perf(self, perf: float) -> float:
        """Time of the cheapest known point reaching ``perf`` (1e50 if none does)."""
        for p in self.optimal_pts:
            if p.perf >= perf:
                return p.t
        return 1e50

    def display(self, only_optimal: bool = True) -> None:
        pts = self.optimal_pts if only_optimal else self.all_pts
        print(f"Tested {len(self.all_pts)} operating points, {len(self.optimal_pts)} ones are Pareto-optimal:")
        for p in pts:
            star = "" if only_optimal or not any(o is p for o in self.optimal_pts) else "*"
            print(f"cno={p.cno} key={p.key} perf={p.perf:.4f} t={p.t:.3f} {star}")


# ----------------------------------------------------------------------
# parameter space
# ----------------------------------------------------------------------

class ParameterRange:
    def __init__(self, name: str):
        self.name = name
        self.values: list[float] = []


def _unwrap(index):
    """(ivf index or None, refine index or None) behind ``index``."""
    refine = index if hasattr(index, "k_factor") and hasattr(index, "base_index") else None
    base = refine.base_index if refine is not None else index
    ivf = base if hasattr(base, "nlist") else None
    return ivf, refine


class ParameterSpace:
    """faiss.ParameterSpace: the tunable search-time parameters of an index."""

    def __init__(self):
        self.parameter_ranges: list[ParameterRange] = []
        self.verbose = 1
        self.n_experiments = 500
        self.batchsize = 1 << 30
        self.thread_over_batches = False      # accepted, unused: the batches run on one HIP stream
        self.min_test_duration = 0.0

    # -- ranges ---------------------------------------------------------
    def add_range(self, name: str) -> ParameterRange:
        for pr in self.parameter_ranges:
            if pr.name == name:
                return pr
        pr = ParameterRange(name)
        self.parameter_ranges.append(pr)
        return pr

    def initialize(self, index) -> None:
        """Default ranges: nprobe = 1, 2, 4 … below nlist (at most 2^12);
        k_factor_rf = 1, 2 … 64 for a refine index."""
        self.parameter_ranges = []
        ivf, refine = _unwrap(index)
        if refine is not None:
            self.add_range("k_factor_rf").values = [float(1 << i) for i in range(7)]
        if ivf is not None:
            pr = self.add_range("nprobe")
            for i in range(13):
                if (1 << i) >= ivf.nlist:
                    break
                pr.values.append(float(1 << i))

    def n_combinations(self) -> int:
        n = 1
        for pr in self.parameter_ranges:
            n *= len(pr.values)
        return n

    def _digits(self, cno: int) -> list[int]:
        out = []
        for pr in self.parameter_ranges:
            out.append(cno % len(pr.values))
            cno //= len(pr.values)
        return out

    def combination_ge(self, c1: int, c2: int) -> bool:
        """True if combination c1 is >= c2 in every parameter."""
        return all(a >= b for a, b in zip(self._digits(c1), self._digits(c2)))

    def combination_name(self, cno: int) -> str:
        return ",".join(f"{pr.name}={pr.values[j]:g}" for pr, j in zip(self.parameter_ranges, self._digits(cno)))

    def display(self) -> None:
        print(f"ParameterSpace, {len(self.parameter_ranges)} parameters, {self.n_combinations()} combinations:")
        for pr in self.parameter_ranges:
            print(f"   {pr.name}: " + " ".join(f"{v:g}" for v in pr.values))

    # -- setting --------------------------------------------------------
    def set_index_parameter(self, index, name: str, value) -> None:
        ivf, refine = _unwrap(index)
        if name == "nprobe" and ivf is not None:
            if int(value) < 1:
                raise ValueError("nprobe must be >= 1")
            ivf.nprobe = int(value)
        elif name == "k_factor_rf" and refine is not None:
            if float(value) < 1:
                raise ValueError("k_factor_rf must be >= 1")
            refine.k_factor = float(value)
        else:
            raise ValueError(f"ParameterSpace: could not set parameter {name!r} on {type(index).__name__}")

    def set_index_parameters(self, index, description) -> None:
        """``description``: "nprobe=16,k_factor_rf=4", or a combination number."""
        if isinstance(description, (int, np.integer)):
            description = self.combination_name(int(description))
        for tok in description.split(","):
            tok = tok.strip()
            if not tok:
                continue
            name, sep, val = tok.partition("=")
            if not sep:
                raise ValueError(f"invalid parameter setting {tok!r} (want name=value)")
            self.set_index_parameter(index, name.strip(), float(val))

    # -- exploration ----------------------------------------------------
    def update_bounds(self, cno: int, op: OperatingPoint, upper_bound_perf: float, lower_bound_t: float):
        """Tighten what experiment ``cno`` can achieve given a finished one."""
        if self.combination_ge(cno, op.cno) and op.t > lower_bound_t:
            lower_bound_t = op.t
        if self.combination_ge(op.cno, cno) and op.perf < upper_bound_perf:
            upper_bound_perf = op.perf
        return upper_bound_perf, lower_bound_t

    def _timed_search(self, index, xq, crit):
        nq = crit.nq
        bs = max(1, min(int(self.batchsize), nq))
        sync = None
        if _is_torch(xq):
            import torch
            sync = torch.cuda.synchronize
            sync()
        index.search(xq[:bs], crit.nnn)       # untimed: first-use allocations of this (k, nprobe)
        if sync:
            sync()
        nrun = 0
        t0 = time.perf_counter()
        while True:
            parts = [index.search(xq[q0:q0 + bs], crit.nnn) for q0 in range(0, nq, bs)]
            if sync:
                sync()
            nrun += 1
            if time.perf_counter() - t0 >= self.min_test_duration:
                break
        t = (time.perf_counter() - t0) / nrun
        I = np.concatenate([_host_i64(p[1]) for p in parts], axis=0)
        return I, t

    def explore(self, index, xq, crit: AutoTuneCriterion, ops: OperatingPoints | None = None) -> OperatingPoints:
        """Run the experiments and return their operating points.  With
        ``n_experiments == 0`` every combination runs in order; otherwise the
        cheapest and the most expensive combination run first and last, the
        rest in a fixed random order, skipping implied outcomes."""
        if len(xq) != crit.nq:
            raise ValueError(f"criterion was built for {crit.nq} queries, got {len(xq)}")
        ops = OperatingPoints() if ops is None else ops
        n_comb = self.n_combinations()
        if n_comb == 0 or not self.parameter_ranges:
            raise RuntimeError("empty parameter space: call initialize(index) first")
        exhaustive = self.n_experiments == 0
        if exhaustive:
            order = list(range(n_comb))
        else:
            n_exp = min(int(self.n_experiments), n_comb)
            middle = (np.random.default_rng(1234).permutation(n_comb - 2) + 1).tolist() if n_comb > 2 else []
            order = [0] + middle[:max(0, n_exp - 2)] + ([n_comb - 1] if n_exp > 1 else [])
        for xp, cno in enumerate(order):
            name = self.combination_name(cno)
            if not exhaustive:
                ub_perf, lb_t = 1.0, 0.0
                for op in ops.all_pts:
                    ub_perf, lb_t = self.update_bounds(cno, op, ub_perf, lb_t)
                if ops.t_for_perf(ub_perf) < lb_t:
                    if self.verbose:
                        print(f"  {xp}/{len(order)}: cno={cno} {name} skip (perf <= {ub_perf:.3f}, t >= {lb_t:.4f})")
                    continue
            self.set_index_parameters(index, name)
            I, t = self._timed_search(index, xq, crit)
            perf = crit.evaluate(None, I)
            keep = ops.add(perf, t, name, cno)
            if self.verbose:
                print(f"  {xp}/{len(order)}: cno={cno} {name} perf={perf:.4f} t={t:.5f} s {'*' if keep else ''}")
        return ops


# ----------------------------------------------------------------------
# the tune step
# ----------------------------------------------------------------------

def tune(index, xq, gt_I, k: int = 10, criterion: str = "intersection", ps: ParameterSpace | None = None,
         min_test_duration: float = 0.0, verbose: int = 0) -> OperatingPoints:
    """What ``index tune`` does with a trained, filled index: explore the
    parameter space on held-out queries against exact neighbours ``gt_I``."""
    nq = len(xq)
    crit = (IntersectionCriterion if criterion == "intersection" else OneRecallAtRCriterion)(nq, k)
    crit.set_groundtruth(None, gt_I)
    if ps is None:
        ps = ParameterSpace()
        ps.initialize(index)
    ps.verbose = verbose
    ps.min_test_duration = min_test_duration
    return ps.explore(index, xq, crit)


def params_for(ops: OperatingPoints, min_perf: float) -> OperatingPoint:
    """Cheapest Pareto point reaching ``min_perf`` (the most accurate one if none does)."""
    for p in ops.optimal_pts:
        if p.perf >= min_perf:
            return p
    return ops.optimal_pts[-1]


def write_params(fname: str, ops: OperatingPoints, min_perf: float | None = None) -> dict:
    """Write the tuning result as JSON.  sidecar-search's ``params.json``
    schema is not in the reference tree; this is the build's own: the Pareto
    front as faiss parameter strings plus, when ``min_perf`` is given, the
    selected one under "index_parameters"."""
    doc = {"optimal_points": [{"index_parameters": p.key, "perf": p.perf, "t": p.t}
                              for p in ops.optimal_pts if p.cno >= 0]}
    if min_perf is not None:
        p = params_for(ops, min_perf)
        doc.update(index_parameters=p.key, perf=p.perf, t=p.t)
    with open(fname, "w") as f:
        json.dump(doc, f, indent=1)
    return doc


def read_params(fname: str, index=None) -> dict:
    """Load ``write_params`` output; apply the selected parameters to ``index`` if given."""
    with open(fname) as f:
        doc = json.load(f)
    if index is not None and doc.get("index_parameters"):
        ParameterSpace().set_index_parameters(index, doc["index_parameters"])
    return doc
