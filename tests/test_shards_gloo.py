"""world_size-2/3 gloo tests (CPU) of the multi-GPU plumbing: vector sharding +
query all-gather + top-k all-gather + merge reproduces the unsharded result.
The local search / merge are injected from the oracle (allowed in tests); on
the GPU box the same ShardedIndex runs the HIP paths (bench.py --gpus N)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    rng = np.random.default_rng(3)
    d, M, nlist, n = 32, 4, 8, 1500
    cent = rng.standard_normal((nlist, d)).astype(np.float32)
    cb = (0.3 * rng.standard_normal((M, 256, d // M))).astype(np.float32)
    x = (cent[rng.integers(0, nlist, n)] + 0.3 * rng.standard_normal((n, d))).astype(np.float32)
    x[n - 20:] = x[:20]       # duplicates: ties must resolve identically when sharded
    return cent, cb, x


def _worker(rank, world, port, mode, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import ivfpq_oracle as O
    from abstracts_search_amd.shards import ShardedIndex, shard_rows
    cent, cb, x = _problem()
    n, nlist, nprobe, k, b = len(x), len(cent), 3, 10, 6
    rows = np.fromiter(shard_rows(n, rank, world), np.int64)
    ln, codes = O.encode(x[rows], cent, cb)
    off, lc, li = O.build_lists(ln, codes, rows, nlist)            # global ids

    def local_search(q, kk):
        D, I = O.search(q.numpy(), cent, cb, off, lc, li, nprobe, kk)
        return torch.from_numpy(D), torch.from_numpy(I)

    def merge(Dp, Ip):
        D, I = O.merge(Dp.numpy(), Ip.numpy())
        return torch.from_numpy(D), torch.from_numpy(I)

    def local_coarse(q, npb, lo, hi):
        D, I = O.flat_ip(q.numpy(), cent[lo:hi], npb)
        I = np.where(I >= 0, I + lo, -1)
        return torch.from_numpy(I.astype(np.int32)), torch.from_numpy(D)

    def local_search_pre(q, kk, cI, cD):
        D, I = O.search_preassigned(q.numpy(), cb, off, lc, li, cI.numpy(), cD.numpy(), kk)
        return torch.from_numpy(D), torch.from_numpy(I)

    if mode == "refine":
        # every shard: IVF-PQ over its rows numbered by position, exact re-ranking of k * 4
        # candidates against its raw vectors, positions -> global row numbers through id_map
        def shard_refine(r, q, kk):
            rr = np.fromiter(shard_rows(n, r, world), np.int64)
            ln_, codes_ = O.encode(x[rr], cent, cb)
            off_, lc_, li_ = O.build_lists(ln_, codes_, np.arange(len(rr)), nlist)   # local ids
            _, cand = O.search(q, cent, cb, off_, lc_, li_, nprobe, kk * 4)
            return O.rerank(q, x[rr], cand, kk)

        def local_refine(q, kk):
            D, I = shard_refine(rank, q.numpy(), kk)
            return torch.from_numpy(D), torch.from_numpy(I)

        sh = ShardedIndex(index=None, local_search=local_refine, merge=merge, id_map=torch.from_numpy(rows))
        rng = np.random.default_rng(100)
        qall = (x[rng.integers(0, n, world * b)] + 0.01 * rng.standard_normal((world * b, x.shape[1]))).astype(np.float32)
        D, I = sh.search(torch.from_numpy(qall[rank * b:(rank + 1) * b]), k)
        qs = qall[rank * b:(rank + 1) * b]
        parts = []
        for r in range(world):
            Dr, Ir = shard_refine(r, qs, k)
            rr = np.fromiter(shard_rows(n, r, world), np.int64)
            parts.append((Dr, np.where(Ir >= 0, rr[np.maximum(Ir, 0)], -1)))
        De, Ie = O.merge(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]))
        # scores are exact inner products of the global rows the ids name
        exact = np.array([[np.float32(0) if i < 0 else O.flat_ip(qs[j:j + 1], x[i:i + 1], 1)[0][0, 0] for i in row]
                          for j, row in enumerate(I.numpy())], np.float32)
        ok = (np.array_equal(I.numpy(), Ie) and np.array_equal(D.numpy().view(np.uint32), De.view(np.uint32))
              and np.array_equal(exact.view(np.uint32)[I.numpy() >= 0], D.numpy().view(np.uint32)[I.numpy() >= 0]))
        ret[rank] = bool(ok)
        dist.barrier()
        dist.destroy_process_group()
        return
    if mode.endswith("+affine"):
        # shard lists numbered by position; the merge's closed-form map (global = local * world + rank)
        # restores the global row numbers
        off_l, lc_l, li_l = O.build_lists(ln, codes, np.arange(len(rows)), nlist)

        def local_search_pos(q, kk):
            D, I = O.search(q.numpy(), cent, cb, off_l, lc_l, li_l, nprobe, kk)
            return torch.from_numpy(D), torch.from_numpy(I)

        sh = ShardedIndex(index=None, local_search=local_search_pos, merge=merge, id_affine=(world, 0, 1))
    else:
        sh = ShardedIndex(index=None, local_search=local_search, merge=merge, shard_coarse=mode.endswith("+coarse"),
                          local_coarse=local_coarse, local_search_pre=local_search_pre, nlist=nlist, nprobe=nprobe)
    rng = np.random.default_rng(100)
    qall = (x[rng.integers(0, n, world * b)] + 0.01 * rng.standard_normal((world * b, x.shape[1]))).astype(np.float32)
    if mode.startswith("own"):
        D, I = sh.search(torch.from_numpy(qall[rank * b:(rank + 1) * b]), k)
        qs = qall[rank * b:(rank + 1) * b]
    else:
        D, I = sh.search_replicated(torch.from_numpy(qall), k)
        qs = qall
    ok2 = True
    if mode.startswith("replicated"):
        # the per-rank step split bench.py prints at N > 1 (every stage named, every rank answers; the probe leaves no state behind)
        sp = sh.probe_split(torch.from_numpy(qall), k, reps=2)
        want = ({"coarse_slice_ms", "coarse_pack_ms", "coarse_all_gather_ms", "coarse_merge_ms", "scan_preassigned_ms", "scan_pack_ms",
                 "scan_all_gather_ms", "scan_merge_ms"} if mode.endswith("+coarse") else {"setup_ms", "local_search_ms", "all_gather_ms", "merge_ms"})
        D2, I2 = sh.search_replicated(torch.from_numpy(qall), k)
        ok2 = set(sp) == want and all(v >= 0 for v in sp.values()) and sh._probe is None and torch.equal(I2, I)
        if not ok2:
            print("probe_split:", sp, sh._probe, torch.equal(I2, I), file=sys.stderr)
    # unsharded expectation
    ln, codes = O.encode(x, cent, cb)
    off, lc, li = O.build_lists(ln, codes, np.arange(n), nlist)
    De, Ie = O.search(qs, cent, cb, off, lc, li, nprobe, k)
    ok = ok2 and np.array_equal(I.numpy(), Ie) and np.array_equal(D.numpy().view(np.uint32), De.view(np.uint32))
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, "own"), (2, "replicated"), (3, "own"), (2, "own+coarse"),
                                        (3, "replicated+coarse"), (2, "refine"), (3, "refine"),
                                        (2, "replicated+affine"), (3, "own+affine")])
def test_sharded_equals_unsharded(world, mode):
    from oracle import ivfpq_oracle as O
    O.build()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), mode, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def _emulate_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from abstracts_search_amd.shards import ShardedIndex
    torch.manual_seed(0)
    nlist, d, nq, nprobe, k, E = 64, 8, 5, 6, 4, 4
    cent, q = torch.randn(nlist, d), torch.randn(nq, d)
    seen = {}

    def coarse(qq, npb, lo, hi):
        D, I = (qq @ cent[lo:hi].T).topk(npb, dim=1)
        return (I + lo).to(torch.int32), D

    def pre(qq, kk, cI, cD):
        seen["cI"] = cI.clone()
        return cD[:, :kk].clone(), cI[:, :kk].to(torch.int64)

    def merge(Dg, Ig):
        W, n, kk = Dg.shape
        Dc, Ic = Dg.permute(1, 0, 2).reshape(n, W * kk), Ig.permute(1, 0, 2).reshape(n, W * kk)
        o = torch.argsort(-Dc, dim=1, stable=True)[:, :kk]
        return Dc.gather(1, o), Ic.gather(1, o)

    class Dummy:
        metric_type = 0
    sh = ShardedIndex(Dummy(), shard_coarse=True, local_coarse=coarse, local_search_pre=pre, nlist=nlist, nprobe=nprobe, merge=merge,
                      id_affine=(E, 0, 1), emulate_world=E)
    D, I = sh.search_replicated(q, k)
    full = (q @ cent.T).topk(nprobe, dim=1)[1]
    ok = torch.equal(torch.sort(seen["cI"].long(), dim=1)[0], torch.sort(full, dim=1)[0])       # the real job's probe lists
    ok = ok and I.shape == (nq, k) and bool((I[:, 0] % E == 0).all()) and bool((I[:, 1] == I[:, 0] + 1).all())   # E copies, ids a rank step apart
    split = sh.probe_split(q, k, reps=2)
    ok = ok and {"coarse_slice_ms", "coarse_all_gather_ms", "coarse_merge_ms", "scan_preassigned_ms", "scan_all_gather_ms", "scan_merge_ms"} <= set(split)
    ret[rank] = ok
    dist.destroy_process_group()


def test_emulated_rank_of_a_larger_job_sees_the_real_probe_lists():
    """ShardedIndex(emulate_world=E) on a one-rank group (bench.py --emulate-rank-of E): rank 0's coarse slice merged with the
    absent ranks' true lists gives the global top-nprobe -- the scan the real job's rank would run --, the result exchange
    merges E blocks, and probe_split names every stage."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    p = ctx.Process(target=_emulate_worker, args=(0, 1, _free_port(), ret))
    p.start()
    p.join(120)
    assert p.exitcode == 0 and ret.get(0) is True
