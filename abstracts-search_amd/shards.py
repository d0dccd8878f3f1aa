"""Vector-sharded IVF-PQ search across the GPUs of one node.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).
Rank r holds a shard of every inverted list (e.g. rows i = r mod N) with
*global* int64 ids; coarse centroids and PQ codebook are replicated.  The path
has exactly one exchange step: an all-gather of the per-shard top-k
(k * 12 bytes per query per rank -- latency-bound, not bandwidth-bound),
followed by a k-way merge under the same (score desc, id asc) order the
single-GPU search uses, so the sharded result is bit-identical to the
unsharded one.  With a re-ranking shard index (IndexRefineFlat over the shard's raw
vectors, `id_map` = the shard's global row numbers) the result is the merge of the
shards' exact re-ranked lists: every shard re-ranks its own k * k_factor candidates.

Two entry points:
  search_replicated(q, k)  every rank passes the SAME queries (a front end
                           broadcast them); one all-gather of (D, I).
  search(q_local, k)       every rank brings its OWN batch (data-parallel
                           clients); the batches are all-gathered first, the
                           per-shard top-k all-gathered after, and every rank
                           merges the slice belonging to its own queries.

`local_search` / `merge` are injectable so that the collective plumbing can be
exercised on CPU with the gloo backend (tests/test_shards_gloo.py); the
defaults are the HIP paths and fail loudly without a GPU.
"""
from __future__ import annotations


class ShardedIndex:
    def __init__(self, index, group=None, local_search=None, merge=None, shard_coarse=False,
                 local_coarse=None, local_search_pre=None, nlist=None, nprobe=None, id_map=None):
        import torch.distributed as dist
        assert dist.is_initialized(), "ShardedIndex needs an initialised process group"
        self.index = index
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._local_search = local_search or self._hip_search
        self._merge = merge or self._hip_merge
        self._bufs = {}
        # shard_coarse: the coarse GEMM (2*d*nlist FLOP per query, replicated on
        # every rank otherwise -- at IVF65536 more work than scanning a shard) is
        # split by centroid range; the per-rank top-nprobe lists are all-gathered
        # and merged (same total order => the same probe list as the full search)
        self.shard_coarse = shard_coarse
        self._local_coarse = local_coarse or (lambda q, nprobe, lo, hi: self.index.coarse_slice(q, nprobe, lo, hi))
        self._local_search_pre = local_search_pre or (lambda q, k, cI, cD: self.index.search_preassigned(q, k, cI, cD))
        self._nlist = nlist if nlist is not None else getattr(index, "nlist", None)
        self._nprobe = nprobe
        # id_map: local result ids -> global ids, applied before the exchange step.  For a
        # shard index that numbers its vectors by position -- an IndexRefineFlat over the
        # shard's raw vectors must -- pass the int64 tensor of the shard's global row numbers
        # (round-robin sharding: torch.arange(rank, n, world)) or a callable on id tensors.
        self._id_map = id_map

    # -- default (HIP) implementations ---------------------------------
    def _hip_search(self, q, k):
        return self.index.search(q, k)

    def _hip_merge(self, Dp, Ip):
        from . import faiss
        return faiss.merge_topk(Dp, Ip)

    def _global_ids(self, I):
        if self._id_map is None:
            return I
        if callable(self._id_map):
            return self._id_map(I)
        import torch
        out = self._id_map.to(I.device)[I.clamp_min(0)]
        return torch.where(I < 0, torch.full_like(out, -1), out)

    def _buf(self, name, shape, dtype, device):
        import torch
        key = (name, tuple(shape), dtype, str(device))
        b = self._bufs.get(key)
        if b is None:
            b = torch.empty(shape, dtype=dtype, device=device)
            self._bufs[key] = b
        return b

    # -- same queries on every rank --------------------------------------
    def search_replicated(self, q, k):
        import torch
        import torch.distributed as dist
        Dl, Il = self._search_sharded_coarse(q, k) if self.shard_coarse else self._local_search(q, k)
        Il = self._global_ids(Il)
        nq = Dl.shape[0]
        Dg = self._buf("Dg", (self.world, nq, k), torch.float32, Dl.device)
        Ig = self._buf("Ig", (self.world, nq, k), torch.int64, Il.device)
        dist.all_gather_into_tensor(Dg.view(-1, k), Dl.contiguous(), group=self.group)
        dist.all_gather_into_tensor(Ig.view(-1, k), Il.contiguous(), group=self.group)
        return self._merge(Dg, Ig)

    # -- a different batch on every rank ----------------------------------
    def search(self, q_local, k):
        import torch
        import torch.distributed as dist
        b, d = q_local.shape
        qall = self._buf("qall", (self.world * b, d), q_local.dtype, q_local.device)
        dist.all_gather_into_tensor(qall, q_local.contiguous(), group=self.group)
        if self.shard_coarse:
            Dl, Il = self._search_sharded_coarse(qall, k)
        else:
            Dl, Il = self._local_search(qall, k)                   # this shard, all queries
        Il = self._global_ids(Il)
        Dg = self._buf("Dg", (self.world, self.world * b, k), torch.float32, Dl.device)
        Ig = self._buf("Ig", (self.world, self.world * b, k), torch.int64, Il.device)
        dist.all_gather_into_tensor(Dg.view(-1, k), Dl.contiguous(), group=self.group)   # the exchange step
        dist.all_gather_into_tensor(Ig.view(-1, k), Il.contiguous(), group=self.group)
        lo, hi = self.rank * b, (self.rank + 1) * b
        return self._merge(Dg[:, lo:hi].contiguous(), Ig[:, lo:hi].contiguous())

    def _search_sharded_coarse(self, qall, k):
        import torch
        import torch.distributed as dist
        nprobe = min(int(self._nprobe if self._nprobe is not None else self.index.nprobe), self._nlist)
        per = (self._nlist + self.world - 1) // self.world
        lo, hi = min(self.rank * per, self._nlist), min((self.rank + 1) * per, self._nlist)
        nq = qall.shape[0]
        if hi > lo:
            cI, cD = self._local_coarse(qall, nprobe, lo, hi)
        else:
            cI = torch.full((nq, nprobe), -1, dtype=torch.int32, device=qall.device)
            cD = torch.full((nq, nprobe), -torch.finfo(torch.float32).max, device=qall.device)
        Ig = self._buf("cIg", (self.world, nq, nprobe), torch.int64, qall.device)
        Dg = self._buf("cDg", (self.world, nq, nprobe), torch.float32, qall.device)
        dist.all_gather_into_tensor(Ig.view(-1, nprobe), cI.to(torch.int64).contiguous(), group=self.group)
        dist.all_gather_into_tensor(Dg.view(-1, nprobe), cD.contiguous(), group=self.group)
        mD, mI = self._merge(Dg, Ig)                                # global top-nprobe lists
        return self._local_search_pre(qall, k, mI.to(torch.int32), mD)

    def search_into(self, q_local, k, D, I):
        Dm, Im = self.search(q_local, k)
        D.copy_(Dm)
        I.copy_(Im)


def shard_rows(n: int, rank: int, world: int):
    """Row indices of shard `rank` (round-robin by vector)."""
    return range(rank, n, world)
