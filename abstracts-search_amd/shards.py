"""Vector-sharded IVF-PQ search across the GPUs of one node.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).
Rank r holds a shard of every inverted list (e.g. rows i = r mod N); coarse
centroids and PQ codebook are replicated.  The path has exactly ONE exchange
step: every rank's search writes its (D, I) into the two halves of one send
buffer (``nq*k*12`` bytes: latency-bound, not bandwidth-bound), ONE
``all_gather_into_tensor`` moves the buffers, and ``mi_merge_topk_gathered``
merges straight out of the receive buffer under the same (score desc, id asc)
order the single-GPU search uses -- so the sharded result is bit-identical to
the unsharded one.  Shard-local ids are translated to global ids inside the
merge (``id_affine``: global = local * mul + add + rank * step, the closed form
of a round-robin or contiguous shard numbering); an arbitrary ``id_map`` table
is applied locally before the exchange instead.  With a re-ranking shard index
(IndexRefineFlat over the shard's raw vectors, which numbers its vectors by
position) the result is the merge of the shards' exact re-ranked lists.

Two entry points:
  search_replicated(q, k)  every rank passes the SAME queries (a front end
                           broadcast them): the one all-gather and nothing else.
  search(q_local, k)       every rank brings its OWN batch (data-parallel
                           clients): the batches are all-gathered first and
                           every rank merges the slice of its own queries.

`local_search` / `merge` are injectable so that the collective plumbing can be
exercised on CPU with the gloo backend (tests/test_shards_gloo.py); the
defaults are the HIP paths and fail loudly without a GPU.
"""
from __future__ import annotations


def _pad8(n: int) -> int:
    return (n + 7) // 8 * 8


class ShardedIndex:
    def __init__(self, index, group=None, local_search=None, merge=None, shard_coarse=False,
                 local_coarse=None, local_search_pre=None, nlist=None, nprobe=None, id_map=None,
                 id_affine=None, emulate_world=None):
        import torch.distributed as dist
        assert dist.is_initialized(), "ShardedIndex needs an initialised process group"
        self.index = index
        # the exchange merges under (score desc, id asc): inner-product results only (an L2 index
        # reports ascending distances; shard it by negating outside, or search its shards unsharded)
        assert getattr(index, "metric_type", 0) == 0, "ShardedIndex: METRIC_INNER_PRODUCT indexes only"
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._local_search = local_search          # None: the HIP index writes into the send buffer
        self._merge = merge                        # None: mi_merge_topk_gathered on the receive buffer
        self._bufs = {}
        # shard_coarse: the coarse GEMM (2*d*nlist FLOP per query, replicated on
        # every rank otherwise -- at IVF65536 as much work as scanning a shard) is
        # split by centroid range; the per-rank top-nprobe lists take the same
        # exchange + merge (same total order => the same probe list as the full search)
        self.shard_coarse = shard_coarse
        self._native_coarse = local_coarse is None and local_search_pre is None   # the HIP index: results straight into the send buffers
        self._local_coarse = local_coarse or (lambda q, nprobe, lo, hi: self.index.coarse_slice(q, nprobe, lo, hi))
        self._local_search_pre = local_search_pre or (lambda q, k, cI, cD: self.index.search_preassigned(q, k, cI, cD))
        self._nlist = nlist if nlist is not None else getattr(index, "nlist", None)
        self._nprobe = nprobe
        # id_affine = (mul, add, step): global id = local * mul + add + rank * step, applied by the
        # merge while it loads the gathered ids (round-robin shards numbered by position:
        # (world, 0, 1)).  id_map: an int64 tensor (local id -> global id) or a callable on id
        # tensors, for numberings without a closed form; applied before the exchange.
        assert id_map is None or id_affine is None, "id_map and id_affine are alternatives"
        self._id_map = id_map
        self._id_affine = tuple(int(v) for v in id_affine) if id_affine is not None else (1, 0, 0)
        # emulate_world = E (a one-rank job only; bench.py --emulate-rank-of E): this process plays rank 0 of an E-rank job on
        # the one GPU it has -- its shard of the rows, its 1 / E slice of the coarse quantiser, the real collective at world
        # size 1, an E-block receive buffer and the E-way merge.  The blocks the E - 1 absent ranks would send are put behind
        # the collective: the TRUE coarse lists of the other centroid slices (computed once per query batch, outside the
        # stages, so that the merged probe lists -- hence the scan -- are the real job's), and copies of this shard's top-k
        # for the result exchange.  What it prices: every fixed cost of a rank's step at that size.  What it cannot: xGMI.
        self.emulate = int(emulate_world) if emulate_world else 0
        assert not self.emulate or self.world == 1, "emulate_world: a one-rank job plays rank 0 of the emulated one"
        self._emu_peer = {}
        self.mworld = self.emulate or self.world                     # blocks in a receive buffer = ways of a merge

    # -- diagnostics ---------------------------------------------------------
    _probe = None

    def _mark(self, what):
        """(probe_split only) a time stamp on the issuing stream: a CUDA event, or the host clock on CPU tensors"""
        if self._probe is None:
            return
        import time
        import torch
        if self._probe["cuda"]:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self._probe["marks"].append((what, e))
        else:
            self._probe["marks"].append((what, time.perf_counter()))

    def probe_split(self, q, k, reps=5):
        """Where this rank's step goes: milliseconds per stage of search_replicated(q, k) -- the local search (with
        shard_coarse: the coarse slice, its exchange and merge as stages of their own), the all-gather as THIS rank sees
        it (a rank that finishes its scan early waits here for the slowest one: the number to look at when a scaling
        run disappoints), the merge -- averaged over `reps` calls, by events on the issuing stream.  A diagnostic: one
        call at a time, nothing else in flight."""
        import torch
        cuda = q.is_cuda
        self.search_replicated(q, k)                                # warm: buffers, communicator
        if cuda:
            torch.cuda.synchronize()
        acc = {}
        for _ in range(reps):
            self._probe = {"cuda": cuda, "marks": []}
            try:
                self.search_replicated(q, k)
                if cuda:
                    torch.cuda.synchronize()
                marks = self._probe["marks"]
            finally:
                self._probe = None
            # one exchange, or two with a sharded coarse quantiser (the probe lists first): name the intervals between marks
            n_ex, prev = 0, None
            for what, t in marks:
                if prev is not None:
                    dt = prev[1].elapsed_time(t) if cuda else (t - prev[1]) * 1e3
                    direct = self.shard_coarse and self._native_coarse and self._id_map is None   # slice / scan write the send buffers themselves
                    if what == "begin":
                        n_ex += 1
                        name = ("coarse_slice" if n_ex == 1 else "scan_preassigned") if self.shard_coarse else "setup"
                        if direct:
                            name = "coarse_setup" if n_ex == 1 else "scan_setup"
                    else:
                        name = {"local": "pack" if self.shard_coarse else "local_search", "exchange": "all_gather", "merge": "merge"}[what]
                        if self.shard_coarse:
                            name = ("coarse_" if n_ex == 1 else "scan_") + name
                        if direct and what == "local":
                            name = "coarse_slice" if n_ex == 1 else "scan_preassigned"
                    acc[name + "_ms"] = acc.get(name + "_ms", 0.0) + dt / reps
                prev = (what, t)
        return {k2: round(v, 4) for k2, v in acc.items()}

    # -- helpers -----------------------------------------------------------
    def _buf(self, name, nbytes, device):
        # one buffer set per (CUDA) stream the search is issued on: batches issued on different streams overlap, and
        # the next batch's local search must not write the send buffer the previous batch's all-gather is reading
        import torch
        dev = torch.device(device)
        sid = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0
        key = (name, int(nbytes), str(device), int(sid))
        b = self._bufs.get(key)
        if b is None:
            b = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            self._bufs[key] = b
        return b

    def _global_ids(self, I):
        if self._id_map is None:
            return I
        if callable(self._id_map):
            return self._id_map(I)
        import torch
        out = self._id_map.to(I.device)[I.clamp_min(0)]
        return torch.where(I < 0, torch.full_like(out, -1), out)

    def _hip_search_into(self, q, k, D, I):
        idx = self.index
        if hasattr(idx, "base_index"):                       # IndexRefineFlat: candidates in scratch
            import torch
            kb = int(k * idx.k_factor)
            kb = max(k, kb - kb % k)
            cD = self._buf("candD", q.shape[0] * kb * 4, q.device).view(torch.float32).view(q.shape[0], kb)
            cI = self._buf("candI", q.shape[0] * kb * 8, q.device).view(torch.int64).view(q.shape[0], kb)
            idx.search_into(q, k, D, I, cD, cI)
        else:
            idx.search_into(q, k, D, I)

    # -- the exchange step ---------------------------------------------------
    def _exchange_merge(self, nq, k, device, fill, q_lo=0, nq_out=None, affine=(1, 0, 0), peers=None):
        """fill(Dview, Iview) writes this rank's [nq, k] lists into the send buffer; one
        all-gather; merged (D, I) of the queries [q_lo, q_lo + nq_out)."""
        import torch
        import torch.distributed as dist
        nq_out = nq if nq_out is None else nq_out
        dbytes = _pad8(nq * k * 4)
        blk = dbytes + nq * k * 8
        send = self._buf("send", blk, device)
        recv = self._buf("recv", blk * self.mworld, device)
        Dv = send[:nq * k * 4].view(torch.float32).view(nq, k)
        Iv = send[dbytes:].view(torch.int64).view(nq, k)
        mark = self._mark
        mark("begin")
        fill(Dv, Iv)
        mark("local")
        if self.emulate:
            dist.all_gather_into_tensor(recv[:blk], send, group=self.group)   # the real collective, at the world size there is
            tail = recv[blk:].view(self.emulate - 1, blk)
            if peers is not None:
                tail.copy_(peers)                                      # the absent ranks' blocks (their true coarse lists)
            else:
                tail.copy_(send.unsqueeze(0).expand(self.emulate - 1, blk))   # ... or copies of this rank's (ids differ by the rank step)
        else:
            dist.all_gather_into_tensor(recv, send, group=self.group)      # the path's one exchange step
        mark("exchange")
        if self._merge is None:
            from . import faiss
            out = faiss.merge_topk_gathered(recv, self.mworld, nq, k, blk, affine, q_lo, nq_out)
            mark("merge")
            return out
        # injected merge (CPU tests): the same views, ids translated with torch
        parts = recv.view(self.mworld, blk)
        Dg = torch.stack([parts[p, :nq * k * 4].view(torch.float32).view(nq, k) for p in range(self.mworld)])
        Ig = torch.stack([parts[p, dbytes:].view(torch.int64).view(nq, k) for p in range(self.mworld)])
        mul, add, step = affine
        if (mul, add, step) != (1, 0, 0):
            off = (add + step * torch.arange(self.mworld, dtype=torch.int64, device=Ig.device)).view(-1, 1, 1)
            Ig = torch.where(Ig < 0, Ig, Ig * mul + off)
        out = self._merge(Dg[:, q_lo:q_lo + nq_out].contiguous(), Ig[:, q_lo:q_lo + nq_out].contiguous())
        mark("merge")
        return out

    def _search_all(self, qall, k, q_lo, nq_out):
        nq = qall.shape[0]
        self._mark("start")
        if self.shard_coarse and self._native_coarse and self._id_map is None:
            mD, mI = self._merged_probe_lists(qall)

            def fill(Dv, Iv):                                       # the scan writes the send buffer (no pack copies)
                self.index.search_preassigned(qall, k, mI, mD, Dv, Iv)
        elif self.shard_coarse:
            Dl, Il = self._search_sharded_coarse(qall, k)

            def fill(Dv, Iv):
                Dv.copy_(Dl)
                Iv.copy_(self._global_ids(Il))
        elif self._local_search is None and self._id_map is None:
            def fill(Dv, Iv):                                       # the search writes the send buffer
                self._hip_search_into(qall, k, Dv, Iv)
        else:
            def fill(Dv, Iv):
                if self._local_search is None:
                    self._hip_search_into(qall, k, Dv, Iv)
                    Iv.copy_(self._global_ids(Iv.clone()))
                else:
                    Dl, Il = self._local_search(qall, k)
                    Dv.copy_(Dl)
                    Iv.copy_(self._global_ids(Il))
        return self._exchange_merge(nq, k, qall.device, fill, q_lo, nq_out, self._id_affine)

    # -- same queries on every rank --------------------------------------
    def search_replicated(self, q, k):
        return self._search_all(q.contiguous(), k, 0, q.shape[0])

    # -- a different batch on every rank ----------------------------------
    def search(self, q_local, k):
        import torch
        import torch.distributed as dist
        b, d = q_local.shape
        if self.world == 1:
            return self._search_all(q_local.contiguous(), k, 0, b)
        qall = self._buf("qall", self.world * b * d * q_local.element_size(), q_local.device) \
            .view(q_local.dtype).view(self.world * b, d)
        dist.all_gather_into_tensor(qall, q_local.contiguous(), group=self.group)
        return self._search_all(qall, k, self.rank * b, b)       # this shard, all queries; own slice merged

    def _search_sharded_coarse(self, qall, k):
        mD, mI = self._merged_probe_lists(qall)
        return self._local_search_pre(qall, k, mI, mD)

    def _merged_probe_lists(self, qall):
        """this rank's slice of the coarse quantiser, exchanged and merged: the global top-nprobe lists (f32 scores, int32 list
        numbers) every rank then scans its shard of"""
        import torch
        nprobe = min(int(self._nprobe if self._nprobe is not None else self.index.nprobe), self._nlist)
        per = (self._nlist + self.mworld - 1) // self.mworld
        lo, hi = min(self.rank * per, self._nlist), min((self.rank + 1) * per, self._nlist)
        nq = qall.shape[0]
        peers = None
        if self.emulate:
            # the other slices' lists for THIS query batch, once (outside the stages probe_split times)
            key = (qall.data_ptr(), nq, nprobe)
            peers = self._emu_peer.get(key)
            if peers is None:
                dbytes = _pad8(nq * nprobe * 4)
                blk = dbytes + nq * nprobe * 8
                peers = torch.empty((self.emulate - 1, blk), dtype=torch.uint8, device=qall.device)
                for p in range(1, self.emulate):
                    plo, phi = min(p * per, self._nlist), min((p + 1) * per, self._nlist)
                    pI, pD = self._local_coarse(qall, nprobe, plo, phi)
                    peers[p - 1, :nq * nprobe * 4].view(torch.float32).view(nq, nprobe).copy_(pD)
                    peers[p - 1, dbytes:].view(torch.int64).view(nq, nprobe).copy_(pI)
                self._emu_peer[key] = peers
        def fill(Dv, Iv):
            if hi > lo and self._native_coarse:
                cI, _ = self.index.coarse_slice(qall, nprobe, lo, hi, D_out=Dv)   # the scores straight into the send buffer
            elif hi > lo:
                cI, cD = self._local_coarse(qall, nprobe, lo, hi)
                Dv.copy_(cD)
            else:
                cI = torch.full((nq, nprobe), -1, dtype=torch.int32, device=qall.device)
                Dv.fill_(-torch.finfo(torch.float32).max)
            Iv.copy_(cI)                                            # int32 -> int64

        mD, mI = self._exchange_merge(nq, nprobe, qall.device, fill, peers=peers)   # global top-nprobe lists
        return mD, mI.to(torch.int32)

    def search_into(self, q_local, k, D, I):
        Dm, Im = self.search(q_local, k)
        D.copy_(Dm)
        I.copy_(Im)


class NativeShardedIndex:
    """The same sharded search with the exchange step inside the C ABI (`mi_shards_search`:
    local search -> one ncclAllGather -> merge, all enqueued on the caller's stream by one
    call): what a host without torch.distributed binds, and the lowest-overhead path for a
    Python host too.  The 128-byte RCCL id is created on rank 0 and broadcast through the
    already initialised torch.distributed group (any backend); RCCL itself is the copy torch
    loaded.  `index` is an IndexIVFPQ or an IndexRefine over one (numbered by position)."""

    def __init__(self, index, group=None, id_affine=None, rank=None, world=None):
        import ctypes
        import os
        import torch
        from .faiss import _Lib, _check
        self._lib, self._check = _Lib.get(), _check
        if world is None:
            import torch.distributed as dist
            world = dist.get_world_size(group) if dist.is_initialized() else 1
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world, self.rank, self.index = int(world), int(rank), index
        base = getattr(index, "base_index", index)
        refine = getattr(index, "refine_index", None)
        rccl = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        rccl = rccl.encode() if os.path.exists(rccl) else None
        uid = torch.zeros(128, dtype=torch.uint8)
        if self.rank == 0:
            buf = (ctypes.c_char * 128)()
            _check(self._lib.mi_shards_unique_id(rccl, buf))
            uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        if self.world > 1:
            import torch.distributed as dist
            dev = torch.device("cuda", base.device) if dist.get_backend(group) == "nccl" else torch.device("cpu")
            t = uid.to(dev)
            dist.broadcast(t, 0, group=group)
            uid = t.cpu()
        mul, add, step = (int(v) for v in (id_affine if id_affine is not None else (1, 0, 0)))
        self._h = ctypes.c_void_p()
        kf = int(getattr(index, "k_factor", 1)) if refine is not None else 1
        _check(self._lib.mi_shards_create(base._h, refine._h if refine is not None else None, kf, self.rank, self.world,
                                          bytes(uid.numpy().tobytes()), rccl, mul, add, step, ctypes.byref(self._h)))
        self._base = base

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.mi_shards_destroy(h)
            except Exception:
                pass

    def search_replicated(self, q, k, D=None, I=None):
        import ctypes
        import torch
        nq = q.shape[0]
        if D is None:
            D = torch.empty((nq, k), dtype=torch.float32, device=q.device)
            I = torch.empty((nq, k), dtype=torch.int64, device=q.device)
        self._check(self._lib.mi_shards_search(self._h, nq, ctypes.c_void_p(q.data_ptr()), int(k), int(self._base.nprobe),
                                               ctypes.c_void_p(D.data_ptr()), ctypes.c_void_p(I.data_ptr()),
                                               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return D, I


def shard_rows(n: int, rank: int, world: int):
    """Row indices of shard `rank` (round-robin by vector)."""
    return range(rank, n, world)
