#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

Default workload (config.workload = "cfg4"): BASELINE.json configs[3], the
configuration the metric is quoted on -- 207 M x 1024-d synthetic clustered
corpus, IVF65536,PQ64 (inner product), batch-1024 queries, nprobe 64, k = 10.
The whole 207 M-vector index (15 GB of codes + ids) is built in HBM from the
chunk-wise regenerated corpus (setup, untimed, ~2-3 min).  A "step" is one
IndexIVFPQ.search of one 1024-query batch (coarse quantise + LUT + PQ-code scan
+ top-k) with queries and outputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): the north
star's layout -- rank r holds rows i = r mod N of every inverted list, every
rank searches the same 1024-query batch over its shard, ONE all-gather of the
per-shard top-k over xGMI (the path's one exchange step) and a k-way merge.
The global batch is fixed, so "scaling" is "strong".  `--multi-gpu-mode
replicas` keeps the query-parallel alternative (no collective).

One JSON line on stdout (rank 0).  Besides the contract's fields:
  roofline        PQ-scan kernel, HIP events on the launch stream
  cpu_baseline    the oracle port (or real faiss-cpu when importable) on the host cores
  recall_at_10    of the timed configuration, against exact search over all 207 M rows
  at_recall_095   the operating point with recall@10 >= 0.95 (IVF-PQ + IndexRefineFlat)
  encode          cfg3: stella_en_1.5B_v5 bf16 batch encode (abstracts/s + its roofline)
  reference_oracles  whether real faiss / sentence-transformers could be imported here

`--workload cfg2` (1 M x 1024, IVF4096,PQ64, batch 64) and `--workload encode`
run the round-1 lines on their own.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Steps are issued round-robin on several HIP streams; the runtime maps streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4, one of which torch's own stream takes), and
# two streams sharing a queue serialise.  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2" if os.environ.get("BENCH_REHEARSAL") else "8")   # rehearsal: several processes share one GPU

D_MODEL, PQ_M = 1024, 64
CH = 1 << 20                      # corpus rows per generated chunk


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# ----------------------------------------------------------------------
# timing: K steps between barrier + synchronize, max over ranks; a region shorter than
# 50 ms is inside the noise of a clock ramp, so the K-step block is then repeated and the
# median block is reported (steps / warmup keep their meaning)
# ----------------------------------------------------------------------
class Clock:
    def __init__(self, torch, dist, world, dev):
        self.torch, self.dist, self.world, self.dev = torch, dist, world, dev

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def block(self, step, first, n):
        self.barrier()
        t0 = time.perf_counter()
        for i in range(n):
            step(first + i)
        t_issue = time.perf_counter() - t0
        self.barrier()
        dt = time.perf_counter() - t0
        if self.world > 1:
            tt = self.torch.tensor([dt], device=self.dev, dtype=self.torch.float64)
            self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, t_issue

    def measure(self, step, steps, warmup, min_region_s=0.05, min_blocks=3, max_blocks=15, max_total_s=2.0):
        """W warmup steps, then blocks of exactly K steps (barrier + synchronize on both sides, max over ranks);
        at least `min_blocks` of them, more while a block is shorter than `min_region_s` (inside the noise of a clock
        ramp); the MEDIAN block is reported.  Every rank takes the same decisions: the times are all-reduced."""
        for i in range(warmup):
            step(i)
        times, issue = [], []
        first = warmup
        while True:
            dt, ti = self.block(step, first, steps)
            first += steps
            times.append(dt)
            issue.append(ti)
            if len(times) < min_blocks:
                continue
            if statistics.median(times) >= min_region_s and len(times) % 2 == 1:
                break
            if len(times) >= max_blocks or (sum(times) >= max_total_s and len(times) % 2 == 1):
                break
        return statistics.median(times), times, statistics.median(issue)


def committed_traffic(name, match):
    """HBM-side bytes of a kernel from the committed counter passes (profiles/<name>: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    their own runs, tools/prof_r06.sh) -- only when `match(doc)` says the file is of this configuration AND the kernel's source
    text is the one the pass profiled (tools/kernel_stamp.py): a stale file yields (None, why), never an old number.
    -> (doc or None, source / reason text)"""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import kernel_stamp
        doc = json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception as e:
        return None, f"no counter file profiles/{name} ({type(e).__name__})"
    if not match(doc):
        return None, f"profiles/{name} is of another configuration"
    ok, why = kernel_stamp.fresh(doc)
    if not ok:
        return None, f"profiles/{name} dropped: {why}"
    return doc, doc.get("source", f"profiles/{name}")


def hbm_gb(torch):
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 1e9, total / 1e9


def recall_at_k(I, gt):
    a, e = I.cpu().numpy(), gt.cpu().numpy()
    hits = sum(len(set(x.tolist()) & set(y.tolist())) for x, y in zip(a, e))
    return hits / float(e.size)


# ----------------------------------------------------------------------
# real reference packages, if this box has them (attempt, never skip silently)
# ----------------------------------------------------------------------
def reference_oracles():
    import importlib.util
    out = {}
    for name in ("faiss", "sentence_transformers", "sidecar_search"):
        try:
            spec = importlib.util.find_spec(name)
        except Exception:
            spec = None
        origin = getattr(spec, "origin", None) or ""
        if spec is None:
            out[name] = "absent"
        elif ROOT in os.path.abspath(origin):
            out[name] = "absent (the name resolves to this repository's drop-in alias)"
        else:
            out[name] = "present: " + origin
    return out


def real_faiss():
    ref = reference_oracles()
    if not ref["faiss"].startswith("present"):
        return None
    try:
        import faiss as real
        return real
    except Exception as e:                                            # pragma: no cover
        log(f"import faiss failed: {e!r}")
        return None


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU, rendezvous on
    127.0.0.1 (the container's hostname may not resolve) -- the command line the driver would have typed."""
    import socket
    if not os.environ.get("BENCH_REHEARSAL"):
        import torch
        have = torch.cuda.device_count()
        if have < n:
            log(f"bench.py --gpus {n}: only {have} GPU(s) visible on this node")
            sys.exit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")                 # RCCL / IPC across processes: dmabuf handles only
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("bench.py: WORLD_SIZE unset, launching " + " ".join(cmd[1:8]) + " ...")
    sys.stdout.flush()
    sys.stderr.flush()
    os.execve(sys.executable, cmd, env)


def rank_census(torch, dist, world, dev, ntotal):
    """What the collective library itself sees: the ranks an all-reduce on the job's backend reaches and every rank's
    share of the index (one all-gather) -- so that the line shows N ranks took part, not N copies of rank 0."""
    if not dist.is_initialized():
        return {"backend": None, "ranks": 1, "index_vectors_per_rank": [int(ntotal)]}
    one = torch.ones(1, device=dev, dtype=torch.int64)
    dist.all_reduce(one)
    mine = torch.tensor([int(ntotal)], device=dev, dtype=torch.int64)
    allv = torch.empty(world, device=dev, dtype=torch.int64)
    dist.all_gather_into_tensor(allv, mine)
    return {"backend": dist.get_backend(), "ranks": int(one.item()), "index_vectors_per_rank": [int(v) for v in allv.tolist()]}


# ----------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: 50 (cfg4), 200 (cfg2), 10 (encode)")
    ap.add_argument("--warmup", type=int, default=None, help="default: 5 (cfg4), 20 (cfg2), 2 (encode)")
    ap.add_argument("--workload", choices=["cfg4", "cfg2", "encode", "search", "cfg5"], default="cfg4",
                    help="cfg4 = the headline line (207M, IVF65536,PQ64, batch 1024); cfg2 = 1M, IVF4096,PQ64, batch 64 "
                         "('search' is its old name); encode = cfg3 (stella_en_1.5B_v5 bf16 batch encode); cfg5 = end-to-end encode + search "
                         "over the cfg4 index at query batches 1 / 16 / 256 (latency and throughput curve)")
    ap.add_argument("--corpus", type=int, default=None, help="corpus rows (default: 207000000 cfg4, 1000000 cfg2)")
    ap.add_argument("--nlist", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--nprobe", type=int, default=None)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--train-iters", type=int, default=None)
    ap.add_argument("--settle-ms", type=float, default=100.0,
                    help="untimed steps issued for this long before warmup (clock ramp after setup)")
    ap.add_argument("--refine", type=int, default=0, metavar="K_FACTOR",
                    help="cfg2 only: IVF4096,PQ64,RFlat with this k_factor as the timed configuration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-recall", action="store_true")
    ap.add_argument("--no-refine-point", action="store_true", help="cfg4: skip the recall >= 0.95 operating point")
    ap.add_argument("--no-encode", action="store_true", help="cfg4: skip the encode half of the metric (and the cfg5 curve)")
    ap.add_argument("--no-cfg5", action="store_true", help="cfg4: skip the end-to-end query curve (BASELINE.json configs[4])")
    ap.add_argument("--refine-store", choices=["auto", "f32", "f16", "sq8"], default="auto",
                    help="cfg4 refine stage: f32 = IndexRefineFlat over the raw vectors (faiss ',RFlat'); f16 = "
                         "',Refine(SQfp16)': IEEE-half store, half the HBM and half the bytes per re-ranked candidate; sq8 = "
                         "',Refine(SQ8)': one byte per component with per-dimension ranges -- all 207 M rows (212 GB) beside "
                         "the index on ONE GPU; auto (default) = sq8 at EVERY N: the one store that holds the whole corpus on one "
                         "GPU, so that the points of a 1 -> 8 GPU curve are the same index (the line records the store)")
    ap.add_argument("--encode-batch", type=int, default=128,
                    help="abstracts per encode step (default 128); 0 = the library's own batching: as many abstracts as fit "
                         "32 768 padded tokens per forward pass (~135; +1 % tokens/s: every GEMM fills whole rounds of the CUs)")
    ap.add_argument("--full", action="store_true",
                    help="--workload encode: also push ALL of configs[2]'s 100 000 synthetic abstracts through one "
                         "encode_tokens() call (length sort, token-budget passes, embeddings back on the host) and report the wall rate")
    ap.add_argument("--full-abstracts", type=int, default=100000, help="--full: how many abstracts (configs[2] says 100k)")
    ap.add_argument("--emulate-rank-of", type=int, default=0, metavar="E",
                    help="cfg4 on ONE GPU as rank 0 of an E-GPU job: rows i = 0 (mod E) of the corpus, the 1/E slice of the coarse "
                         "quantiser, both exchanges through the real collective at world size 1 with E-block receive buffers, the "
                         "E-way merges; per-stage ms in `step_split`.  A rehearsal of a rank's fixed costs, never a scaling result")
    ap.add_argument("--no-full", action="store_true",
                    help="N = 1 runs configs[2] at its stated size by default (all 100 000 abstracts through one encode call, ~57 s: "
                         "`full_run` beside the per-batch rate); this skips it")
    ap.add_argument("--encode-steps", type=int, default=24, help="cfg4 line: encode steps (x encode-batch abstracts)")
    ap.add_argument("--encode-streams", type=int, default=1, help="encode steps issued round-robin on this many HIP streams")
    ap.add_argument("--multi-gpu-mode", choices=["shards", "replicas"], default="shards",
                    help="N>1: shards = vector-sharded index + one all-gather of top-k (default, the north star's "
                         "layout); replicas = query-parallel, no collective")
    ap.add_argument("--exchange", choices=["torch", "native"], default="torch",
                    help="N>1 shards: torch = one torch.distributed all_gather_into_tensor + mi_merge_topk_gathered (default); "
                         "native = the whole step inside the C ABI (mi_shards_search: ncclAllGather bound by dlopen)")
    ap.add_argument("--shard-coarse", type=int, default=None,
                    help="N>1 shards: also split the coarse quantiser across ranks by centroid range (one more exchange of "
                         "nq*nprobe*12 bytes per rank, ~50 us; every rank then multiplies 1/N of the 65536 centroids). "
                         "Default: on from 4 ranks (a replicated coarse stage is 0.34 of a 0.65 ms step at N = 8), off below")
    ap.add_argument("--streams", type=int, default=None,
                    help="HIP streams the steps are issued round-robin on (default: cfg4 4 on one GPU and 2 sharded, 4 cfg2)")
    args = ap.parse_args()
    if args.workload == "search":
        args.workload = "cfg2"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)                                        # does not return
    if args.emulate_rank_of:
        assert args.gpus == 1 and args.workload == "cfg4" and args.emulate_rank_of >= 2, "--emulate-rank-of E: one GPU, cfg4, E >= 2"
        os.environ["BENCH_FORCE_SHARDED"] = "1"                       # a process group of one rank (RCCL)
        args.no_refine_point = args.no_recall = args.no_encode = args.no_cfg5 = args.no_cpu_baseline = True
        if args.shard_coarse is None:
            args.shard_coarse = 1 if args.emulate_rank_of >= 4 else 0
    if args.shard_coarse is None:
        args.shard_coarse = 1 if (args.gpus >= 4 and args.workload == "cfg4") else 0
    # stdout carries exactly one JSON line: whatever native libraries print to fd 1 (RCCL's
    # version banner, flushed at exit) goes to stderr instead
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import faulthandler
    import signal
    faulthandler.register(signal.SIGUSR1, all_threads=True)            # a stalled rank says where: kill -USR1 <pid> (tests do on a timeout)

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # BENCH_REHEARSAL=1 (tests/test_bench_multirank_gpu.py): the N ranks share GPU 0 and talk over gloo --
    # the whole N > 1 control flow (rank != 0 paths, dealt shards, the exchange step) on a one-GPU box.
    # Never a measurement: the JSON line carries "rehearsal": true.
    rehearsal = bool(os.environ.get("BENCH_REHEARSAL"))
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("BENCH_FORCE_SHARDED"):
        for name, val in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29511"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(name, val)           # (one process playing a rank: whatever the launcher did not set)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    ctx = dict(np=np, torch=torch, dist=dist, world=world, rank=rank, local_rank=local_rank, dev=dev,
               clock=Clock(torch, dist, world, dev))

    if args.workload == "encode":
        args.steps = 10 if args.steps is None else args.steps
        args.warmup = 2 if args.warmup is None else args.warmup
        out = encode_workload(args, ctx, args.steps, args.warmup, with_cpu=not args.no_cpu_baseline)
    elif args.workload == "cfg2":
        out = cfg2_workload(args, ctx)
    elif args.workload == "cfg5":
        out = cfg5_workload(args, ctx)
    else:
        out = cfg4_workload(args, ctx)
    if rank == 0 and out is not None:
        if rehearsal:
            out["rehearsal"] = True
        print(json.dumps(out), file=json_out, flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


# ======================================================================
# cfg4: 207 M x 1024, IVF65536,PQ64, batch 1024 -- the configuration the metric is quoted on
# ======================================================================
def cfg4_workload(args, ctx):
    np, torch, dist = ctx["np"], ctx["torch"], ctx["dist"]
    world, rank, local_rank, dev, clock = ctx["world"], ctx["rank"], ctx["local_rank"], ctx["dev"], ctx["clock"]
    import abstracts_search_amd.faiss as faiss
    import abstracts_search_amd.synth as synth
    from abstracts_search_amd.shards import ShardedIndex

    N = 207_000_000 if args.corpus is None else args.corpus
    nlist = 65536 if args.nlist is None else args.nlist
    batch = 1024 if args.batch is None else args.batch
    nprobe = 64 if args.nprobe is None else args.nprobe
    steps = 50 if args.steps is None else args.steps
    warmup = 5 if args.warmup is None else args.warmup
    train_iters = 4 if args.train_iters is None else args.train_iters
    d, M, k = D_MODEL, PQ_M, args.k
    force_sharded = bool(os.environ.get("BENCH_FORCE_SHARDED"))       # exercise the N>1 plumbing on one GPU
    use_shards = (world > 1 and args.multi_gpu_mode == "shards") or force_sharded
    replicas = world > 1 and not use_shards
    nsh = world if use_shards else 1                                   # shards the corpus is dealt into
    my = rank if use_shards else 0
    emu = int(getattr(args, "emulate_rank_of", 0) or 0)
    if emu:
        nsh, my = emu, 0                                               # this GPU holds what rank 0 of the E-rank job holds
    assert CH % max(nsh, 8) == 0
    t0 = time.time()

    # ---- train on rank 0 (first 4 M rows: 64 points per centroid), broadcast the tables
    index = faiss.IndexIVFPQ(d, nlist, M, 8, faiss.METRIC_INNER_PRODUCT, device=local_rank)
    index.cp.niter = index.pq.cp.niter = train_iters
    cent = torch.empty((nlist, d), dtype=torch.float32, device=dev)
    cb = torch.empty((M, 256, d // M), dtype=torch.float32, device=dev)
    sq_train_rows = None
    if rank == 0:
        ntrain = min(N, max(4 * CH, 64 * nlist))
        xs = synth.corpus_cuda(ntrain, d, device=local_rank)
        index.train(xs)
        cent.copy_(torch.from_numpy(index.get_centroids()))
        cb.copy_(torch.from_numpy(index.get_codebook()))
        sq_train_rows = xs                                             # also the ScalarQuantizer's training set (',Refine(SQ8)')
        del xs
    if dist.is_initialized() and world > 1:
        dist.broadcast(cent, 0)
        dist.broadcast(cb, 0)
        index.set_centroids(cent)        # bit-identical tables on every rank (k-means uses atomic scatter-adds)
        index.set_codebook(cb)
    del cent, cb
    t_train = time.time() - t0
    log(f"[rank {rank}] train {t_train:.1f}s")

    # ---- queries: perturbed corpus rows from the middle of the corpus (outside the training sample)
    NB = 8
    q0 = (N // 2) // CH * CH
    xq = synth.corpus_cuda(min(CH, N - q0), d, device=local_rank, row0=q0)
    qpool = synth.queries_cuda(xq, NB * batch, seed=4321).view(NB, batch, d)
    del xq
    if replicas:                                                       # every rank its own batches
        g = torch.Generator(device=dev).manual_seed(977 + rank)
        qpool = qpool[torch.randperm(NB, generator=g, device=dev)].contiguous()
    q_gt = qpool[0].contiguous()                                       # recall is measured on this batch
    q_sel = qpool[1].contiguous()                                      # held out: the recall >= 0.95 point's (nprobe, k_factor) is chosen on this one
    q_gt2 = torch.cat([q_gt, q_sel])                                   # exact top-k is kept for both

    # ---- the refine stage's raw vectors (recall >= 0.95 operating point, IndexRefineFlat): the
    # shard's own rows when they fit beside the index, else the 1/8 sub-shard a GPU of the
    # 8-GPU job would hold (rows i = rank mod 8)
    per_rank = (N + nsh - 1) // nsh
    torch.cuda.empty_cache()
    hbm_free = torch.cuda.mem_get_info()[0] + (sq_train_rows.numel() * 4 if sq_train_rows is not None else 0)   # the training sample is freed below
    if dist.is_initialized() and world > 1:                           # every rank must take the same layout decision
        t = torch.tensor([float(hbm_free)], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        hbm_free = float(t.item())
    want_refine = not args.no_refine_point and not replicas
    # what this rank's HBM has left for the refine store: the index itself is 152 B per vector (append log 80 + scan image
    # 72), plus ~25 GB of corpus chunks, ground-truth store and search workspaces (measured: 277 of 309 GB in use at N = 1).
    # From the memory that is FREE now (another tenant of the GPU shrinks the store's choice, it does not break the run:
    # without room for the whole shard the point falls back to the 1/8 sub-shard layout)
    room = hbm_free - per_rank * 152 - 25e9
    store = args.refine_store
    if store == "auto":                                                # the same index at every N (a curve compares like with like)
        store = "sq8"
    relem = {"f32": 4, "f16": 2, "sq8": 1}[store]
    refine_own = want_refine and per_rank * d * relem <= room
    if os.environ.get("BENCH_FORCE_SUBSHARD"):                         # rehearsals: the sub-shard layout on a small corpus
        refine_own = False
    sub_mod = 8 if (want_refine and not refine_own) else 0
    assert not sub_mod or (8 % nsh == 0), "the 1/8 sub-shard needs N in {1, 2, 4, 8}"
    flat_r = sub = None
    if want_refine:
        if store == "sq8":
            flat_r = faiss.IndexScalarQuantizer(d, faiss.ScalarQuantizer.QT_8bit, faiss.METRIC_INNER_PRODUCT, device=local_rank)
            # ScalarQuantizer.train on the index's own training sample (rank 0), the ranges broadcast like the other tables
            tr = torch.empty(2 * d, dtype=torch.float32, device=dev)
            if rank == 0:
                flat_r.train(sq_train_rows)
                tr.copy_(torch.from_numpy(flat_r.sq.trained))
            if dist.is_initialized() and world > 1:
                dist.broadcast(tr, 0)
                flat_r.sq.trained = tr.cpu().numpy()
            del tr
        elif store == "f16":
            flat_r = faiss.IndexScalarQuantizer(d, faiss.ScalarQuantizer.QT_fp16, faiss.METRIC_INNER_PRODUCT, device=local_rank)
        else:
            flat_r = faiss.IndexFlatIP(d, device=local_rank)
        n_r = per_rank if refine_own else (N + 7) // 8
        flat_r.reserve(n_r + 1)
        if sub_mod:
            sub = faiss.IndexIVFPQ(d, nlist, M, 8, faiss.METRIC_INNER_PRODUCT, device=local_rank)
            sub.set_centroids(torch.from_numpy(index.get_centroids()).to(dev))
            sub.set_codebook(torch.from_numpy(index.get_codebook()).to(dev))
            sub.reserve(n_r + 1)
    del sq_train_rows
    index.reserve(per_rank + 1)

    # ---- build: regenerate the corpus chunk by chunk; add this rank's rows; exact top-k of the
    # recall queries over the same rows (running merge) -- every chunk is generated once
    want_gt = not args.no_recall
    flat_gt = faiss.IndexFlatIP(d, device=local_rank)
    flat_gt.reserve(CH)
    neg = -torch.finfo(torch.float32).max

    def empty_gt():
        return (torch.full((2 * batch, k), neg, dtype=torch.float32, device=dev),
                torch.full((2 * batch, k), -1, dtype=torch.int64, device=dev))

    def fold(run, rows, gid_of):
        flat_gt.reset()
        flat_gt.add(rows)
        Dg, Ig = flat_gt.search(q_gt2, k)
        Ig = torch.where(Ig < 0, Ig, gid_of(Ig))
        return faiss.merge_topk(torch.stack([run[0], Dg]), torch.stack([run[1], Ig]))

    gt, gt_sub = empty_gt(), empty_gt()
    t1 = time.time()
    for c0 in range(0, N, CH):
        m = min(CH, N - c0)
        x = synth.corpus_cuda(m, d, device=local_rank, row0=c0)
        mine = x if nsh == 1 else x[my::nsh].contiguous()
        index.add(mine)                                                # local position p <-> global row p * nsh + my
        if want_gt:
            gt = fold(gt, mine, lambda I, c0=c0: I * nsh + (c0 + my))
        if refine_own:
            flat_r.add(mine)
        elif sub_mod:
            srows = x[rank % 8::8].contiguous()                        # subset of this rank's shard (8 % nsh == 0)
            sub.add(srows)
            flat_r.add(srows)
            if want_gt:
                gt_sub = fold(gt_sub, srows, lambda I, c0=c0: I + c0 // 8)   # positions in `sub` (what IndexRefineFlat returns)
        del x, mine
        if rank == 0 and (c0 // CH) % 32 == 31:
            log(f"  built {c0 + m} rows, {time.time() - t1:.0f}s")
    torch.cuda.synchronize()
    t_build = time.time() - t1
    del flat_gt
    torch.cuda.empty_cache()                                           # the chunk buffers: the scan image needs the room at N = 1
    if want_gt and use_shards and world > 1:                           # exact top-k over all shards
        Dall = torch.empty((world, 2 * batch, k), dtype=torch.float32, device=dev)
        Iall = torch.empty((world, 2 * batch, k), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(Dall.view(-1, k), gt[0].contiguous())
        dist.all_gather_into_tensor(Iall.view(-1, k), gt[1].contiguous())
        gt = faiss.merge_topk(Dall, Iall)
    index.nprobe = nprobe
    t2 = time.time()
    index.search(q_gt[:8].contiguous(), k)                             # builds the scan image of the lists
    torch.cuda.synchronize()
    t_image = time.time() - t2
    index.seal()                                                       # filled, searched from here on: the append log (80 B per vector) goes
    used, total = hbm_gb(torch)
    log(f"[rank {rank}] ntotal={index.ntotal} add {t_build:.1f}s ({index.ntotal / t_build / 1e6:.2f} M vec/s incl. "
        f"generation + ground truth), scan image {t_image:.2f}s, HBM in use {used:.1f} of {total:.0f} GB")

    # ---- the timed configuration
    sharded = None
    if use_shards and args.exchange == "native" and not args.shard_coarse:
        from abstracts_search_amd.shards import NativeShardedIndex
        sharded = NativeShardedIndex(index, id_affine=(nsh, 0, 1))
    elif use_shards:
        sharded = ShardedIndex(index, shard_coarse=bool(args.shard_coarse), id_affine=(nsh, 0, 1), emulate_world=emu or None)
    # batches are independent: 2 streams, also on the sharded path (the exchange of one batch and the coarse stage of
    # the next overlap the other's scan; ShardedIndex keeps a buffer set per stream, torch.distributed orders the
    # collectives by issue order on every rank).  BENCH_SHARD_STREAMS=1 issues the sharded steps on one stream.
    # (round 6: 4 on one GPU -- with the exact list pruning a step is four short stages of different bounds (MFMA GEMM, HBM-bound
    # chains, LDS-bound scan): 2 / 3 / 4 streams 0.472 / 0.452 / 0.447 ms, profiles/r06_exact_list_pruning_ab.txt)
    S = max(1, (4 if not use_shards else 2) if args.streams is None else args.streams)
    if sharded is not None:
        S = max(1, int(os.environ.get("BENCH_SHARD_STREAMS", "0")) or S)
        # (the native exchange keeps a buffer set per stream too -- mi_shards::Bufs -- and RCCL orders the collectives of one
        # communicator by issue order: two batches in flight like the torch path)
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else [torch.cuda.current_stream(dev)]
    sptr = [int(s_.cuda_stream) for s_ in streams]
    Ds = [torch.empty((batch, k), dtype=torch.float32, device=dev) for _ in range(S)]
    Is = [torch.empty((batch, k), dtype=torch.int64, device=dev) for _ in range(S)]
    my_q = [qpool[b] for b in range(NB)]

    def step(b):
        if sharded is None:
            j = b % S
            index.search_into(my_q[b % NB], k, Ds[j], Is[j], None, sptr[j])
        elif S > 1:
            with torch.cuda.stream(streams[b % S]):
                sharded.search_replicated(my_q[b % NB], k)
        else:
            sharded.search_replicated(my_q[b % NB], k)

    def settle(fn, ms):
        for b in range(max(2, 2 * S)):
            fn(b)
        torch.cuda.synchronize()
        t_s, b = time.perf_counter(), 0
        while time.perf_counter() - t_s < ms * 1e-3:
            for _ in range(4):
                fn(b)
                b += 1
            torch.cuda.synchronize()

    settle(step, args.settle_ms)
    torch.cuda.synchronize()
    index.prune_stats(reset=True)
    dt, blocks, t_issue = clock.measure(step, steps, warmup)
    qps = steps * batch * (world if replicas else 1) / dt
    torch.cuda.synchronize()
    pst = index.prune_stats(reset=True)

    # ---- the same loop with the exact list pruning off (MI_SCAN_PRUNE=0): every probed list scanned, as faiss's
    # IndexIVFPQ.search does -- the step the rounds before 6 timed, and the launch the scan kernel's roofline is quoted on
    pruning, exhaustive = None, None
    pruned_here = pst["queries"] > 0
    if world > 1 and dist.is_initialized():                  # (every rank takes the same legs: the steps below hold collectives)
        t = torch.tensor([1 if pruned_here else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        pruned_here = bool(t.item())
    if pruned_here and not os.environ.get("BENCH_NO_EXHAUSTIVE"):   # (BENCH_NO_EXHAUSTIVE: kernel traces of the pruned step alone)
        g_all, g_2 = pst["groups_all_probes"], pst["groups_second_phase"]
        early = batch >= 512 and nprobe <= 64 and os.environ.get("MI_SCAN_PRUNE_MODE", "0") != "1"
        pruning = {"exact": True, "form": "early stop inside the scan kernel (one workgroup per query)" if early else "two scan launches",
                   "queries": pst["queries"], "groups_all_probes_per_query": round(g_all / pst["queries"], 1),
                   ("groups_scanned_per_query" if early else "groups_second_launch_per_query"): round(g_2 / pst["queries"], 2),
                   ("scanned_fraction" if early else "second_launch_fraction"): round(g_2 / max(1, g_all), 4),
                   "what": "by-residual inner product: a code scores <q, centroid> + a chain of 64 table entries, so <q, centroid> + the same "
                           "chain over the tables' row maxima bounds every code of a list (rounded addition is monotone: no slack term); a list "
                           "whose bound is below a score k found codes already reach provably holds no result.  The lists of a query come in "
                           "descending coarse order: a wave stops at the first such list (batches >= 512, nprobe <= 64), or -- caller-ordered "
                           "lists, smaller batches -- a first launch scans the best lists and a second one only the lists that can still "
                           "matter.  (D, I) are the exhaustive scan's bits (tests: test_exact_list_pruning_is_bit_identical; the whole index "
                           "suite with the pruning forced on; parity_vs_oracle below runs on the pruned path).  How much goes depends on "
                           "the data: this corpus is SURVEY 8(d)'s clustered one"}
        os.environ["MI_SCAN_PRUNE"] = "0"
        faiss.reload_env()
        settle(step, args.settle_ms)
        dt_x, blocks_x, _ = clock.measure(step, steps, warmup)
        torch.cuda.synchronize()
        exhaustive = {"queries_per_s": round(steps * batch * (world if replicas else 1) / dt_x, 1), "ms_per_step": round(dt_x / steps * 1e3, 5),
                      "what": "MI_SCAN_PRUNE=0: the same K steps scanning all nprobe lists of every query"}
        del os.environ["MI_SCAN_PRUNE"]
        faiss.reload_env()

    # ---- the reference's own call shape, reported beside the metric and never as `value`: numpy queries in, numpy (D, I)
    # out through IndexIVFPQ.search, as a faiss caller writes it (the timed loop above hands over device tensors and keeps
    # the results in HBM).  The same K steps on ONE stream, every call synchronous: H2D of batch x 4 KiB from pageable host
    # memory, the search, D2H of batch x k x 12 B.
    host_io = None
    if sharded is None and rank == 0:
        q_np = [qb.cpu().numpy() for qb in my_q[:4]]
        for b in range(2):
            index.search(q_np[b % 4], k)
        torch.cuda.synchronize()
        t_h = time.perf_counter()
        for b in range(steps):
            Dh, Ih = index.search(q_np[b % 4], k)
        dt_h = time.perf_counter() - t_h
        assert isinstance(Dh, np.ndarray) and Ih.shape == (batch, k)
        host_io = {"queries_per_s": round(steps * batch / dt_h, 1), "ms_per_step": round(dt_h / steps * 1e3, 5), "steps": steps,
                   "what": "index.search(numpy [%d, 1024] f32, %d) -> numpy (D, I), back to back, one call in flight: the call the "
                           "reference makes (faiss Index.search); PCIe both ways and the host's synchronisation inside every step -- "
                           "a reported figure, never `value`" % (batch, k)}

    # recall of the timed configuration against the exact search
    recall = None
    if want_gt and not replicas:
        if sharded is None:
            _, Ia = index.search(q_gt, k)
        else:
            _, Ia = sharded.search_replicated(q_gt, k)
        recall = recall_at_k(Ia, gt[1][:batch])

    # ---- roofline of the dominant kernel (PQ-code scan): the library re-launches the scan of the last step back to back between
    # two HIP events recorded on the launch stream.  With the exact list pruning the timed step's launch reads what its waves
    # reach (device-counted); the exhaustive launch -- the kernel at the size SURVEY 8(d) prices -- is profiled beside it.
    torch.cuda.synchronize()
    reps_p = max(3, min(steps, 20))
    prof_p = prof_x = None
    if pruned_here:
        index.search_into(my_q[0], k, Ds[0], Is[0], None, sptr[0])
        prof_p = index.profile_scan(reps_p, sptr[0])
        torch.cuda.synchronize()
    if not pruned_here or exhaustive is not None:
        if pruned_here:
            os.environ["MI_SCAN_PRUNE"] = "0"
            faiss.reload_env()
        index.search_into(my_q[0], k, Ds[0], Is[0], None, sptr[0])
        prof_x = index.profile_scan(reps_p, sptr[0])
        torch.cuda.synchronize()
        if pruned_here:
            del os.environ["MI_SCAN_PRUNE"]
            faiss.reload_env()
    cfg_key = (N, nlist, batch, nprobe, k, nsh)

    def scan_roofline(prof, pmc_file, pruned_launch):
        ms, nbytes = prof["scan_ms_avg"], prof["scan_bytes"]
        ach = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        pmc, src = committed_traffic(pmc_file, lambda d_: cfg_key == tuple(d_["config"]))
        out = {"kernel": "scan_kernel<64,8,false>", "bound": "hbm", "achieved": round(ach, 1), "peak": 8000.0,
               "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": int(pmc["corrected_bytes_per_launch"]) if pmc else None,
               "traffic_source": src, "bytes_per_launch": int(nbytes), "avg_launch_ms": round(ms, 5),
               # since round 6 the kernel reads the id of a code only if the code survives its workgroup's selection: what it
               # MOVES is the code bytes -- below the algorithmic figure, which `traffic` (PMC) shows
               "bytes_moved_per_launch": int(nbytes * 64 // 72),
               "frac_of_bytes_moved": round(nbytes * 64 / 72 / (ms * 1e-3) / 1e9 / 8000.0, 4) if ms > 0 else None}
        if pruned_launch:
            out["launch"] = ("the timed step's scan launch: exact list pruning on, one workgroup per query walks the probed lists in coarse order "
                             "and every wave stops at the first list that provably holds no result (`pruning`)")
            out["algorithmic_bytes"] = ("the 64-code groups the waves reached (device-counted) x 64 codes x (64 B code + 8 B id): SURVEY 8(d)'s per-code "
                                        "figure on the codes this launch reads; PMC traffic adds the 64 KiB look-up table every workgroup stages")
            out["note"] = ("a workgroup's table staging and selection tail are spread over ~200 groups where a workgroup of the exhaustive launch "
                           "has ~500: the rate on this launch is lower than on `exhaustive_launch`, the step 5-6 x shorter")
        else:
            out["algorithmic_bytes"] = ("codes of the probed lists x (64 B code + 8 B id), device-counted: SURVEY 8(d)'s per-code figure (what "
                                        "the reference's scan reads)")
            out["bytes_moved_note"] = ("64 B per code (ids are fetched for the <= 3 k survivors of a workgroup only): the kernel is bound by its "
                                       "LDS gather (64 table reads per code, random banks) at about this rate")
        return out

    if prof_p is not None:
        roofline = scan_roofline(prof_p, "r06_cfg4_pruned_scan_pmc.json", True)
        if prof_x is not None:
            roofline["exhaustive_launch"] = dict(scan_roofline(prof_x, "r06_cfg4_scan_pmc.json", False),
                                                 launch="MI_SCAN_PRUNE=0: all nprobe lists of every query (`exhaustive_scan`) -- the kernel at the size "
                                                        "SURVEY 8(d) prices, the launch rounds 1-5 quoted")
    else:
        roofline = scan_roofline(prof_x, "r06_cfg4_scan_pmc.json", False)
    # (what the per-rank split and the scaling model below are written in: the exhaustive launch when it was measured)
    prof_m = prof_x if prof_x is not None else prof_p
    scan_ms, scan_bytes = prof_m["scan_ms_avg"], prof_m["scan_bytes"]

    # ---- recall >= 0.95 operating point: IVF-PQ proposes k * k_factor candidates, exact re-ranking
    at095 = None
    if want_refine:
        at095 = refine_point(args, ctx, faiss, ShardedIndex, index, sub, flat_r, refine_own, nsh, my_q, q_gt, q_sel,
                             gt if refine_own else gt_sub, batch, k, steps, warmup, settle, store)

    census = rank_census(torch, dist, world, dev, index.ntotal)

    # ---- N > 1: where every rank's step goes, measured, beside what the single-GPU numbers predict -- so that a scaling
    # run that disappoints can be read from its one line (a slow rank shows as the OTHER ranks' all-gather time)
    split = None
    if sharded is not None and hasattr(sharded, "probe_split"):
        mine_split = dict(sharded.probe_split(my_q[0], k, reps=5), rank=rank, index_vectors=index.ntotal,
                          scan_bytes=int(scan_bytes), scan_ms=round(scan_ms, 4))
        gathered = [None] * world
        if dist.is_initialized() and world > 1:
            dist.all_gather_object(gathered, mine_split)
        else:
            gathered = [mine_split]
        coarse_pred = 0.23 / ((emu or world) if args.shard_coarse else 1)       # ms: the measured N = 1 coarse stage (f16 slab GEMM + selection) at batch 1024
        split = {"per_rank": gathered,
                 "predicted_ms": {"scan": round(scan_bytes / (0.72 * 8e12) * 1e3, 4), "coarse": round(coarse_pred, 4), "all_gather": 0.05, "merge": 0.02,
                                  "model": "scan = this rank's scanned bytes / (0.72 x 8 TB/s: the fraction measured at N = 1); coarse = 0.23 ms at N = 1 "
                                           "(/ N with --shard-coarse); all-gather of nq*k*12 B per rank: latency-bound, ~50 us over xGMI"},
                 "note": "one call at a time (events on the issuing stream); the timed loop overlaps two batches, so ms_per_step is below the sum"}

    # ---- CPU baseline + parity spot check (rank 0, N = 1): the oracle on the same index / queries
    cpu, parity = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, parity = cpu_baseline_ivfpq(index, q_gt, nprobe, k, np, torch, "cfg4")

    out = None
    if rank == 0:
        par = "1 GPU" if world == 1 else (
            f"vector-sharded x{world} (rows i mod {world}), same {batch}-query batch on every rank, one all-gather of "
            f"per-shard top-k + merge" if use_shards else f"query-parallel replicas x{world} (no collective)")
        out = {
            "metric": "queries/sec, IVF-PQ search (IVF%d,PQ64, %dx1024-d, batch %d, nprobe %d, k %d)" % (nlist, N, batch, nprobe, k),
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(dt / steps * 1e3, 5), "higher_is_better": True,
            "scaling": "weak" if replicas else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cfg4: %dx1024 clustered synthetic corpus, IVF%d,PQ64, batch-%d queries (BASELINE.json configs[3])"
                                   % (N, nlist, batch),
                       "corpus": N, "nlist": nlist, "M": 64, "nprobe": nprobe, "k": k, "global_batch": batch * (world if replicas else 1),
                       "parallelism": par, "shard_coarse": bool(args.shard_coarse) if use_shards else None,
                       "exchange": (args.exchange if use_shards else None),
                       "launch": "eager", "streams": S, "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                       "host_issue_ms_per_step": round(t_issue / steps * 1e3, 5),
                       "timed_blocks": len(blocks), "block_ms": [round(b * 1e3, 3) for b in blocks],
                       "timing": "median of the K-step blocks" if len(blocks) > 1 else "one K-step block",
                       "setup_s": {"train": round(t_train, 1), "generate+add+ground_truth": round(t_build, 1),
                                   "scan_image": round(t_image, 2)},
                       "index_vectors_this_rank": index.ntotal, "index_vectors_per_rank": census["index_vectors_per_rank"],
                       "collective_backend": census["backend"], "rccl_ranks": census["ranks"],
                       "rccl_ranks_note": "ranks one all-reduce on the job's process group reached (backend nccl = RCCL; 1 = no collective, single GPU)",
                       "hbm_in_use_gb": round(used, 1),
                       "hbm_note": "after IndexIVFPQ.seal(): the append log the lists were built from is freed (an export / add rebuilds it from the scan image)"},
            "recall_at_10": None if recall is None else round(recall, 4),
            "recall_note": "against exact inner-product search over all %d rows, %d queries" % (N, batch),
            "roofline": roofline, "pruning": pruning, "exhaustive_scan": exhaustive, "host_io": host_io, "cpu_baseline": cpu, "parity_vs_oracle": parity,
            "at_recall_095": at095, "step_split": split, "reference_oracles": reference_oracles(),
        }
        if emu:
            out["rehearsal"] = True
            out["emulation"] = {"rank_of": emu, "index_vectors": index.ntotal, "coarse_slice_centroids": (nlist + emu - 1) // emu,
                                "what": "ONE GPU playing rank 0 of a %d-GPU job: its shard (rows i = 0 mod %d), its slice of the coarse quantiser, the "
                                        "collective at world size 1 with %d-block receive buffers (the absent ranks' blocks are put behind it: their "
                                        "true coarse lists, copies of this shard's top-k), the %d-way merges.  `value` is what ONE such rank sustains, "
                                        "i.e. the job's rate if every rank kept up and xGMI cost nothing -- a rehearsal of a rank's fixed costs, "
                                        "NOT a scaling measurement" % (emu, emu, emu, emu)}
    # ---- BASELINE.json configs[4] in the same line: encode + search at query batches 1 / 16 / 256 over THIS index (before it
    # is freed); the model is the one the encode half below times
    pack = None
    if not args.no_encode:
        do_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
        host_w = {} if do_cpu else None
        model, mcfg = stella_random_model(args, ctx, keep=host_w)
        pack = (model, mcfg, host_w)
        if not args.no_cfg5 and not replicas:
            c5 = cfg5_curve(args, ctx, model, mcfg, host_w, index, sharded, nprobe, k, 20, 3, do_cpu)
            if out is not None:
                out["cfg5"] = dict(c5, workload="cfg5: end-to-end encode + search over the cfg4 index above, query batches 1 / 16 / 256 "
                                                "(BASELINE.json configs[4]); latency of one call, throughput of back-to-back calls")
    # ---- the other half of the metric: embed abstracts/sec (cfg3), in the same line
    del sharded, index, sub, flat_r
    torch.cuda.empty_cache()
    if not args.no_encode:
        enc = encode_workload(args, ctx, args.encode_steps, 2, with_cpu=not args.no_cpu_baseline, pack=pack)
        if out is not None and enc is not None:
            out["encode"] = {"abstracts_per_s": enc["value"], "tokens_per_s": enc["config"]["tokens_per_sec"],
                             "ms_per_step": enc["ms_per_step"], "steps": enc["steps"], "batch": enc["config"]["batch"],
                             "sample": enc["config"]["sample"], "full_run": enc["full_run"], "dtype": enc["dtype"], "data": enc["data"],
                             "roofline": enc["roofline"], "cpu_baseline": enc["cpu_baseline"],
                             "parity_vs_oracle": enc["parity_vs_oracle"]}
    return out


def refine_point(args, ctx, faiss, ShardedIndex, index, sub, flat_r, refine_own, nsh, my_q, q_gt, q_sel, gt, batch, k,
                 steps, warmup, settle, store_name):
    """Smallest-cost (nprobe, k_factor_rf) from a short ascending list that reaches recall@10 >= 0.95 ON A HELD-OUT
    QUERY BATCH (q_sel; its exact top-k are rows [batch, 2 batch) of gt), timed with the same loop; the recall the line
    reports is then measured on q_gt, which took no part in the choice.  refine_own: the job's real layout (every rank
    re-ranks its own shard, one exchange of the exact lists); otherwise the 1/8 sub-shard one GPU of the 8-GPU job holds."""
    np, torch, dist, world, rank, dev, clock = ctx["np"], ctx["torch"], ctx["dist"], ctx["world"], ctx["rank"], ctx["dev"], ctx["clock"]
    base = index if refine_own else sub
    args_nprobe = index.nprobe                                         # restored below
    ref = faiss.IndexRefine(base, flat_r)
    torch.cuda.synchronize()
    ref.release_workspaces()                                           # the main line's scratch sets: this loop brings its own
    qt = getattr(flat_r, "qtype", None)
    relem = 1 if qt == faiss.ScalarQuantizer.QT_8bit else 2 if qt == faiss.ScalarQuantizer.QT_fp16 else 4
    store = {1: "8-bit (SQ8, per-dimension ranges)", 2: "IEEE-half (SQfp16)", 4: "raw f32"}[relem]
    suffix = {1: "Refine(SQ8)", 2: "Refine(SQfp16)", 4: "RFlat"}[relem]
    sharded = ShardedIndex(ref, id_affine=(nsh, 0, 1)) if (refine_own and nsh > 1) else None
    # Two knobs, explored in ascending cost: the length of the candidate list (k_factor_rf: what PQ64's ranking of the ~N/16384
    # near-identical cluster mates needs) and, only when a longer list stops helping, the number of probes (list coverage).  On
    # the 207 M configuration nprobe 8 already holds the neighbours (16 / 32 / 64 probes change no digit at a given k_factor).
    # (steps of ~8 % from 320 up: the re-rank streams k_factor_rf KiB per query, so the grid's resolution there is the step's --
    # round 5's 400 -> 512 jump chose 512 where 480 already holds the recall)
    kfs = [f for f in (64, 72, 80, 100, 128, 160, 200, 256, 320, 352, 384, 416, 448, 480, 512, 560, 640, 720, 800) if k * f <= 8192]
    if os.environ.get("BENCH_REFINE_KFS"):                            # counter passes: the timed shape only (tools/prof_r06.sh)
        kfs = [int(v) for v in os.environ["BENCH_REFINE_KFS"].split(",")]
    best, curve, i_kf, done = None, [], 0, False

    def recall_on(qs, rows):
        _, Ia = (sharded.search_replicated(qs, k) if sharded is not None else ref.search(qs, k))
        return recall_at_k(Ia, gt[1][rows])

    def evaluate(nprobe, kf):
        base.nprobe, ref.k_factor = nprobe, kf
        r = recall_on(q_sel, slice(batch, 2 * batch))
        if world > 1 and not refine_own:                               # sub-shards differ per rank: agree on the worst
            t = torch.tensor([r], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            r = float(t.item())
        log(f"  refine point nprobe={nprobe} k_factor_rf={kf}: recall@10 {r:.4f}")
        curve.append([nprobe, kf, round(r, 4)])
        return r

    nprobes = [int(v) for v in os.environ.get("BENCH_REFINE_NPROBES", "8,16,32,64,128,256").split(",")]
    for nprobe in nprobes:
        if nprobe > base.nlist:
            break
        prev = None
        while True:
            kf = kfs[min(i_kf, len(kfs) - 1)]
            r = evaluate(nprobe, kf)
            best = (nprobe, kf, r)
            if r >= 0.95:
                done = True
                break
            if (prev is not None and r - prev < 0.002) or i_kf >= len(kfs) - 1:
                break                                                  # a longer list no longer helps at this many probes
            prev = r
            i_kf += 1
        if done:
            # bisect between the last k_factor that missed the recall and the first that held it, to a resolution of 16 (the
            # re-rank streams k_factor KiB per query: a 3 % step is 3 % of the stage), still on the held-out batch only
            lo_kf = kfs[i_kf - 1] if i_kf > 0 and len(curve) > 1 and curve[-2][0] == nprobe and curve[-2][2] < 0.95 else None
            hi_kf, hi_r = kf, r
            while lo_kf is not None and hi_kf - lo_kf > 16 and not os.environ.get("BENCH_REFINE_KFS"):
                mid = (lo_kf + hi_kf) // 2 // 8 * 8
                if mid <= lo_kf or mid >= hi_kf:
                    break
                rm = evaluate(nprobe, mid)
                if rm >= 0.95:
                    hi_kf, hi_r = mid, rm
                else:
                    lo_kf = mid
            best = (nprobe, hi_kf, hi_r)
            break
    nprobe, kf, r_sel = best
    base.nprobe, ref.k_factor = nprobe, kf
    r = recall_on(q_gt, slice(0, batch))                               # the reported recall: a batch the choice never saw
    if world > 1 and not refine_own:
        t = torch.tensor([r], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        r = float(t.item())
    log(f"  refine point chosen on the held-out batch: nprobe={nprobe} k_factor_rf={kf} (recall {r_sel:.4f} there); reported batch: {r:.4f}")
    kb = k * kf
    # batches are independent: the unsharded loop issues them round-robin on 4 streams (the library keeps a workspace set per
    # stream) -- the HBM-bound re-rank and scan of some batches beside the MFMA- and latency-bound coarse stage and set
    # selection of others.  Measured on the 207 M index: 1 stream 2.08 ms per batch, 2: 1.85, 3: 1.79, 4: 1.74
    # (profiles/r04_refine_streams.txt); the two stages of consecutive batches on two streams linked by events (a stage
    # pipeline) measured 2.05 -- no overlap at all on this runtime, although spin kernels linked the same way do overlap
    # (tools/micro/stream_event_overlap.py).
    S2 = max(1, int(os.environ.get("BENCH_REFINE_STREAMS", "0")) or (4 if args.streams is None else args.streams))
    if sharded is not None:
        S2 = max(1, int(os.environ.get("BENCH_SHARD_STREAMS", "0")) or S2)
    rstreams = [torch.cuda.Stream(device=dev) for _ in range(S2)] if S2 > 1 else [torch.cuda.current_stream(dev)]
    rptr = [int(s_.cuda_stream) for s_ in rstreams]
    D = [torch.empty((batch, k), dtype=torch.float32, device=dev) for _ in range(S2)]
    I = [torch.empty((batch, k), dtype=torch.int64, device=dev) for _ in range(S2)]
    cD = [torch.empty((batch, kb), dtype=torch.float32, device=dev) for _ in range(S2)]
    cI = [torch.empty((batch, kb), dtype=torch.int64, device=dev) for _ in range(S2)]
    NBq = len(my_q)

    def step(b):
        if sharded is not None and S2 > 1:
            with torch.cuda.stream(rstreams[b % S2]):
                sharded.search_replicated(my_q[b % NBq], k)
        elif sharded is not None:
            sharded.search_replicated(my_q[b % NBq], k)
        else:
            j = b % S2
            ref.search_into(my_q[b % NBq], k, D[j], I[j], cD[j], cI[j], rptr[j])

    settle(step, args.settle_ms)
    dt, blocks, _ = clock.measure(step, steps, warmup)

    # ---- roofline of the step's dominant kernel (the streaming re-rank: HBM-bound) and the step's split: each stage alone,
    # back to back on one launch stream between two HIP events recorded on that stream
    def stage_ms(fn, reps=10):
        with torch.cuda.stream(rstreams[0]):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

    qs0 = my_q[0]
    have_cand = hasattr(base, "search_candidates_into")
    roofline = None
    if have_cand:
        ms_cand = stage_ms(lambda: base.search_candidates_into(qs0, kb, cI[0], None, rptr[0]))
        ms_rr = stage_ms(lambda: flat_r.rerank(qs0, cI[0], k, D[0], I[0], rptr[0]))
        ms_both = stage_ms(lambda: ref.search_into(qs0, k, D[0], I[0], cD[0], cI[0], rptr[0]))
        torch.cuda.synchronize()
        ncand = int((cI[0] >= 0).sum().item())
        rr_bytes = ncand * D_MODEL * relem                               # every candidate's stored row, read once
        ach = rr_bytes / (ms_rr * 1e-3) / 1e9
        # the re-rank kernel's HBM-side bytes at THIS shape (counter passes with the sweep pinned: tools/prof_r06.sh cfg4)
        pmc, rr_traffic_src = committed_traffic("r06_cfg4_refine_pmc.json", lambda d_: relem == 1 and world == 1 and
                                                 tuple(d_["config"]) == (int(base.ntotal), int(base.nlist), batch, nprobe, kf, k))
        roofline = {"kernel": {1: "rerank_sq8_kernel", 2: "rerank_f16_kernel", 4: "rerank_f32_kernel"}[relem] + " (streaming re-rank of k*k_factor candidates per query)",
                    "bound": "hbm", "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4),
                    "traffic": int(pmc["corrected_bytes_per_launch"]) if pmc else None, "traffic_source": rr_traffic_src,
                    "bytes_per_launch": int(rr_bytes), "avg_launch_ms": round(ms_rr, 5),
                    "algorithmic_bytes": "candidates x d x %d B (the stored row of every candidate, read once)" % relem,
                    "step_split_ms": {"candidates (coarse + LUT + all-scores scan + set selection)": round(ms_cand, 4),
                                      "re-rank (query table + streaming kernel + top-k)": round(ms_rr, 4),
                                      "both stages on one stream": round(ms_both, 4),
                                      "timed step (batches round-robin on %d streams)" % S2: round(dt / steps * 1e3, 4)},
                    "timing": "each stage alone, 10 launches back to back on one launch stream between two HIP events recorded on that stream"}

    # ---- parity of the timed shape: the re-rank of the first 128 timed queries recomputed by the oracle from nothing but
    # the candidate rows' STORED bytes (mi_flat_get_rows: the 212 GB store cannot be exported) and the trained ranges
    parity = None
    if rank == 0 and have_cand and relem in (1, 4) and not args.no_cpu_baseline and (sharded is None):
        from oracle import ivfpq_oracle as O
        O.build()
        nqp = min(128, batch)
        ref.search_into(qs0, k, D[0], I[0], cD[0], cI[0], rptr[0])
        torch.cuda.synchronize()
        cs = cI[0][:nqp].cpu().numpy()
        Dh, Ih = D[0][:nqp].cpu().numpy(), I[0][:nqp].cpu().numpy()
        valid = cs >= 0
        uniq, inv = np.unique(cs[valid], return_inverse=True)            # sorted: the remap keeps the (score desc, id asc) order
        cl = np.full(cs.shape, -1, np.int64)
        cl[valid] = inv
        rows = flat_r.get_rows(uniq)
        t0 = time.perf_counter()
        if relem == 1:
            De, Ie = O.rerank_sq8(qs0[:nqp].cpu().numpy(), rows, flat_r.sq.trained, cl, k)
        else:
            De, Ie = O.rerank(qs0[:nqp].cpu().numpy(), rows.view(np.float32), cl, k)
        t_or = time.perf_counter() - t0
        Ie = np.where(Ie >= 0, uniq[np.maximum(Ie, 0)], -1)
        parity = {"against": "oracle/ivfpq_oracle.c " + ("oracle_rerank_sq8" if relem == 1 else "oracle_rerank") + " on the candidate rows' stored bytes",
                  "queries": int(nqp), "candidates_per_query": int(kb), "distinct_rows_exported": int(uniq.shape[0]),
                  "ids_equal": bool(np.array_equal(Ih, Ie)),
                  "scores_bit_equal": bool(np.array_equal(Dh.view(np.uint32), De.view(np.uint32))), "oracle_s": round(t_or, 1),
                  "note": "the candidate SETS themselves are the PQ scan's, whose parity is the main line's parity_vs_oracle "
                          "(tests/test_ivfpq_gpu.py::test_refine_sq8_at_the_timed_shape_matches_oracle checks both at this shape)"}
    base.nprobe = index.nprobe = args_nprobe
    # the scratch sets of the streams used so far (2 of the main line + S2 here: ~2-3 GB each) go back to the allocator: the
    # cfg5 leg's CPU baseline exports the lists through a 15 GB device staging buffer
    torch.cuda.synchronize()
    ref.release_workspaces()
    gb = flat_r.ntotal * D_MODEL * relem / 1e9
    if refine_own:
        scope = (f"the whole job: every rank re-ranks k*k_factor candidates of its own shard against the shard's {store} "
                 f"vectors ({gb:.0f} GB per GPU), one all-gather of the exact per-shard lists"
                 if nsh > 1 else f"whole index: all {index.ntotal} vectors + their {store} refine store ({gb:.0f} GB) on one GPU")
        recall_note = "against exact search over the whole corpus"
    else:
        scope = (f"one GPU's share of the 8-GPU job: the 1/8 sub-shard (rows i mod 8 = rank, {flat_r.ntotal} vectors + their "
                 f"{store} vectors, {gb:.0f} GB) -- the refine store of all {index.ntotal * nsh} rows "
                 f"({index.ntotal * nsh * D_MODEL * relem / 1e9:.0f} GB) does not fit {world} GPU(s); every GPU of the 8-GPU job "
                 f"sees every query, so its job rate is this rate less one all-gather")
        recall_note = "against exact search over the same sub-shard"
    return {"index": "IVF%d,PQ64,%s" % (base.nlist, suffix), "refine_store": store_name,
            "refine_store_note": "--refine-store auto = sq8 at every N: the points of a 1 -> 8 GPU curve are the same index",
            "nprobe": nprobe, "k_factor_rf": kf, "recall_at_10": round(r, 4),
            "recall_at_10_selection_batch": round(r_sel, 4),
            "reached": bool(r >= 0.95), "qps": round(steps * batch / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4),
            "timed_blocks": len(blocks), "streams": S2, "scope": scope,
            "recall_note": recall_note + "; (nprobe, k_factor_rf) chosen on a held-out query batch, recall reported on another",
            "roofline": roofline, "parity_vs_oracle": parity,
            "explored": curve, "explored_note": "[nprobe, k_factor_rf, recall@10 on the held-out batch] in ascending cost, then a bisection (to 16) between the last k_factor that missed 0.95 and the first that held it; the cheapest point at or above 0.95 is timed"}


# ======================================================================
# cfg5: end to end -- encode + search over the cfg4 index, query batches 1 / 16 / 256
# ======================================================================
def stella_random_weights(cfg, torch, dev, seed=7):
    """Random-init bf16 weights of a stella-shaped configuration, generated on the GPU, as (name, tensor) pairs in load
    order (no checkpoint is reachable from the build / bench boxes)."""
    g = torch.Generator(device=dev).manual_seed(seed)

    def rnd(shape, scale):
        return (torch.randn(shape, generator=g, device=dev) * scale).bfloat16()

    H, I = cfg["hidden"], cfg["intermediate"]
    qc, kc = cfg["n_heads"] * cfg["head_dim"], cfg["n_kv_heads"] * cfg["head_dim"]
    yield "embed_tokens.weight", rnd((cfg["vocab_size"], H), 0.3)
    yield "norm.weight", torch.ones(H, device=dev)
    yield "dense.weight", rnd((cfg["dense_out"], H), H ** -0.5)
    yield "dense.bias", torch.zeros(cfg["dense_out"], device=dev)
    for l in range(cfg["n_layers"]):
        p = f"layers.{l}."
        yield p + "input_layernorm.weight", torch.ones(H, device=dev)
        yield p + "post_attention_layernorm.weight", torch.ones(H, device=dev)
        yield p + "self_attn.q_proj.weight", rnd((qc, H), H ** -0.5)
        yield p + "self_attn.q_proj.bias", rnd((qc,), 0.1)
        yield p + "self_attn.k_proj.weight", rnd((kc, H), H ** -0.5)
        yield p + "self_attn.k_proj.bias", rnd((kc,), 0.1)
        yield p + "self_attn.v_proj.weight", rnd((kc, H), H ** -0.5)
        yield p + "self_attn.v_proj.bias", rnd((kc,), 0.1)
        yield p + "self_attn.o_proj.weight", rnd((H, qc), qc ** -0.5)
        yield p + "mlp.gate_proj.weight", rnd((I, H), H ** -0.5)
        yield p + "mlp.up_proj.weight", rnd((I, H), H ** -0.5)
        yield p + "mlp.down_proj.weight", rnd((H, I), I ** -0.5)


def stella_random_model(args, ctx, keep=None):
    """stella_en_1.5B_v5 architecture with random-init bf16 weights; `keep`: a dict that receives fp32 host copies of
    the weights (what the CPU oracle multiplies: the same bf16-representable values)."""
    torch, dev, local_rank = ctx["torch"], ctx["dev"], ctx["local_rank"]
    import abstracts_search_amd.sentence_transformers as st
    cfg = dict(st.STELLA_EN_1_5B_V5)
    model = st.SentenceTransformer(config=cfg, device=f"cuda:{local_rank}")
    for name, t in stella_random_weights(cfg, torch, dev, 7):
        model.load_weights({name: t})
        if keep is not None:
            keep[name] = t.float().cpu()
    return model, cfg


def cfg5_curve(args, ctx, model, cfg, host_w, index, sharded, nprobe, k, steps, warmup, do_cpu):
    """The query-time path at batches 1 / 16 / 256 (BASELINE.json configs[4]): prompted queries of 16-48 tokens arrive as
    token ids on the host, are encoded (batch 1: the weight-streaming kernels of csrc/encoder_few.h) and searched over
    `index` at `nprobe`.  Per batch: latency of one call, throughput of back-to-back calls, and the encoder's roofline
    with the bound that applies -- below ~312 tokens a forward pass is a pass over the weights (HBM: 2 FLOP per weight
    byte and token against 2.5 PFLOP/s : 8 TB/s), above it the MFMA rate."""
    np, torch = ctx["np"], ctx["torch"]
    clock, rank = ctx["clock"], ctx["rank"]
    H, I = cfg["hidden"], cfg["intermediate"]
    qc, kc = cfg["n_heads"] * cfg["head_dim"], cfg["n_kv_heads"] * cfg["head_dim"]
    # what one forward pass must stream at least once: the bf16 weights of the 28 layers + the Dense module
    nparams = cfg["n_layers"] * ((qc + 2 * kc) * H + H * qc + 3 * I * H) + cfg["dense_out"] * H
    weight_bytes = 2 * nparams
    budget0 = model.token_budget
    model.token_budget = 32768
    rng = np.random.default_rng(1)                                      # the same queries on every rank
    curve, batch_toks = [], {}
    for batch in (1, 16, 256):
        toks = [rng.integers(0, cfg["vocab_size"], int(rng.integers(16, 49))).tolist() for _ in range(batch)]
        batch_toks[batch] = toks

        def once(_=0):
            e = model.encode_tokens(toks, batch_size=batch, normalize_embeddings=True, as_tensor=True)
            return index.search(e, k) if sharded is None else sharded.search_replicated(e, k)

        for i in range(warmup):
            once()
        lat = []
        for i in range(steps):                                          # latency: one call at a time
            clock.barrier()
            t1 = time.perf_counter()
            once()
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t1)
        dt, blocks, _ = clock.measure(once, steps, 1)                   # throughput: back to back
        t1 = time.perf_counter()
        for i in range(steps):
            model.encode_tokens(toks, batch_size=batch, normalize_embeddings=True, as_tensor=True)
        torch.cuda.synchronize()
        enc = (time.perf_counter() - t1) / steps
        lat.sort()
        ntok = sum(len(t) for t in toks)
        flops = 2.0 * nparams * ntok + 4.0 * cfg["n_layers"] * qc * sum(len(t) ** 2 for t in toks)
        hbm_bound = ntok < 312                                          # 2 FLOP per weight byte and token vs 2.5 PF / 8 TB/s
        if hbm_bound:
            roof = {"bound": "hbm", "achieved": round(weight_bytes / enc / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(weight_bytes / enc / 8e12, 4), "bytes_per_launch": int(weight_bytes),
                    "algorithmic_bytes": "the bf16 weights of the 28 decoder layers + Dense, read once per forward pass"}
        else:
            roof = {"bound": "mfma", "achieved": round(flops / enc / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s",
                    "frac": round(flops / enc / 2.5e15, 4), "flops_per_launch": flops,
                    "algorithmic_flops": "2 x parameters x tokens + 4 x layers x q_cols x sum(len^2) (attention)"}
        qp, q_src = committed_traffic("r06_encode_query_pmc.json", lambda d_: any(r_.get("tokens") == ntok for r_ in d_["regimes"].values()))
        q_traffic = [r_["bytes_per_pass"] for r_ in qp["regimes"].values() if r_.get("tokens") == ntok][0] if qp else None
        roof.update({"kernel": "the encoder forward pass (%s)" % ("csrc/encoder_few.h: five weight-streaming launches per layer, the attention inside the O projection's" if ntok <= 32 else
                                                                    "csrc/encoder_few.h: six weight-streaming launches per layer" if ntok <= 48 else
                                                                    "general path: MFMA GEMM tiles by token count"),
                     "traffic": q_traffic, "traffic_source": q_src, "avg_launch_ms": round(enc * 1e3, 4), "what": "encode alone, back-to-back calls"})
        curve.append({"batch": batch, "latency_ms_p50": round(lat[len(lat) // 2] * 1e3, 3),
                      "latency_ms_p95": round(lat[min(len(lat) - 1, int(0.95 * len(lat)))] * 1e3, 3),
                      "queries_per_s": round(steps * batch / dt, 1), "encode_alone_ms": round(enc * 1e3, 3),
                      "tokens": ntok, "roofline": roof})
        log(f"  cfg5 batch {batch}: {curve[-1]}")
    model.token_budget = budget0
    cpu = parity = None
    if do_cpu and rank == 0:
        from oracle import encoder_oracle as E
        par = {}
        t_enc = None
        for bsz in (1, 16):                                             # the two batches the query-time kernels / mid-batch tiles serve
            toks = batch_toks[bsz]
            cu = np.concatenate([[0], np.cumsum([len(t) for t in toks])])
            e_gpu = model.encode_tokens(toks, batch_size=bsz, normalize_embeddings=True, as_tensor=True)
            with torch.no_grad():
                t1 = time.perf_counter()
                ref = E.encode(E.EncoderConfig(**cfg), host_w, np.concatenate(toks), cu, True).numpy()
                t_e = time.perf_counter() - t1
            cos = (e_gpu.cpu().numpy() * ref).sum(1)
            par["batch_%d" % bsz] = {"tokens": int(cu[-1]), "min_cosine_vs_oracle": round(float(cos.min()), 7), "ok": bool(cos.min() >= 1 - 1e-3)}
            if bsz == 16:
                t_enc, e16, cu16 = t_e, e_gpu, cu
        cpu_s, par_s = cpu_baseline_ivfpq(index, e16, nprobe, k, np, torch, "cfg5")
        t_search = 16.0 / cpu_s["value"]
        cpu = {"value": round(16.0 / (t_enc + t_search), 3), "unit": "queries/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"the batch of 16 queries ({int(cu16[-1])} tokens) end to end on the host cores: oracle encode at full depth {t_enc:.1f}s "
                         f"(oracle/encoder_oracle.py, torch fp32) + oracle search {t_search * 1e3:.0f} ms ({cpu_s['sample'][:60]}...)"}
        parity = {"encode": par, "tolerance": "cosine >= 1 - 1e-3, full depth", "search": par_s}
    return {"curve": curve, "cpu_baseline": cpu, "parity_vs_oracle": parity,
            "query_tokens": "16-48 per query (prompt + question)", "nprobe": nprobe, "k": k, "steps": steps}


def cfg5_workload(args, ctx):
    """BASELINE.json configs[4]: a query batch arrives as token ids on the host (what the tokenizer
    hands over), is encoded (prompted query, 16-48 tokens) and searched over the cfg4 index
    (IVF65536,PQ64, nprobe 64, k 10); N > 1: the index is vector-sharded, every rank encodes the
    same batch (replicated queries) and the per-shard top-k go through the one exchange step.
    Reported per batch size: latency of one call (median / p95 over the timed calls) and the
    throughput of back-to-back calls."""
    np, torch, dist = ctx["np"], ctx["torch"], ctx["dist"]
    world, rank, local_rank, dev, clock = ctx["world"], ctx["rank"], ctx["local_rank"], ctx["dev"], ctx["clock"]
    import abstracts_search_amd.faiss as faiss
    import abstracts_search_amd.synth as synth
    from abstracts_search_amd.shards import ShardedIndex
    N = 207_000_000 if args.corpus is None else args.corpus
    nlist = 65536 if args.nlist is None else args.nlist
    nprobe = 64 if args.nprobe is None else args.nprobe
    steps = 30 if args.steps is None else args.steps
    warmup = 3 if args.warmup is None else args.warmup
    d, M, k = D_MODEL, PQ_M, args.k
    nsh = world
    t0 = time.time()
    index = faiss.IndexIVFPQ(d, nlist, M, 8, faiss.METRIC_INNER_PRODUCT, device=local_rank)
    index.cp.niter = index.pq.cp.niter = 4 if args.train_iters is None else args.train_iters
    cent = torch.empty((nlist, d), dtype=torch.float32, device=dev)
    cb = torch.empty((M, 256, d // M), dtype=torch.float32, device=dev)
    if rank == 0:
        index.train(synth.corpus_cuda(min(N, max(4 * CH, 64 * nlist)), d, device=local_rank))
        cent.copy_(torch.from_numpy(index.get_centroids()))
        cb.copy_(torch.from_numpy(index.get_codebook()))
    if world > 1:
        dist.broadcast(cent, 0)
        dist.broadcast(cb, 0)
        index.set_centroids(cent)
        index.set_codebook(cb)
    del cent, cb
    index.reserve((N + nsh - 1) // nsh + 1)
    for c0 in range(0, N, CH):
        x = synth.corpus_cuda(min(CH, N - c0), d, device=local_rank, row0=c0)
        index.add(x if nsh == 1 else x[rank::nsh].contiguous())
        del x
    index.nprobe = nprobe
    torch.cuda.synchronize()
    log(f"[rank {rank}] index ntotal={index.ntotal} built in {time.time() - t0:.0f}s")
    sharded = ShardedIndex(index, id_affine=(nsh, 0, 1)) if world > 1 else None
    do_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    host_w = {} if do_cpu else None
    model, cfg = stella_random_model(args, ctx, keep=host_w)
    c5 = cfg5_curve(args, ctx, model, cfg, host_w, index, sharded, nprobe, k, steps, warmup, do_cpu)
    if rank != 0:
        return None
    curve, cpu, parity = c5["curve"], c5["cpu_baseline"], c5["parity_vs_oracle"]
    roofline = dict(curve[1]["roofline"], note="the batch-16 point (BASELINE.json configs[4]'s middle point); every batch carries its own in `curve`")
    return {"metric": "queries/sec end to end (stella_en_1.5B_v5 encode + IVF%d,PQ64 search over %dx1024-d, batch 256)" % (nlist, N),
            "value": curve[-1]["queries_per_s"], "unit": "queries/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(256e3 / curve[-1]["queries_per_s"], 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "bf16 encoder, f32 search", "data": "synthetic (random-init encoder weights, synthetic token ids and corpus)",
            "config": {"workload": "cfg5: end-to-end encode + search, %dx1024 IVF%d,PQ64 index, query batches 1/16/256 (BASELINE.json configs[4])" % (N, nlist),
                       "nprobe": nprobe, "k": k, "query_tokens": "16-48 per query (prompt + question)",
                       "parallelism": "1 GPU" if world == 1 else f"index vector-sharded x{world}, queries replicated, one all-gather of top-k"},
            "curve": curve, "roofline": roofline, "cpu_baseline": cpu, "parity_vs_oracle": parity, "reference_oracles": reference_oracles()}


# ======================================================================
# cfg2: 1 M x 1024, IVF4096,PQ64, batch 64 (round 1's line; `--workload cfg2`)
# ======================================================================
def cfg2_workload(args, ctx):
    np, torch, dist = ctx["np"], ctx["torch"], ctx["dist"]
    world, rank, local_rank, dev, clock = ctx["world"], ctx["rank"], ctx["local_rank"], ctx["dev"], ctx["clock"]
    import abstracts_search_amd.faiss as faiss
    import abstracts_search_amd.synth as synth
    from abstracts_search_amd.shards import ShardedIndex

    N = 1_000_000 if args.corpus is None else args.corpus
    nlist = 4096 if args.nlist is None else args.nlist
    batch = 64 if args.batch is None else args.batch
    nprobe = 16 if args.nprobe is None else args.nprobe
    steps = 200 if args.steps is None else args.steps
    warmup = 20 if args.warmup is None else args.warmup
    d, M, k = D_MODEL, PQ_M, args.k
    t0 = time.time()
    x = synth.corpus_cuda(N, d, device=local_rank)
    index = faiss.IndexIVFPQ(d, nlist, M, 8, faiss.METRIC_INNER_PRODUCT, device=local_rank)
    index.cp.niter = index.pq.cp.niter = 10 if args.train_iters is None else args.train_iters
    force_sharded = bool(os.environ.get("BENCH_FORCE_SHARDED"))
    use_shards = (world > 1 and args.multi_gpu_mode == "shards") or force_sharded
    if world > 1 or force_sharded:
        cent = torch.empty((nlist, d), dtype=torch.float32, device=dev)
        cb = torch.empty((M, 256, d // M), dtype=torch.float32, device=dev)
        if rank == 0:
            index.train(x)
            cent.copy_(torch.from_numpy(index.get_centroids()))
            cb.copy_(torch.from_numpy(index.get_codebook()))
        if dist.is_initialized():
            dist.broadcast(cent, 0)
            dist.broadcast(cb, 0)
        index.set_centroids(cent)
        index.set_codebook(cb)
        index.add(x[rank::world].contiguous() if use_shards else x)
    else:
        index.train(x)
        index.add(x)
    index.nprobe = nprobe
    sharded = ShardedIndex(index, shard_coarse=bool(args.shard_coarse), id_affine=(world, 0, 1)) if use_shards else None
    log(f"[rank {rank}] setup {time.time() - t0:.1f}s ntotal={index.ntotal}")

    NB = 16
    qpool = synth.queries_cuda(x, NB * batch * world, seed=4321).view(NB, world, batch, d)
    my_q = [qpool[b, 0 if use_shards else rank].contiguous() for b in range(NB)]
    S = max(1, 4 if args.streams is None else args.streams) if not use_shards else 1
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else [torch.cuda.current_stream(dev)]
    Ds = [torch.empty((batch, k), dtype=torch.float32, device=dev) for _ in range(S)]
    Is = [torch.empty((batch, k), dtype=torch.int64, device=dev) for _ in range(S)]
    sptr = [int(s_.cuda_stream) for s_ in streams]
    torch.cuda.synchronize()

    refine = None
    if args.refine > 1 and sharded is None:
        flat_r = faiss.IndexFlatIP(d, device=local_rank)
        flat_r.add(x)
        refine = faiss.IndexRefineFlat(index, flat_r)
        refine.k_factor = args.refine
        kb = k * args.refine
        cDs = [torch.empty((batch, kb), dtype=torch.float32, device=dev) for _ in range(S)]
        cIs = [torch.empty((batch, kb), dtype=torch.int64, device=dev) for _ in range(S)]

    def step(b):
        j = b % S
        if refine is not None:
            refine.search_into(my_q[b % NB], k, Ds[j], Is[j], cDs[j], cIs[j], sptr[j])
        elif sharded is None:
            index.search_into(my_q[b % NB], k, Ds[j], Is[j], None, sptr[j])
        else:
            sharded.search_replicated(my_q[b % NB], k)

    for b in range(max(warmup, 2 * S)):
        step(b)
    torch.cuda.synchronize()
    t_settle, b = time.perf_counter(), 0
    while time.perf_counter() - t_settle < args.settle_ms * 1e-3:
        for _ in range(64):
            step(b)
            b += 1
        torch.cuda.synchronize()
    dt, blocks, t_issue = clock.measure(step, steps, warmup)
    replicas = world > 1 and not use_shards
    qps = steps * batch * (world if replicas else 1) / dt

    torch.cuda.synchronize()
    index.search_into(my_q[0], k, Ds[0], Is[0], None, sptr[0])
    prof = index.profile_scan(max(steps, 50), sptr[0])
    torch.cuda.synchronize()
    scan_ms, scan_bytes = prof["scan_ms_avg"], prof["scan_bytes"]
    achieved = scan_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    pmc, traffic_src2 = committed_traffic("r04_cfg2_scan_pmc.json", lambda d: (N, nlist, batch, nprobe, k, world) == (1_000_000, 4096, 64, 16, 10, 1)
                                          and not os.environ.get("MI_NSLICE"))
    traffic = int(pmc["corrected_bytes_per_launch"]) if pmc else None
    roofline = {"kernel": "scan_kernel<64,8,false>", "bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0,
                "unit": "GB/s", "frac": round(achieved / 8000.0, 4), "traffic": traffic,
                "traffic_source": traffic_src2,
                "note": "a 14-15 us launch: a latency chain (tables + LUT staging, three rounds of gathers, rank / publish / merge), not a "
                        "bandwidth number -- DESIGN.md section 5",
                "bytes_per_launch": int(scan_bytes), "avg_launch_ms": round(scan_ms, 5)}
    if rank != 0:
        return None
    recall = None
    if not args.no_recall and world == 1 and sharded is None:
        flat = faiss.IndexFlatIP(d, device=local_rank)
        flat.add(x)
        hits = tot = 0
        for b in range(4):
            _, Ia = (refine if refine is not None else index).search(my_q[b], k)
            _, Ie = flat.search(my_q[b], k)
            hits += recall_at_k(Ia, Ie) * Ie.numel()
            tot += Ie.numel()
        recall = hits / tot
        del flat
    cpu = parity = None
    if not args.no_cpu_baseline and world == 1:
        cpu, parity = cpu_baseline_ivfpq(index, torch.cat(my_q), nprobe, k, np, torch, "cfg2")
    return {
        "metric": "queries/sec, IVF-PQ search (IVF%d,PQ64%s, %dx1024-d, batch %d, nprobe %d, k %d)"
                  % (nlist, ",RFlat x%d" % args.refine if refine is not None else "", N, batch, nprobe, k),
        "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(dt / steps * 1e3, 5), "higher_is_better": True,
        "scaling": "weak" if replicas else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "cfg2: 1Mx1024 clustered synthetic corpus, IVF4096,PQ64, batch-64 queries (BASELINE.json configs[1])",
                   "corpus": N, "nlist": nlist, "M": 64, "nprobe": nprobe, "k": k, "global_batch": batch * (world if replicas else 1),
                   "parallelism": "1 GPU" if world == 1 else (f"vector-sharded x{world} + one all-gather of top-k" if use_shards
                                                              else f"query-parallel replicas x{world} (no collective)"),
                   "launch": "eager (a hipGraph replay of the step measured slower)", "streams": S,
                   "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                   "host_issue_ms_per_step": round(t_issue / steps * 1e3, 5),
                   "timed_blocks": len(blocks), "timing": "median of the K-step blocks" if len(blocks) > 1 else "one K-step block"},
        "recall_at_10": None if recall is None else round(recall, 4),
        "roofline": roofline, "cpu_baseline": cpu, "parity_vs_oracle": parity, "reference_oracles": reference_oracles(),
    }


# ======================================================================
# cfg3: stella_en_1.5B_v5 bf16 batch encode
# ======================================================================
def encode_workload(args, ctx, steps, warmup, with_cpu=True, pack=None):
    """stella_en_1.5B_v5 architecture (random-init bf16 weights -- no checkpoint is reachable
    from the build/bench boxes), synthetic abstracts with clipped log-normal token counts
    (median 220, max 512).  A step encodes one batch of `--encode-batch` abstracts: embedding
    gather, 28 decoder layers, mean pooling, Dense 1536->1024, L2 normalise; token ids start on
    the host (as they do after tokenisation), embeddings stay in HBM.  N > 1: replicas, every
    rank encodes its own batches (no collective on this path)."""
    np, torch, dist = ctx["np"], ctx["torch"], ctx["dist"]
    world, rank, local_rank, dev, clock = ctx["world"], ctx["rank"], ctx["local_rank"], ctx["dev"], ctx["clock"]
    import abstracts_search_amd.sentence_transformers as st
    do_cpu = rank == 0 and world == 1 and with_cpu
    if pack is not None:                                              # the caller's model (the cfg4 line builds it once)
        model, cfg, host_w = pack
        do_cpu = do_cpu and host_w is not None
    else:
        host_w = {} if do_cpu else None                               # fp32 host copies for the full-depth oracle leg
        model, cfg = stella_random_model(args, ctx, keep=host_w)
    model.token_budget = 32768
    bs = args.encode_batch
    rng = np.random.default_rng(7 + rank)
    NBATCH = 8
    batches = []
    if bs > 0:                                                        # a fixed number of abstracts per step
        model.token_budget = None
        for _ in range(NBATCH):
            lens = np.clip(np.exp(rng.normal(np.log(220), 0.45, bs)), 8, 512).astype(int)
            batches.append([rng.integers(0, cfg["vocab_size"], L).tolist() for L in lens])
    else:
        # the library's own batching: abstracts in arrival order, cut where the next one would take the forward pass
        # past SentenceTransformer.token_budget padded tokens (32 768: every GEMM fills whole rounds of the CUs)
        budget = model.token_budget
        cur, tok = [], 0
        while len(batches) < NBATCH:
            L = int(np.clip(np.exp(rng.normal(np.log(220), 0.45)), 8, 512))
            t = L
            if cur and tok + t > budget:
                batches.append(cur)
                cur, tok = [], 0
            cur.append(rng.integers(0, cfg["vocab_size"], L).tolist())
            tok += t
    ntok = [sum(len(t) for t in b) for b in batches]
    nabs = [len(b) for b in batches]

    # batches are independent: like the search loops, steps may be issued round-robin on S streams (the library keeps a
    # workspace set per stream) -- the partial last round of one batch's GEMM tiles then shares the chip with the other's
    ES = max(1, args.encode_streams)
    estreams = [torch.cuda.Stream(device=dev) for _ in range(ES)] if ES > 1 else None

    def step(i):
        b = batches[i % NBATCH]
        if estreams is None:
            return model.encode_tokens(b, batch_size=len(b), normalize_embeddings=True, as_tensor=True)
        with torch.cuda.stream(estreams[i % ES]):
            return model.encode_tokens(b, batch_size=len(b), normalize_embeddings=True, as_tensor=True)

    dt, blocks, _ = clock.measure(step, steps, max(warmup, 1))
    toks = sum(ntok[(warmup + i) % NBATCH] for i in range(steps))
    n_abs = sum(nabs[(warmup + i) % NBATCH] for i in range(steps))
    bs_txt = bs if bs > 0 else "token budget %d per forward pass (%d-%d abstracts)" % (model.token_budget, min(nabs), max(nabs))
    # roofline of the dominant kernel (bf16 MFMA GEMMs): HIP events around the GEMM launches
    model.profile(True)
    step(0)
    torch.cuda.synchronize()
    pr = model.profile_read()
    model.profile(False)
    tf = pr["gemm_flops"] / (pr["gemm_ms"] * 1e-3) / 1e12
    # separate --pmc passes of `bench.py --workload encode` (tools/prof_r06.sh), dropped when the slab kernel changed since
    pmc, traffic_src = committed_traffic("r06_cfg3_encoder_gemm_pmc.json", lambda d: d.get("batch") == bs and d.get("tokens_step0") == ntok[0])
    traffic, xcheck = None, None
    if pmc:
        traffic = int(pmc["hbm_bytes_per_step"])
        kt = pmc["same_run_timing"]["gemm_ms_per_step_kernel_trace_sum"]
        xcheck = {"gemm_ms_per_step": kt, "achieved": round(pr["gemm_flops"] / (kt * 1e-3) / 1e12, 1),
                  "frac": round(pr["gemm_flops"] / (kt * 1e-3) / 1e12 / 2500.0, 4),
                  "what": "the same FLOPs over the SUM of rocprofv3's per-kernel durations of the four GEMM kernels in the committed "
                          "kernel-trace run of this workload (" + str(pmc.get("kernel_stats_file", "profiles/")) + "): a kernel's traced duration "
                          "includes its drain tail and end-of-kernel cache write-back, during which the next launch already runs -- the "
                          "sum reads 3-4 % above what the launches occupy back to back, and bounds `frac` from below"}
    roofline = {"kernel": "gemm_bf16_slab_kernel (QKV / O / gate-up+SwiGLU / down, 112 launches per step; their epilogues carry the RMSNorms "
                          "and the rotary embedding: no rmsnorm_kernel / rope_kernel in the pass)",
                "bound": "mfma", "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s",
                "frac": round(tf / 2500.0, 4), "traffic": traffic, "traffic_source": traffic_src,
                "flops_per_step": pr["gemm_flops"], "gemm_ms_per_step": round(pr["gemm_ms"], 3), "kernel_trace_cross_check": xcheck,
                "whole_step": {"achieved": round(pr["gemm_flops"] / (dt / steps) / 1e12, 1), "frac": round(pr["gemm_flops"] / (dt / steps) / 1e12 / 2500.0, 4),
                               "what": "the same GEMM FLOPs over the whole timed step (attention, pooling and launch gaps included): what the fused "
                                       "epilogues buy shows here -- they lengthen the GEMM launches (frac above falls ~3 %) and shorten the step (~1 %)"},
                "timing": "the GEMM launches of one profiled step replayed back to back on the launch stream between two HIP events "
                          "(one warm pass, three timed; mi_encoder_profile_read), after the timed blocks"}
    # configs[2] at its stated size: at N = 1 by default (the driver's line times the whole 100 000-abstract run, not a sample)
    do_full = getattr(args, "full", False) or (world == 1 and not getattr(args, "no_full", False))
    full_run = encode_full_run(args, ctx, model, cfg) if do_full else None
    cpu = parity = None
    if do_cpu:
        cpu, parity = encode_cpu_baseline(model, cfg, host_w, batches, ctx)
    del model, host_w
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    return {
        "metric": "abstracts/sec, stella_en_1.5B_v5 bf16 batch encode (synthetic abstracts, median 220 tokens)",
        "value": round(n_abs * world / dt, 1), "unit": "abstracts/s", "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
        "data": "synthetic (random-init weights of the real architecture, synthetic token ids)",
        "config": {"workload": "cfg3: stella_en_1.5B_v5 bf16 batch encode (BASELINE.json configs[2])", "batch": bs_txt,
                   "sample": "%d abstracts (%d steps) of the config's 100k: the rate is per batch, batches are independent"
                             % (n_abs, steps),
                   "tokens_per_sec": round(toks * world / dt, 0), "parallelism": "replicas" if world > 1 else "1 GPU",
                   "timed_blocks": len(blocks)},
        "full_run": full_run,
        "roofline": roofline, "cpu_baseline": cpu, "parity_vs_oracle": parity, "reference_oracles": reference_oracles()}


def encode_full_run(args, ctx, model, cfg):
    """configs[2] as a RUN, not a sample: every one of the 100 000 synthetic abstracts goes through the call
    `sidecar-search build` makes (reference Makefile:65: SentenceTransformer.encode over the whole input) -- here
    encode_tokens(), i.e. encode() behind the tokenizer (token ids are synthetic: there is no text to tokenise) -- with
    the library's own glue inside the timed region: sort by length, cut into forward passes by `token_budget`, pack
    ids on the host, stage them through pinned memory, scatter the embeddings to their input rows, copy the
    [n, 1024] float32 result back to the host.  Wall clock, one call, nothing rotated or replayed."""
    np, torch, rank = ctx["np"], ctx["torch"], ctx["rank"]
    n = int(args.full_abstracts)
    rng = np.random.default_rng(1007 + rank)
    lens = np.clip(np.exp(rng.normal(np.log(220), 0.45, n)), 8, 512).astype(int)
    flat = rng.integers(0, cfg["vocab_size"], int(lens.sum()), dtype=np.int32)
    off = np.concatenate([[0], np.cumsum(lens)])
    toks = [flat[off[i]:off[i + 1]] for i in range(n)]                # what a tokenizer hands over: one id array per text
    budget0 = model.token_budget
    model.token_budget = 32768
    order = sorted(range(n), key=lambda i: -len(toks[i]))
    passes = model._passes(order, toks, None)
    model.encode_tokens(toks[:256], normalize_embeddings=True)         # warm (workspaces at the pass size are allocated lazily)
    model.encode_tokens([toks[i] for i in passes[0]], normalize_embeddings=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    emb = model.encode_tokens(toks, normalize_embeddings=True)         # -> numpy on the host
    wall = time.perf_counter() - t0
    model.token_budget = budget0
    norms = np.linalg.norm(emb, axis=1)
    assert emb.shape == (n, 1024) and np.isfinite(emb).all() and abs(norms - 1).max() < 1e-3
    # the glue must not change a result: rows of a mid-run pass recomputed alone
    probe = passes[len(passes) // 2][:64]
    again = model.encode_tokens([toks[i] for i in probe], normalize_embeddings=True)
    cos = float((again * emb[probe]).sum(1).min())
    return {"abstracts": n, "tokens": int(lens.sum()), "forward_passes": len(passes),
            "abstracts_per_pass": [min(len(p) for p in passes), max(len(p) for p in passes)],
            "wall_s": round(wall, 2), "abstracts_per_s": round(n / wall, 1), "tokens_per_s": round(float(lens.sum()) / wall, 0),
            "min_cosine_vs_same_rows_encoded_alone": round(cos, 7),
            "what": "ONE encode_tokens() call over all abstracts, wall clock: length sort + token-budget passes + host packing + "
                    "pinned staging + row scatter + the [n, 1024] f32 copy to the host (tokenisation excluded: ids are synthetic)"}


def encode_cpu_baseline(model, cfg, host_w, batches, ctx):
    """The oracle port (oracle/encoder_oracle.py, torch fp32 on the host cores; sentence-transformers and the stella
    checkpoint are on no box of this pool -- `reference` records the attempt), two legs:

    full depth   the bench's own 28-layer model on 4 abstracts of the timed workload: the CPU baseline (abstracts/s,
                 nothing extrapolated) and the parity of the whole stack at depth;
    bulk path    the forward pass bench.py TIMES -- one 128-abstract batch (~29 k tokens, sequences up to 512: the
                 256x256 slab GEMMs at K = 1536 / 8960, >= 8 attention chunks, GEMM pooling) -- through a model of
                 stella's widths cut to 2 layers and a 4096-row vocabulary (what bounds the oracle's time), default
                 dispatch, against the oracle on the same batch.
    Tolerance: cosine >= 1 - 1e-3 per embedding (BASELINE.json north star)."""
    np, torch, dev, local_rank = ctx["np"], ctx["torch"], ctx["dev"], ctx["local_rank"]
    import abstracts_search_amd.sentence_transformers as st
    from oracle import encoder_oracle as E
    st_state = reference_oracles()["sentence_transformers"]
    # ---- full depth
    toks = batches[0][:4]
    cu = np.concatenate([[0], np.cumsum([len(t) for t in toks])])
    ids = np.concatenate(toks)
    ocfg = E.EncoderConfig(**cfg)
    e_gpu = model.encode_tokens(toks, batch_size=len(toks), normalize_embeddings=True)
    with torch.no_grad():
        t0 = time.perf_counter()
        ref = E.encode(ocfg, host_w, ids, cu, True).numpy()
        dt = time.perf_counter() - t0
    cos_full = (e_gpu * ref).sum(1)
    cpu = {"value": round(len(toks) / dt, 3), "unit": "abstracts/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"{len(toks)} abstracts ({int(cu[-1])} tokens) of the timed workload through all {cfg['n_layers']} layers "
                     f"(oracle/encoder_oracle.py, torch fp32), {dt:.1f}s, nothing extrapolated",
           "reference": "sentence_transformers " + st_state + "; the stella_en_1.5B_v5 checkpoint is not on this box"}
    # ---- the bulk path
    small = dict(cfg)
    small["n_layers"], small["vocab_size"] = 2, 4096
    w2 = dict(stella_random_weights(small, torch, dev, 11))
    m2 = st.SentenceTransformer(config=small, weights=w2, device=f"cuda:{local_rank}")
    m2.token_budget = model.token_budget
    b0 = [[t % 4096 for t in s] for s in batches[0]]
    before = st.debug_counter("tail_split_launches")
    e2 = m2.encode_tokens(b0, batch_size=len(b0), normalize_embeddings=True)
    tails = st.debug_counter("tail_split_launches") - before
    cu2 = np.concatenate([[0], np.cumsum([len(t) for t in b0])])
    with torch.no_grad():
        t0 = time.perf_counter()
        ref2 = E.encode(E.EncoderConfig(**small), {k: v.float().cpu() for k, v in w2.items()}, np.concatenate(b0), cu2, True).numpy()
        dt2 = time.perf_counter() - t0
    cos_bulk = (e2 * ref2).sum(1)
    del m2, w2
    parity = {"against": "oracle/encoder_oracle.py (torch fp32 on the host; sentence_transformers " + st_state + ")",
              "tolerance": "cosine >= 1 - 1e-3",
              "full_depth": {"abstracts": len(toks), "tokens": int(cu[-1]), "layers": cfg["n_layers"], "min_cosine": round(float(cos_full.min()), 7),
                             "ok": bool(cos_full.min() >= 1 - 1e-3)},
              "bulk_path": {"abstracts": len(b0), "tokens": int(cu2[-1]), "longest": int(max(len(t) for t in b0)), "layers": 2,
                            "min_cosine": round(float(cos_bulk.min()), 7), "ok": bool(cos_bulk.min() >= 1 - 1e-3),
                            "k_split_tail_launches": int(tails), "oracle_s": round(dt2, 1),
                            "path": "default dispatch: 256x256 slab GEMMs (K = 1536 and 8960), paired-head attention, GEMM pooling"}}
    return cpu, parity


def cpu_baseline_ivfpq(index, queries, nprobe, k, np, torch, tag):
    """CPU path on the host cores, same index and queries, bounded to roughly 10-30 s: real
    faiss-cpu when it is importable on this box (the index goes through write_index ->
    faiss.read_index, which is also the file format's first contact with real faiss),
    otherwise the oracle port (oracle/ivfpq_oracle.c, OpenMP over queries).  Also returns the
    parity check of the HIP result against whichever ran."""
    qs = queries.cpu().numpy()
    D_hip, I_hip = index.search(queries, k, nprobe=nprobe)
    D_hip, I_hip = D_hip.cpu().numpy(), I_hip.cpu().numpy()
    real = real_faiss()
    if real is not None:
        import tempfile
        import abstracts_search_amd.faiss as mine
        with tempfile.TemporaryDirectory() as tmp:
            f = os.path.join(tmp, "index.faiss")
            mine.write_index(index, f)
            ridx = real.read_index(f)
        ridx.nprobe = nprobe
        t0 = time.perf_counter()
        Dr, Ir = ridx.search(qs, k)
        one = time.perf_counter() - t0
        reps = int(max(1, min(50, 15.0 / max(one, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(reps):
            ridx.search(qs, k)
        dt = time.perf_counter() - t0
        same = I_hip == Ir
        mism = ~same
        dmax = float(np.abs(D_hip - Dr).max())
        # a mismatch is a rounding tie when the two scores at that rank differ by no more than the f32 spacing
        tie = np.abs(D_hip - Dr)[mism] <= 4 * np.spacing(np.abs(Dr[mism]).astype(np.float32)) if mism.any() else np.array([], bool)
        parity = {"against": "faiss-cpu " + getattr(real, "__version__", "?"), "queries": int(qs.shape[0]),
                  "ids_exact_match_rate": float(same.mean()), "max_abs_score_diff": dmax,
                  "mismatches_that_are_rounding_ties": int(tie.sum()), "mismatches": int(mism.sum())}
        cpu = {"value": round(reps * qs.shape[0] / dt, 1), "unit": "queries/s", "cores": real.omp_get_max_threads(),
               "kind": "faiss-cpu", "sample": f"{reps} calls of {qs.shape[0]} queries, same index (write_index -> faiss.read_index), "
                                                f"nprobe {nprobe}, k {k}, {dt:.1f}s"}
        return cpu, parity
    from oracle import ivfpq_oracle as O
    O.build()
    cent, cb = index.get_centroids(), index.get_codebook()
    sizes = index.list_sizes()
    off = np.zeros(index.nlist + 1, np.int64)
    np.cumsum(sizes, out=off[1:])
    t0 = time.perf_counter()
    codes, ids = index.export_lists()                                   # list order, insertion order inside a list
    t_exp = time.perf_counter() - t0
    t0 = time.perf_counter()
    De, Ie = O.search(qs, cent, cb, off, codes, ids, nprobe, k)         # warm + calibrate + parity
    one = time.perf_counter() - t0
    reps = int(max(1, min(200, 12.0 / max(one, 1e-4))))
    t0 = time.perf_counter()
    for _ in range(reps):
        O.search(qs, cent, cb, off, codes, ids, nprobe, k)
    dt = time.perf_counter() - t0
    parity = {"against": "oracle/ivfpq_oracle.c (CPU restatement; faiss absent on this box)", "queries": int(qs.shape[0]),
              "ids_equal": bool(np.array_equal(I_hip, Ie)),
              "scores_bit_equal": bool(np.array_equal(D_hip.view(np.uint32), De.view(np.uint32)))}
    cpu = {"value": round(reps * qs.shape[0] / dt, 1), "unit": "queries/s", "cores": O.num_threads(), "kind": "port",
           "sample": f"{reps} calls of {qs.shape[0]} queries on the same {tag} index ({int(sizes.sum())} vectors exported from HBM in "
                     f"{t_exp:.1f}s), nprobe {nprobe}, k {k}, {dt:.1f}s (oracle/ivfpq_oracle.c, OpenMP over queries; "
                     f"faiss-cpu is not importable on this box)"}
    return cpu, parity


if __name__ == "__main__":
    main()
