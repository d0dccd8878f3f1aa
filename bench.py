#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

Workload (config.workload = "cfg2"): BASELINE.json configs[1] -- 1M x 1024-d
synthetic clustered corpus, IVF4096,PQ64 (inner product), batch-64 queries,
k = 10.  A "step" is one IndexIVFPQ.search of one 64-query batch (coarse
quantise + LUT + PQ-code scan + top-k), queries and outputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): the corpus
is sharded by vector across the ranks (rank r holds rows i = r mod N); every
rank brings its own 64-query batch; an all-gather shares the queries, every
rank scans its shard for all 64*N queries, an all-gather of the per-shard
top-k (the path's one exchange step) is merged on every rank.  Per-rank scan
work is constant in N ("weak"); value = all queries of all ranks / time.

One JSON line on stdout (rank 0).  Extra objects: "roofline" (PQ-scan kernel,
HIP events on the launch stream) and "cpu_baseline" (oracle port on the host
cores, bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--corpus", type=int, default=1_000_000)
    ap.add_argument("--nlist", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--nprobe", type=int, default=16)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--train-iters", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-recall", action="store_true")
    ap.add_argument("--streams", type=int, default=4,
                    help="HIP streams the steps are issued round-robin on (batches overlap on the GPU)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    import abstracts_search_amd.faiss as faiss
    import abstracts_search_amd.synth as synth
    from abstracts_search_amd.shards import ShardedIndex

    d, M, k = 1024, 64, args.k
    t0 = time.time()
    # ---- corpus + index (setup, untimed).  The corpus is cfg2's 1M vectors at
    # every N; rank r indexes rows r mod N.
    x = synth.corpus_cuda(args.corpus, d, device=local_rank)
    index = faiss.IndexIVFPQ(d, args.nlist, M, 8, faiss.METRIC_INNER_PRODUCT, device=local_rank)
    index.cp.niter = args.train_iters
    index.train(x)                       # identical on every rank (same data, same seeds)
    if world > 1:
        ids = torch.arange(rank, args.corpus, world, device=dev)
        index.add_with_ids(x[rank::world].contiguous(), ids)
    else:
        index.add(x)
    index.nprobe = args.nprobe
    sharded = ShardedIndex(index) if world > 1 else None
    log(f"[rank {rank}] setup {time.time() - t0:.1f}s ntotal={index.ntotal}")

    NB = 16                               # pool of distinct query batches
    qpool = synth.queries_cuda(x, NB * args.batch * world, seed=4321).view(NB, world, args.batch, d)
    my_q = [qpool[b, rank].contiguous() for b in range(NB)]
    nq_out = args.batch
    # steps are independent query batches: they are issued round-robin on S
    # streams (each with its own output buffers and library workspaces) so that
    # consecutive batches overlap on the GPU, as a serving loop would run them
    S = max(1, args.streams) if world == 1 else 1
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else [torch.cuda.current_stream(dev)]
    Ds = [torch.empty((nq_out, k), dtype=torch.float32, device=dev) for _ in range(S)]
    Is = [torch.empty((nq_out, k), dtype=torch.int64, device=dev) for _ in range(S)]
    D, I = Ds[0], Is[0]
    torch.cuda.synchronize()

    sptr = [int(s_.cuda_stream) for s_ in streams]

    def step(b):
        if sharded is None:
            j = b % S
            index.search_into(my_q[b % NB], k, Ds[j], Is[j], None, sptr[j])
        else:
            sharded.search_into(my_q[b % NB], k, D, I)

    for b in range(max(args.warmup, 2 * S)):
        step(b)
    torch.cuda.synchronize()

    def run(nsteps, first=0):
        for i in range(nsteps):
            step(first + i)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    barrier()
    t1 = time.perf_counter()
    run(args.steps, args.warmup)
    barrier()
    dt = time.perf_counter() - t1
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    total_queries = args.steps * args.batch * world
    qps = total_queries / dt

    # ---- roofline of the dominant kernel (PQ-code scan): the library re-launches
    # the scan of the last step K times back to back between two HIP events
    # recorded on the launch stream (per-launch events cost more than the kernel)
    torch.cuda.synchronize()
    index.search_into(my_q[0], k, Ds[0], Is[0], None, sptr[0])
    prof = index.profile_scan(args.steps, sptr[0])
    torch.cuda.synchronize()
    scan_ms = prof["scan_ms_avg"]
    scan_bytes = prof["scan_bytes"]
    achieved = scan_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    roofline = {"kernel": "scan_kernel<64>", "bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0,
                "unit": "GB/s", "frac": round(achieved / 8000.0, 4), "traffic": None,
                "bytes_per_launch": int(scan_bytes), "avg_launch_ms": round(scan_ms, 5),
                "launches": args.steps}

    out = None
    if rank == 0:
        # ---- recall@10 against exact search (untimed)
        recall = None
        if not args.no_recall and world == 1:
            flat = faiss.IndexFlatIP(d, device=local_rank)
            flat.add(x)
            hits = tot = 0
            for b in range(4):
                _, Ia = index.search(my_q[b], k)
                _, Ie = flat.search(my_q[b], k)
                for a, e in zip(Ia.cpu().numpy(), Ie.cpu().numpy()):
                    hits += len(set(a.tolist()) & set(e.tolist()))
                    tot += k
            recall = hits / tot
            del flat
        cpu = None
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(index, my_q, args, np)
        out = {
            "metric": "queries/sec, IVF-PQ search (IVF%d,PQ64, %dx1024-d, batch %d, nprobe %d, k %d)"
                      % (args.nlist, args.corpus, args.batch, args.nprobe, k),
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "cfg2: 1Mx1024 clustered synthetic corpus, IVF4096,PQ64, batch-64 queries",
                       "corpus": args.corpus, "nlist": args.nlist, "M": 64, "nprobe": args.nprobe,
                       "k": k, "batch_per_rank": args.batch, "global_batch": args.batch * world,
                       "parallelism": "1 GPU" if world == 1 else f"vector-sharded x{world} + all-gather top-k",
                       "launch": "eager (a hipGraph replay of the step measured slower)", "streams": S},
            "recall_at_10": None if recall is None else round(recall, 4),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(index, my_q, args, np):
    """The oracle port (oracle/ivfpq_oracle.c, OpenMP) on the host cores, same
    index and queries, bounded to roughly 10-20 s."""
    from oracle import ivfpq_oracle as O
    O.build()
    cent, cb = index.get_centroids(), index.get_codebook()
    sizes = np.array([index.list_size(l) for l in range(index.nlist)], np.int64)
    off = np.zeros(index.nlist + 1, np.int64)
    np.cumsum(sizes, out=off[1:])
    codes = np.empty((int(off[-1]), 64), np.uint8)
    ids = np.empty(int(off[-1]), np.int64)
    for l in range(index.nlist):
        if sizes[l]:
            c, i = index.get_list(l)
            codes[off[l]:off[l + 1]], ids[off[l]:off[l + 1]] = c, i
    qs = [q.cpu().numpy() for q in my_q]
    t0 = time.perf_counter()
    O.search(qs[0], cent, cb, off, codes, ids, args.nprobe, args.k)     # warm + calibrate
    one = time.perf_counter() - t0
    reps = int(max(2, min(len(qs) * 64, 12.0 / max(one, 1e-4))))
    t0 = time.perf_counter()
    for r in range(reps):
        O.search(qs[r % len(qs)], cent, cb, off, codes, ids, args.nprobe, args.k)
    dt = time.perf_counter() - t0
    return {"value": round(reps * qs[0].shape[0] / dt, 1), "unit": "queries/s", "cores": O.num_threads(),
            "kind": "port",
            "sample": f"{reps} batches of {qs[0].shape[0]} queries, same index/nprobe/k, {dt:.1f}s "
                      f"(oracle/ivfpq_oracle.c, OpenMP over queries; faiss-cpu not installable here)"}


if __name__ == "__main__":
    main()
