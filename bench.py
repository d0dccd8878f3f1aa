#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

Workload (config.workload = "cfg2"): BASELINE.json configs[1] -- 1M x 1024-d
synthetic clustered corpus, IVF4096,PQ64 (inner product), batch-64 queries,
k = 10.  A "step" is one IndexIVFPQ.search of one 64-query batch (coarse
quantise + LUT + PQ-code scan + top-k), queries and outputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL).  Query
batches are independent units, and the whole cfg2 index is 72 MB (the 207 M
index of cfg4 is 15 GB) against 288 GB of HBM per GPU, so the default
`--multi-gpu-mode replicas` shards the *queries*: every rank holds the index and
searches its own 64-query batches, no collective on the data path, per-rank work
constant in N ("weak"); value = all queries of all ranks / max-over-ranks time.
`--multi-gpu-mode shards` runs the north star's vector-sharded variant instead
(rank r holds rows i = r mod N; all-gather of the ranks' query batches, every
rank scans its shard for all 64*N queries, all-gather of the per-shard top-k --
the path's one exchange step -- and a k-way merge); it is what a 207 M index
uses to cut single-batch latency, and it is latency-bound by its collectives at
cfg2 sizes.

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

# The steps are issued round-robin on several HIP streams; the runtime maps streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4, one of which torch's own stream takes), and
# two streams sharing a queue serialise.  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


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
    ap.add_argument("--settle-ms", type=float, default=100.0,
                    help="untimed steps issued for this long before warmup (clock ramp after setup)")
    ap.add_argument("--refine", type=int, default=0, metavar="K_FACTOR",
                    help="IVF4096,PQ64,RFlat: re-rank k*K_FACTOR PQ candidates with exact inner products "
                         "(faiss IndexRefineFlat); 0 = plain IVF-PQ, the headline configuration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-recall", action="store_true")
    ap.add_argument("--workload", choices=["search", "encode"], default="search",
                    help="search = cfg2 (the headline line); encode = cfg3 (stella_en_1.5B_v5 bf16 batch encode)")
    ap.add_argument("--encode-batch", type=int, default=128, help="abstracts per encode step")
    ap.add_argument("--multi-gpu-mode", choices=["replicas", "shards"], default="replicas",
                    help="N>1: replicas = query-parallel, no collective (default); shards = vector-sharded index + all-gather")
    ap.add_argument("--shard-coarse", type=int, default=0,
                    help="N>1: also split the coarse quantiser across ranks (pays off at IVF65536, not at cfg2)")
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
    if world > 1 or os.environ.get("BENCH_FORCE_SHARDED"):
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29511", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=dev)

    if args.workload == "encode":
        return encode_workload(args, np, torch, dist, world, rank, local_rank, dev)

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
    force_sharded = bool(os.environ.get("BENCH_FORCE_SHARDED"))   # exercise the N>1 plumbing on one GPU
    use_shards = (world > 1 and args.multi_gpu_mode == "shards") or force_sharded
    if world > 1 or force_sharded:
        # rank 0 trains; centroids and codebook are broadcast so that every rank
        # quantises with bit-identical tables (k-means uses atomic scatter-adds)
        cent = torch.empty((args.nlist, d), dtype=torch.float32, device=dev)
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
        if use_shards:
            ids = torch.arange(rank, args.corpus, world, device=dev)
            index.add_with_ids(x[rank::world].contiguous(), ids)
        else:
            index.add(x)                     # replica: the whole corpus on every rank
    else:
        index.train(x)
        index.add(x)
    index.nprobe = args.nprobe
    sharded = ShardedIndex(index, shard_coarse=bool(args.shard_coarse)) if use_shards else None
    log(f"[rank {rank}] setup {time.time() - t0:.1f}s ntotal={index.ntotal}")

    NB = 16                               # pool of distinct query batches (different per rank)
    qpool = synth.queries_cuda(x, NB * args.batch * world, seed=4321).view(NB, world, args.batch, d)
    my_q = [qpool[b, rank].contiguous() for b in range(NB)]
    nq_out = args.batch
    # steps are independent query batches: they are issued round-robin on S
    # streams (each with its own output buffers and library workspaces) so that
    # consecutive batches overlap on the GPU, as a serving loop would run them
    S = max(1, args.streams) if not use_shards else 1
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else [torch.cuda.current_stream(dev)]
    Ds = [torch.empty((nq_out, k), dtype=torch.float32, device=dev) for _ in range(S)]
    Is = [torch.empty((nq_out, k), dtype=torch.int64, device=dev) for _ in range(S)]
    D, I = Ds[0], Is[0]
    torch.cuda.synchronize()

    sptr = [int(s_.cuda_stream) for s_ in streams]

    refine = None
    if args.refine > 1 and sharded is None:
        # second stage over the raw vectors (4 GB at cfg2), candidates in per-stream scratch
        flat_r = faiss.IndexFlatIP(d, device=local_rank)
        flat_r.add(x)
        refine = faiss.IndexRefineFlat(index, flat_r)
        refine.k_factor = args.refine
        kb = k * args.refine
        cDs = [torch.empty((nq_out, kb), dtype=torch.float32, device=dev) for _ in range(S)]
        cIs = [torch.empty((nq_out, kb), dtype=torch.int64, device=dev) for _ in range(S)]

    def step(b):
        if refine is not None:
            j = b % S
            refine.search_into(my_q[b % NB], k, Ds[j], Is[j], cDs[j], cIs[j], sptr[j])
        elif sharded is None:
            j = b % S
            index.search_into(my_q[b % NB], k, Ds[j], Is[j], None, sptr[j])
        else:
            sharded.search_into(my_q[b % NB], k, D, I)

    for b in range(max(args.warmup, 2 * S)):
        step(b)
    torch.cuda.synchronize()
    # settle: the timed region is only a few milliseconds at the default K, shorter than
    # the GPU's clock ramp after the idle setup phase -- run untimed steps for a fixed
    # wall time first (part of setup; the W warmup steps and the K timed steps follow)
    t_settle = time.perf_counter()
    b = 0
    while time.perf_counter() - t_settle < args.settle_ms * 1e-3:
        for _ in range(64):
            step(b)
            b += 1
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
    t_issue = time.perf_counter() - t1     # host time to issue the steps (diagnostic only)
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
    # HBM traffic per launch: PMC counters cannot be collected from inside the timed
    # process; the value comes from the committed separate --pmc passes of this same
    # command (profiles/r01_cfg2_scan_pmc.json) and is reported only for that config
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_cfg2_scan_pmc.json")))
        if (args.corpus, args.nlist, args.batch, args.nprobe, k, world) == (1_000_000, 4096, 64, 16, 10, 1) \
                and not os.environ.get("MI_NSLICE"):
            traffic = int(pmc["corrected_bytes_per_launch"])
    except Exception:
        traffic = None
    roofline = {"kernel": "scan_kernel<64,8>", "bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0,
                "unit": "GB/s", "frac": round(achieved / 8000.0, 4), "traffic": traffic,
                "traffic_source": "rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, separate passes, gfx950 x2 read correction "
                                  "(profiles/r01_cfg2_scan_pmc.json)" if traffic else None,
                "bytes_per_launch": int(scan_bytes), "avg_launch_ms": round(scan_ms, 5),
                "launches": args.steps}

    out = None
    if rank == 0:
        # ---- recall@10 against exact search (untimed)
        recall = None
        if not args.no_recall and world == 1 and sharded is None:
            flat = faiss.IndexFlatIP(d, device=local_rank)
            flat.add(x)
            hits = tot = 0
            for b in range(4):
                _, Ia = (refine if refine is not None else index).search(my_q[b], k)
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
            "metric": "queries/sec, IVF-PQ search (IVF%d,PQ64%s, %dx1024-d, batch %d, nprobe %d, k %d)"
                      % (args.nlist, ",RFlat x%d" % args.refine if refine is not None else "", args.corpus, args.batch,
                         args.nprobe, k),
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "cfg2: 1Mx1024 clustered synthetic corpus, IVF4096,PQ64, batch-64 queries",
                       "corpus": args.corpus, "nlist": args.nlist, "M": 64, "nprobe": args.nprobe,
                       "k": k, "batch_per_rank": args.batch, "global_batch": args.batch * world,
                       "parallelism": "1 GPU" if world == 1 else (f"vector-sharded x{world} + all-gather top-k" if use_shards
                                                                  else f"query-parallel replicas x{world} (no collective)"),
                       "launch": "eager (a hipGraph replay of the step measured slower)", "streams": S,
                       "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                       "host_issue_ms_per_step": round(t_issue / args.steps * 1e3, 5)},
            "recall_at_10": None if recall is None else round(recall, 4),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def encode_workload(args, np, torch, dist, world, rank, local_rank, dev):
    """cfg3: stella_en_1.5B_v5 architecture (random-init bf16 weights -- no
    checkpoint is reachable from the build/bench boxes), synthetic abstracts with
    clipped log-normal token counts (median 220, max 512).  A step encodes one
    batch of `--encode-batch` abstracts: embedding gather, 28 decoder layers,
    mean pooling, Dense 1536->1024, L2 normalise; token ids start on the host
    (as they do after tokenisation), embeddings stay in HBM.  N > 1: replicas,
    every rank encodes its own batches (no collective on this path)."""
    import abstracts_search_amd.sentence_transformers as st
    cfg = dict(st.STELLA_EN_1_5B_V5)
    model = st.SentenceTransformer(config=cfg, device=f"cuda:{local_rank}")
    g = torch.Generator(device=dev).manual_seed(7)

    def rnd(shape, scale):
        return (torch.randn(shape, generator=g, device=dev) * scale).bfloat16()

    H, I = cfg["hidden"], cfg["intermediate"]
    qc, kc = cfg["n_heads"] * cfg["head_dim"], cfg["n_kv_heads"] * cfg["head_dim"]
    model.load_weights({"embed_tokens.weight": rnd((cfg["vocab_size"], H), 0.3), "norm.weight": torch.ones(H, device=dev),
                        "dense.weight": rnd((cfg["dense_out"], H), H ** -0.5), "dense.bias": torch.zeros(cfg["dense_out"], device=dev)})
    for l in range(cfg["n_layers"]):
        p = f"layers.{l}."
        model.load_weights({
            p + "input_layernorm.weight": torch.ones(H, device=dev), p + "post_attention_layernorm.weight": torch.ones(H, device=dev),
            p + "self_attn.q_proj.weight": rnd((qc, H), H ** -0.5), p + "self_attn.q_proj.bias": rnd((qc,), 0.1),
            p + "self_attn.k_proj.weight": rnd((kc, H), H ** -0.5), p + "self_attn.k_proj.bias": rnd((kc,), 0.1),
            p + "self_attn.v_proj.weight": rnd((kc, H), H ** -0.5), p + "self_attn.v_proj.bias": rnd((kc,), 0.1),
            p + "self_attn.o_proj.weight": rnd((H, qc), qc ** -0.5), p + "mlp.gate_proj.weight": rnd((I, H), H ** -0.5),
            p + "mlp.up_proj.weight": rnd((I, H), H ** -0.5), p + "mlp.down_proj.weight": rnd((H, I), I ** -0.5)})
    bs = args.encode_batch
    rng = np.random.default_rng(7 + rank)
    NBATCH = 8
    batches = []
    for _ in range(NBATCH):
        lens = np.clip(np.exp(rng.normal(np.log(220), 0.45, bs)), 8, 512).astype(int)
        batches.append([rng.integers(0, cfg["vocab_size"], L).tolist() for L in lens])
    ntok = [sum(len(t) for t in b) for b in batches]

    def step(i):
        return model.encode_tokens(batches[i % NBATCH], batch_size=bs, normalize_embeddings=True, as_tensor=True)

    for i in range(max(args.warmup, 1)):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t1
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    toks = sum(ntok[(args.warmup + i) % NBATCH] for i in range(args.steps))
    # roofline of the dominant kernel (bf16 MFMA GEMMs): HIP events around the GEMM launches
    model.profile(True)
    step(0)
    torch.cuda.synchronize()
    pr = model.profile_read()
    model.profile(False)
    tf = pr["gemm_flops"] / (pr["gemm_ms"] * 1e-3) / 1e12
    roofline = {"kernel": "gemm_bf16_ring_kernel<EPI,8,4,2,4,4> (QKV / O / gate-up+SwiGLU / down, 112 launches per step)",
                "bound": "mfma", "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s",
                "frac": round(tf / 2500.0, 4), "traffic": None,
                "flops_per_step": pr["gemm_flops"], "gemm_ms_per_step": round(pr["gemm_ms"], 3)}
    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline:
            cpu = encode_cpu_baseline(model, cfg, batches, torch, np)
        print(json.dumps({
            "metric": "abstracts/sec, stella_en_1.5B_v5 bf16 batch encode (synthetic abstracts, median 220 tokens)",
            "value": round(args.steps * bs * world / dt, 1), "unit": "abstracts/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic (random-init weights of the real architecture, synthetic token ids)",
            "config": {"workload": "cfg3: stella_en_1.5B_v5 bf16 batch encode", "batch": bs,
                       "tokens_per_sec": round(toks * world / dt, 0), "parallelism": "replicas" if world > 1 else "1 GPU"},
            "roofline": roofline, "cpu_baseline": cpu}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def encode_cpu_baseline(model, cfg, batches, torch, np):
    """oracle port (torch fp32 on the host cores) on a bounded sample: the same
    architecture cut to 2 layers and a 4096-row vocabulary (a 1.5 B-parameter fp32
    copy is 6 GB and tens of seconds per batch), scaled by 28/2 layers."""
    from oracle import encoder_oracle as E
    small = dict(cfg)
    small["n_layers"], small["vocab_size"] = 2, 4096
    W = E.synth_weights(E.EncoderConfig(**small), 3)
    toks = [[t % 4096 for t in s] for s in batches[0][:16]]
    cu = np.concatenate([[0], np.cumsum([len(t) for t in toks])])
    ids = np.concatenate(toks)
    with torch.no_grad():
        E.encode(E.EncoderConfig(**small), W, ids, cu, True)
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            E.encode(E.EncoderConfig(**small), W, ids, cu, True)
        dt = (time.perf_counter() - t0) / reps
    full = dt * cfg["n_layers"] / 2
    return {"value": round(len(toks) / full, 2), "unit": "abstracts/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"16 abstracts through 2 of 28 layers (oracle/encoder_oracle.py, torch fp32), "
                                      f"{dt:.2f}s per pass, scaled x14 to full depth"}


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
    # all 16 distinct batches per call (1024 queries) so that every host core has
    # work: the oracle parallelises over queries, as faiss-cpu does
    qs = np.concatenate([q.cpu().numpy() for q in my_q])
    t0 = time.perf_counter()
    O.search(qs, cent, cb, off, codes, ids, args.nprobe, args.k)     # warm + calibrate
    one = time.perf_counter() - t0
    reps = int(max(2, min(200, 12.0 / max(one, 1e-4))))
    t0 = time.perf_counter()
    for r in range(reps):
        O.search(qs, cent, cb, off, codes, ids, args.nprobe, args.k)
    dt = time.perf_counter() - t0
    return {"value": round(reps * qs.shape[0] / dt, 1), "unit": "queries/s", "cores": O.num_threads(),
            "kind": "port",
            "sample": f"{reps} calls of {qs.shape[0]} queries (16 batches of {args.batch}), same index/nprobe/k, {dt:.1f}s "
                      f"(oracle/ivfpq_oracle.c, OpenMP over queries; faiss-cpu not installable here)"}


if __name__ == "__main__":
    main()
