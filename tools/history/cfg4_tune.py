"""HISTORY (rounds 1-3): kept for the record of how a number in DESIGN_HISTORY.md / profiles/ was produced.  NOT maintained: knobs it
sets may no longer exist (MI_RERANK, MI_REFINE_DEBUG, MI_SCAN_DEBUG ... are silent no-ops now) and paths may have moved."""
"""The `tune` step on one shard of BASELINE.json configs[3] (207M x 1024, IVF65536,PQ64 over
8 GPUs = 25.9M vectors on this GPU) with a flat refine stage over the shard's raw vectors
(106 GB of the 288 GB HBM): which (nprobe, k_factor_rf) reaches recall@10 >= 0.95 against
the exact search over the same 25.9M vectors, and at what rate.  GPU box; ~4 minutes.
usage: python tools/cfg4_tune.py [nq] [batch] [target]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import abstracts_search_amd.faiss as faiss
import abstracts_search_amd.autotune as autotune
import abstracts_search_amd.synth as synth

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
target = float(sys.argv[3]) if len(sys.argv) > 3 else 0.95
N = int(os.environ.get("NSHARD_VEC", 207_000_000 // 8)); NLIST = int(os.environ.get("NLIST", 65536))
CH = 65536 * 16
t0 = time.time()
base = faiss.IndexIVFPQ(1024, NLIST, 64, 8, faiss.METRIC_INNER_PRODUCT)
xs = synth.corpus_cuda(4 * 1024 * 1024, 1024)
base.cp.niter = int(os.environ.get("NITER", 4))
base.train(xs)
print(f"train {time.time()-t0:.1f}s", flush=True)
idx = faiss.IndexRefineFlat(base)
idx.refine_index.reserve(N)
t1 = time.time()
done = 0
while done < N:
    m = min(CH, N - done)
    idx.add(synth.corpus_cuda(m, 1024, row0=done))
    done += m
torch.cuda.synchronize()
print(f"add {N} vectors to IVF-PQ + flat store {time.time()-t1:.1f}s, HBM in use {torch.cuda.mem_get_info()[1]/1e9 - torch.cuda.mem_get_info()[0]/1e9:.1f} GB", flush=True)
q = synth.queries_cuda(xs, nq)
del xs
t2 = time.time()
gt = torch.cat([idx.refine_index.search(q[i:i + 64], 10)[1] for i in range(0, nq, 64)])
torch.cuda.synchronize()
print(f"exact top-10 of {nq} queries over {N} vectors {time.time()-t2:.1f}s", flush=True)
ps = faiss.ParameterSpace()
ps.initialize(idx)
ps.parameter_ranges[0].values = [float(v) for v in os.environ.get('KF', '1,6,16,32,64,100').split(',')]   # k_factor_rf (one scan pass per 64 of k_base)
ps.parameter_ranges[1].values = [float(v) for v in os.environ.get('NPROBE', '8,16,32,64,256').split(',')]
ps.batchsize = batch
ps.min_test_duration = 0.0
ps.display()
crit = faiss.IntersectionCriterion(nq, 10)
crit.set_groundtruth(None, gt)
ops = ps.explore(idx, q, crit)
print(f"\nPareto front ({nq} queries in batches of {batch}, one stream, one shard-GPU):")
for p in ops.optimal_pts[1:]:
    print(f"  recall@10 {p.perf:.4f}  {p.t * 1e3:9.2f} ms  {nq / p.t:10.0f} QPS/shard-GPU   {p.key}")
doc = autotune.write_params("gpurun_out/params_cfg4.json", ops, min_perf=target)
print(f"target recall {target}: {doc['index_parameters']} -> recall {doc['perf']:.4f}, {nq / doc['t']:.0f} QPS/shard-GPU")
print("max torch mem GB", torch.cuda.max_memory_allocated() / 1e9)
