"""The reference's `index tune` step (Makefile:32 -> params.json) on the bench corpus:
faiss-style ParameterSpace exploration (nprobe x k_factor_rf) of IVF4096,PQ64,RFlat against
exact neighbours, Pareto front of recall@10 vs time, params for a recall target.  GPU box.
usage: python tools/tune.py [n_vectors] [nq] [batch] [target_recall] [out.json]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import abstracts_search_amd.faiss as faiss
import abstracts_search_amd.autotune as autotune
import abstracts_search_amd.synth as synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
target = float(sys.argv[4]) if len(sys.argv) > 4 else 0.95
out = sys.argv[5] if len(sys.argv) > 5 else "gpurun_out/params.json"
nlist, k = 4096, 10
x = synth.corpus_cuda(n, 1024)
idx = faiss.index_factory(1024, f"IVF{nlist},PQ64,RFlat", faiss.METRIC_INNER_PRODUCT)
idx.base_index.cp.niter = idx.base_index.pq.cp.niter = 10
idx.train(x)
idx.add(x)
q = synth.queries_cuda(x, nq, seed=4321)
gt = torch.cat([idx.refine_index.search(q[i:i + 256], k)[1] for i in range(0, nq, 256)])
ps = faiss.ParameterSpace()
ps.initialize(idx)
ps.parameter_ranges[0].values = [1.0, 2.0, 4.0, 6.0, 8.0, 16.0]       # k_factor_rf (k_base > 64: two scan passes)
ps.parameter_ranges[1].values = [float(1 << i) for i in range(3, 11)]  # nprobe 8 .. 1024
ps.batchsize = batch
ps.min_test_duration = 0.05
ps.verbose = 1
ps.display()
crit = faiss.IntersectionCriterion(nq, k)
crit.set_groundtruth(None, gt)
ops = ps.explore(idx, q, crit)
ops.display()
print(f"\nPareto front ({nq} queries in batches of {batch}, one stream, device-resident queries and results):")
for p in ops.optimal_pts[1:]:
    print(f"  recall@10 {p.perf:.4f}  {p.t * 1e3:8.2f} ms  {nq / p.t:10.0f} QPS   {p.key}")
doc = autotune.write_params(out, ops, min_perf=target)
print(f"target recall {target}: {doc['index_parameters']} -> recall {doc['perf']:.4f}, {nq / doc['t']:.0f} QPS; wrote {out}")
