"""Time IndexScalarQuantizer(QT_8bit).rerank alone at the recall point's shape (1024 queries x 4 640 random candidates of a
64 M-row store), for A/B runs of compile-time variants of rerank_sq8_kernel:
    MI_IVFPQ_LIB=libmi_ivfpq_nst2.so python tools/micro/rerank_sq8_ab.py
"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
faiss = importlib.import_module("abstracts-search_amd.faiss")

d, nb, nq, kc, k = 1024, int(os.environ.get("NB", 64_000_000)), 1024, 4640, 10
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
ix = faiss.IndexScalarQuantizer(d, faiss.ScalarQuantizer.QT_8bit)
ix.train(torch.randn(65536, d, device=dev, generator=g))
for i in range(0, nb, 1 << 20):
    ix.add(torch.randn(min(1 << 20, nb - i), d, device=dev, generator=g))
q = torch.randn(nq, d, device=dev, generator=g)
cand = torch.randint(0, nb, (nq, kc), device=dev, generator=g, dtype=torch.int64)
D = torch.empty(nq, k, device=dev); I = torch.empty(nq, k, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
for _ in range(3):
    ix.rerank(q, cand, k, D, I, st)
best = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ix.rerank(q, cand, k, D, I, st)
    e1.record(); e1.synchronize()
    best.append(e0.elapsed_time(e1) / 20)
ms = sorted(best)[len(best) // 2]
print("%s: rerank %.4f ms (median of 5 x 20), %.0f GB/s of candidate rows; checksum %d" % (
    os.environ.get("MI_IVFPQ_LIB", "default"), ms, nq * kc * d / ms / 1e6, int(I.sum().item())))
