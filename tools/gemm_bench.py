"""bf16 MFMA GEMM kernel throughput at the encoder's shapes (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import abstracts_search_amd.sentence_transformers as st
M = int(os.environ.get("M", 32768))
for name, N, K in (("qkv", 2048, 1536), ("o", 1536, 1536), ("gate_up", 17920, 1536), ("down", 1536, 8960), ("square", 4096, 4096)):
    A = torch.randn((M, K), device="cuda").bfloat16(); W = (torch.randn((N, K), device="cuda") / K ** 0.5).bfloat16()
    if os.environ.get("ZERO") == "1":          # zero operands toggle no data lines: what the same schedule does without the power throttle
        A.zero_(); W.zero_()
    for _ in range(3): C = st.gemm_bf16(A, W)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps): C = st.gemm_bf16(A, W)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:8s} M={M} N={N:6d} K={K:5d}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)
    ref = torch.matmul(A, W.T)  # hipBLASLt (reference point only, not used by the product)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): ref = torch.matmul(A, W.T)
    e1.record(); torch.cuda.synchronize()
    ms2 = e0.elapsed_time(e1) / reps
    print(f"{'':8s} torch.matmul (hipBLASLt) {ms2*1e3:8.1f} us  {2*M*N*K/ms2/1e9:7.1f} TFLOP/s   maxdiff {(C.float()-ref.float()).abs().max().item():.3f}", flush=True)
