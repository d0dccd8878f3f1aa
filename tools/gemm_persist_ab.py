"""Sustained throughput of the slab GEMM launched one tile per workgroup (MI_GEMM_PERSIST=0), as one persistent workgroup
per CU with a stream-K tail (default) and of hipBLASLt, on the encoder's shapes: each contender runs back to back for SECS
seconds, alternating A B C A B C on ONE box (GPU box; plain-store epilogue -- the encoder's fused epilogues are timed by
`bench.py --workload encode` under the same knob)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import abstracts_search_amd.sentence_transformers as st
M = int(os.environ.get("M", 27958))
SECS = float(os.environ.get("SECS", 0.4))
shapes = {"qkv": (2048, 1536), "o": (1536, 1536), "gate_up": (17920, 1536), "down": (1536, 8960), "square": (4096, 4096)}


def set_persist(on):
    if on:
        os.environ.pop("MI_GEMM_PERSIST", None)
    else:
        os.environ["MI_GEMM_PERSIST"] = "0"
    st.reload_env()


for name in os.environ.get("SHAPES", "gate_up,down,qkv,o,square").split(","):
    N, K = shapes[name]
    A = torch.randn((M, K), device="cuda").bfloat16(); W = (torch.randn((N, K), device="cuda") / K ** 0.5).bfloat16()
    C = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    fns = {"tile/wg": (False, lambda: st.gemm_bf16(A, W)), "persist": (True, lambda: st.gemm_bf16(A, W)),
           "hipblaslt": (None, lambda: torch.matmul(A, W.T, out=C))}
    for who, (p, f) in fns.items():
        if p is not None:
            set_persist(p)
        f()
    torch.cuda.synchronize()
    line = []
    for rnd in range(3):
        for who, (p, f) in fns.items():
            if p is not None:
                set_persist(p)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 0
            t0 = time.time()
            e0.record()
            while time.time() - t0 < SECS:
                for _ in range(20):
                    f()
                reps += 20
                torch.cuda.current_stream().synchronize() if reps % 200 == 0 else None
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            line.append(f"{who} {2*M*N*K/ms/1e9:6.0f}")
    set_persist(True)
    print(f"{name:8s} M={M} N={N:6d} K={K:5d} TFLOP/s per {SECS:.1f}-s block: " + " | ".join(line), flush=True)
print("sk_giveups", st.debug_counter("sk_giveups"))
