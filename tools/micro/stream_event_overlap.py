"""Do two HIP streams linked by events overlap on this runtime?  A: k(b) -> event; B: wait event -> k'(b), for 10 batches, each
kernel a ~5 ms single-thread spin (torch.cuda._sleep).  Overlapped: ~55 ms; serialised: ~100 ms.  Also the same with whole
batches round-robin on two streams, and with more streams alive than GPU_MAX_HW_QUEUES.  usage: python tools/micro/stream_event_overlap.py"""
import os, time, torch
dev = torch.device("cuda", 0)
cyc = int(5e-3 * 100e6)            # s_memtime ticks at 100 MHz
def t(fn, n=10):
    fn(0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range(n): fn(b)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
def make_pipe(prio=0, depth=3):
    A, B = torch.cuda.Stream(device=dev, priority=prio), torch.cuda.Stream(device=dev)
    evA = [torch.cuda.Event() for _ in range(depth)]; evB = [torch.cuda.Event() for _ in range(depth)]
    def step(b):
        j = b % depth
        A.wait_event(evB[j])
        with torch.cuda.stream(A): torch.cuda._sleep(cyc)
        evA[j].record(A)
        B.wait_event(evA[j])
        with torch.cuda.stream(B): torch.cuda._sleep(cyc)
        evB[j].record(B)
    return step
def make_rr(ns):
    ss = [torch.cuda.Stream(device=dev) for _ in range(ns)]
    def step(b):
        with torch.cuda.stream(ss[b % ns]): torch.cuda._sleep(cyc); torch.cuda._sleep(cyc)
    return step
torch.cuda._sleep(cyc); torch.cuda.synchronize()
t0 = time.perf_counter(); torch.cuda._sleep(cyc); torch.cuda.synchronize(); one = (time.perf_counter() - t0) * 1e3
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}; one spin kernel {one:.2f} ms; 10 batches x 2 kernels serial = {20 * one:.1f} ms")
print(f"pipe (fresh process, first streams): {t(make_pipe()):.1f} ms")
print(f"pipe prio -1: {t(make_pipe(-1)):.1f} ms")
print(f"rr2: {t(make_rr(2)):.1f} ms")
keep = [make_rr(4) for _ in range(3)]
for k in keep: t(k, 4)
print(f"pipe after 14 more streams were used: {t(make_pipe()):.1f} ms")
