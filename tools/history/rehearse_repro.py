"""HISTORY (rounds 1-3): kept for the record of how a number in DESIGN_HISTORY.md / profiles/ was produced.  NOT maintained: knobs it
sets may no longer exist (MI_RERANK, MI_REFINE_DEBUG, MI_SCAN_DEBUG ... are silent no-ops now) and paths may have moved."""
"""Repeat the N-rank rehearsal of bench.py and, when a run stalls, dump every rank's Python stack (SIGABRT + faulthandler)."""
import os, signal, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 150
for rep in range(reps):
    env = dict(os.environ, BENCH_REHEARSAL="1", PYTHONFAULTHANDLER="1", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-X", "faulthandler", "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(29700 + rep), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1",
           "--corpus", str(2 * 1048576), "--nlist", "1024", "--batch", "256", "--nprobe", "16", "--no-encode"]
    t0 = time.time()
    p = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=limit)
        print(f"rep {rep}: rc {p.returncode} in {time.time()-t0:.0f}s", flush=True)
        if p.returncode:
            print(err[-3000:])
    except subprocess.TimeoutExpired:
        print(f"rep {rep}: STALLED after {limit}s -- dumping stacks", flush=True)
        os.killpg(p.pid, signal.SIGABRT)
        try:
            out, err = p.communicate(timeout=30)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)
            out, err = p.communicate()
        lines = err.splitlines()
        keep = [l for l in lines if "File" in l or "Thread" in l or "rank" in l.lower() or "Fatal" in l]
        print("\n".join(keep[-120:]))
        break
