"""Throughput of the OpenAlex text filter: the reference binary (oracle/_ref/oa_jsonl, fgetc loop,
one core) against abstracts-search_amd/oa_jsonl_mt at 1/2/4/8 threads, same input through a
pipe-free file redirect, outputs compared.  CPU only.  usage: python tools/oa_jsonl_bench.py [records]"""
import os, random, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oa_jsonl_corpus import random_record
import abstracts_search_amd._native as nat

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
rng = random.Random(1)
base = [random_record(rng) for _ in range(2000)]
with tempfile.NamedTemporaryFile("w", suffix=".jsonl", delete=False, encoding="utf-8") as f:
    for i in range(n):
        f.write(base[i % len(base)] + "\n")
    path = f.name
size = os.path.getsize(path)
print(f"{n} records, {size/1e6:.0f} MB, {os.cpu_count()} host cores")
ref, mine = os.path.join(ROOT, "oracle", "_ref", "oa_jsonl"), nat.build_oa_jsonl()

def run(cmd):
    best, out = 1e9, None
    for _ in range(3):
        t = time.perf_counter()
        out = subprocess.run(cmd, stdin=open(path, "rb"), capture_output=True, check=True).stdout
        best = min(best, time.perf_counter() - t)
    return best, out

want = None
if os.path.exists(ref):
    t, want = run([ref])
    print(f"reference oa_jsonl (1 core)      {t:6.2f} s  {size/t/1e6:7.0f} MB/s  {n/t/1e3:7.0f} k records/s  -> {want.count(10)} documents")
for th in (1, 2, 4, 8, 16):
    if th > 2 * (os.cpu_count() or 1):
        break
    t, out = run([mine, "-t", str(th)])
    same = "" if want is None else ("  output identical" if out == want else "  OUTPUT DIFFERS")
    print(f"oa_jsonl_mt -t {th:<2d}                {t:6.2f} s  {size/t/1e6:7.0f} MB/s  {n/t/1e3:7.0f} k records/s{same}")
os.unlink(path)
