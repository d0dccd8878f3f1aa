"""SURVEY section 5 (race detection / sanitizers: the reference has none, `assert` only): the two pieces of plain-C
test / host infrastructure run under AddressSanitizer + UndefinedBehaviorSanitizer in the CPU suite.

  * oracle/ivfpq_oracle.c -- the checker every parity claim rests on: its own CPU tests (golden vectors, float64 brute
    force, edge cases, L2, SQ8, merges; tests/test_oracle_ivfpq.py) re-run in a subprocess against a
    `-fsanitize=address,undefined -fno-sanitize-recover` build of the same source;
  * csrc/oa_jsonl_mt.c -- the multi-threaded text filter: the golden input at several thread counts and block sizes
    through a sanitized build, output byte-identical to the reference binary's.
(The HIP kernels have no compute-sanitizer equivalent on ROCm: they are covered by run-twice / shuffled-order bit
comparisons in the GPU suite.)"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]


def _cc():
    cc = shutil.which("gcc") or shutil.which("cc")
    if not cc:
        pytest.skip("no C compiler")
    return cc


def _runtime(cc, name):
    p = subprocess.run([cc, f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    if not p or not os.path.isabs(p) or not os.path.exists(p):
        pytest.skip(f"{name} not installed with this compiler")
    return p


def test_oracle_under_asan_ubsan(tmp_path):
    cc = _cc()
    so = str(tmp_path / "libivfpq_oracle_san.so")
    subprocess.check_call([cc, *SAN, "-mavx2", "-mfma", "-ffp-contract=off", "-fno-math-errno", "-fopenmp", "-fPIC",
                           "-shared", "-o", so, os.path.join(ROOT, "oracle", "ivfpq_oracle.c"), "-lm"])
    env = dict(os.environ, MI_ORACLE_SO=so, LD_PRELOAD=_runtime(cc, "libasan.so") + " " + _runtime(cc, "libubsan.so"),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               PYTHONPATH=ROOT, OMP_NUM_THREADS="4", HYPOTHESIS_MAX_EXAMPLES="10")
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_ivfpq.py")],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = (p.stdout + p.stderr)[-4000:]
    assert p.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
    assert " passed" in p.stdout, tail


def test_oa_jsonl_mt_under_asan_ubsan(tmp_path):
    cc = _cc()
    exe = str(tmp_path / "oa_jsonl_mt_san")
    subprocess.check_call([cc, *SAN, "-pthread", "-o", exe, os.path.join(ROOT, "abstracts-search_amd", "csrc", "oa_jsonl_mt.c")])
    gold = os.path.join(ROOT, "tests", "golden")
    want = open(os.path.join(gold, "oa_jsonl_expected.jsonl"), "rb").read()
    data = open(os.path.join(gold, "oa_jsonl_input.jsonl"), "rb").read()
    # (leak checking off: the tool's block / carry buffers are process-lifetime by design and die with the process)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    for args in (("-t", "1"), ("-t", "4"), ("-t", "3", "-B", "4096"), ("-t", "8", "-B", "256"), ("-t", "2", "-B", "64")):
        p = subprocess.run([exe, *args], input=data, env=env, capture_output=True, timeout=120)
        assert p.returncode == 0, p.stderr.decode()[-3000:]
        assert b"Sanitizer" not in p.stderr and b"runtime error" not in p.stderr, p.stderr.decode()[-3000:]
        assert p.stdout == want, args
