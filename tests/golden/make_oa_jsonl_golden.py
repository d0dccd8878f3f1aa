"""Generates tests/golden/oa_jsonl_input.jsonl (synthetic OpenAlex-style `works` records:
hand-written edge cases + seeded random records) and tests/golden/oa_jsonl_expected.jsonl =
the output of the REAL reference tool on it (oracle/_ref/oa_jsonl, compiled from
/root/reference/oa_jsonl.c by `make -C oracle ref`).  Run in the build container:

    python tests/golden/make_oa_jsonl_golden.py
"""
import json
import os
import random
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oa_jsonl_corpus import edge_cases, random_record   # noqa: E402


def main():
    rng = random.Random(20260928)
    lines = edge_cases() + [random_record(rng) for _ in range(400)]
    src = os.path.join(HERE, "oa_jsonl_input.jsonl")
    with open(src, "w", encoding="utf-8", newline="") as f:
        f.write("\n".join(lines) + "\n")
    ref = os.path.join(ROOT, "oracle", "_ref", "oa_jsonl")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    out = subprocess.run([ref], stdin=open(src, "rb"), capture_output=True, check=True).stdout
    with open(os.path.join(HERE, "oa_jsonl_expected.jsonl"), "wb") as f:
        f.write(out)
    print(f"{len(lines)} records in, {out.count(10)} documents out, {len(out)} bytes")


if __name__ == "__main__":
    main()
