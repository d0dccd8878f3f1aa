"""the buffered, multi-threaded OpenAlex text filter
(abstracts-search_amd/csrc/oa_jsonl_mt.c) against the REAL reference tool: byte for byte
equal to the committed output of oracle/_ref/oa_jsonl (compiled from the reference's own
oa_jsonl.c; generator: tests/golden/make_oa_jsonl_golden.py), for every thread count and
block size; and, where the reference binary itself is present (this container), against it
on a larger seeded corpus.  CPU only."""
import os
import random
import subprocess

import pytest

from conftest import ROOT

GOLD = os.path.join(ROOT, "tests", "golden")
REF = os.path.join(ROOT, "oracle", "_ref", "oa_jsonl")


@pytest.fixture(scope="module")
def tool():
    import abstracts_search_amd._native as nat
    return nat.build_oa_jsonl()


def run(exe, data: bytes, *args) -> bytes:
    return subprocess.run([exe, *args], input=data, capture_output=True, check=True, timeout=300).stdout


@pytest.mark.parametrize("args", [("-t", "1"), ("-t", "3"), ("-t", "8"), ("-t", "4", "-B", "64"), ("-t", "2", "-B", "5000"),
                                  ("-t", "1", "-B", "1000000")])
def test_matches_the_reference_output(tool, args):
    src = open(os.path.join(GOLD, "oa_jsonl_input.jsonl"), "rb").read()
    want = open(os.path.join(GOLD, "oa_jsonl_expected.jsonl"), "rb").read()
    assert run(tool, src, *args) == want


def test_stream_edges(tool):
    rec = b'{"id":"W1","title":"t","language":"en","abstract_inverted_index":{"a":[0]}}'
    out = b'{"id":"W1","document":"t a"}\n'
    assert run(tool, b"") == b""
    assert run(tool, rec) == out                       # no trailing newline
    assert run(tool, rec + b"\n") == out
    assert run(tool, rec + b"\n\n" + rec + b"\n") == out          # an empty line ends the stream (oa_jsonl.c:357-360)
    assert run(tool, (rec + b"\n") * 1000, "-t", "5", "-B", "777") == out * 1000


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/oa_jsonl is built where /root/reference exists (make -C oracle ref)")
def test_against_the_reference_binary_on_a_larger_corpus(tool):
    from oa_jsonl_corpus import random_record
    rng = random.Random(7)
    data = ("\n".join(random_record(rng) for _ in range(5000)) + "\n").encode("utf-8")
    want = run(REF, data)
    assert want.count(b"\n") > 2000
    for args in (("-t", "1"), ("-t", "8"), ("-t", "3", "-B", "100000")):
        assert run(tool, data, *args) == want
