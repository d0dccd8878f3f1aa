"""The import alias of INTEGRATION.md section 1 (integration/sitecustomize.py), exercised in a
subprocess the way the reference pipeline would pick it up: PYTHONPATH + one environment
variable, caller code that only knows the names `faiss` / `sentence_transformers`."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _env(on=True):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "integration"), ROOT, env.get("PYTHONPATH", "")])
    if on:
        env["ABSTRACTS_SEARCH_BACKEND"] = "mi355x"
    else:
        env.pop("ABSTRACTS_SEARCH_BACKEND", None)
    return env


def test_alias_resolves_the_reference_names():
    code = ("import faiss, sentence_transformers as st;"
            "print(faiss.__name__, st.__name__, hasattr(faiss, 'index_factory'), hasattr(faiss, 'read_index'),"
            " hasattr(faiss, 'IndexIVFPQ'), hasattr(faiss, 'ParameterSpace'), hasattr(st, 'SentenceTransformer'))")
    out = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ["abstracts_search_amd.faiss", "abstracts_search_amd.sentence_transformers"] + ["True"] * 5
    # without the switch the names are whatever the box has (here: nothing)
    off = subprocess.run([sys.executable, "-c", "import faiss"], env=_env(False), capture_output=True, text=True, timeout=300)
    assert off.returncode != 0 or "abstracts_search_amd" not in off.stdout


@pytest.mark.gpu
def test_sidecar_shaped_script_runs_unchanged(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sidecar_shaped_script.py"), str(tmp_path)],
                         env=_env(), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["module"] == "abstracts_search_amd.faiss"
    assert r["ntotal"] == 20000 and r["shape"] == [50, 10] and r["sorted"] and r["ctor_signature_ok"]
    print(r)
    assert r["self_hit"] > 0.2 and r["recall_at_10"] > 0.1       # plumbing check (PQ8 over tight clusters), parity is elsewhere
    assert os.path.exists(tmp_path / "index.faiss")
