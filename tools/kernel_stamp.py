"""Provenance stamps for the committed counter files (profiles/*_pmc.json): the sha1 of the SOURCE TEXT of the kernel a PMC pass
profiled (its definition in csrc/*.h, from the `__global__` line to the closing brace in column 0), so that bench.py can tell a
`traffic` number that belongs to the kernel it is timing from one that belongs to an older kernel (and drop the latter instead of
quoting it).  Device helpers a kernel calls are outside the stamp: a change there that moves bytes needs a new PMC pass anyway.

    python tools/kernel_stamp.py profiles/r05_cfg4_scan_pmc.json ivfpq_kernels.h scan_kernel     # writes "kernel_source" into the file
"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "abstracts-search_amd", "csrc")


def kernel_source_sha1(header: str, kernel: str) -> str | None:
    """sha1 of the definition of `kernel` in csrc/<header>; None when it is not found"""
    try:
        lines = open(os.path.join(CSRC, header)).read().split("\n")
    except OSError:
        return None
    pat = re.compile(r"\b" + re.escape(kernel) + r"\s*\(")
    for i, l in enumerate(lines):
        if pat.search(l) and ("__global__" in l or (i and "__global__" in lines[i - 1]) or (i > 1 and "__global__" in lines[i - 2])):
            j = i
            while j < len(lines) and lines[j] != "}":
                j += 1
            start = i
            while start and ("__global__" not in lines[start]):
                start -= 1
            return hashlib.sha1("\n".join(lines[start:j + 1]).encode()).hexdigest()
    return None


def stamp(header: str, kernel: str) -> dict:
    return {"header": header, "kernel": kernel, "sha1": kernel_source_sha1(header, kernel)}


def fresh(doc: dict) -> tuple[bool, str]:
    """whether a committed PMC document still describes the kernel in the tree"""
    ks = doc.get("kernel_source")
    if not ks:
        return False, "the counter file carries no kernel_source stamp (pre-round-5 file)"
    stamps = ks if isinstance(ks, list) else [ks]
    for s in stamps:
        now = kernel_source_sha1(s["header"], s["kernel"])
        if now != s["sha1"]:
            return False, f"{s['kernel']} in {s['header']} changed since the counter pass (stamp {str(s['sha1'])[:10]}, tree {str(now)[:10]})"
    return True, "kernel source unchanged since the counter pass"


if __name__ == "__main__":
    f, pairs = sys.argv[1], sys.argv[2:]
    doc = json.load(open(f))
    doc["kernel_source"] = [stamp(pairs[i], pairs[i + 1]) for i in range(0, len(pairs), 2)]
    assert all(s["sha1"] for s in doc["kernel_source"]), doc["kernel_source"]
    json.dump(doc, open(f, "w"), indent=1)
    print(json.dumps(doc["kernel_source"]))
