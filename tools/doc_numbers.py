"""Fill the @@NAME@@ placeholders of DESIGN.md / README.md from a bench line (the closing commit's `python bench.py` output):
   python tools/doc_numbers.py profiles/r06_cfg4_bench_final.json
The two documents quote ONE run; this is how its numbers get there (no hand-copied figures)."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
a, e, c5 = d["at_recall_095"], d["encode"], d["cfg5"]["curve"]
k = lambda v: f"{v / 1e3:.1f} k"
vals = {
    "QPS": k(d["value"]), "MS": f"{d['ms_per_step']:.3f}", "SCAN_TBS": f"{d['roofline']['achieved'] / 1e3:.2f}",
    "SCAN_FRAC": f"{d['roofline']['frac']:.3f}", "SCAN_TRAFFIC": f"{(d['roofline']['traffic'] or 0) / 1e9:.2f}",
    "HOSTQPS": k(d["host_io"]["queries_per_s"]), "HOSTMS": f"{d['host_io']['ms_per_step']:.3f}",
    "R095": f"{a['recall_at_10']:.4f}", "R095S": f"{a['recall_at_10_selection_batch']:.4f}", "QPS095": k(a["qps"]),
    "MS095": f"{a['ms_per_step']:.3f}", "NP095": str(a["nprobe"]), "KF095": str(a["k_factor_rf"]),
    "ENC": f"{e['abstracts_per_s']:,.0f}".replace(",", " "), "ENCMS": f"{e['ms_per_step']:.1f}", "ENCFRAC": f"{e['roofline']['frac']:.3f}",
    "FULL": f"{e['full_run']['abstracts_per_s']:,.0f}".replace(",", " "), "FULLS": f"{e['full_run']['wall_s']:.1f}",
    "C5": " / ".join(f"{c['latency_ms_p50']:.2f}" for c in c5), "C5E": " / ".join(f"{c['encode_alone_ms']:.2f}" for c in c5),
    "HBM": f"{d['config']['hbm_in_use_gb']:.0f}", "CPU": f"{d['cpu_baseline']['value']:.0f}" if d.get("cpu_baseline") else "n/a",
}
for name in ("DESIGN.md", "README.md"):
    p = os.path.join(ROOT, name)
    s = open(p).read()
    missing = set(re.findall(r"@@(\w+)@@", s)) - set(vals)
    assert not missing, missing
    for key, v in vals.items():
        s = s.replace(f"@@{key}@@", v)
    open(p, "w").write(s)
    print(name, "filled")
