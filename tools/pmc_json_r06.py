"""gpurun_out/r06_prof/ (tools/prof_r06.sh) -> the stamped counter files bench.py reads for `roofline.traffic`:
  profiles/r06_cfg4_scan_pmc.json          scan_kernel<64, 8, false, false> at cfg4 (the headline's dominant kernel)
  profiles/r06_cfg4_refine_pmc.json        rerank_sq8_kernel / scan_kernel<..., true, ...> / select_pairs_kernel at the recall-0.95 point's timed shape
  profiles/r06_cfg3_encoder_gemm_pmc.json  the four slab GEMMs of the bulk encode (via tools/pmc_encode_json.py)
  profiles/r06_encode_query_pmc.json       whole forward passes of the query-time regimes: one query (31 tokens), 16 queries (563 tokens)
Every file carries `kernel_source` (tools/kernel_stamp.py): bench.py drops a number whose kernel text changed since the pass.
FETCH_SIZE: x2 on gfx950 (a wide coalesced read is counted at half its bytes: MI355X_MICROARCH.md, section HBM), KiB units;
WRITE_SIZE as is.   usage: [PMC_VER=v2] python tools/pmc_json_r06.py [dir = gpurun_out/r06_prof]
Only the parts found under `dir` are rewritten: the query-time file keeps the regimes (and their stamps) a run did not profile again;
PMC_VER names the copies of the summaries it cites (profiles/r06_<part>_kernel_stats_<ver>.csv ...)."""
import csv, json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_stamp
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r06_prof")
prof = os.path.join(ROOT, "profiles")
VER = os.environ.get("PMC_VER", "v1")
CORR = "gfx950: FETCH_SIZE counts a wide coalesced read at half its bytes (MI355X_MICROARCH.md, section HBM) -> read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 taken as is; both are the L2's memory-side requests (Infinity-Cache hits are inside the count)"


def parse(fn):
    d, k = {}, None
    if not os.path.exists(fn):
        return d
    for l in open(fn):
        if not l.startswith("   "):
            k = l.strip()
        else:
            m = re.search(r"(\w+)\s+n=\s*(\d+) mean=\s*([\d.]+)", l)
            d.setdefault(k, {})[m.group(1)] = (int(m.group(2)), float(m.group(3)))
    return d


def stats(fn):
    return {r["Name"]: r for r in csv.DictReader(open(fn))} if os.path.exists(fn) else {}


def avg_us(st, frag):
    rows = [r for n, r in st.items() if frag in n]
    return round(float(rows[0]["AverageNs"]) / 1e3, 2) if rows else None


def keep(stem):   # copy the summaries the json files cite
    for suf in ("_kernel_stats.csv", "_FETCH_SIZE.txt", "_WRITE_SIZE.txt"):
        f = os.path.join(src, stem + suf)
        if os.path.exists(f):
            dst = os.path.join(prof, "r06_" + stem + suf.replace("_kernel_stats", "_kernel_stats_" + VER).replace("_SIZE.txt", "_SIZE_" + VER + ".txt"))
            open(dst, "w").write(open(f).read())


made = []
# ---- cfg4: scan + the refine point's kernels
F, W, st = parse(src + "/cfg4_FETCH_SIZE.txt"), parse(src + "/cfg4_WRITE_SIZE.txt"), stats(src + "/cfg4_kernel_stats.csv")
if F and W:
    keep("cfg4")
    line = json.loads([l for l in open(src + "/cfg4_under_FETCH_SIZE.out") if l.startswith("{")][-1])
    cfg, alg = line["config"], line["roofline"]["bytes_per_launch"]
    def one(frag):
        k = [n for n in F if frag in n]
        if not k:
            return None
        k = k[0]
        kw = [n for n in W if frag in n][0]
        return {"kernel": k, "dispatches": F[k]["FETCH_SIZE"][0], "FETCH_SIZE_KiB_mean": F[k]["FETCH_SIZE"][1], "WRITE_SIZE_KiB_mean": W[kw]["WRITE_SIZE"][1],
                "corrected_bytes_per_launch": int(F[k]["FETCH_SIZE"][1] * 2048 + W[kw]["WRITE_SIZE"][1] * 1024)}
    sc = one("scan_kernel<64, 8, false, false>")
    doc = dict(sc, config=[cfg["corpus"], cfg["nlist"], cfg["global_batch"], cfg["nprobe"], cfg["k"], 1],
               config_text="cfg4: 207Mx1024 IVF65536,PQ64, batch 1024, nprobe 64, k 10, 1 GPU (round 6: the default line's search half, the refine point pinned to its timed shape)",
               how="two separate rocprofv3 --pmc passes (tools/prof_r06.sh cfg4: FETCH_SIZE, then WRITE_SIZE; --kernel-trace only, --kernel-include-regex 'scan_kernel|rerank_sq8|select_pairs') of `bench.py --no-encode --no-cpu-baseline --streams 1 --no-recall --steps 10`; means over the dispatches of this kernel (profiles/r06_cfg4_FETCH_SIZE_v1.txt, _WRITE_SIZE_v1.txt)",
               correction=CORR, algorithmic_bytes_per_launch=alg, traffic_over_algorithmic=round(sc["corrected_bytes_per_launch"] / alg, 4),
               source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 x2 read correction (profiles/r06_cfg4_scan_pmc.json)",
               kernel_trace_same_command={"scan_avg_ms_batch1024": round((avg_us(st, "scan_kernel<64, 8, false, false>") or 0) / 1e3, 4), "file": "profiles/r06_cfg4_kernel_stats_v1.csv"},
               kernel_source=[kernel_stamp.stamp("ivfpq_kernels.h", "scan_kernel")])
    json.dump(doc, open(prof + "/r06_cfg4_scan_pmc.json", "w"), indent=1); made.append("r06_cfg4_scan_pmc.json")
    a95 = line.get("at_recall_095") or {}
    rr, sa, sp = one("rerank_sq8_kernel"), one("scan_kernel<64, 8, true, false>"), one("select_pairs_kernel")
    if rr and a95:
        rl = a95.get("roofline") or {}
        doc = {"config": [cfg["corpus"], cfg["nlist"], cfg["global_batch"], a95["nprobe"], a95["k_factor_rf"], cfg["k"]],
               "config_text": "the recall-0.95 point at its timed shape (nprobe %d, k_factor %d): every launch of the pass is of this shape (BENCH_REFINE_NPROBES / BENCH_REFINE_KFS pin the sweep)" % (a95["nprobe"], a95["k_factor_rf"]),
               "rerank": rr, "all_scores_scan": sa, "set_selection": sp, "correction": CORR,
               "corrected_bytes_per_launch": rr["corrected_bytes_per_launch"], "algorithmic_bytes_per_launch": rl.get("bytes_per_launch"),
               "traffic_over_algorithmic": round(rr["corrected_bytes_per_launch"] / rl["bytes_per_launch"], 4) if rl.get("bytes_per_launch") else None,
               "avg_us_kernel_trace": {"rerank_sq8_kernel": avg_us(st, "rerank_sq8_kernel"), "scan_kernel<64, 8, true, false>": avg_us(st, "scan_kernel<64, 8, true, false>"),
                                       "select_pairs_kernel": avg_us(st, "select_pairs_kernel"), "note": "profiles/r06_cfg4_kernel_stats_v1.csv: that run includes the (nprobe, k_factor) sweep's smaller launches"},
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 x2 read correction (profiles/r06_cfg4_refine_pmc.json)",
               "kernel_source": [kernel_stamp.stamp("ivfpq_kernels.h", "rerank_sq8_kernel")]}
        json.dump(doc, open(prof + "/r06_cfg4_refine_pmc.json", "w"), indent=1); made.append("r06_cfg4_refine_pmc.json")
# ---- cfg4 with the exact list pruning on: the timed step's own scan launch (BENCH_NO_EXHAUSTIVE=1 passes: every dispatch is of that kind)
Fp, Wp, stp = parse(src + "/cfg4_pruned_FETCH_SIZE.txt"), parse(src + "/cfg4_pruned_WRITE_SIZE.txt"), stats(src + "/cfg4_pruned_kernel_stats.csv")
if Fp and Wp:
    for suf in ("_kernel_stats.csv", "_FETCH_SIZE.txt", "_WRITE_SIZE.txt"):
        f = os.path.join(src, "cfg4_pruned" + suf)
        if os.path.exists(f):
            open(os.path.join(prof, "r06_cfg4_pruned_step" + suf.replace("_kernel_stats", "_kernel_stats_" + VER).replace("_SIZE.txt", "_SIZE_" + VER + ".txt")), "w").write(open(f).read())
    line = json.loads([l for l in open(src + "/cfg4_pruned_under_FETCH_SIZE.out") if l.startswith("{")][-1])
    cfg, rl = line["config"], line["roofline"]
    kf = [n for n in Fp if "scan_kernel<64, 8, false, false>" in n][0]
    kw = [n for n in Wp if "scan_kernel<64, 8, false, false>" in n][0]
    corrected = int(Fp[kf]["FETCH_SIZE"][1] * 2048 + Wp[kw]["WRITE_SIZE"][1] * 1024)
    doc = {"kernel": kf, "dispatches": Fp[kf]["FETCH_SIZE"][0], "FETCH_SIZE_KiB_mean": Fp[kf]["FETCH_SIZE"][1], "WRITE_SIZE_KiB_mean": Wp[kw]["WRITE_SIZE"][1],
           "corrected_bytes_per_launch": corrected, "config": [cfg["corpus"], cfg["nlist"], cfg["global_batch"], cfg["nprobe"], cfg["k"], 1],
           "config_text": "cfg4 with the exact list pruning on (the default): the timed step's scan launch -- one workgroup per query, every wave stops at the first list that provably holds no result",
           "how": "two separate rocprofv3 --pmc passes (tools/prof_r06.sh cfg4: FETCH_SIZE, then WRITE_SIZE; --kernel-trace only, --kernel-include-regex 'scan_kernel') of `BENCH_NO_EXHAUSTIVE=1 bench.py --no-encode --no-cpu-baseline --streams 1 --no-refine-point --no-recall --steps 10`: no exhaustive launch in the process; means over the dispatches of this kernel (profiles/r06_cfg4_pruned_step_FETCH_SIZE_v1.txt, _WRITE_SIZE_v1.txt)",
           "correction": CORR, "algorithmic_bytes_per_launch": rl["bytes_per_launch"], "traffic_over_algorithmic": round(corrected / rl["bytes_per_launch"], 4),
           "what_the_difference_is": "every workgroup stages its query's 64 KiB look-up table (1024 x 64 KiB = 67 MB) and reads whole code groups two ahead of the stop",
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 x2 read correction (profiles/r06_cfg4_pruned_scan_pmc.json)",
           "kernel_trace_same_command": {"scan_avg_ms_batch1024": round((avg_us(stp, "scan_kernel<64, 8, false, false>") or 0) / 1e3, 4), "file": "profiles/r06_cfg4_pruned_step_kernel_stats_v1.csv"},
           "kernel_source": [kernel_stamp.stamp("ivfpq_kernels.h", "scan_kernel")]}
    json.dump(doc, open(prof + "/r06_cfg4_pruned_scan_pmc.json", "w"), indent=1); made.append("r06_cfg4_pruned_scan_pmc.json")
# ---- bulk encode
if os.path.exists(src + "/encode_gemm_FETCH_SIZE.txt"):
    keep("encode"); keep("encode_gemm")
    dst = prof + "/r06_cfg3_encoder_gemm_pmc.json"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_encode_json.py"), src, dst])
    d = json.load(open(dst))
    d["command"] = d["command"].replace("tools/prof_r04.sh", "tools/prof_r06.sh enc")
    d["kernel_stats_file"] = "profiles/r06_encode_kernel_stats_v1.csv"
    d["kernel_source"] = [kernel_stamp.stamp("encoder_kernels.h", "gemm_bf16_slab_kernel")]
    json.dump(d, open(dst, "w"), indent=1); made.append("r06_cfg3_encoder_gemm_pmc.json")
# ---- the query-time regimes: bytes of a whole forward pass
doc = {"correction": CORR, "regimes": {}, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 x2 read correction (profiles/r06_encode_query_pmc.json)",
       "how": "tools/prof_r06.sh b1 / mid: every kernel of the pass summed (per-kernel mean x launches per forward pass = dispatches / passes)"}
stamps = []
try:                                                              # regimes an earlier run profiled stay (with their stamps) unless this run has them again
    old = json.load(open(prof + "/r06_encode_query_pmc.json"))
except Exception:
    old = {"regimes": {}, "kernel_source": []}
REG = (("b1", "few_", "encoder_few.h", ["few_qkv8_kernel", "few_ao_kernel", "few_gu8_kernel", "few_d_kernel", "few_row_kernel"]),
       ("mid", "mienc", "encoder_mid.h", ["mid_gemm_kernel"]), ("mid256", "mienc", "encoder_kernels.h", ["gemm_bf16_slab_kernel"]))
again = [tag for tag, *_ in REG if os.path.exists(src + f"/{tag}_FETCH_SIZE.txt")]
for tag, r in old.get("regimes", {}).items():
    if tag not in again:
        doc["regimes"][tag] = r
        stamps += r.get("kernel_source") or [s_ for s_ in old.get("kernel_source", []) if (tag == "b1") == s_["kernel"].startswith("few_")]
for tag, frag, hdr, kerns in REG:
    F, W, st = parse(src + f"/{tag}_FETCH_SIZE.txt"), parse(src + f"/{tag}_WRITE_SIZE.txt"), stats(src + f"/{tag}_kernel_stats.csv")
    if not F:
        continue
    keep(tag)
    txt = open(src + f"/{tag}_plain.out").read() if os.path.exists(src + f"/{tag}_plain.out") else ""
    m = re.search(r"tokens (\d+)", txt) or re.search(r"(\d+) tokens each", txt)
    ntok = int(m.group(1)) if m else None
    layer_k = [k for k in F if frag in k and not any(x in k for x in ("import_rows", "interleave", "few_tile", "tile_weights"))]
    # launches per pass: a per-layer kernel is dispatched 28 x passes times; passes = dispatches of the first per-layer GEMM / 28
    gem = [k for k in layer_k if ("few_qkv8_kernel" in k or "few_gemm_kernel<0" in k or "mid_gemm_kernel<0" in k or "gemm_bf16_slab_kernel<3" in k)]
    passes = F[gem[0]]["FETCH_SIZE"][0] / 28.0 if gem else None
    tot_f = tot_w = 0.0
    per = {}
    for k in layer_k:
        n, mean = F[k]["FETCH_SIZE"]
        per_pass = n / passes
        f = mean * 2048 * per_pass
        w = (W.get(k, {}).get("WRITE_SIZE", (0, 0.0))[1]) * 1024 * per_pass
        tot_f += f; tot_w += w
        per[k] = {"launches_per_pass": round(per_pass, 2), "fetch_bytes_per_pass": int(f), "write_bytes_per_pass": int(w), "avg_us": avg_us(st, k[:40])}
    doc["regimes"][tag] = {"tokens": ntok, "passes_profiled": passes, "fetch_bytes_per_pass": int(tot_f), "write_bytes_per_pass": int(tot_w) if W else None,
                           "bytes_per_pass": int(tot_f + tot_w), "kernels": per, "kernel_stats_file": f"profiles/r06_{tag}_kernel_stats_{VER}.csv"}
    mine = [kernel_stamp.stamp(hdr, k) for k in kerns]
    if tag == "mid":
        mine.append(kernel_stamp.stamp("encoder_kernels.h", "gemm_bf16_slab_kernel"))
    doc["regimes"][tag]["kernel_source"] = mine
    stamps += mine
if doc["regimes"]:
    seen = set()
    doc["kernel_source"] = [s_ for s_ in stamps if not ((s_["header"], s_["kernel"]) in seen or seen.add((s_["header"], s_["kernel"])))]
    json.dump(doc, open(prof + "/r06_encode_query_pmc.json", "w"), indent=1); made.append("r06_encode_query_pmc.json")
print("wrote", made)
