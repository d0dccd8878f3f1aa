"""bench.py at N > 1, rehearsed on one GPU: the ranks of `python -m torch.distributed.run ... bench.py
--gpus N` share GPU 0 and talk over gloo (BENCH_REHEARSAL=1), so the control flow the driver's
scaling run takes -- rank != 0 paths, the corpus dealt into shards, the broadcast tables, the packed
exchange step, the sharded coarse quantiser from 4 ranks, the sub-shard refine point -- executes end to
end on the 1-GPU box the tests run on.  It checks the contract of the one JSON line and that the sharded
top-k equals the oracle's on the gathered index (bench.py's own `parity_vs_oracle` check); it is not a
measurement (the line says "rehearsal": true)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# The N = 2 rehearsal of the driver's own command shape (`python bench.py --gpus 2`, no launcher: bench.py re-executes itself
# under torch.distributed.run) runs in the default `-m gpu` suite under a hard timeout.  The wider ones (4 ranks, replicas,
# the forced sub-shard layout) stay opt-in (MI_RUN_REHEARSAL=1): several processes sharing one GPU is not a configuration
# the job ever runs in, and with FOUR of them the ROCm runtime stalled once in four runs on this pool (every rank parked
# inside the driver, not killable) -- two ranks never did in any run.  A run that does not finish within 300 s FAILS and
# prints every rank's Python stacks (faulthandler on SIGUSR1): a hang on a real node must never read as a skip.
opt_in = pytest.mark.skipif(os.environ.get("MI_RUN_REHEARSAL") != "1",
                            reason="the wider multi-process-per-GPU rehearsals are opt-in: MI_RUN_REHEARSAL=1")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(n, extra, self_launch=False, **env_extra):
    env = dict(os.environ, BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT, **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    launcher = [] if self_launch else ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
                                       "--master-addr", "127.0.0.1", "--master-port", str(_free_port())]
    cmd = [sys.executable] + launcher + [os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1"] + extra
    p = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=300)
    except subprocess.TimeoutExpired:
        # a stall is a FAILURE, and it says where: bench.py registers faulthandler on SIGUSR1, so every rank (and the
        # launcher) dumps the stack of every thread to stderr before the group is killed
        import signal
        os.killpg(p.pid, signal.SIGUSR1)
        try:
            out, err = p.communicate(timeout=10)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)
            out, err = p.communicate()
        pytest.fail("the %d-rank rehearsal did not finish within 300 s; stacks of the ranks at the time:\n%s" % (n, err[-12000:]))
    assert p.returncode == 0, err[-3000:]
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, out[-2000:]                                # exactly one JSON line, from rank 0
    return json.loads(lines[0])


SMALL = ["--corpus", str(2 * 1048576), "--nlist", "1024", "--batch", "256", "--nprobe", "16", "--no-encode"]


@pytest.fixture(scope="module")
def one_gpu_line():
    """The same small configuration on one rank (no rehearsal): bench.py's own oracle check + the recall the
    sharded runs must reproduce exactly (same deterministic training, sharded top-k == unsharded top-k)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1"] + SMALL,
                       cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    par = out["parity_vs_oracle"]
    assert par["ids_equal"] is True and par["scores_bit_equal"] is True and "rehearsal" not in out
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, pytest.param(4, marks=opt_in)])
def test_cfg4_line_at_n_ranks(n, one_gpu_line):
    out = _run(n, SMALL, self_launch=(n == 2))                         # N = 2: `python bench.py --gpus 2`, as the driver types it
    assert out["rehearsal"] is True and out["n_gpus"] == n and out["steps"] == 3 and out["warmup"] == 1
    assert out["scaling"] == "strong" and out["higher_is_better"] is True
    assert out["value"] > 0 and out["ms_per_step"] > 0
    cfg = out["config"]
    assert cfg["workload"] == one_gpu_line["config"]["workload"]
    assert "vector-sharded x%d" % n in cfg["parallelism"] and cfg["exchange"] == "torch"
    assert cfg["shard_coarse"] is (n >= 4)                             # the sliced coarse quantiser from 4 ranks
    assert cfg["index_vectors_this_rank"] == 2 * 1048576 // n
    # what the collective library saw: n ranks answered an all-reduce, every rank holds its share of the index
    assert cfg["rccl_ranks"] == n and cfg["index_vectors_per_rank"] == [2 * 1048576 // n] * n and cfg["collective_backend"] == "gloo"
    assert cfg["timed_blocks"] >= 3
    # the per-rank step split (one entry per rank, every stage named) and the single-GPU prediction beside it
    sp = out["step_split"]
    assert [r["rank"] for r in sp["per_rank"]] == list(range(n)) and all(r["index_vectors"] == 2 * 1048576 // n for r in sp["per_rank"])
    key = "scan_all_gather_ms" if n >= 4 else "all_gather_ms"
    assert all(r[key] >= 0 and r["scan_ms"] > 0 for r in sp["per_rank"]) and sp["predicted_ms"]["scan"] > 0
    # sharded top-k == unsharded top-k, so recall against the exact search is the same number
    assert out["recall_at_10"] == one_gpu_line["recall_at_10"], (out["recall_at_10"], one_gpu_line["recall_at_10"])
    assert out["roofline"]["bound"] == "hbm" and out["roofline"]["achieved"] > 0
    a, b = out["at_recall_095"], one_gpu_line["at_recall_095"]
    # every shard proposes its own k * k_factor candidates: at least the unsharded candidate set's recall
    # (this small corpus tops out near 0.9 at the end of the list; the 207 M line reaches 0.95 at the second point)
    assert a is not None and b is not None and a["recall_at_10"] >= b["recall_at_10"] - 0.01 > 0.8


@pytest.mark.gpu
@opt_in
def test_cfg4_line_replicas_mode_two_ranks():
    out = _run(2, SMALL + ["--no-refine-point", "--multi-gpu-mode", "replicas"])
    assert out["n_gpus"] == 2 and out["value"] > 0 and "replicas" in json.dumps(out["config"])


@pytest.mark.gpu
@opt_in
def test_cfg4_line_two_ranks_subshard_refine_point():
    """At 207 M the refine store of a whole shard does not fit N = 1 or 2 GPUs; the line then times the 1/8
    sub-shard a GPU of the 8-GPU job holds and the ranks agree on the worst recall (an all-reduce)."""
    out = _run(2, SMALL, BENCH_FORCE_SUBSHARD="1")
    a = out["at_recall_095"]
    assert a is not None and a["recall_at_10"] > 0.8 and "1/8 sub-shard" in a["scope"] and a["qps"] > 0


@pytest.mark.gpu
def test_cfg4_line_emulating_rank_0_of_8(one_gpu_line):
    """`python bench.py --emulate-rank-of 8` (round 6): one process, one GPU, RCCL at world size 1 -- rank 0's shard (rows i = 0
    mod 8), its 1/8 slice of the coarse quantiser, 8-block exchanges, 8-way merges.  The line is labelled a rehearsal, names
    every stage of the step, and scans an eighth of the bytes the whole index's step scans (the merged probe lists are the real
    job's: the absent ranks' TRUE coarse lists stand behind the collective)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--emulate-rank-of", "8", "--steps", "3", "--warmup", "1"] + SMALL,
                       cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["rehearsal"] is True and out["emulation"]["rank_of"] == 8 and out["n_gpus"] == 1
    assert out["emulation"]["index_vectors"] == 2 * 1048576 // 8 and out["config"]["shard_coarse"] is True
    assert out["config"]["rccl_ranks"] == 1 and out["config"]["collective_backend"] == "nccl"
    sp = out["step_split"]["per_rank"][0]
    for key in ("coarse_slice_ms", "coarse_all_gather_ms", "coarse_merge_ms", "scan_preassigned_ms", "scan_all_gather_ms", "scan_merge_ms"):
        assert sp[key] >= 0, key
    whole = one_gpu_line["roofline"]["bytes_per_launch"]
    assert 0.08 * whole < sp["scan_bytes"] < 0.18 * whole, (sp["scan_bytes"], whole)   # the same lists, an eighth of every one
    assert out["recall_at_10"] is None and out["at_recall_095"] is None and "encode" not in out
