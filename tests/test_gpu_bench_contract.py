"""bench.py prints exactly one JSON line with the driver's contract fields, the roofline and the CPU-baseline objects (small run)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("extra", [["--cpu-pairs", "2", "--verify"], ["--no-cpu-baseline", "--workload", "orb_match", "--streams", "1"]])
def test_bench_json_contract(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--pairs", "16"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "profiled_pass", "rccl_ranks"):
        assert k in d, k
    assert d["roofline"]["kernel"].startswith("k_")               # the profiled kernel by its symbol
    assert d["config"]["hip_streams"]["caller"] >= 1
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"] and d["value"] > 0
    assert abs(d["value"] - 16 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    roof = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in roof, k
    assert roof["bound"] in ("hbm", "mfma") and roof["peak"] > 0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    if "--no-cpu-baseline" in extra:
        assert d["cpu_baseline"] is None
    else:
        cb = d["cpu_baseline"]
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in cb, k
        assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 1


def test_bench_two_ranks_on_one_gpu():
    """The N>1 path of bench.py (frame sharding, id-range sharded loop DB + all-gather, barrier, max over ranks, one JSON line from
    rank 0) launched exactly as the driver launches it, with both ranks on the only GPU of the box and gloo carrying the collectives."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--pairs", "16",
           "--backend", "gloo", "--verify"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["cpu_baseline"] is None        # CPU baseline: rank 0 at N=1 only
    assert abs(d["value"] - 2 * 16 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]       # whole-job rate: both ranks' pairs
    assert "x2" in d["config"]["parallelism"] and d["collective_ranks"] == 2


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` without torch.distributed.run: bench.py starts its ranks itself (gloo here: both ranks share the
    box's only GPU; with the default nccl backend each rank takes its own GPU and the JSON carries rccl_ranks)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--pairs", "16", "--backend", "gloo",
           "--no-extra-passes"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["collective_ranks"] == 2 and d["collective_backend"] == "gloo" and d["value"] > 0
