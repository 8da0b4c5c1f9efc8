"""bench.py's contract, end to end on the GPU: one JSON line on stdout with the fields the driver reads, for the default
single-GPU invocation and for the N > 1 code path on a one-rank RCCL communicator."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _run(*flags):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--prewarm-steps", "20",
                        "--gaussians", "20000", "--width", "320", "--height", "240", "--no-cpu-baseline", *flags],
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600,
                       env=dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines  # exactly one line on stdout: the JSON
    return json.loads(lines[0])


def test_default_invocation_prints_one_contract_line():
    d = _run()
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "frames/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["scaling"] == "weak"
    assert d["value"] > 0 and abs(d["value"] - 1e3 / d["ms_per_step"]) <= 1e-2 * d["value"]
    assert "workload" in d["config"] and d["config"]["num_rendered"] > 0 and d["config"]["prewarm_steps"] == 20
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "stage_ms"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert d["cpu_baseline"] is None  # --no-cpu-baseline


def test_one_rank_rccl_invocation_runs_the_collectives():
    d = _run("--rccl-one-rank", "--tune-allreduce")
    assert d["n_gpus"] == 1 and d["value"] > 0
    c = d["config"]
    assert c["allreduce_ms"] is not None and c["allreduce_ms"] >= 0 and c["allreduce_chunks"] in (1, 2, 4)
    assert set(c["allreduce_chunks_tuning_ms"]) == {"1", "2", "4"} and c["allreduce_tuning_error"] is None
    assert "all-reduce" in c["step"]
