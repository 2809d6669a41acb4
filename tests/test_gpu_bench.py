"""bench.py as the driver starts it: one JSON line with the contract's fields; `--gpus 2` without a launcher starts
its own rank processes (here both on the one GPU, gloo standing in for RCCL) and reports how many ranks met."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=280,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields():
    d = run_bench("--workload", "glass-of-water-256", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-pmc", "--no-extras")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and d["dtype"] == "f64" and d["config"]["ranks_seen"] == 1
    assert "glass-of-water.yaml" in d["config"]["scene_source"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["traffic"] is None and r["frac"] is None      # no counter passes asked for
    assert r["algorithmic"]["bytes_per_launch"] > 0 and r["avg_launch_ms"] > 0


def test_two_ranks_start_themselves_and_agree_with_one():
    one = run_bench("--workload", "glass-of-water-256", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-roofline", "--no-extras")
    two = run_bench("--gpus", "2", "--workload", "glass-of-water-256", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-roofline",
                    "--no-extras", env={"BDPT_BENCH_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and two["config"]["ranks_seen"] == 2 and len(two["config"]["rank_devices"]) == 2
    assert two["config"]["collective_backend"] == "gloo" and "2 GPU(s)" in two["config"]["sharding"]
    assert two["value"] > 0 and one["n_gpus"] == 1
