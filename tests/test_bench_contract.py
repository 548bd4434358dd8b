"""bench.py's contract with the driver: without a GPU it fails loudly (no CPU path), on a GPU it prints ONE JSON line
carrying every required key, the roofline object and the CPU baseline."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 3
    assert "no CPU path" in out.stderr and out.stdout.strip() == ""


def test_bench_rejects_a_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         timeout=600, env=env)
    assert out.returncode == 2 and "WORLD_SIZE" in out.stderr


def test_bench_gpus_2_launches_its_own_ranks_cpu_rehearsal():
    """`python bench.py --gpus 2` invoked directly, with no launcher and no WORLD_SIZE: bench.py must start its own two
    ranks (children, before anything touches a GPU) and rank 0 must print exactly one JSON line with n_gpus = 2.  On
    CPU this is the --rehearsal mode (gloo, the host emulator of the kernels): plumbing only, value = null."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--rehearsal"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1
    assert d["rehearsal"] is True and d["value"] is None and d["ms_per_step"] is None      # never a measurement
    assert d["config"]["collective"] == {"backend": "gloo", "ranks": 2}
    assert d["sharded_vs_whole_rel_err"] <= 1e-13


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "evals/s" and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] - 1000.0) < 1e-6 * 1000.0
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.3 < r["frac"] < 1.0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    assert d["value"] > 50 * c["value"]          # sanity: the GPU path is not the CPU path
    # the reference driver's blocking loop beside the headline; the PMC traffic only when its profile matches the build
    assert 0 < d["blocking_call"]["value"] <= d["value"] * 1.02 and d["overlapped"] is None
    assert d["config"]["collective"] is None and d["config"]["collective_overlap"] is False
    assert (r["traffic"] is None) or (r["traffic_source"].startswith("profiles/") and r["traffic"] > 0)
    # the per-kernel HIP-event times are taken in the regime of the timed loop (evaluations queued back to back), so
    # they add up to no more than a step (3 %: the event records themselves, and a 3-step headline sample)
    assert abs(sum(k["ms_per_eval"] for k in r["per_kernel"].values()) - r["per_kernel_sum_ms"]) < 1e-9
    assert r["per_kernel_sum_ms"] <= r["profiled_eval_ms"], (r["per_kernel_sum_ms"], r["profiled_eval_ms"])
    assert r["per_kernel_sum_ms"] <= 1.03 * d["ms_per_step"], (r["per_kernel_sum_ms"], d["ms_per_step"])
    rp = d["repeats"]
    assert rp["n"] == 5 and rp["min"] <= rp["median"] <= rp["max"] and 0.8 * d["value"] < rp["median"] < 1.25 * d["value"]
