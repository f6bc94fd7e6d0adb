"""bench.py's contract with the driver, on the GPU: ONE JSON line on stdout with the fields the round summary reads, at a
tiny workload (64x64 / 96x64, a few steps) in a fresh process per case."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1"] + list(flags),
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_has_the_contract_fields_and_a_live_roofline():
    j = _bench("--size", "64", "--cpu-frames", "5")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1 and j["dtype"] == "f32" and j["vs_baseline"] is None
    assert j["value"] > 0 and abs(j["value"] - 1e3 / j["ms_per_step"]) < 1e-6 * j["value"]
    r = j["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0 < r["frac"] <= 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["live_frac"] >= r["frac"] and r["class_kernel_ms_per_step"] > 0
    assert r["kernel_time_sum_ms_per_step"] > 0 and r["launches_per_step"] == j["config"]["launches_per_step"]
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1
    assert j["parity"]["max_abs_img"] <= 1e-3 and j["parity"]["max_abs_mask"] <= 1e-3
    from render_in_between_amd import _native
    assert j["config"]["build"] == _native.build_info()["raw"]


@pytest.mark.gpu
def test_bench_other_modes_and_shapes():
    j = _bench("--no-cpu-baseline", "--height", "64", "--width", "96", "--mode", "chain", "--frames", "4", "--batch", "2")
    assert j["config"]["height"] == 64 and j["config"]["width"] == 96 and j["config"]["frames_per_step_per_gpu"] == 8
    assert j["cpu_baseline"] is None and j["value"] > 0
    j = _bench("--no-cpu-baseline", "--size", "64", "--mode", "clips", "--frames", "3", "--graph", "--dtype", "bf16")
    g = j["config"]["graph_replay"]
    assert g["captures"] >= 1 and g["captures"] + g["replays"] == 1 + 3      # warm-up + timed steps (the profiling pass launches kernel by kernel)
    assert j["config"]["replica_check"]["all_equal"] and j["roofline"]["bound"] == "hbm"
    assert len(j["config"]["per_rank_host_enqueue_ms_per_step"]) == 1
