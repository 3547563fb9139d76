"""The line the driver records: `python3 bench.py --gpus 1 --steps K --warmup W` prints ONE JSON line with the
contract's keys, the roofline object, the order of the legs, and the exact sum of the synthetic workload."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def test_driver_command_line(orc):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_FORCE_DIST")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
           "--no-cpu-baseline", "--no-extras"]
    r = subprocess.run(cmd, env=env, capture_output=True, timeout=200)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    n = 1 << 20
    assert out["unit"] == "points/s" and out["n_gpus"] == 1 and out["steps"] == 20 and out["warmup"] == 5
    assert out["higher_is_better"] is True and out["scaling"] == "weak" and out["vs_baseline"] is None
    assert out["data"].startswith("synthetic") and "workload" in out["config"] and "model" not in out["config"]
    assert abs(out["value"] - n * 20 / (out["ms_per_step"] * 20e-3)) < 1e-6 * out["value"]
    assert bytes.fromhex(out["config"]["result_xy"]) == orc.synth_msm_closed_form(0x5EED0001, 0x5EED0002, n)
    # the legs and their order: blocking calls (medians over >= 50) before the warm-up steps and the timed steps
    assert out["config"]["order"].startswith("set-up") and "5 warm-up steps; 20 timed steps" in out["config"]["order"]
    sc = out["single_call"]
    assert sc["same_result"] is True and sc["calls"] >= 50 and 0.5 < sc["median_ms"] < 5
    rf = out["roofline"]
    assert rf["bound"] == "valu_int" and rf["kernel"] == "k_accumulate" and rf["unit"] == "IMAD/s"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0.5 < rf["frac"] < 1.0
    assert abs(rf["achieved"] - n * 23120 / (rf["kernel_ms"] * 1e-3)) < 1e-6 * rf["achieved"]
    assert rf["frac_vs_r01_peak"] > rf["frac"] and rf["traffic"] is None or rf["traffic"] > 1e9
    ph = out["phases_ms"]
    assert abs(ph["accumulate"] - rf["kernel_ms"]) < 1e-9 and ph["accumulate"] < ph["device_total"] < sc["median_ms"] * 1.05
