"""bench.py itself, end to end on the GPU at a small size: the line the driver parses must come out whole -- every leg, the parity object
green, exit code 0 -- in both regimes of the headline step (K below / above the length from which the two queues pay)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(*argv):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    return json.loads(lines[0])


@pytest.mark.parametrize("steps", [6, 48])
def test_bench_line_is_whole_and_its_parity_object_green(built_lib, steps):
    d = _run("--frames", "16384", "--utts", "4096", "--steps", str(steps), "--warmup", "3", "--settle-ms", "10")
    # the contract's keys
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == 3 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["higher_is_better"] is True
    assert d["value"] > 0 and abs(d["value"] - 16384 / (d["ms_per_step"] * 1e-3)) <= 1e-2 * d["value"]     # (ms_per_step is rounded to 0.1 us on the line)
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s" and 0 < rf["frac"] < 1 and rf["kernel_ms"] <= d["ms_per_step"] * 1.001
    assert abs(rf["achieved"] - 2100 * 16384 / (rf["kernel_ms"] * 1e-3) / 1e9) <= 0.02 * rf["achieved"]     # algorithmic bytes / the event time (rounded to 0.1 us on the line)
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    # what the timed steps left behind agrees with the oracle
    p = d["parity"]
    assert p["ok"] is True and p["mfcc_b"]["within_tolerance"] and p["mfcc_a"]["within_tolerance"] and p["q15"]["bit_exact"]
    assert p["kws"]["cnn_bit_exact_on_gpu_features"] and p["kws"]["argmax_flips"] == 0
    # the headline's queue choice: two queues only when a pair was kept AND the region is long enough
    assert d["config"]["queues"] in (1, 2)
    if steps < 40:
        assert d["config"]["queues"] == 1
        if d["queue_calibration"].get("pair") is not None:
            assert "two_queues_same_wk" in d and "queues_why" in d["config"]
    else:
        assert (d["config"]["queues"] == 2) == (d["queue_calibration"].get("pair") is not None)
    # every side leg ran (an error object is what a failed leg leaves)
    for k in ("serial_cold", "serial", "settled", "rows_launch", "batch_list", "two_queues", "mfcc_variant_a", "mfcc_variant_tf", "mfcc_variant_d", "mfcc_q15", "kws", "streaming"):
        assert k in d and "error" not in d[k], (k, d.get(k))
    assert d["batch_list"]["outputs_bit_identical_to_one_call_per_batch"] and d["two_queues"]["outputs_bit_identical_to_serial"]
    assert d["kws"]["collective_path"].startswith("none")
