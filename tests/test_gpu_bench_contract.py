"""bench.py keeps the driver's contract: one JSON line with the agreed fields (short run, no CPU baseline)."""
import json
import os
import subprocess
import sys

import pytest

from .conftest import REPO

pytestmark = pytest.mark.gpu


def test_bench_json_line():
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "5", "--warmup", "2",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.3 < r["frac"] < 1.0
    # value = hypotheses of all steps / wall time; consistent with ms_per_step
    assert abs(d["value"] - 50000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
