"""-m gpu: bench.py's output contract -- exactly one JSON line on stdout with the keys and types the driver reads, the
`roofline` and `cpu_baseline` objects, and the library (not the oracle) as the thing that was measured."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_bench_prints_one_json_line_with_the_contract_keys():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "3"],
                       cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[:500]
    d = json.loads(lines[0])
    for k, t in dict(metric=str, value=float, unit=str, n_gpus=int, steps=int, warmup=int, ms_per_step=float, higher_is_better=bool,
                     scaling=str, dtype=str, data=str, config=dict, roofline=dict, cpu_baseline=dict).items():
        assert isinstance(d[k], t), (k, d[k])
    assert "vs_baseline" in d and d["vs_baseline"] is None                      # BASELINE.md holds no published number
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 3 and d["higher_is_better"] is True
    assert d["unit"] == "frames/s" and d["data"] == "synthetic" and d["dtype"] == "f32" and "workload" in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    assert d["value"] > 30.0                                                     # north_star's floor at 512^3 x 4 streams, by two orders of magnitude
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-9 and r["achieved"] > 0 and (r["traffic"] is None or r["traffic"] > 0)
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "frames/s" and c["sample"]
    assert d["value"] > 100 * c["value"]                                        # the GPU line is not the oracle's
