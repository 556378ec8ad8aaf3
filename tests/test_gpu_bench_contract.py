"""-m gpu: bench.py's output contract -- exactly one JSON line on stdout with the keys and types the driver reads, the
`roofline` and `cpu_baseline` objects, and the library (not the oracle) as the thing that was measured."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def c_one(d):
    c = d["cpu_baseline"]
    return c["one_thread"]["cores"] == 1 and 0 < c["one_thread"]["value"] <= c["value"] * 1.05


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
    # algorithmic bytes are those of the units the launch processes (active tiles), not the dense volume: a fraction of the peak
    assert 0.0 < r["frac"] < 1.0 and r["algorithmic_bytes"] < 200e6 and abs(r["achieved"] - r["algorithmic_bytes"] / (r["avg_launch_ms"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    # the dense configuration's two kernels ride along in the same line
    c1 = d["roofline_c1"]
    for k, dense in (("integrate", 4 * 256 ** 3 + 4 * 16 * 128 ** 3 + 4 * 16 * 640 * 480), ("march", 4 * 256 ** 3 + 24 * 1280 * 720)):
        assert c1[k]["algorithmic_bytes"] == dense and 0.0 < c1[k]["frac"] < 1.0 and c1[k]["avg_launch_ms"] > 0
    # value: a NEW frame every step (re-laid out inside the step), the scene moving; the static case and round 2's definition (two frames
    # already in the kernels' layout) are reported beside it, and the re-layout is a stage of the frame
    assert "NEW frame every step" in d["config"]["scene"] and d["static"]["value"] > 0 and d["long_run"]["steps"] >= 1000
    assert d["resident_frames"]["value"] > 0 and d["stage_ms"]["0repack"] > 0 and d["warmup_effective"] >= d["warmup"]
    st = d["stage_ms"]
    # the stages (timed with stage overlap off) are the frame: their sum is the serial frame time
    assert 0.8 < (st["0repack"] + st["bricks"] + st["2integrate"] + st["3recon"]) / d["serial"]["ms_per_step"] < 1.5    # (every recorded event adds ~1-2 us to the stage pass)
    assert d["stage_overlap"] is True and d["value"] > d["serial"]["value"] and d["frame_device_ms"]["median"] > d["ms_per_step"]    # three lanes: latency > 1 / rate
    rf = d["roofline_frame"]
    assert 0.0 < rf["frac"] < 1.0 and abs(rf["bytes"] - sum(rf["stage_bytes"].values())) < 1.0
    assert 0.0 < r["with_helper"]["frac"] < r["frac"]
    assert c_one(d)
    assert d["pcie_overlapped"]["frames_per_s"] > d["pcie_inclusive_frames_per_s"] * 0.9
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "frames/s" and c["sample"]
    assert d["value"] > 100 * c["value"]                                        # the GPU line is not the oracle's


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("world,extra,what", [(2, [], "1 Z-slab(s) on ranks 1..1 + rank 0 as dedicated compositor"),
                                              (3, [], "2 Z-slab(s) on ranks 1..2 + rank 0 as dedicated compositor"),
                                              (2, ["--compositor", "shared"], "ONE volume in 2 Z-slabs")])
def test_bench_with_more_ranks_runs_the_slab_partition_by_default(world, extra, what):
    """`--gpus N` without further flags = ONE volume in Z-slabs on ranks 1..N-1 with rank 0 as the compositing rank (`--compositor
    shared`: N slabs, rank 0 composites as well), strong scaling, and the run itself checks the composite against an unpartitioned
    context before timing.  N processes on the one GPU of the box, gloo standing in for RCCL (RCCL refuses two ranks on one
    device): everything else is the driver's command."""
    env = dict(os.environ, RR_BENCH_BACKEND="gloo", RR_BENCH_DEVICE="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(29571 + world + len(extra)),
                        os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "10", "--warmup", "2", "--long-steps", "0"] + extra,
                       cwd=ROOT, capture_output=True, text=True, timeout=1400, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, p.stdout[:500]
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["scaling"] == "strong" and what in d["config"]["parallelism"], d["config"]["parallelism"]
    assert "bit-identical" in d["slab_check"] and d["regathers"] == 0
    assert d["frame_replicas"]["scaling"] == "weak" and d["frame_replicas"]["value"] > 0     # an extra key beside the slab value, never instead of it
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"] and d["value"] > 30.0
    assert d["stage_ms"]["k_integrate_tiles"] > 0                                # (from a rank that owns a slab)
    r = d["roofline"]                                                            # the slowest slab launch, measured on its own rank
    assert r["bound"] == "hbm" and 0.0 < r["frac"] < 1.0 and r["traffic"] is None and "slab" in r["note"]
    assert abs(r["achieved"] - r["algorithmic_bytes"] / (r["avg_launch_ms"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    if not extra:
        assert d["stage_ms_compositor"]["holefill"] > 0 and "k_integrate_tiles" not in d["stage_ms_compositor"]
