"""A/B of whole frames across ENVIRONMENT variants of one library:  python tools/env_ab.py c2 "" RR_K1_RECT=0 "RR_K1_RECT=0 RR_DEEP=0"
Prints per variant the moving-scene rate, the serial (one stream) rate and the stage times.  AB_ARGS="--preprocess": extra bench.py arguments."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
config = sys.argv[1]
steps = os.environ.get("AB_STEPS", "300")
for var in sys.argv[2:]:
    env = dict(os.environ)
    for kv in var.split():
        k, v = kv.split("=", 1)
        env[k] = v
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--no-cpu-baseline", "--no-c1", "--long-steps", "0", "--steps", steps] + os.environ.get("AB_ARGS", "").split(),
                       env=env, capture_output=True, text=True)
    try:
        d = json.loads(p.stdout.strip().splitlines()[-1])
        print(f"{config} [{var or 'default'}] value {d['value']:.1f} serial {d.get('serial', {}).get('value', 0):.1f}", {k: round(v * 1e3, 1) for k, v in d["stage_ms"].items()},
              "k1", round(d["roofline"]["avg_launch_ms"] * 1e3, 1), flush=True)
    except Exception:
        print(config, var, "FAILED", p.stdout[-300:], p.stderr[-800:], flush=True)
