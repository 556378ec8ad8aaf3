"""A/B of whole c2 frames (static scene A) across variant libraries: frame time + stage times.  python tools/c2_ab.py default lib1.so ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for lib in sys.argv[1:]:
    env = dict(os.environ)
    if lib != "default":
        env["RGBDR_LIB"] = os.path.abspath(lib)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-c1", "--long-steps", "0", "--steps", "300"], env=env, capture_output=True, text=True)
    try:
        d = json.loads(p.stdout.strip().splitlines()[-1])
        print(lib, "moving", round(d["value"], 1), "static", round(d["static"]["value"], 1), {k: round(v * 1e3, 1) for k, v in d["stage_ms"].items()}, flush=True)
    except Exception:
        print(lib, "FAILED", p.stderr[-500:], flush=True)
