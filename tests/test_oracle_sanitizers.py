"""CPU: the oracle's known-answer and golden tests once more under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY.md section 5:
sanitizers belong on the CPU build; GPU sanitizers are not available on the pool).  `make -C oracle asan` builds
oracle/libtsdf_oracle_asan.so; a child interpreter preloads the sanitizer runtimes, points oracle/oracle.py at that library
(RGBDR_ORACLE_LIB) and runs the oracle tests with halt_on_error: any report fails the run."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TESTS = ["test_oracle_k1.py", "test_oracle_bricks.py", "test_oracle_primitives.py", "test_oracle_raymarch.py", "test_oracle_inpaint.py",
         "test_oracle_stereo.py", "test_oracle_points.py", "test_oracle_trigrid.py", "test_oracle_preprocess.py", "test_golden.py"]


def _runtime(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_known_answers_under_asan_and_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("no sanitizer runtimes next to gcc")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    lib = os.path.join(ROOT, "oracle", "libtsdf_oracle_asan.so")
    env = dict(os.environ, LD_PRELOAD=f"{asan} {ubsan}", ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               RGBDR_ORACLE_LIB=lib, OMP_NUM_THREADS="4")
    # the child must really be running the instrumented library
    probe = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from oracle import oracle; oracle.lib(); "
                            "print(any('libtsdf_oracle_asan' in l for l in open('/proc/self/maps')))" % ROOT], env=env, capture_output=True, text=True)
    assert probe.returncode == 0 and probe.stdout.strip() == "True", probe.stderr[-2000:]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + [os.path.join(ROOT, "tests", t) for t in TESTS],
                       env=env, capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:] + r.stderr[-3000:])
    assert "passed" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
