"""CPU sanitizer run: the oracle and the product's host translation unit are rebuilt with
-fsanitize=address,undefined (make -C oracle asan) and the host + oracle tests run again on those builds."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(os.environ.get("UGRT_ORACLE_LIB") is not None, reason="already inside the sanitizer run")
def test_host_and_oracle_under_asan_ubsan():
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("gcc has no libasan here")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True, capture_output=True)
    d = os.path.join(ROOT, "oracle", "_asan")
    env = dict(os.environ)
    env.update(LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", UGRT_LIB=os.path.join(d, "libugrt_host_asan.so"),
               UGRT_HOST_ONLY="1", UGRT_ORACLE_LIB=os.path.join(d, "libugrt_oracle_asan.so"), OMP_NUM_THREADS="4")
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                        "tests/test_host.py", "tests/test_oracle_pipeline.py", "tests/test_oracle_kat.py"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    out = p.stdout + p.stderr
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-6000:]
    assert p.returncode == 0, out[-6000:]
    assert " passed" in out
