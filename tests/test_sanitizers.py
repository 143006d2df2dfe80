"""Sanitizer runs of the product's host side (CPU, no GPU).  `make -C oracle asan|tsan` compiles the HIP
library's own sources with hipcc, host code instrumented (-fsanitize=address,undefined / -fsanitize=thread,
-fno-gpu-sanitize: GPU AddressSanitizer is not available on the pool), together with tests/c/san_driver.cpp, which
exercises parameter/table generation, context life cycle, the ModUp constant caches, the automorphism tables and the
acehip_hw_batch planner from one and from four threads (own contexts and one shared context).  A sanitizer report makes
the driver exit non-zero.  (ASan found and fixed: the context leaked when acehip_ctx_create_host failed.)"""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_host_side_under_sanitizer(kind):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), kind])
    exe = os.path.join(ROOT, "oracle", "_san", "san_driver_" + kind)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "san_driver: OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "ERROR: " not in r.stderr and "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
