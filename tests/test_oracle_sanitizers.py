"""The oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build; the reference has real
out-of-bounds reads -- solve_3d.cc:1161, trp_wrapper.cpp:221 -- that the restatement must define away)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_clean_under_asan_ubsan():
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "selftest"], capture_output=True, text=True, env=env,
                       timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "selftest ok" in out and "ERROR: AddressSanitizer" not in out and "runtime error" not in out, out[-3000:]
