"""SURVEY.md §5.2 (race detection / sanitizers): the C++ host runtime under AddressSanitizer +
UndefinedBehaviorSanitizer.  CPU only — GPU sanitizers are not available on this platform."""
import os
import shutil
import subprocess

import pytest


@pytest.mark.skipif(shutil.which("g++") is None, reason="no host compiler")
def test_host_runtime_under_asan_ubsan():
    from distributedpytorch_amd.csrc import build
    exe = build.build_host_sanitized()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", OMP_NUM_THREADS="2")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host sanitize: ok" in r.stdout
