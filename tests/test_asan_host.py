"""SURVEY.md section 5 (race / memory checking), CPU build only: the host side of libdmh_hip.so -- argument checks, workspace
sizing, error formatting: everything that runs before a launch -- compiled with AddressSanitizer and driven through the C ABI
by tests/asan/host_checks.c.  GPU AddressSanitizer is not available on this pool; the device side is covered by the parity
tests and by bitwise run-to-run determinism (tests/test_gpu_trainer.py)."""
import os
import subprocess


def test_host_side_under_address_sanitizer():
    from depthmodelhardening_amd.build import ASAN_LIB_PATH, build_asan
    driver = build_asan()
    assert os.path.exists(ASAN_LIB_PATH)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1")
    r = subprocess.run([driver], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "host_checks: ok" in r.stdout
