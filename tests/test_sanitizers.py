"""The committed sanitizer recipe (cf. the reference's own: CMakeLists.txt:30-33,106-110).  Both legs are CPU builds:

* `make -C oracle asan`      the C restatement under ASan + UBSan, driven through the whole golden suite
* `make -C tests/cxx asan`   the host-only C++ cases (detail/field.hpp -- the arithmetic source of the kernels -- and
                             the generic host paths of the mirror, incl. the user-defined GF(7) field) under ASan + UBSan

GPU AddressSanitizer is not available on this pool."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libasan():
    out = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def test_oracle_golden_suite_under_asan_ubsan():
    asan = _libasan()
    if asan is None:
        pytest.skip("gcc has no libasan here")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True)
    so = os.path.join(ROOT, "oracle", "_build", "libscl_oracle_asan.so")
    env = dict(os.environ, LD_PRELOAD=asan, SCL_ORACLE_SO=so,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"),
                        os.path.join(ROOT, "tests", "test_plugin_field_pins.py"), "-q", "-x", "-m", "not gpu",
                        "-k", "not live_reference", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "passed" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


def test_cxx_host_cases_under_asan_ubsan():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "cxx"), "asan"], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([os.path.join(ROOT, "tests", "cxx", "_build", "test_scl_api_asan"), "--host-only"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "0 failures" in r.stdout and "Berlekamp-Welch" in r.stdout
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
