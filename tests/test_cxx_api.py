"""The C++ mirror of the reference's API (include/scl_hip/) exercised by tests/cxx/test_scl_api.cc,
a restatement of the reference's own Catch2 cases for this path."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CXX = os.path.join(ROOT, "tests", "cxx")
BIN = os.path.join(CXX, "_build", "test_scl_api")


def _build():
    subprocess.run(["make", "-s", "-C", CXX], check=True)
    return BIN


def test_cxx_api_host_only():
    """scalar FF / Polynomial / Matrix / Lagrange-table cases: no kernel launches"""
    r = subprocess.run([_build(), "--host-only"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failures" in r.stdout


@pytest.mark.gpu
def test_cxx_api_full():
    """Vector / Matrix / PRG / ss:: cases through the C ABI on the GPU"""
    r = subprocess.run([_build()], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failures" in r.stdout and "ss::shamir*" in r.stdout
