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


@pytest.mark.gpu
def test_cxx_open_over_a_world_of_threads():
    """hip::Communicator / hip::open / hip::openByPartialSums -- the C++ mirror of the C ABI's open step -- with worlds of 2, 3,
    4 and 8: the ranks are std::threads of the test binary on this one GPU, RCCL is tests/cxx/fake_rccl.cc bound through
    SCL_HIP_RCCL_LIBRARY (an all-gather = rendezvous + device-to-device copies).  Every rank must end with every secret."""
    fake = os.path.join(CXX, "_build", "libfake_rccl.so")
    exe = _build()
    assert os.path.exists(fake)
    r = subprocess.run([exe, "--open-world"], capture_output=True, text=True, env=dict(os.environ, SCL_HIP_RCCL_LIBRARY=fake),
                       timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "1 cases" in r.stdout and "0 failures" in r.stdout and "skipped" not in r.stdout


def _per_secret(args):
    subprocess.run(["make", "-s", "-C", CXX], check=True)
    r = subprocess.run([os.path.join(CXX, "_build", "bench_per_secret")] + args, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("per_secret")][-1]
    return {k: v for k, v in (kv.split("=") for kv in line.split()[1:])}


def test_per_secret_signatures_cost_what_the_reference_does():
    """A caller that keeps the reference's loop-over-secrets shape -- shamirSecretShare(secret, t, n, prg) then
    shamirRecoverP(shares) per secret (include/scl/ss/shamir.h:51-68,99-104) -- must not pay a device round trip per call.
    The mirror's per-secret signatures run on the host (FF's operators = detail/field.hpp, host AES); timed against the real
    reference (oracle/_ref's time_shamir) on the same count: within 2x per call, bit-identical (mismatches = 0)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    try:
        ref = O.Ref()
    except Exception:
        pytest.skip("oracle/_ref is not built here")
    count = 100_000
    best = None
    for _ in range(3):     # a shared CI core is noisy: best of three, both sides
        m = _per_secret([str(count)])
        assert m["mode"] == "host" and m["mismatches"] == "0"
        r = ref.time_shamir(O.M61, count, 3, 10)
        assert r["mismatches"] == 0
        mine = float(m["share_ns"]) + float(m["recover_ns"])
        theirs = 1e9 * (r["share_s"] + r["recover_s"]) / count
        if best is None or mine / theirs < best[0]:
            best = (mine / theirs, m, theirs)
    assert best[0] <= 2.0, f"mirror {best[1]} vs reference {best[2]:.0f} ns per share + recover"


@pytest.mark.gpu
def test_per_secret_signatures_through_the_kernels_for_comparison():
    """the same loop with the host threshold at 0: Vector::random and innerProd inside every call go through their kernels
    (what every call did before the host path) -- still bit-identical, two to three orders of magnitude slower per call"""
    dev = _per_secret(["2000", "--device"])
    host = _per_secret(["2000"])
    assert dev["mode"] == "device" and dev["mismatches"] == "0" and host["mismatches"] == "0"
    assert float(dev["share_ns"]) + float(dev["recover_ns"]) > 10 * (float(host["share_ns"]) + float(host["recover_ns"]))


def test_reference_gf7_compiles_in_place_against_the_mirror(tmp_path):
    """The reference's own example of its field plug-in boundary -- /root/reference/test/scl/gf7.h + gf7.cc:26-103, the traits
    struct and the ff:: specialisations of the integers modulo 7 -- handed to the compiler WHERE THEY LIE, with an include
    directory in which `scl` links to include/scl_hip: the reference's `#include "scl/math/fields/ff_ops.h"` resolves to the
    mirror's.  Runs the reference's Berlekamp-Welch Wikipedia case (test/scl/ss/test_shamir.cc:144-160) and field / Vector /
    sharing identities through FF<GaloisField7> on the mirror's generic host paths.  No reference file is copied."""
    ref = "/root/reference/test/scl"
    if not os.path.exists(os.path.join(ref, "gf7.cc")):
        pytest.skip("the reference is not on this machine")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "secure-computation-library_amd", "csrc"), "-j8"], check=True)
    alias = tmp_path / "alias"
    alias.mkdir()
    os.symlink(os.path.join(ROOT, "include", "scl_hip"), alias / "scl")
    lib = os.path.join(ROOT, "secure-computation-library_amd", "scl_amd")
    exe = str(tmp_path / "ref_gf7_inplace")
    cmd = ["g++", "-std=c++20", "-O1", "-Wall", "-Wextra", "-Wno-unknown-pragmas", f"-I{alias}", f"-I{ROOT}/include", f"-I{ref}",
           "-o", exe, os.path.join(CXX, "ref_gf7_inplace.cc"), os.path.join(ref, "gf7.cc"),
           f"-L{lib}", "-lscl_hip", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib"]
    b = subprocess.run(cmd, capture_output=True, text=True)
    assert b.returncode == 0, b.stderr[-4000:]
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "0 failures" in r.stdout, r.stdout + r.stderr


def test_mont128_small_node_fold_against_long_arithmetic(tmp_path):
    """Mont128::sacc_fold -- the one reduction of the small-node share kernel for the 128-bit Montgomery prime, since round 4 with
    a quotient estimate sharp enough for ONE conditional subtraction -- on two million sums with extreme operands and five
    full-width moduli, against a 192-bit sum reduced bit by bit (tests/cxx/mont128_fold_check.cc).  The GPU half of the same
    arithmetic: tests/test_plugin_field_pins.py (Python big integers) and every Mont128 sharing test."""
    exe = str(tmp_path / "mont128_fold_check")
    b = subprocess.run(["g++", "-std=c++20", "-O2", f"-I{ROOT}/include", "-o", exe, os.path.join(CXX, "mont128_fold_check.cc")],
                       capture_output=True, text=True)
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, r.stdout + r.stderr


def test_mont128_interleaved_product_against_long_arithmetic(tmp_path):
    """Mont128::mul -- since round 6 the Montgomery product interleaved over 32-bit words (the reference's montyModMul form,
    ff_ops_gmp.h:174-191, half the vector instructions of the two-step form) -- against that two-step form and against plain
    long arithmetic (a b R^-1 mod p, bit by bit) over fourteen moduli of every shape and edge operands
    (tests/cxx/mont_mul_check.cc, 840 k products); Mont256::mul over both secp256k1 primes
    the same way against the limb-by-limb form it replaced (600 k products).  The kernels compile the same function; the reference's own values for it:
    tests/golden/golden_mont128.json."""
    exe = str(tmp_path / "mont_mul_check")
    b = subprocess.run(["g++", "-std=c++20", "-O2", f"-I{ROOT}/include", "-o", exe, os.path.join(CXX, "mont_mul_check.cc")],
                       capture_output=True, text=True)
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout + r.stderr


def test_gf2_128_host_arithmetic_against_a_bit_serial_multiplier(tmp_path):
    """Gf128::mul (4-bit-window comb), sqr (bit spread + fold) and inv (Itoh-Tsujii chain) of detail/field.hpp -- the code behind
    FF<GF2_128> on the host and, for sqr / inv, in the kernels -- against a shift-xor multiplier and the 254-product ladder
    (tests/cxx/gf128_host_check.cc): 400 k comparisons, corners included.  CPU only."""
    exe = str(tmp_path / "gf128_host_check")
    b = subprocess.run(["g++", "-std=c++20", "-O2", "-w", f"-I{ROOT}/include", "-o", exe, os.path.join(CXX, "gf128_host_check.cc")],
                       capture_output=True, text=True)
    assert b.returncode == 0, b.stderr
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout

