"""CPU-side checks of the drop-in boundary: the library loads, exports every symbol that
include/scl_hip.h declares, and its host-only entry points (tables, messages) behave like the
reference.  No kernels are launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def scl():
    import scl_amd
    return scl_amd


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "scl_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(scl_hip_\w+)\s*\(", src)))


def test_every_declared_symbol_is_exported(scl):
    names = declared_symbols()
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(scl.lib, n)]
    assert not missing, missing


def test_nothing_but_the_declared_symbols_is_exported(scl):
    """The library's dynamic symbol table is the header's prototypes and nothing else (csrc/exports.map, written by
    tools/gen_capi_route.py): the per-family unit names (scl_hip_ew__m61 ..) and the state the units share cannot be called
    or interposed from outside."""
    import subprocess
    so = os.path.join(ROOT, "secure-computation-library_amd", "scl_amd", "libscl_hip.so")
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert exported == set(declared_symbols()), sorted(exported ^ set(declared_symbols()))


def test_metadata_and_messages(scl):
    assert scl.lib.scl_hip_abi_version() == 2
    assert [scl.limbs(f) for f in range(6)] == [1, 2, 2, 2, 4, 4]
    # names pinned by test/scl/math/test_mersenne61.cc:28-33, test_mersenne127.cc:28-33
    assert scl.field_name(0) == "Mersenne61" and scl.field_name(1) == "Mersenne127"
    assert scl.field_name(4) == "secp256k1_order"  # include/scl/math/fields/secp256k1_scalar.h
    assert scl.field_name(5) == "secp256k1_field"  # include/scl/math/fields/secp256k1_field.h
    msg = lambda s: scl.lib.scl_hip_status_message(s).decode()
    assert msg(scl.ERR_ZERO_INVERSE) == "0 not invertible modulo prime"
    assert msg(scl.ERR_SIZE_MISMATCH) == "Vec sizes mismatch"
    assert msg(scl.ERR_ERROR_DETECTED) == "error detected during recovery"
    assert msg(scl.ERR_NOT_ENOUGH_SHARES) == "not enough shares provided to detect errors"
    assert msg(scl.ERR_MATMUL_DIMS) == "matmul: this->cols() != that->rows()"
    assert msg(scl.ERR_VANDERMONDE_XS) == "|xs| != number of rows"
    assert msg(scl.ERR_INVALID_RANGE) == "invalid range"


@pytest.mark.parametrize("f", [O.M61, O.M127, O.MONT128, O.GF2_128, O.SECP256K1_SCALAR, O.SECP256K1_FIELD])
def test_host_lagrange_basis_matches_oracle(scl, f):
    """the hoisted basis (host table code in csrc/capi.hip, Fermat inverses) == the oracle's
    per-factor Euclid divisions (lagrange.h:54-71)"""
    port = O.Port()
    L = O.LIMBS[f]
    for m in (1, 2, 4, 10, 40, 128):
        if f in (O.MONT128, O.GF2_128, O.SECP256K1_SCALAR, O.SECP256K1_FIELD) and m > 40:
            continue
        nodes = O.from_ints(list(range(1, m + 1)), L) if f == O.GF2_128 else np.stack(
            [port.from_int(f, i + 1) for i in range(m)])
        want = port.lagrange_basis(f, nodes, port.from_int(f, 0))
        assert np.array_equal(scl.lagrange_basis(f, m), want), (f, m)
    nodes = port.vector_random(f, b"nodes", 9)
    x = port.vector_random(f, b"x", 1)[0]
    assert np.array_equal(scl.lagrange_basis(f, 9, nodes, x), port.lagrange_basis(f, nodes, x))
    nodes[3] = nodes[7]
    with pytest.raises(scl.SclError) as ei:
        scl.lagrange_basis(f, 9, nodes, x)
    assert ei.value.status == scl.ERR_ZERO_INVERSE


def test_mont128_prime_roundtrip(scl):
    p = np.zeros(2, dtype=np.uint64)
    scl.lib.scl_hip_mont128_get_prime(p.ctypes.data_as(C.POINTER(C.c_uint64)))
    assert O.to_ints(p.reshape(1, 2))[0] == 2 ** 128 - 159
    with pytest.raises(scl.SclError):
        scl.set_mont128_prime(2 ** 100)  # even


def test_mont128_default_latch_goes_stale_loudly(scl):
    """The Mont128 modulus rule (scl_hip.h): a thread that never set a prime latches the process-wide default at its first
    Mont128 call and keeps it -- and when another thread changes the default afterwards, the latched thread's next Mont128
    call fails (SCL_ERR_BAD_ARG) instead of computing over the old modulus in silence, until it re-latches or sets its own.
    A thread that set its own modulus is never disturbed.  Host-only entry point (the Lagrange table), so this runs on CPU."""
    from concurrent.futures import ThreadPoolExecutor
    p0, p1, p2 = 2 ** 128 - 159, 2 ** 127 - 1, 2 ** 61 - 1
    f = scl.MONT128
    try:
        scl.set_mont128_prime(p0)
        with ThreadPoolExecutor(max_workers=1) as worker:          # one thread, reused for every submit
            assert worker.submit(scl.mont128_prime).result() == p0     # the worker latches the default
            base0 = worker.submit(scl.lagrange_basis, f, 3).result()
            scl.set_mont128_prime(p1)                                  # the main thread moves the default
            with pytest.raises(scl.SclError) as ei:
                worker.submit(scl.lagrange_basis, f, 3).result()
            assert ei.value.status == scl.ERR_BAD_ARG and "latched" in str(ei.value)
            assert worker.submit(scl.mont128_prime).result() == p0     # (reporting the thread's modulus never fails)
            worker.submit(scl.mont128_relatch).result()
            assert worker.submit(scl.mont128_prime).result() == p1
            base1 = worker.submit(scl.lagrange_basis, f, 3).result()
            assert not np.array_equal(base0, base1) and np.array_equal(base1, scl.lagrange_basis(f, 3))
            worker.submit(scl.set_mont128_prime, p2).result()          # a modulus of the worker's own (also the new default)
            scl.set_mont128_prime(p0)
            assert worker.submit(scl.mont128_prime).result() == p2
            worker.submit(scl.lagrange_basis, f, 3).result()           # undisturbed by the main thread's change
    finally:
        scl.set_mont128_prime(p0)


def test_null_handles_are_an_error_not_a_crash(scl):
    """the small host-side entry points check their pointers before they touch the runtime: a NULL handle or output is
    SCL_ERR_BAD_ARG with a message, never a dereference"""
    lib = scl.lib
    bad = 3  # SCL_ERR_BAD_ARG
    null = C.c_void_p(None)
    ms = C.c_float(0)
    assert lib.scl_hip_timer_start(null, null) == bad and b"NULL" in lib.scl_hip_last_error()
    assert lib.scl_hip_timer_stop(null, null) == bad
    assert lib.scl_hip_timer_elapsed_ms(null, C.byref(ms)) == bad
    assert lib.scl_hip_timer_create(null) == bad
    assert lib.scl_hip_timer_destroy(null) == 0          # like free(NULL)
    assert lib.scl_hip_stream_create(null) == bad
    assert lib.scl_hip_mont128_set_prime(null) == bad
    assert lib.scl_hip_mont128_get_prime(null) == bad
    assert lib.scl_hip_malloc(null, C.c_size_t(16)) == bad
    assert lib.scl_hip_comm_destroy(null) == 0


def test_batch_calls_fail_loudly_without_a_gpu(scl):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cnt = C.c_int(-1)
    st = scl.lib.scl_hip_device_count(C.byref(cnt))
    assert st != 0 or cnt.value == 0
    # a device pointer is required: host tensors are rejected, nothing is computed on the CPU
    a = torch.zeros(4, 1, dtype=torch.int64)
    with pytest.raises(scl.SclError):
        scl.ew(O.M61, O.ADD, a, a)


def test_bench_cpu_baseline_leg():
    """bench.py's cpu_baseline object (the reference where oracle/_ref is built, else the port) on a tiny sample"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cb = bench.cpu_baseline("m61", 10, 3, 2000)
    assert cb["unit"] == "reconstructions/s" and cb["cores"] == 1 and cb["kind"] in ("reference", "port")
    assert cb["value"] > 1e4 and "2000 secrets" in cb["sample"]


def test_recover_detect_refuses_a_node_range_it_does_not_have(scl):
    """shamirRecoverD(shares, alphas, t = 0, d, x) with exactly d shares passes the reference's size test
    (shares.size() >= d + t, shamir.h:122-124) and then interpolates from d + 1 shares and nodes -- past the end of
    both vectors (alphas.subVector(d + 1) only checks start <= end).  The boundary refuses it with the text of the
    range check before anything is launched; one share short of d + t is still "not enough shares"."""
    dummy = (C.c_uint64 * 8)()
    bad = C.c_size_t(7)
    args = lambda m, t, d: (0, dummy, dummy, dummy, 4, m, 4, t, d, None, None, C.byref(bad), None)
    assert scl.lib.scl_hip_shamir_recover_detect(*args(3, 0, 3)) == scl.ERR_INVALID_RANGE
    assert scl.lib.scl_hip_last_error().decode() == "invalid range"
    assert scl.lib.scl_hip_shamir_recover_detect(*args(1, 0, 1)) == scl.ERR_INVALID_RANGE
    assert scl.lib.scl_hip_shamir_recover_detect(*args(5, 3, 3)) == scl.ERR_NOT_ENOUGH_SHARES
    assert scl.lib.scl_hip_shamir_recover_detect(*args(0, 0, 0)) == scl.ERR_INVALID_RANGE


def test_prototypes_come_from_the_header(scl):
    """every entry point has argtypes / restype taken from include/scl_hip.h: a Python int in a pointer slot of the wrong
    kind or a missing argument is an error at the call, not a wild pointer on the device"""
    names = declared_symbols()
    assert scl._NPROTO == len(names)
    assert scl.lib.scl_hip_shamir_share.argtypes[2] is C.c_size_t and scl.lib.scl_hip_wire_size.restype is C.c_size_t
    with pytest.raises((C.ArgumentError, TypeError)):
        scl.lib.scl_hip_limbs()                      # too few arguments
    with pytest.raises((C.ArgumentError, TypeError)):
        scl.lib.scl_hip_limbs("m61")                 # not an int


def test_route_tables_are_current():
    """csrc/capi_names.inc and csrc/capi_route.cc (the per-family unit names and the forwarding entry points) are what
    tools/gen_capi_route.py writes from include/scl_hip.h today"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_capi_route.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
