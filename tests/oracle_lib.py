"""ctypes faces over the two CPU checkers (TEST INFRASTRUCTURE).

* ``Port``  -- oracle/_build/libscl_oracle.so, our plain-C restatement (oracle/scl_oracle.c)
* ``Ref``   -- oracle/_ref/libscl_ref.so, the real reference compiled from /root/reference
               (only where that tree or a prebuilt .so exists)

Both expose the same Python methods; elements travel as numpy uint64 arrays of
shape (..., limbs) and share matrices as AoS [secret][party][limb].
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
PORT_SO = os.path.join(ORACLE_DIR, "_build", "libscl_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libscl_ref.so")

M61, M127, MONT128, GF2_128, SECP256K1_SCALAR, SECP256K1_FIELD = 0, 1, 2, 3, 4, 5
ADD, SUB, MUL, NEG, INV, DIV = range(6)


def Z2K(bits: int) -> int:
    """tag of the ring Z2k<bits> (include/scl/math/z2k.h); the reference harness instantiates REF_RING_BITS"""
    assert 1 <= bits <= 128
    return 0x100 + bits


REF_RING_BITS = (1, 32, 62, 64, 65, 123, 128)


def is_ring(field: int) -> bool:
    return 0x100 < field <= 0x100 + 128


class _Limbs(dict):
    def __missing__(self, field):
        if is_ring(field):
            return 1 if field - 0x100 <= 64 else 2
        raise KeyError(field)


LIMBS = _Limbs({M61: 1, M127: 2, MONT128: 2, GF2_128: 2, SECP256K1_SCALAR: 4, SECP256K1_FIELD: 4})


def byte_size(field: int) -> int:
    """T::byteSize(): what read / write / Vector::random step by"""
    return (field - 0x100 - 1) // 8 + 1 if is_ring(field) else 8 * LIMBS[field]


P = {M61: (1 << 61) - 1, M127: (1 << 127) - 1}

u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_ubyte)
szp = C.POINTER(C.c_size_t)


def build_port() -> str:
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "port"], check=True)
    return PORT_SO


def build_ref() -> str | None:
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "ref"], check=True, stdout=subprocess.DEVNULL)
    return REF_SO if os.path.exists(REF_SO) else None


def _p(a: np.ndarray):
    return a.ctypes.data_as(u64p)


def _b(a):
    return a.ctypes.data_as(u8p)


def _arr(x, limbs: int) -> np.ndarray:
    a = np.ascontiguousarray(x, dtype=np.uint64)
    assert a.shape[-1] == limbs, (a.shape, limbs)
    return a


def to_ints(a: np.ndarray) -> list[int]:
    """(..., limbs) uint64 -> flat list of Python ints"""
    a = np.asarray(a, dtype=np.uint64)
    flat = a.reshape(-1, a.shape[-1])
    return [sum(int(flat[i, j]) << (64 * j) for j in range(flat.shape[1])) for i in range(flat.shape[0])]


def from_ints(vals, limbs: int) -> np.ndarray:
    out = np.zeros((len(vals), limbs), dtype=np.uint64)
    for i, v in enumerate(vals):
        for j in range(limbs):
            out[i, j] = (int(v) >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return out


class OracleError(Exception):
    def __init__(self, status: int, message: str):
        super().__init__(message)
        self.status = status
        self.message = message


class _Base:
    prefix = ""
    has_err = False

    def __init__(self, path: str):
        self.lib = C.CDLL(path)
        self.path = path

    def _f(self, name):
        return getattr(self.lib, self.prefix + name)

    # -- helpers for the two calling conventions (ref passes an error buffer) --
    def _call(self, name, *args, err=False):
        fn = self._f(name)
        fn.restype = C.c_int
        if err and self.has_err:
            buf = C.create_string_buffer(256)
            st = fn(*args, buf, C.c_size_t(256))
            if st:
                raise OracleError(st, buf.value.decode())
            return
        st = fn(*args)
        if st:
            raise OracleError(st, self.status_message(st))

    def status_message(self, st: int) -> str:
        return f"status {st}"

    def limbs(self, field):
        return LIMBS[field]

    def ew(self, field, op, a, b=None):
        L = LIMBS[field]
        a = _arr(a, L)
        bb = _arr(b, L) if b is not None else None
        dst = np.empty_like(a)
        self._call("ew", C.c_int(field), C.c_int(op), _p(dst), _p(a), _p(bb) if bb is not None else None,
                   C.c_size_t(a.size // L), err=True)
        return dst

    def from_int(self, field, v: int):
        dst = np.zeros(LIMBS[field], dtype=np.uint64)
        self._call("from_int", C.c_int(field), C.c_int(v), _p(dst))
        return dst

    def from_bytes(self, field, raw: bytes):
        L = LIMBS[field]
        n = len(raw) // byte_size(field)
        src = np.frombuffer(raw + bytes(16), dtype=np.uint8).copy()  # Z2k::read loads a whole word (z2k_ops.h:109)
        dst = np.zeros((n, L), dtype=np.uint64)
        self._call("from_bytes", C.c_int(field), _b(src), C.c_size_t(n), _p(dst))
        return dst

    def from_hex(self, field, s: str):
        dst = np.zeros(LIMBS[field], dtype=np.uint64)
        self._call("from_hex", C.c_int(field), s.encode(), _p(dst), err=True)
        return dst

    def to_hex(self, field, a):
        a = _arr(a, LIMBS[field])
        buf = C.create_string_buffer(96)
        self._call("to_hex", C.c_int(field), _p(a), buf, C.c_size_t(96))
        return buf.value.decode()

    def exp(self, field, base, e: int):
        base = _arr(base, LIMBS[field])
        dst = np.zeros(LIMBS[field], dtype=np.uint64)
        self._call("exp", C.c_int(field), _p(base), C.c_size_t(e), _p(dst))
        return dst

    def prg(self, seed: bytes, sizes) -> bytes:
        sz = (C.c_size_t * len(sizes))(*sizes)
        out = np.zeros(max(1, sum(sizes)), dtype=np.uint8)
        self._call("prg", seed, C.c_size_t(len(seed)), sz, C.c_size_t(len(sizes)), _b(out))
        return out[: sum(sizes)].tobytes()

    def vector_random(self, field, seed: bytes, n: int):
        out = np.zeros((n, LIMBS[field]), dtype=np.uint64)
        self._call("vector_random", C.c_int(field), seed, C.c_size_t(len(seed)), C.c_size_t(n), _p(out))
        return out

    def shamir_share(self, field, seed: bytes, secrets, t, n):
        L = LIMBS[field]
        secrets = _arr(secrets, L)
        N = secrets.shape[0]
        out = np.zeros((N, n, L), dtype=np.uint64)
        self._call("shamir_share", C.c_int(field), seed, C.c_size_t(len(seed)), _p(secrets), C.c_size_t(N),
                   C.c_size_t(t), C.c_size_t(n), _p(out))
        return out

    def shamir_share_packed(self, field, seed: bytes, secrets, t, n):
        """shamirSecretShare over Array<FF, W>: secrets [N][W][L] -> shares [N][n][W][L]"""
        L = LIMBS[field]
        secrets = np.ascontiguousarray(secrets, dtype=np.uint64)
        N, W = secrets.shape[0], secrets.shape[1]
        out = np.zeros((N, n, W, L), dtype=np.uint64)
        self._call("shamir_share_packed", C.c_int(field), seed, C.c_size_t(len(seed)), _p(secrets), C.c_size_t(N),
                   C.c_size_t(t), C.c_size_t(n), C.c_size_t(W), _p(out))
        return out

    def shamir_recover(self, field, shares):
        L = LIMBS[field]
        shares = _arr(shares, L)
        N, n = shares.shape[0], shares.shape[1]
        out = np.zeros((N, L), dtype=np.uint64)
        self._call("shamir_recover", C.c_int(field), _p(shares), C.c_size_t(n), C.c_size_t(N), _p(out))
        return out

    def shamir_recover_at(self, field, shares, alphas, x):
        L = LIMBS[field]
        shares, alphas, x = _arr(shares, L), _arr(alphas, L), _arr(x, L)
        N, m = shares.shape[0], shares.shape[1]
        out = np.zeros((N, L), dtype=np.uint64)
        self._call("shamir_recover_at", C.c_int(field), _p(shares), _p(alphas), _p(x), C.c_size_t(m),
                   C.c_size_t(N), _p(out), err=True)
        return out

    def shamir_recover_d(self, field, shares, t):
        L = LIMBS[field]
        shares = _arr(shares, L)
        N, n = shares.shape[0], shares.shape[1]
        out = np.zeros((N, L), dtype=np.uint64)
        status = np.zeros(N, dtype=np.uint8)
        self._call("shamir_recover_d", C.c_int(field), _p(shares), C.c_size_t(n), C.c_size_t(t), C.c_size_t(N),
                   _p(out), _b(status))
        return out, status

    def shamir_recover_c(self, field, shares, alphas=None):
        """shamirRecoverC per secret: shares [N][count][L] -> (f [N][3t+1][L], err [N][t+1][L], status [N], nerr [N])"""
        L = LIMBS[field]
        shares = np.ascontiguousarray(shares, dtype=np.uint64)
        N, count = shares.shape[0], shares.shape[1]
        t = (count - 1) // 3
        n = 3 * t + 1
        f = np.zeros((N, n, L), dtype=np.uint64)
        e = np.zeros((N, t + 1, L), dtype=np.uint64)
        status = np.zeros(N, dtype=np.uint8)
        nerr = np.zeros(N, dtype=np.uint32)
        al = _arr(alphas, L) if alphas is not None else None
        self._call("shamir_recover_c", C.c_int(field), _p(shares), _p(al) if al is not None else None, C.c_size_t(count),
                   C.c_size_t(N), _p(f), _p(e), status.ctypes.data_as(u8p), nerr.ctypes.data_as(C.POINTER(C.c_uint32)))
        return f, e, status, nerr

    def lagrange_basis(self, field, nodes, x):
        L = LIMBS[field]
        nodes, x = _arr(nodes, L), _arr(x, L)
        out = np.zeros_like(nodes)
        self._call("lagrange_basis", C.c_int(field), _p(nodes), C.c_size_t(nodes.shape[0]), _p(x), _p(out),
                   err=True)
        return out

    def additive_share(self, field, seed: bytes, secrets, n):
        L = LIMBS[field]
        secrets = _arr(secrets, L)
        N = secrets.shape[0]
        out = np.zeros((N, n, L), dtype=np.uint64)
        self._call("additive_share", C.c_int(field), seed, C.c_size_t(len(seed)), _p(secrets), C.c_size_t(N),
                   C.c_size_t(n), _p(out))
        return out

    def additive_recover(self, field, shares):
        L = LIMBS[field]
        shares = _arr(shares, L)
        N, n = shares.shape[0], shares.shape[1]
        out = np.zeros((N, L), dtype=np.uint64)
        self._call("additive_recover", C.c_int(field), _p(shares), C.c_size_t(n), C.c_size_t(N), _p(out))
        return out

    def dot(self, field, a, b):
        L = LIMBS[field]
        a, b = _arr(a, L), _arr(b, L)
        out = np.zeros(L, dtype=np.uint64)
        self._call("dot", C.c_int(field), _p(a), _p(b), C.c_size_t(a.shape[0]), _p(out))
        return out

    def sum(self, field, a):
        L = LIMBS[field]
        a = _arr(a, L)
        out = np.zeros(L, dtype=np.uint64)
        self._call("sum", C.c_int(field), _p(a), C.c_size_t(a.shape[0]), _p(out))
        return out

    def scalar_mul(self, field, a, scalar):
        L = LIMBS[field]
        a, scalar = _arr(a, L), _arr(scalar, L)
        out = np.zeros_like(a)
        self._call("scalar_mul", C.c_int(field), _p(a), _p(scalar), C.c_size_t(a.shape[0]), _p(out))
        return out

    def poly_eval(self, field, coeffs, xs):
        L = LIMBS[field]
        coeffs, xs = _arr(coeffs, L), _arr(xs, L)
        out = np.zeros_like(xs)
        self._call("poly_eval", C.c_int(field), _p(coeffs), C.c_size_t(coeffs.shape[0]), _p(xs),
                   C.c_size_t(xs.shape[0]), _p(out))
        return out

    def vandermonde(self, field, n, m, xs=None):
        L = LIMBS[field]
        out = np.zeros((n, m, L), dtype=np.uint64)
        xp = _p(_arr(xs, L)) if xs is not None else None
        self._call("vandermonde", C.c_int(field), C.c_size_t(n), C.c_size_t(m), xp, _p(out))
        return out

    def matmul(self, field, A, B):
        L = LIMBS[field]
        A, B = _arr(A, L), _arr(B, L)
        n, k = A.shape[0], A.shape[1]
        m = B.shape[1]
        assert B.shape[0] == k
        out = np.zeros((n, m, L), dtype=np.uint64)
        self._call("matmul", C.c_int(field), _p(A), _p(B), C.c_size_t(n), C.c_size_t(k), C.c_size_t(m), _p(out))
        return out

    def wire_vector(self, field, elems) -> bytes:
        L = LIMBS[field]
        elems = _arr(elems, L).reshape(-1, L)
        n = elems.shape[0]
        out = np.zeros(4 + n * 8 * L, dtype=np.uint8)
        if self.has_err:
            ln = C.c_size_t(0)
            self._call("wire_vector", C.c_int(field), _p(elems), C.c_size_t(n), _b(out), C.byref(ln))
            assert ln.value == out.size
        else:
            fn = self._f("wire_vector")
            fn.restype = C.c_size_t
            assert fn(C.c_int(field), _p(elems), C.c_size_t(n), _b(out)) == out.size
        return out.tobytes()

    def unwire_vector(self, field, raw: bytes):
        L = LIMBS[field]
        src = np.frombuffer(raw, dtype=np.uint8).copy()
        cap = max(1, (len(raw) - 4) // (8 * L))
        out = np.zeros((cap, L), dtype=np.uint64)
        n = C.c_size_t(0)
        if self.has_err:
            self._call("unwire_vector", C.c_int(field), _b(src), _p(out), C.byref(n))
        else:
            self._call("unwire_vector", C.c_int(field), _b(src), C.c_size_t(len(raw)), _p(out), C.c_size_t(cap), C.byref(n))
        return out[: n.value]

    def wire_matrix(self, field, mat) -> bytes:
        L = LIMBS[field]
        mat = np.ascontiguousarray(mat, dtype=np.uint64)
        rows, cols = mat.shape[0], mat.shape[1]
        out = np.zeros(12 + rows * cols * 8 * L, dtype=np.uint8)
        if self.has_err:
            ln = C.c_size_t(0)
            self._call("wire_matrix", C.c_int(field), _p(mat), C.c_size_t(rows), C.c_size_t(cols), _b(out), C.byref(ln))
            assert ln.value == out.size
        else:
            fn = self._f("wire_matrix")
            fn.restype = C.c_size_t
            assert fn(C.c_int(field), _p(mat), C.c_size_t(rows), C.c_size_t(cols), _b(out)) == out.size
        return out.tobytes()

    def unwire_matrix(self, field, raw: bytes):
        L = LIMBS[field]
        src = np.frombuffer(raw, dtype=np.uint8).copy()
        cap = max(1, (len(raw) - 12) // (8 * L))
        out = np.zeros((cap, L), dtype=np.uint64)
        r, c = C.c_size_t(0), C.c_size_t(0)
        if self.has_err:
            self._call("unwire_matrix", C.c_int(field), _b(src), _p(out), C.byref(r), C.byref(c))
        else:
            self._call("unwire_matrix", C.c_int(field), _b(src), C.c_size_t(len(raw)), _p(out), C.c_size_t(cap),
                       C.byref(r), C.byref(c))
        return out[: r.value * c.value].reshape(r.value, c.value, L)

    def frame(self, field, elems, as_matrix=False) -> bytes:
        """TcpChannel frame (u32 packet size || packet) of a Packet holding one Vector / Matrix"""
        L = LIMBS[field]
        a = np.ascontiguousarray(elems, dtype=np.uint64)
        rows, cols = (a.shape[0], a.shape[1]) if as_matrix else (0, a.reshape(-1, L).shape[0])
        inner = self.wire_matrix(field, a) if as_matrix else self.wire_vector(field, a)
        if not self.has_err:   # the port: the frame is the length prefix on the image (tcp_channel.h:127-137)
            return len(inner).to_bytes(4, "little") + inner
        out = np.zeros(4 + len(inner), dtype=np.uint8)
        ln = C.c_size_t(0)
        self._call("frame", C.c_int(field), _p(a), C.c_size_t(rows), C.c_size_t(cols), C.c_int(1 if as_matrix else 0),
                   _b(out), C.byref(ln))
        assert ln.value == out.size
        return out.tobytes()

    def time_shamir(self, field, N, t, n, seed: bytes = b"scl-bench"):
        ss, rs = C.c_double(), C.c_double()
        bad, chk = C.c_uint64(), C.c_uint64()
        self._call("time_shamir", C.c_int(field), C.c_size_t(N), C.c_size_t(t), C.c_size_t(n), seed,
                   C.c_size_t(len(seed)), C.byref(ss), C.byref(rs), C.byref(bad), C.byref(chk))
        return {"share_s": ss.value, "recover_s": rs.value, "mismatches": bad.value, "checksum": chk.value}

    def time_additive(self, field, N, n, seed: bytes = b"scl-bench"):
        """per secret additiveShare + Vector::sum over N secrets FF(int(s)): seconds in each half.  The reference library
        times its own per-secret calls; the port has the batch form only and is timed around it here."""
        if self.has_err:
            ss, rs = C.c_double(), C.c_double()
            bad, chk = C.c_uint64(), C.c_uint64()
            self._call("time_additive", C.c_int(field), C.c_size_t(N), C.c_size_t(n), seed, C.c_size_t(len(seed)),
                       C.byref(ss), C.byref(rs), C.byref(bad), C.byref(chk))
            return {"share_s": ss.value, "recover_s": rs.value, "mismatches": bad.value, "checksum": chk.value}
        import time
        L = LIMBS[field]
        secrets = np.zeros((N, L), dtype=np.uint64)
        secrets[:, 0] = np.arange(N, dtype=np.uint64) & np.uint64(0x7fffffff)
        t0 = time.perf_counter()
        sh = self.additive_share(field, seed, secrets, n)
        t1 = time.perf_counter()
        out = self.additive_recover(field, sh)
        t2 = time.perf_counter()
        return {"share_s": t1 - t0, "recover_s": t2 - t1, "mismatches": int((out != secrets).any(axis=1).sum()),
                "checksum": int(out[:, 0].sum(dtype=np.uint64))}

    def time_shamir_hoisted(self, field, N, t, n, seed: bytes = b"scl-bench"):
        """as time_shamir with the Lagrange basis computed once (reference library only)"""
        ss, rs = C.c_double(), C.c_double()
        bad, chk = C.c_uint64(), C.c_uint64()
        self._call("time_shamir_hoisted", C.c_int(field), C.c_size_t(N), C.c_size_t(t), C.c_size_t(n), seed,
                   C.c_size_t(len(seed)), C.byref(ss), C.byref(rs), C.byref(bad), C.byref(chk))
        return {"share_s": ss.value, "recover_s": rs.value, "mismatches": bad.value, "checksum": chk.value}


class Port(_Base):
    prefix = "sclo_"
    has_err = False

    def __init__(self, path: str | None = None):
        if path is None and os.environ.get("SCL_ORACLE_SO"):   # the sanitizer run points this at the instrumented build
            path = os.environ["SCL_ORACLE_SO"]
        if path is None:
            path = PORT_SO if os.path.exists(PORT_SO) and os.path.getmtime(PORT_SO) >= os.path.getmtime(
                os.path.join(ORACLE_DIR, "scl_oracle.c")) else build_port()
        super().__init__(path)
        self.lib.sclo_status_message.restype = C.c_char_p

    def status_message(self, st):
        return self.lib.sclo_status_message(C.c_int(st)).decode()

    def aes_force(self, mode: int):
        self.lib.sclo_aes_force(C.c_int(mode))

    def prg_blocks(self, seed: bytes, counter0: int, nblocks: int) -> bytes:
        out = np.zeros(16 * max(1, nblocks), dtype=np.uint8)
        self._call("prg_blocks", seed, C.c_size_t(len(seed)), C.c_uint64(counter0), C.c_size_t(nblocks), _b(out))
        return out[: 16 * nblocks].tobytes()

    def shamir_share_coeffs(self, field, secrets, coeffs, n):
        L = LIMBS[field]
        secrets, coeffs = _arr(secrets, L), _arr(coeffs, L)
        N, t = coeffs.shape[0], coeffs.shape[1]
        out = np.zeros((N, n, L), dtype=np.uint64)
        self._call("shamir_share_coeffs", C.c_int(field), _p(secrets), _p(coeffs), C.c_size_t(N), C.c_size_t(t),
                   C.c_size_t(n), _p(out))
        return out

    def shamir_recover_lambda(self, field, shares, lam):
        L = LIMBS[field]
        shares, lam = _arr(shares, L), _arr(lam, L)
        N, n = shares.shape[0], shares.shape[1]
        out = np.zeros((N, L), dtype=np.uint64)
        self._call("shamir_recover_lambda", C.c_int(field), _p(shares), _p(lam), C.c_size_t(n), C.c_size_t(N), _p(out))
        return out

    def mont128_set_prime(self, p: int):
        a = from_ints([p], 2)[0]
        self._call("mont128_set_prime", _p(a))

    def mont128_get_prime(self) -> int:
        a = np.zeros(2, dtype=np.uint64)
        self.lib.sclo_mont128_get_prime(_p(a))
        return to_ints(a.reshape(1, 2))[0]


class Ref(_Base):
    prefix = "sclref_"
    has_err = True

    def __init__(self, path: str | None = None, fresh: bool = False):
        """fresh=True loads a private COPY of the library (its own statics): the reference's FF::one() / zero() are
        function-local statics that latch the modulus of field tag 2 at first use, so a second Mont128 prime needs a second
        instance"""
        if path is None:
            path = REF_SO if os.path.exists(REF_SO) else build_ref()
        if path is None or not os.path.exists(path):
            raise FileNotFoundError("oracle/_ref/libscl_ref.so unavailable (no /root/reference here)")
        self._tmp = None
        if fresh:
            import shutil
            import tempfile
            fd, tmp = tempfile.mkstemp(prefix="libscl_ref_", suffix=".so")
            os.close(fd)
            shutil.copyfile(path, tmp)
            self._tmp = path = tmp
        super().__init__(path)
        if self._tmp:            # (the mapping stays valid after the name is gone)
            os.unlink(self._tmp)

    def mont128_set_prime(self, p: int):
        """modulus of field tag 2: the reference's Montgomery templates (ff_ops_gmp.h) instantiated at two limbs
        (oracle/ref_harness.cc); 2^128 - 159 until set"""
        a = from_ints([p], 2)[0]
        self._call("mont128_set_prime", _p(a))

    def mont128_get_prime(self) -> int:
        a = np.zeros(2, dtype=np.uint64)
        self.lib.sclref_mont128_get_prime(_p(a))
        return to_ints(a.reshape(1, 2))[0]


def ref_available() -> bool:
    return os.path.exists(REF_SO) or os.path.isdir("/root/reference/include/scl")
