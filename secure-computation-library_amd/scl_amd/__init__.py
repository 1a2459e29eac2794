"""scl_amd -- Python harness over libscl_hip.so, the MI355X-native finite-field /
secret-sharing engine behind SCL's API (include/scl_hip.h is the boundary).

This module is plumbing: it hands torch device buffers (`data_ptr()`) and the
current HIP stream to the C ABI.  All arithmetic runs in the HIP kernels of
csrc/; there is no CPU or torch fallback -- if the extension is missing the
import fails, and if no GPU is present every batch call raises SclError.

Conventions: an element tensor has dtype int64 (the bits of the uint64 limbs)
and trailing dimension `limbs(field)`; a share matrix is SoA `[party][secret][limb]`.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch  # imported BEFORE the extension so both share torch's libamdhip64 (same SONAME)

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libscl_hip.so")
if not os.path.exists(_SO):
    raise ImportError(
        f"{_SO} is missing: build the HIP extension first "
        "(python -c 'import __graft_entry__ as g; g.build()' or make -C secure-computation-library_amd/csrc)")
lib = C.CDLL(_SO)

M61, M127, MONT128, GF2_128, SECP256K1_SCALAR, SECP256K1_FIELD = 0, 1, 2, 3, 4, 5


def Z2K(bits: int) -> int:
    """tag of the ring scl::math::Z2k<bits> (include/scl/math/z2k.h), 1 <= bits <= 128; see SCL_Z2K in scl_hip.h"""
    if not 1 <= bits <= 128:
        raise ValueError("Z2k bit size must be in 1..128")
    return 0x100 + bits


def byte_size(field) -> int:
    """T::byteSize(): the stride of read / Vector::random -- (K-1)/8 + 1 for Z2k<K>, 8 * limbs for the fields"""
    return (field - 0x100 - 1) // 8 + 1 if 0x100 < field <= 0x100 + 128 else 8 * limbs(field)

ADD, SUB, MUL, NEG, INV, DIV = range(6)
OK, ERR_SIZE_MISMATCH, ERR_ZERO_INVERSE, ERR_BAD_ARG, ERR_HIP, ERR_NO_DEVICE, ERR_ERROR_DETECTED, \
    ERR_NOT_ENOUGH_SHARES, ERR_MATMUL_DIMS, ERR_VANDERMONDE_XS, ERR_INVALID_RANGE = range(11)



def _declare_prototypes():
    """argtypes / restype of every entry point, read from the prototypes of include/scl_hip.h (the boundary's one
    source of truth): a forgotten c_size_t wrap or a swapped argument is then a ctypes.ArgumentError, not a wild
    pointer.  Every pointer parameter is c_void_p (accepts ints, data_ptr() wrappers, bytes, byref() and None)."""
    import re
    import warnings
    # the repository's header, or the copy the csrc Makefile leaves next to the .so (a package moved without the tree)
    candidates = [os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "scl_hip.h"), os.path.join(_HERE, "scl_hip.h")]
    hdr = next((c for c in candidates if os.path.exists(c)), None)
    if hdr is None:
        # no header to read: result types only (status ints, the two size_t / string getters); arguments unchecked
        warnings.warn("scl_amd: include/scl_hip.h not found beside the package; ctypes argument checking is off")
        for name, ret in (("scl_hip_last_error", C.c_char_p), ("scl_hip_status_message", C.c_char_p),
                          ("scl_hip_field_name", C.c_char_p), ("scl_hip_wire_size", C.c_size_t),
                          ("scl_hip_wire_size_matrix", C.c_size_t), ("scl_hip_frame_size", C.c_size_t)):
            if hasattr(lib, name):
                getattr(lib, name).restype = ret
        return 0
    src = re.sub(r"/\*.*?\*/", "", open(hdr).read(), flags=re.S)
    # the boundary's version first: a library built from an older header fails HERE, with a message that says so, not at the
    # lookup of a symbol it does not have
    want = re.search(r"#define\s+SCL_HIP_ABI_VERSION\s+(\d+)", src)
    lib.scl_hip_abi_version.restype = C.c_int
    have = lib.scl_hip_abi_version()
    if want and have != int(want.group(1)):
        raise ImportError(f"scl_amd: {_SO} implements ABI version {have}, {hdr} declares {want.group(1)}: rebuild the extension "
                          "(make -C secure-computation-library_amd/csrc)")
    scalars = {"int": C.c_int, "long": C.c_long, "size_t": C.c_size_t, "uint64_t": C.c_uint64, "unsigned": C.c_uint,
               "float": C.c_float}
    rets = {"int": C.c_int, "size_t": C.c_size_t, "const char*": C.c_char_p}
    n = 0
    for m in re.finditer(r"\b(int|size_t|const char\s*\*)\s*(scl_hip_\w+)\s*\(([^;{]*?)\)\s*;", src):
        ret, name, params = m.group(1).replace(" ", "").replace("constchar*", "const char*"), m.group(2), m.group(3)
        fn = getattr(lib, name)
        fn.restype = rets[ret]
        argt = []
        for prm in [x.strip() for x in params.split(",")]:
            if prm in ("void", ""):
                continue
            if "*" in prm or "[" in prm:
                argt.append(C.c_void_p)
            else:
                base = re.sub(r"\bconst\b", "", prm).split()
                if base[0] not in scalars:
                    raise ImportError(f"scl_amd: {name}: parameter type {base[0]!r} in {hdr} has no ctypes mapping here")
                argt.append(scalars[base[0]])
        fn.argtypes = argt
        n += 1
    return n


_NPROTO = _declare_prototypes()

_u64p = C.POINTER(C.c_uint64)


class SclError(RuntimeError):
    """status + the message the reference would put in its C++ exception"""

    def __init__(self, status: int, detail: str):
        self.status = status
        self.reference_message = lib.scl_hip_status_message(status).decode()
        super().__init__(f"[{status}] {self.reference_message}" + (f" ({detail})" if detail and detail != self.reference_message else ""))


def _chk(status: int):
    if status != OK:
        raise SclError(status, lib.scl_hip_last_error().decode())


def limbs(field: int) -> int:
    n = lib.scl_hip_limbs(field)
    if n < 0:
        raise SclError(ERR_BAD_ARG, "unknown field tag")
    return n


def field_name(field: int) -> str:
    return lib.scl_hip_field_name(field).decode()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t: torch.Tensor):
    if not t.is_cuda:
        raise SclError(ERR_BAD_ARG, "tensor is not on the GPU")
    if not t.is_contiguous():
        raise SclError(ERR_BAD_ARG, "tensor is not contiguous")
    return C.c_void_p(t.data_ptr())


def _want(t: torch.Tensor, shape, what: str, like: torch.Tensor | None = None):
    """out= / operand checks before a raw pointer crosses the boundary: exact shape, int64 limbs, contiguous, on the
    same device as `like` -- a short buffer would otherwise be a silent out-of-bounds HBM access"""
    if t.dtype != torch.int64:
        raise SclError(ERR_BAD_ARG, f"{what}: dtype {t.dtype}, expected int64 limbs")
    if tuple(t.shape) != tuple(shape):
        raise SclError(ERR_SIZE_MISMATCH, f"{what}: shape {tuple(t.shape)}, expected {tuple(shape)}")
    if like is not None and t.device != like.device:
        raise SclError(ERR_BAD_ARG, f"{what}: on {t.device}, expected {like.device}")
    return t


def _dev_rows(t: torch.Tensor):
    """(device pointer, row stride in elements) of a share matrix [rows][N][L] whose rows are dense but may sit a pitch
    apart -- a window of a larger matrix (a chunk, a shard), the C ABI's `stride >= N`"""
    if not t.is_cuda:
        raise SclError(ERR_BAD_ARG, "tensor is not on the GPU")
    rows, N, L = t.shape
    # (the stride of a dimension of extent 1 means nothing: a [rows][1][L] view of a transposed array carries any value there)
    if N and ((L > 1 and t.stride(2) != 1) or (N > 1 and t.stride(1) != L) or
              (rows > 1 and (t.stride(0) % L or t.stride(0) < N * L))):
        raise SclError(ERR_BAD_ARG, "share matrix rows must be dense (a row pitch is allowed)")
    return C.c_void_p(t.data_ptr()), (t.stride(0) // L if rows > 1 else max(N, 1))


def _host(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64)


def _hp(a: np.ndarray):
    return a.ctypes.data_as(_u64p)


def to_device(a, device="cuda") -> torch.Tensor:
    """numpy uint64 (..., limbs) -> int64 device tensor"""
    return torch.from_numpy(_host(a).view(np.int64)).to(device)


def to_host(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().numpy().view(np.uint64)


def empty(field: int, *shape, device="cuda") -> torch.Tensor:
    return torch.empty(*shape, limbs(field), dtype=torch.int64, device=device)


def set_tuning(key: str, value: int):
    _chk(lib.scl_hip_set_tuning(key.encode(), C.c_long(value)))


def set_mont128_prime(p: int):
    """Modulus of the MONT128 field for the calling thread AND the process-wide default (scl_hip.h).  A thread that never
    calls this -- a ThreadPoolExecutor worker, say -- LATCHES the default at its first Mont128 call and keeps it; when the
    default changes afterwards that worker's next Mont128 call raises SclError (ERR_BAD_ARG) instead of computing on over
    the old modulus in silence: the worker then calls set_mont128_prime (its own modulus) or mont128_relatch (the current
    default).  A thread that has called this keeps its own modulus whatever the others do."""
    a = np.array([p & (2 ** 64 - 1), p >> 64], dtype=np.uint64)
    _chk(lib.scl_hip_mont128_set_prime(_hp(a)))


def mont128_relatch():
    """the calling thread takes the current process-wide MONT128 modulus (see set_mont128_prime)"""
    _chk(lib.scl_hip_mont128_relatch())


def mont128_prime() -> int:
    a = np.zeros(2, dtype=np.uint64)
    _chk(lib.scl_hip_mont128_get_prime(_hp(a)))
    return int(a[0]) | (int(a[1]) << 64)


# ---- element-wise: scl::math::Vector<FF> ------------------------------------------------------------
def ew(field, op, a, b=None, out=None):
    L = limbs(field)
    if b is not None and b.shape != a.shape:
        raise SclError(ERR_SIZE_MISMATCH, "")  # Vector::ensureCompatible
    if out is None:
        out = torch.empty_like(a)
    n = a.numel() // L
    _chk(lib.scl_hip_ew(field, op, _dev(out), _dev(a), _dev(b) if b is not None else None, C.c_size_t(n), _stream()))
    return out


def ew_status_buffer(device="cuda") -> torch.Tensor:
    """one cleared 32-bit device word for ew_status (scl_hip_ew_status's status_dev)"""
    return torch.zeros(1, dtype=torch.int32, device=device)


def ew_status(field, op, a, b, status, out=None):
    """scl_hip_ew_status: the element-wise call without a host synchronisation -- a zero (even, in a ring) operand of INV / DIV
    ORs 1 into the device word `status` (an int32 tensor of one element, never cleared by the call) instead of raising; read it
    with status.item() when the caller synchronises anyway.  Capturable into a hipGraph."""
    L = limbs(field)
    if b is not None and b.shape != a.shape:
        raise SclError(ERR_SIZE_MISMATCH, "")
    if out is None:
        out = torch.empty_like(a)
    if status is not None and (status.dtype != torch.int32 or status.numel() != 1 or not status.is_cuda):
        raise SclError(ERR_BAD_ARG, "status: one int32 element on the GPU")
    n = a.numel() // L
    _chk(lib.scl_hip_ew_status(field, op, _dev(out), _dev(a), _dev(b) if b is not None else None, C.c_size_t(n),
                               C.c_void_p(status.data_ptr()) if status is not None else None, _stream()))
    return out


def scalar_mul(field, a, scalar, out=None):
    L = limbs(field)
    if out is None:
        out = torch.empty_like(a)
    s = _host(scalar).reshape(L)
    _chk(lib.scl_hip_scalar_mul(field, _dev(out), _dev(a), _hp(s), C.c_size_t(a.numel() // L), _stream()))
    return out


def vsum(field, a) -> np.ndarray:
    L = limbs(field)
    out = np.zeros(L, dtype=np.uint64)
    _chk(lib.scl_hip_sum(field, _hp(out), _dev(a), C.c_size_t(a.numel() // L), _stream()))
    return out


def dot(field, a, b) -> np.ndarray:
    L = limbs(field)
    if b.shape != a.shape:
        raise SclError(ERR_SIZE_MISMATCH, "")
    out = np.zeros(L, dtype=np.uint64)
    _chk(lib.scl_hip_dot(field, _hp(out), _dev(a), _dev(b), C.c_size_t(a.numel() // L), _stream()))
    return out


def equals(field, a, b) -> bool:
    if a.shape != b.shape:
        return False  # Vector::equals (vector.h:559-561)
    eq = C.c_int(0)
    _chk(lib.scl_hip_equals(field, C.byref(eq), _dev(a), _dev(b), C.c_size_t(a.numel() // limbs(field)), _stream()))
    return bool(eq.value)


# ---- randomness: scl::util::PRG ------------------------------------------------------------------------
def prg_blocks(nblocks: int, seed: bytes, counter0: int = 0, device="cuda", out=None) -> torch.Tensor:
    """PRG::next: blocks counter0 .. counter0 + nblocks - 1 of the stream as bytes; `out` (uint8, >= 16 nblocks bytes) is reused"""
    if out is None:
        out = torch.empty(max(nblocks, 1) * 16, dtype=torch.uint8, device=device)
    elif out.dtype != torch.uint8 or out.numel() < nblocks * 16:
        raise SclError(ERR_BAD_ARG, "prg_blocks out: uint8 tensor of at least 16 * nblocks bytes")
    _chk(lib.scl_hip_prg_blocks(_dev(out), C.c_size_t(nblocks), seed, C.c_size_t(len(seed)), C.c_uint64(counter0),
                                _stream()))
    return out[: nblocks * 16]


def from_bytes(field, raw: torch.Tensor) -> torch.Tensor:
    n = raw.numel() // byte_size(field)
    out = empty(field, n, device=raw.device)
    _chk(lib.scl_hip_from_bytes(field, _dev(out), _dev(raw), C.c_size_t(n), _stream()))
    return out


def vector_random(field, n: int, seed: bytes, counter0: int = 0, device="cuda", out=None) -> torch.Tensor:
    if out is None:
        out = empty(field, n, device=device)
    else:
        _want(out, (n, limbs(field)), "vector_random out")
    _chk(lib.scl_hip_vector_random(field, _dev(out), C.c_size_t(n), seed, C.c_size_t(len(seed)),
                                   C.c_uint64(counter0), _stream()))
    return out


# ---- Shamir: scl::ss -----------------------------------------------------------------------------------------
def lagrange_basis(field, m: int, alphas=None, x=None) -> np.ndarray:
    L = limbs(field)
    out = np.zeros((m, L), dtype=np.uint64)
    al = _host(alphas).reshape(m, L) if alphas is not None else None
    xx = _host(x).reshape(L) if x is not None else None
    _chk(lib.scl_hip_lagrange_basis(field, _hp(out), _hp(al) if al is not None else None, C.c_size_t(m),
                                    _hp(xx) if xx is not None else None))
    return out


def shamir_share(field, secrets, coeffs, n: int, alphas=None, out=None):
    """secrets [N][L]; coeffs [t][N][L] (c_1..c_t) -> shares [n][N][L]"""
    L = limbs(field)
    N = secrets.shape[0]
    _want(secrets, (N, L), "shamir_share secrets")
    t = 0 if coeffs is None else coeffs.shape[0]
    if coeffs is not None:
        _want(coeffs, (t, N, L), "shamir_share coeffs", secrets)
    if out is None:
        out = empty(field, n, N, device=secrets.device)
    else:
        _want(out, (n, N, L), "shamir_share out", secrets)
    al = _host(alphas).reshape(n, L) if alphas is not None else None
    _chk(lib.scl_hip_shamir_share(field, _dev(out), C.c_size_t(N), _dev(secrets),
                                  _dev(coeffs) if t else None, C.c_size_t(N), C.c_size_t(N), C.c_size_t(t),
                                  C.c_size_t(n), _hp(al) if al is not None else None, _stream()))
    return out


def blocks_per_secret(field, t: int) -> int:
    """AES blocks one shamirSecretShare call consumes: ceil((t+1)*byteSize/16)"""
    return ((t + 1) * 8 * limbs(field) + 15) // 16


def shamir_share_prg(field, secrets, t: int, n: int, seed: bytes, first_secret: int = 0, out=None, counter0=None):
    """first_secret: index of secrets[0] in a longer single-PRG run (counter0 = first_secret * B);
    counter0 overrides it with an explicit PRG block counter"""
    N = secrets.shape[0]
    _want(secrets, (N, limbs(field)), "shamir_share_prg secrets")
    if counter0 is None:
        counter0 = first_secret * blocks_per_secret(field, t)
    if out is None:
        out = empty(field, n, N, device=secrets.device)
    else:
        _want(out, (n, N, limbs(field)), "shamir_share_prg out", secrets)
    _chk(lib.scl_hip_shamir_share_prg(field, _dev(out), C.c_size_t(N), _dev(secrets), C.c_size_t(N), C.c_size_t(t),
                                      C.c_size_t(n), seed, C.c_size_t(len(seed)), C.c_uint64(counter0),
                                      _stream()))
    return out


def shamir_recover(field, shares, lam=None, out=None):
    """shares [m][N][L] -> out [N][L]; lam defaults to the basis for nodes 1..m at x=0"""
    L = limbs(field)
    m, N = shares.shape[0], shares.shape[1]
    _want(shares, (m, N, L), "shamir_recover shares")
    if lam is None:
        lam = lagrange_basis(field, m)
    lam = _host(lam).reshape(m, L)
    if out is None:
        out = empty(field, N, device=shares.device)
    else:
        _want(out, (N, L), "shamir_recover out", shares)
    sp, stride = _dev_rows(shares)
    _chk(lib.scl_hip_shamir_recover(field, _dev(out), sp, C.c_size_t(stride), _hp(lam), C.c_size_t(m),
                                    C.c_size_t(N), _stream()))
    return out


def shamir_recover_detect(field, shares, t: int, d: int | None = None, alphas=None, x=None):
    """-> (out [N][L], status uint8 [N], num_bad); raises nothing for detected errors (see status)"""
    L = limbs(field)
    m, N = shares.shape[0], shares.shape[1]
    d = t if d is None else d
    out = empty(field, N, device=shares.device)
    status = torch.empty(max(N, 1), dtype=torch.uint8, device=shares.device)
    al = _host(alphas).reshape(m, L) if alphas is not None else None
    xx = _host(x).reshape(L) if x is not None else None
    bad = C.c_size_t(0)
    st = lib.scl_hip_shamir_recover_detect(field, _dev(out), _dev(status), _dev(shares), C.c_size_t(N), C.c_size_t(m),
                                           C.c_size_t(N), C.c_size_t(t), C.c_size_t(d),
                                           _hp(al) if al is not None else None, _hp(xx) if xx is not None else None,
                                           C.byref(bad), _stream())
    if st not in (OK, ERR_ERROR_DETECTED):
        _chk(st)
    return out, status[:N], bad.value


def shamir_share_prg_packed(field, secrets, t: int, n: int, seed: bytes, counter0: int = 0):
    """shamirSecretShare over Array<FF, W>: secrets [W][N][L] -> shares [W][n][N][L] (component-major SoA)"""
    W, N = secrets.shape[0], secrets.shape[1]
    out = empty(field, W, n, N, device=secrets.device)
    _chk(lib.scl_hip_shamir_share_prg_packed(field, _dev(out), C.c_size_t(N), _dev(secrets), C.c_size_t(N), C.c_size_t(N),
                                             C.c_size_t(t), C.c_size_t(n), C.c_size_t(W), seed, C.c_size_t(len(seed)),
                                             C.c_uint64(counter0), _stream()))
    return out


def shamir_recover_correct(field, shares, alphas=None):
    """Batched shamirRecoverC (Berlekamp-Welch).  shares [m][N][L] -> dict(f [3t+1][N][L], err [t+1][N][L],
    status uint8 [N], nerr int32 [N], queued, failed); the secrets are f[0]."""
    L = limbs(field)
    m, N = shares.shape[0], shares.shape[1]
    t = (m - 1) // 3
    n = 3 * t + 1
    f = empty(field, n, N, device=shares.device)
    e = empty(field, t + 1, N, device=shares.device)
    status = torch.empty(max(N, 1), dtype=torch.uint8, device=shares.device)
    nerr = torch.empty(max(N, 1), dtype=torch.int32, device=shares.device)
    al = _host(alphas).reshape(-1, L) if alphas is not None else None
    queued, failed = C.c_size_t(0), C.c_size_t(0)
    _chk(lib.scl_hip_shamir_recover_correct(field, _dev(f), C.c_size_t(N), _dev(e), C.c_size_t(N), _dev(status), _dev(nerr),
                                            _dev(shares), C.c_size_t(N), C.c_size_t(m), C.c_size_t(N),
                                            _hp(al) if al is not None else None, C.byref(queued), C.byref(failed),
                                            _stream()))
    return {"f": f, "err": e, "status": status[:N], "nerr": nerr[:N], "queued": queued.value, "failed": failed.value}


# ---- additive ----------------------------------------------------------------------------------------------------
def additive_share(field, secrets, rnd, n: int, out=None):
    N, L = secrets.shape[0], limbs(field)
    _want(secrets, (N, L), "additive_share secrets")
    if rnd is not None:
        _want(rnd, (n - 1, N, L), "additive_share rnd", secrets)
    if out is None:
        out = empty(field, n, N, device=secrets.device)
    else:
        _want(out, (n, N, L), "additive_share out", secrets)
    _chk(lib.scl_hip_additive_share(field, _dev(out), C.c_size_t(N), _dev(secrets),
                                    _dev(rnd) if rnd is not None else None, C.c_size_t(N), C.c_size_t(N),
                                    C.c_size_t(n), _stream()))
    return out


def additive_share_prg(field, secrets, n: int, seed: bytes, first_secret: int = 0, out=None, counter0=None):
    N = secrets.shape[0]
    _want(secrets, (N, limbs(field)), "additive_share_prg secrets")
    if counter0 is None:
        counter0 = first_secret * (n - 1) * ((8 * limbs(field) + 15) // 16)  # FF::random burns whole blocks
    if out is None:
        out = empty(field, n, N, device=secrets.device)
    else:
        _want(out, (n, N, limbs(field)), "additive_share_prg out", secrets)
    _chk(lib.scl_hip_additive_share_prg(field, _dev(out), C.c_size_t(N), _dev(secrets), C.c_size_t(N), C.c_size_t(n),
                                        seed, C.c_size_t(len(seed)), C.c_uint64(counter0), _stream()))
    return out


def additive_recover(field, shares, out=None):
    n, N = shares.shape[0], shares.shape[1]
    _want(shares, (n, N, limbs(field)), "additive_recover shares")
    if out is None:
        out = empty(field, N, device=shares.device)
    else:
        _want(out, (N, limbs(field)), "additive_recover out", shares)
    _chk(lib.scl_hip_additive_recover(field, _dev(out), _dev(shares), C.c_size_t(N), C.c_size_t(n), C.c_size_t(N),
                                      _stream()))
    return out


# ---- matrices ---------------------------------------------------------------------------------------------------------
def vandermonde(field, n: int, m: int, xs=None, device="cuda"):
    L = limbs(field)
    out = empty(field, n, m, device=device)
    x = _host(xs).reshape(n, L) if xs is not None else None
    _chk(lib.scl_hip_vandermonde(field, _dev(out), C.c_size_t(n), C.c_size_t(m), _hp(x) if x is not None else None,
                                 _stream()))
    return out


def matmul(field, A, B, out=None):
    M, K = A.shape[0], A.shape[1]
    if B.shape[0] != K:
        raise SclError(ERR_MATMUL_DIMS, "")
    N = B.shape[1]
    if out is None:
        out = empty(field, M, N, device=A.device)
    _chk(lib.scl_hip_matmul(field, _dev(out), C.c_size_t(N), _dev(A), C.c_size_t(K), _dev(B), C.c_size_t(N),
                            C.c_size_t(M), C.c_size_t(K), C.c_size_t(N), _stream()))
    return out


# ---- layout -----------------------------------------------------------------------------------------------------------
def aos_to_soa(field, aos):
    N, n = aos.shape[0], aos.shape[1]
    out = empty(field, n, N, device=aos.device)
    _chk(lib.scl_hip_aos_to_soa(field, _dev(out), C.c_size_t(N), _dev(aos), C.c_size_t(N), C.c_size_t(n), _stream()))
    return out


def soa_to_aos(field, soa):
    n, N = soa.shape[0], soa.shape[1]
    out = empty(field, N, n, device=soa.device)
    _chk(lib.scl_hip_soa_to_aos(field, _dev(out), _dev(soa), C.c_size_t(N), C.c_size_t(N), C.c_size_t(n), _stream()))
    return out


# ---- wire image: seri::Serializer<Vector<FF>> -------------------------------------------------------------------------
def wire_pack(field, a: torch.Tensor) -> torch.Tensor:
    """elements [n][L] -> uint8 device buffer: u32 count || n * byteSize bytes (FF::write images)"""
    L = limbs(field)
    n = a.numel() // L
    lib.scl_hip_wire_size.restype = C.c_size_t
    out = torch.empty(lib.scl_hip_wire_size(field, C.c_size_t(n)), dtype=torch.uint8, device=a.device)
    _chk(lib.scl_hip_wire_pack(field, _dev(out), _dev(a) if n else None, C.c_size_t(n), _stream()))
    return out


def wire_unpack(field, raw: torch.Tensor, capacity: int | None = None) -> torch.Tensor:
    L = limbs(field)
    cap = (raw.numel() - 4) // (8 * L) if capacity is None else capacity
    out = empty(field, max(cap, 1), device=raw.device)
    n = C.c_size_t(0)
    _chk(lib.scl_hip_wire_unpack(field, _dev(out), C.c_size_t(cap), _dev(raw), C.c_size_t(raw.numel()), C.byref(n),
                                 _stream()))
    return out[: n.value]


def wire_pack_matrix(field, m: torch.Tensor) -> torch.Tensor:
    """row-major matrix [rows][cols][L] -> uint8 device buffer: u32 rows || u32 cols || vector image (matrix.h:910-963)"""
    rows, cols = (m.shape[0], m.shape[1]) if m.numel() else (0, 0)
    lib.scl_hip_wire_size_matrix.restype = C.c_size_t
    out = torch.empty(lib.scl_hip_wire_size_matrix(field, C.c_size_t(rows), C.c_size_t(cols)), dtype=torch.uint8,
                      device=m.device)
    _chk(lib.scl_hip_wire_pack_matrix(field, _dev(out), _dev(m) if m.numel() else None, C.c_size_t(cols), C.c_size_t(rows),
                                      C.c_size_t(cols), _stream()))
    return out


def wire_unpack_matrix(field, raw: torch.Tensor, capacity=None) -> torch.Tensor:
    """inverse of wire_pack_matrix; capacity = (rows, cols) of the destination (default: sized from the image)"""
    L = limbs(field)
    if capacity is None:
        hdr = raw[:12].cpu().numpy().view("<u4")
        capacity = (int(hdr[0]), int(hdr[1]))
    cr, cc = capacity
    out = empty(field, max(cr, 1), max(cc, 1), device=raw.device)
    r, c = C.c_size_t(0), C.c_size_t(0)
    _chk(lib.scl_hip_wire_unpack_matrix(field, _dev(out), C.c_size_t(max(cc, 1)), C.c_size_t(cr), _dev(raw),
                                        C.c_size_t(raw.numel()), C.byref(r), C.byref(c), _stream()))
    return out[: r.value, : c.value]


def frame_pack(field, a: torch.Tensor, as_matrix: bool = False) -> torch.Tensor:
    """TcpChannel frame (u32 packet size || packet) of a Packet holding this Vector [n][L] / Matrix [rows][cols][L]"""
    L = limbs(field)
    lib.scl_hip_wire_size.restype = lib.scl_hip_wire_size_matrix.restype = lib.scl_hip_frame_size.restype = C.c_size_t
    if as_matrix:
        rows, cols = (a.shape[0], a.shape[1]) if a.numel() else (0, 0)
        image = lib.scl_hip_wire_size_matrix(field, C.c_size_t(rows), C.c_size_t(cols))
        out = torch.empty(lib.scl_hip_frame_size(C.c_size_t(image)), dtype=torch.uint8, device=a.device)
        _chk(lib.scl_hip_frame_pack_matrix(field, _dev(out), _dev(a) if a.numel() else None, C.c_size_t(cols),
                                           C.c_size_t(rows), C.c_size_t(cols), _stream()))
        return out
    n = a.numel() // L
    image = lib.scl_hip_wire_size(field, C.c_size_t(n))
    out = torch.empty(lib.scl_hip_frame_size(C.c_size_t(image)), dtype=torch.uint8, device=a.device)
    _chk(lib.scl_hip_frame_pack(field, _dev(out), _dev(a) if n else None, C.c_size_t(n), _stream()))
    return out


def frame_unpack(field, raw: torch.Tensor) -> torch.Tensor:
    L = limbs(field)
    cap = max(0, (raw.numel() - 8) // (8 * L))
    out = empty(field, max(cap, 1), device=raw.device)
    n = C.c_size_t(0)
    _chk(lib.scl_hip_frame_unpack(field, _dev(out), C.c_size_t(cap), _dev(raw), C.c_size_t(raw.numel()), C.byref(n),
                                  _stream()))
    return out[: n.value]


def stream_copy(dst: torch.Tensor, src: torch.Tensor):
    _chk(lib.scl_hip_stream_copy(_dev(dst), _dev(src), C.c_size_t(src.numel() * src.element_size()), _stream()))


class Timer:
    """HIP-event timer on the stream the kernels are launched on (torch's current stream)"""

    def __init__(self):
        self._h = C.c_void_p()
        _chk(lib.scl_hip_timer_create(C.byref(self._h)))

    def start(self):
        _chk(lib.scl_hip_timer_start(self._h, _stream()))

    def stop(self):
        _chk(lib.scl_hip_timer_stop(self._h, _stream()))

    def elapsed_ms(self) -> float:
        ms = C.c_float()
        _chk(lib.scl_hip_timer_elapsed_ms(self._h, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            lib.scl_hip_timer_destroy(self._h)
        except Exception:
            pass
